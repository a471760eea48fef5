/* esr_io.h -- host-side input decoding for the GloVe step (libesr_io.so, plain C, no device code).
 *
 * Replaces the per-row protobuf parse inside the reference's Python generator
 * (wikipedia/cooccurrence_matrix.py:62-78: bz2 text, one base64 line per `CooccurrenceRow`,
 * proto/nlp.proto:44-48: uint64 index = 1; repeated uint64 other_index = 2; repeated float count = 3).
 * A maintainer of the reference would bind it with ctypes exactly as
 * esrecsys_amd/wikipedia/cooccurrence_matrix.py does (decode_lines). */
#ifndef ESR_IO_H_
#define ESR_IO_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int esr_io_version(void);

/* Decode the complete lines of text[0, len): every line is base64 of one CooccurrenceRow; its pairs
 * (index, other_index[i], count[i]) are appended to t1 / t2 / cnt (capacity cap, ids truncated to int32 as the
 * reference's int32 batch arrays do).  A row whose pairs do not fit, and an incomplete last line, are left
 * unconsumed.  scratch: len + 8 bytes.  Returns the number of pairs written, or -1 for malformed base64 / wire data;
 * *consumed = bytes of text used (a whole number of lines; on -1 the offset of the offending line). */
int64_t esr_cooccur_decode_lines(const uint8_t* text, int64_t len, int32_t* t1, int32_t* t2, float* cnt,
                                 int64_t cap, uint8_t* scratch, int64_t* consumed);

#ifdef __cplusplus
}
#endif
#endif /* ESR_IO_H_ */
