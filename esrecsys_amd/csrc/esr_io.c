/* esr_io.c -- host-side decoder of the co-occurrence line files that feed the GloVe step
 * (wikipedia/cooccurrence_matrix.py:62-78 of the reference: one base64 line per `CooccurrenceRow` protobuf,
 * proto/nlp.proto:44-48: `uint64 index = 1; repeated uint64 other_index = 2; repeated float count = 3`).
 *
 * The reference walks these with generated protobuf classes inside a Python generator; the pure-Python wire decoder of
 * esrecsys_amd/wikipedia/cooccurrence_matrix.py does 0.23 M pairs/s, three orders of magnitude below what the HIP step
 * consumes.  This file is the same decoder in plain C (no protobuf / TensorFlow dependency): base64 -> wire format ->
 * three flat arrays, whole lines at a time.  Built with gcc into libesr_io.so by esrecsys_amd/build.py; loaded with
 * ctypes.  The Python decoder stays as the readable restatement the tests compare this one against.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/esr_io.h"

#define ESR_IO_EFORMAT (-1) /* malformed base64 / wire data; *consumed = offset of the offending line */

static int8_t b64val[256];
/* filled when the library is loaded: decoder calls come from several reader threads at once */
__attribute__((constructor)) static void b64_init(void) {
  memset(b64val, -1, sizeof(b64val));
  const char* a = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  for (int i = 0; i < 64; ++i) b64val[(uint8_t)a[i]] = (int8_t)i;
}

/* decode one base64 line (no newline) into out; returns decoded length or -1 */
static int64_t b64_decode(const uint8_t* s, int64_t n, uint8_t* out) {
  while (n > 0 && (s[n - 1] == '\r' || s[n - 1] == ' ')) --n;
  if (n % 4 != 0) return -1;
  int64_t o = 0;
  for (int64_t i = 0; i < n; i += 4) {
    const int a = b64val[s[i]], b = b64val[s[i + 1]];
    const int pad2 = s[i + 2] == '=', pad3 = s[i + 3] == '=';
    const int c = pad2 ? 0 : b64val[s[i + 2]], d = pad3 ? 0 : b64val[s[i + 3]];
    if (a < 0 || b < 0 || c < 0 || d < 0) return -1;
    if ((pad2 || pad3) && i + 4 != n) return -1;
    if (pad2 && !pad3) return -1;
    const uint32_t v = ((uint32_t)a << 18) | ((uint32_t)b << 12) | ((uint32_t)c << 6) | (uint32_t)d;
    out[o++] = (uint8_t)(v >> 16);
    if (!pad2) out[o++] = (uint8_t)(v >> 8);
    if (!pad3) out[o++] = (uint8_t)v;
  }
  return o;
}

static inline int varint(const uint8_t* b, int64_t n, int64_t* pos, uint64_t* val) {
  uint64_t r = 0;
  int shift = 0;
  while (*pos < n && shift < 70) {
    const uint8_t x = b[(*pos)++];
    r |= (uint64_t)(x & 0x7F) << (shift < 64 ? shift : 63);
    if (x < 0x80) {
      *val = r;
      return 0;
    }
    shift += 7;
  }
  return -1;
}

/* One CooccurrenceRow -> pairs (index, other_index[i], count[i]).  Two passes over the message: field order is not
 * guaranteed, packed and unpacked encodings of the repeated fields are both legal.  Returns the number of pairs, -1 on
 * malformed data, -2 if they do not fit (cap). */
static int64_t row_pairs(const uint8_t* b, int64_t n, int32_t* t1, int32_t* t2, float* cnt, int64_t cap) {
  int64_t pos = 0, nother = 0, ncount = 0;
  uint64_t index = 0;
  while (pos < n) {
    uint64_t key, v;
    if (varint(b, n, &pos, &key)) return -1;
    const uint64_t field = key >> 3;
    const int wire = (int)(key & 7);
    if (wire == 0) {
      if (varint(b, n, &pos, &v)) return -1;
      if (field == 1) index = v;
      else if (field == 2) {
        if (nother >= cap) return -2;
        t2[nother++] = (int32_t)v;
      }
    } else if (wire == 2) {
      if (varint(b, n, &pos, &v)) return -1;
      if (v > (uint64_t)(n - pos)) return -1;
      const int64_t end = pos + (int64_t)v;
      if (field == 2) {
        while (pos < end) {
          uint64_t o;
          if (varint(b, end, &pos, &o)) return -1;
          if (nother >= cap) return -2;
          t2[nother++] = (int32_t)o;
        }
      } else if (field == 3) {
        const int64_t k = (int64_t)v / 4;
        if (ncount + k > cap) return -2;
        memcpy(cnt + ncount, b + pos, (size_t)k * 4); /* little-endian host (x86-64), as the wire format */
        ncount += k;
      }
      pos = end;
    } else if (wire == 5) {
      if (pos + 4 > n) return -1;
      if (field == 3) {
        if (ncount >= cap) return -2;
        memcpy(cnt + ncount, b + pos, 4);
        ++ncount;
      }
      pos += 4;
    } else if (wire == 1) {
      if (pos + 8 > n) return -1;
      pos += 8;
    } else {
      return -1;
    }
  }
  if (ncount < nother) return -1; /* the reference indexes count[i] for every other_index[i] */
  for (int64_t i = 0; i < nother; ++i) t1[i] = (int32_t)index;
  return nother;
}

/* Decode the complete lines of text[0, len).  Pairs are appended to t1 / t2 / cnt (capacity cap); a line whose pairs
 * do not fit is left unconsumed.  Returns the number of pairs written (>= 0) or ESR_IO_EFORMAT; *consumed = bytes of
 * text used (always a whole number of lines).  scratch: at least (longest line) * 3 / 4 + 4 bytes -- pass len. */
int64_t esr_cooccur_decode_lines(const uint8_t* text, int64_t len, int32_t* t1, int32_t* t2, float* cnt, int64_t cap,
                                 uint8_t* scratch, int64_t* consumed) {
  int64_t pos = 0, npairs = 0;
  *consumed = 0;
  while (pos < len) {
    const uint8_t* nl = (const uint8_t*)memchr(text + pos, '\n', (size_t)(len - pos));
    if (!nl) break; /* incomplete last line: the caller supplies it again with more data */
    const int64_t line_len = (int64_t)(nl - (text + pos));
    if (line_len > 0) {
      const int64_t m = b64_decode(text + pos, line_len, scratch);
      if (m < 0) return ESR_IO_EFORMAT;
      const int64_t k = row_pairs(scratch, m, t1 + npairs, t2 + npairs, cnt + npairs, cap - npairs);
      if (k == -1) return ESR_IO_EFORMAT;
      if (k == -2) break; /* out of room: stop before this line */
      npairs += k;
    }
    pos += line_len + 1;
    *consumed = pos;
  }
  return npairs;
}

int esr_io_version(void) { return 100; }
