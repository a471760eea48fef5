"""Co-occurrence line-file reader -- drop-in for ``CooccurrenceGenerator`` of the reference's
``wikipedia/cooccurrence_matrix.py:57-115`` ("next" row N2 of SURVEY.md 8f: the step before the hot path).

Input files are ``*.cooccur.pb.b64.bz2``: bz2 text, one base64 line per ``CooccurrenceRow`` protobuf
(proto/nlp.proto:44-48: ``uint64 index = 1; repeated uint64 other_index = 2; repeated float count = 3``).
The reference decodes them with generated protobuf classes inside a ``tf.data`` generator; neither
TensorFlow nor the generated module is a dependency here: the three fields are decoded directly from
the wire format (packed or unpacked repeated fields), and ``get_dataset`` returns a small iterator
object with the ``.prefetch(n).as_numpy_iterator()`` call shape of ``tf.data`` backed by a reader thread.

Batches have the reference's layout: ``([int32[B], int32[B]], float32[B])`` from ``get_batch`` and
``(int32[2, B], float32[B])`` from the dataset (cooccurrence_matrix.py:95-106,110-114).
"""
import base64
import bz2
import ctypes
import glob
import os
import queue
import struct
import threading

import numpy as np

_IO_LIB = None


def _io_lib():
    """libesr_io.so (esrecsys_amd/csrc/esr_io.c, built by esrecsys_amd/build.py): the same wire decoder in C."""
    global _IO_LIB
    if _IO_LIB is None:
        from ..build import IO_LIB_PATH, build_io_library
        if not os.path.exists(IO_LIB_PATH):
            build_io_library()
        lib = ctypes.CDLL(IO_LIB_PATH)
        lib.esr_cooccur_decode_lines.restype = ctypes.c_int64
        lib.esr_cooccur_decode_lines.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                                 ctypes.POINTER(ctypes.c_int64)]
        _IO_LIB = lib
    return _IO_LIB


def decode_lines(text, cap=None):
    """All complete base64 lines of `text` (bytes) -> (index int32[n], other int32[n], count float32[n], consumed).
    The C counterpart of parse_cooccurrence_row over a whole buffer; a row of more than `cap` pairs (default: enough
    for any row the buffer can hold) is never split."""
    lib = _io_lib()
    n = len(text)
    cap = int(cap) if cap else max(1024, n // 2)   # >= 2 base64 bytes per pair even for 1-byte ids and no counts
    t1 = np.empty(cap, np.int32)
    t2 = np.empty(cap, np.int32)
    cnt = np.empty(cap, np.float32)
    scratch = np.empty(n + 8, np.uint8)
    consumed = ctypes.c_int64(0)
    buf = (ctypes.c_char * n).from_buffer_copy(text) if not isinstance(text, (bytearray, memoryview)) else \
        (ctypes.c_char * n).from_buffer(text)
    got = lib.esr_cooccur_decode_lines(ctypes.addressof(buf), n, t1.ctypes.data, t2.ctypes.data, cnt.ctypes.data, cap,
                                       scratch.ctypes.data, ctypes.byref(consumed))
    if got < 0:
        raise ValueError("malformed CooccurrenceRow line at byte %d of the buffer" % consumed.value)
    return t1[:got], t2[:got], cnt[:got], consumed.value


def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if b < 0x80:
            return result, pos
        shift += 7


def parse_cooccurrence_row(serialized):
    """Decode one CooccurrenceRow.  Returns (index, other_index list, count list)."""
    buf = memoryview(serialized)
    n, pos = len(buf), 0
    index, others, counts = 0, [], []
    while pos < n:
        key, pos = _varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:  # varint
            val, pos = _varint(buf, pos)
            if field == 1:
                index = val
            elif field == 2:
                others.append(val)
        elif wire == 2:  # length-delimited: packed repeated field
            ln, pos = _varint(buf, pos)
            end = pos + ln
            if field == 2:
                while pos < end:
                    val, pos = _varint(buf, pos)
                    others.append(val)
            elif field == 3:
                counts.extend(struct.unpack_from("<%df" % (ln // 4), buf, pos))
            pos = end
        elif wire == 5:  # 32-bit: unpacked float
            if field == 3:
                counts.append(struct.unpack_from("<f", buf, pos)[0])
            pos += 4
        elif wire == 1:
            pos += 8
        else:
            raise ValueError("unsupported wire type %d in CooccurrenceRow" % wire)
    return index, others, counts


class CooccurrenceMatrix:
    """The whole co-occurrence file as a sparse matrix in host memory -- the reference's debug consumer
    (wikipedia/cooccurrence_matrix.py:18-55): ``CooccurrenceMatrix(input_file)`` loads, ``debug_print(max_rows,
    token_dictionary, num_terms)`` prints, per token, its num_terms heaviest partners.  Rows are kept as the reference
    keeps them (one list of (other_index, count) per row index, in file order; repeated row indices append; a row without
    pairs keeps its empty list), decoded by parse_cooccurrence_row instead of protobuf."""

    def __init__(self, input_file):
        self.load(input_file)

    def _reset(self):
        self._matrix = {}

    def load(self, input_file):
        """Loads a co-occurrence file (``*.cooccur.pb.b64.bz2``: one base64 CooccurrenceRow per line), line by line as the
        reference does (cooccurrence_matrix.py:38-52): EVERY row registers its index, also one without pairs (the block
        decoder yields pairs only, so it cannot stand in here), and the file is never held in memory whole."""
        import base64
        import bz2
        self._reset()
        with bz2.open(input_file, "rb") as f:
            for line in f:
                index, others, counts = parse_cooccurrence_row(base64.b64decode(line.rstrip(b"\n")))
                row = self._matrix.setdefault(index, [])
                row.extend(zip(others, counts))

    def rows(self):
        """{row index: [(other index, count), ...]} -- what the reference holds in its private ``__matrix``."""
        return self._matrix

    def debug_print(self, max_rows, token_dictionary, num_terms):
        count = 0
        for key in self._matrix.keys():
            row = sorted(self._matrix[key], key=lambda x: x[1], reverse=True)
            token = token_dictionary.get_token_from_embedding_index(key)
            print("Token [%s]" % token)
            for i in range(min(num_terms, len(row))):
                token = token_dictionary.get_token_from_embedding_index(row[i][0])
                print(" %s : %f" % (token, row[i][1]))
            count = count + 1
            if count > max_rows:  # (the reference's own off-by-one: max_rows + 1 tokens are printed)
                break


class _Dataset:
    """The slice of the tf.data API the reference's trainer touches (train_cooccurence.py:165-166)."""

    def __init__(self, make_iter, depth=0):
        self._make_iter = make_iter
        self._depth = depth

    def prefetch(self, buffer_size):
        return _Dataset(self._make_iter, 8 if buffer_size is None or buffer_size < 0 else max(1, int(buffer_size)))

    def as_numpy_iterator(self):
        it = self._make_iter()
        if not self._depth:
            return it
        q = queue.Queue(maxsize=self._depth)

        def worker():
            # whatever ends the producer -- exhaustion, no files matching the glob, a malformed line, a bz2 / IO
            # error -- reaches the consumer as a sentinel: a silent thread death would leave next() blocked forever
            try:
                for item in it:
                    q.put(("item", item))
                q.put(("end", None))
            except BaseException as e:  # noqa: BLE001 -- re-raised on the consumer side
                q.put(("err", e))

        threading.Thread(target=worker, daemon=True).start()

        def gen():
            while True:
                kind, payload = q.get()
                if kind == "item":
                    yield payload
                elif kind == "err":
                    raise payload
                else:
                    return
        return gen()

    __iter__ = as_numpy_iterator


AUTOTUNE = -1


class CooccurrenceGenerator:
    def __init__(self, input_pattern):
        self._input_files = glob.glob(input_pattern)
        self._total_files = len(self._input_files)

    def get_item(self):
        """Gets a single item of i, j, count -- cycles over the files forever (cooccurrence_matrix.py:62-78)."""
        if not self._input_files:
            raise FileNotFoundError("no co-occurrence files match the input pattern")
        while True:
            for input_file in self._input_files:
                with bz2.open(input_file, "rb") as file:
                    for line in file:
                        index, others, counts = parse_cooccurrence_row(base64.b64decode(line[:-1]))
                        for i in range(len(others)):
                            yield (index, others[i], counts[i])

    @staticmethod
    def _file_blocks(input_file, chunk_bytes):
        """One file -> blocks (index int32[k], other int32[k], count float32[k]) in file order."""
        tail = b""
        with bz2.open(input_file, "rb") as file:
            while True:
                chunk = file.read(chunk_bytes)
                data = tail + chunk
                if not chunk and data and not data.endswith(b"\n"):
                    data += b"\n"  # a last line without a newline is still a line
                if not data:
                    break
                t1, t2, cnt, used = decode_lines(data)
                if used == 0 and not chunk:
                    break
                tail = data[used:]
                if len(t1):
                    yield t1, t2, cnt
                if not chunk:
                    break

    def get_item_blocks(self, chunk_bytes=1 << 22, workers=None):
        """The same stream as get_item, as blocks of arrays (index int32[k], other int32[k], count float32[k]):
        files are decompressed `chunk_bytes` at a time and decoded by libesr_io.so, whole lines at a time.
        bz2 decompression (~20 MB/s of text per core, GIL released) is what bounds one file, so with several
        files the next `workers` of them are decoded ahead on threads (default: up to 8, half the cores) and handed
        over in file order -- the item order of the reference's sequential walk is kept; memory: up to `workers`
        decoded files (12 bytes per pair)."""
        if not self._input_files:
            raise FileNotFoundError("no co-occurrence files match the input pattern")
        files = self._input_files
        if workers is None:
            workers = min(8, max(1, (os.cpu_count() or 2) // 2), len(files))
        if workers <= 1 or len(files) == 1:
            while True:
                for input_file in files:
                    yield from self._file_blocks(input_file, chunk_bytes)
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=workers)
        window, nxt = deque(), 0
        while True:
            while len(window) < workers:
                window.append(pool.submit(lambda f: list(self._file_blocks(f, chunk_bytes)), files[nxt % len(files)]))
                nxt += 1
            yield from window.popleft().result()

    def _take(self, blocks, pending, n):
        """n items from the block stream (pending = leftover arrays of the last block) -> three arrays, new pending."""
        parts, have = [], 0
        while have < n:
            if pending is None or len(pending[0]) == 0:
                pending = next(blocks)
            k = min(n - have, len(pending[0]))
            parts.append(tuple(a[:k] for a in pending))
            pending = tuple(a[k:] for a in pending)
            have += k
        if len(parts) == 1:
            return parts[0], pending
        return tuple(np.concatenate([p[i] for p in parts]) for i in range(3)), pending

    def get_shuffled_items(self, num_items):
        """cooccurrence_matrix.py:80-87: the item stream shuffled in buffers of `num_items`.  The same draw from the global
        NumPy RNG as the reference's ``np.random.shuffle(items)`` (one permutation of `num_items` positions per buffer),
        applied to the decoded array blocks; yields ``(index, other_index, count)`` tuples."""
        blocks, pending = self.get_item_blocks(), None
        while True:
            buf, pending = self._take(blocks, pending, num_items)
            order = np.arange(num_items)
            np.random.shuffle(order)
            for j in order:
                yield (int(buf[0][j]), int(buf[1][j]), float(buf[2][j]))

    def get_batch(self, batch_size, shuffle_size=0):
        """cooccurrence_matrix.py:89-106.  Same batches as the reference's item-at-a-time loop -- including the
        buffer shuffle: fill `shuffle_size` items, np.random.shuffle (global NumPy RNG, one call per buffer: a
        permutation of positions drawn exactly as shuffling the item list draws it), drain -- built from array
        blocks decoded in C instead of one Python tuple per pair."""
        blocks = self.get_item_blocks()
        pending = None
        if not shuffle_size:
            while True:
                (t1, t2, cnt), pending = self._take(blocks, pending, batch_size)
                yield ([np.ascontiguousarray(t1), np.ascontiguousarray(t2)], np.ascontiguousarray(cnt))
        buf, pos = None, 0
        while True:
            out = [np.empty(batch_size, np.int32), np.empty(batch_size, np.int32), np.empty(batch_size, np.float32)]
            filled = 0
            while filled < batch_size:
                if buf is None or pos == shuffle_size:
                    buf, pending = self._take(blocks, pending, shuffle_size)
                    order = np.arange(shuffle_size)
                    np.random.shuffle(order)
                    buf = tuple(a[order] for a in buf)
                    pos = 0
                k = min(batch_size - filled, shuffle_size - pos)
                for o, a in zip(out, buf):
                    o[filled:filled + k] = a[pos:pos + k]
                filled += k
                pos += k
            yield ([out[0], out[1]], out[2])

    def get_dataset(self, batch_size, shuffle_size=0):
        """Returns the batches as a tf.data-shaped dataset of ``(int32[2, B], float32[B])``
        (cooccurrence_matrix.py:108-115)."""
        def make_iter():
            for x, y in self.get_batch(batch_size, shuffle_size):
                yield np.stack(x), y
        return _Dataset(make_iter)
