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
import glob
import queue
import struct
import threading

import numpy as np


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
            for item in it:
                q.put(item)

        threading.Thread(target=worker, daemon=True).start()

        def gen():
            while True:
                yield q.get()
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

    def get_shuffled_items(self, num_items):
        """Pre-fetches and shuffles num_items of stuff (cooccurrence_matrix.py:80-87; global NumPy RNG, as there)."""
        iterator = self.get_item()
        while True:
            items = [next(iterator) for _ in range(num_items)]
            np.random.shuffle(items)
            for item in items:
                yield item

    def get_batch(self, batch_size, shuffle_size=0):
        """cooccurrence_matrix.py:89-106."""
        iterator = self.get_shuffled_items(shuffle_size) if shuffle_size else self.get_item()
        while True:
            token1 = np.empty(batch_size, np.int32)
            token2 = np.empty(batch_size, np.int32)
            token_count = np.empty(batch_size, np.float32)
            for k in range(batch_size):
                token1[k], token2[k], token_count[k] = next(iterator)
            yield ([token1, token2], token_count)

    def get_dataset(self, batch_size, shuffle_size=0):
        """Returns the batches as a tf.data-shaped dataset of ``(int32[2, B], float32[B])``
        (cooccurrence_matrix.py:108-115)."""
        def make_iter():
            for x, y in self.get_batch(batch_size, shuffle_size):
                yield np.stack(x), y
        return _Dataset(make_iter)
