"""TEST INFRASTRUCTURE: the item-at-a-time batching of the reference's CooccurrenceGenerator
(/root/reference wikipedia/cooccurrence_matrix.py:80-106), restated as free functions over the product's
``get_item`` stream, so the product's block-wise ``get_batch`` can be compared against it under the same
NumPy seed.  Not imported by the product."""
import itertools

import numpy as np


def shuffled_items(generator, num_items):
    """Fill a buffer of num_items items, np.random.shuffle it (global NumPy RNG, one call per buffer), drain."""
    stream = generator.get_item()
    while True:
        buffer = list(itertools.islice(stream, num_items))
        np.random.shuffle(buffer)
        yield from buffer


def batches_item_by_item(generator, batch_size, shuffle_size=0):
    """([token1 int32[B], token2 int32[B]], count float32[B]) built one item per next(), as the reference does."""
    stream = shuffled_items(generator, shuffle_size) if shuffle_size else generator.get_item()
    while True:
        rows = list(itertools.islice(stream, batch_size))
        t1 = np.fromiter((r[0] for r in rows), np.int32, batch_size)
        t2 = np.fromiter((r[1] for r in rows), np.int32, batch_size)
        cnt = np.fromiter((r[2] for r in rows), np.float32, batch_size)
        yield ([t1, t2], cnt)
