"""numpy batch helpers (counterparts of reference src/util_np.py:5-33).

Golden vectors captured from the reference functions: tests/golden/util_np_golden.json."""
import numpy as np


def vpack(arrays, shape, fill, dtype=None):
    """stack ragged 1-d ``arrays`` into an array of ``shape`` padded at the end with ``fill``
    (src/util_np.py:5-13).  Rows beyond ``shape[0]`` are ignored; a row longer than
    ``shape[1]`` raises, as in the reference."""
    out = np.full(shape, fill, dtype)
    for row, arr in zip(out, arrays):
        row[:len(arr)] = arr
    return out


def partition(n, m, discard=False):
    """yields (i, j) bounds cutting range(n) into chunks of ``m``; the final short chunk is
    yielded unless ``discard`` (src/util_np.py:16-24)."""
    full = n // m
    for k in range(full):
        yield k * m, (k + 1) * m
    if n % m and not discard:
        yield full * m, n


def sample(n, seed=0):
    """endless index stream (src/util_np.py:27-33): every epoch re-seeds the legacy numpy
    generator with ``seed`` and shuffles the ALREADY shuffled list in place, so epoch k applies the
    same permutation k times.  A private RandomState(seed) reproduces ``np.random.seed(seed);
    np.random.shuffle(data)`` without touching the global generator."""
    data = list(range(n))
    while True:
        np.random.RandomState(seed).shuffle(data)
        yield from data
