"""synthetic IBM/IAC-shaped batches.

ids int32 (B, S): tokens i.i.d. Zipf(s=1.0) over [3, V-1] (0/1/2 = unk/eos/bos, reference
src/util_sp.py:17).  FULL: every row has length S.  RAGGED: len = clip(round(LogNormal(ln 24, 0.5)), 2, S),
eos padded like util_np.vpack.  src = tgt (no SentencePiece sampling)."""
import numpy as np


def zipf_ids(rng, n, V):
    ranks = np.arange(1, V - 3 + 1, dtype=np.float64)
    p = 1.0 / ranks
    p /= p.sum()
    return (rng.choice(V - 3, size=n, p=p) + 3).astype(np.int32)


def batch(B, S, V=8192, ragged=False, seed=0, eos=1, len_median=24.0, len_sigma=0.5):
    rng = np.random.default_rng(seed)
    ids = zipf_ids(rng, B * S, V).reshape(B, S)
    if ragged:
        lens = np.clip(np.rint(rng.lognormal(np.log(len_median), len_sigma, B)), 2, S).astype(int)
        for b, n in enumerate(lens):
            ids[b, n:] = eos
    return ids
