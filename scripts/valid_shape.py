"""the reference's validation geometry (src/train.py:104-113: chunks of 200 rows, rows up to 512 pieces): one eval
call at B=200, S=512 with config.json dims (D=512, R=1024) -- finite outputs, time per chunk."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from argsim_amd.model import VAE
rng = np.random.default_rng(0)
B, S = 200, 512
lens = np.clip(np.rint(rng.lognormal(np.log(60.0), 0.8, B)), 2, S).astype(int); lens[0] = S
ids = np.ones((B, S), np.int32)
for b, n in enumerate(lens): ids[b, :n] = rng.integers(3, 8192, n)
for dt in ('f32', 'f32s'):
    m = VAE('valid', seed=0, dtype=dt, dim_tgt=8192, dim_emb=512, dim_rep=1024, rnn_layers=3)
    for i in range(2): errt, lgen, lkld = m.eval(ids, ids)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3): errt, lgen, lkld = m.eval(ids, ids)
    torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / 3
    assert np.isfinite(lgen).all() and np.isfinite(lkld).all() and lgen.shape == (int(lens.sum()) + B,)
    print(dt, 'eval B=200 S=512: %.1f ms per chunk, %d tokens, mean CE %.4f, error rate %.4f' % (1e3 * dt_, lgen.size, lgen.mean(), errt.mean()))
    del m; torch.cuda.empty_cache()
