"""greedy decoding, steps = 512 (explore_centroids.py:40): persistent launch vs one launch sequence per token"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from argsim_amd.model import VAE
m = VAE('infer', seed=2, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
steps = 512
for b in (1, 16, 64, 128):
    z = np.random.default_rng(b).standard_normal((b, 128)).astype(np.float32)
    for mode in (1, 0):
        m.set_option('persistent', mode)
        y = m.decode(z, steps=steps)
        t0 = time.perf_counter()
        y = m.decode(z, steps=steps)
        dt = time.perf_counter() - t0
        print('b %4d  %-10s tokens kept %4d  %.1f ms  %.1f us/token-step  %.0f tokens/s' %
              (b, 'persistent' if mode else 'stepwise', y.shape[1], dt * 1e3, dt * 1e6 / max(y.shape[1], 1), b * y.shape[1] / dt), flush=True)
