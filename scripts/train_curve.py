"""trains the benchmark-shape model for a few hundred steps on a synthetic LANGUAGE (a sparse random bigram chain over
the 8k vocabulary, so that there is structure to learn) in the exact-fp32 mode and in the split-bf16 fp32 mode with
identical seeds -- and the exact-fp32 mode a second time -- and prints the loss curves side by side: the split mode
leaves the exact-fp32 curve no earlier and no further than a second run of the exact-fp32 mode itself does."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from argsim_amd.model import VAE

V, B, S, STEPS, EVERY = 8192, 256, 64, int(os.environ.get('STEPS', '400')), 25
rng = np.random.default_rng(0)
succ = rng.integers(3, V, (V, 8))                       # 8 successors per token
p = 1.0 / np.arange(1, 9); p /= p.sum()


def batch(seed):
    r = np.random.default_rng(seed)
    x = np.empty((B, S), np.int32)
    x[:, 0] = r.integers(3, V, B)
    for t in range(1, S):
        x[:, t] = succ[x[:, t - 1], r.choice(8, B, p=p)]
    if os.environ.get('RAGGED'):                  # LogNormal lengths, eos padded (SURVEY 8d's RAGGED set): exercises the padding skip
        lens = np.clip(np.rint(r.lognormal(np.log(24.0), 0.5, B)), 2, S).astype(int)
        for b, n in enumerate(lens):
            x[b, n:] = 1
    return x


# MODES=round2: the default path (layers fed by embedding rows project the present ids) against the per-token
# projection (table_l1 = 0) and the bf16-operand mode, same seeds
# MODES=round3: the default path against the reference's graph executed as written (table_l1 = 0, enc_top1 = 0: all steps of the top
# encoder layer's backward direction, skip_pad = 0: every step of every row, compact = 0: the padded layout) and the bf16-operand mode (bf16 gate gradients, saved
# gates, h / h_prev; transposing-load GEMMs), same seeds
ROUND2 = os.environ.get('MODES') in ('round2', 'round3')
RUNS = (('f32', 'f32', {}), ('f32 again', 'f32', {'table_l1': 0}), ('f32s', 'bf16', {})) if os.environ.get('MODES') == 'round2' else \
       (('f32', 'f32', {}), ('f32 again', 'f32', {'table_l1': 0, 'enc_top1': 0, 'skip_pad': 0, 'compact': 0}), ('f32s', 'bf16', {})) if os.environ.get('MODES') == 'round3' else \
       (('f32', 'f32', {}), ('f32 again', 'f32', {}), ('f32s', 'f32s', {}))
curves = {}
for tag, dt, opts in RUNS:
    m = VAE('train', seed=0, dtype=dt, dim_tgt=V, dim_emb=512, dim_rep=128, rnn_layers=3)
    for k, v in opts.items():
        m.set_option(k, v)
    out = []
    for i in range(STEPS):
        x = batch(i)
        m.train_step(x, x, seed=i)
        if i % EVERY == 0 or i == STEPS - 1:
            out.append((i,) + tuple(m.losses()))
    curves[tag] = out
    del m
    torch.cuda.empty_cache()
# 'f32 again' = the same exact-fp32 mode a second time: gradients use float atomics, so two runs of ONE mode drift apart too
print('step   loss_gen f32   f32 again (rel diff)     f32s (rel diff)   |  loss_kld f32   f32 again        f32s' if not ROUND2 else
      'step   loss_gen f32   per-token projection (rel diff)   bf16 mode (rel diff)   |  loss_kld f32   per-token      bf16')
for a, c, b in zip(curves['f32'], curves['f32 again'], curves['f32s']):
    print('%4d   %12.6f   %10.6f (%7.1e)   %10.6f (%7.1e) |  %12.6f  %10.6f  %10.6f' %
          (a[0], a[1], c[1], abs(a[1] - c[1]) / abs(a[1]), b[1], abs(a[1] - b[1]) / abs(a[1]), a[2], c[2], b[2]))
