"""A/B of a library option inside one process: alternating blocks of timed training steps.  usage: [DTYPE=bf16] ab_option.py KEY [B S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import synth
from argsim_amd.model import VAE
key = sys.argv[1]
B, S = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 64)
m = VAE('train', seed=0, dtype=os.environ.get('DTYPE', 'f32'), dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
m.step = 20000
_np = synth.batch(B, S, 8192, ragged=bool(os.environ.get('RAGGED')), seed=0, len_median=float(os.environ.get('LEN_MEDIAN', '24')), len_sigma=float(os.environ.get('LEN_SIGMA', '0.5')))
print('fill %.3f' % (float((_np != 1).sum()) / _np.size), flush=True)
ids = torch.as_tensor(_np).cuda()
for i in range(5): m.train_step(ids, ids, seed=i)
torch.cuda.synchronize()
NSTEP = int(os.environ.get('NSTEP', '15'))
VALS = [int(x) for x in os.environ.get('VALS', '1,0').split(',')]
res = {v: [] for v in VALS}
for rep in range(int(os.environ.get('REPS', '6'))):
    for v in VALS:
        m.set_option(key, v)
        for i in range(2): m.train_step(ids, ids, seed=i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(NSTEP): m.train_step(ids, ids, seed=100 + i)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / NSTEP * 1e3)
        print(key, v, '%.3f ms' % res[v][-1], 'losses', m.losses(), flush=True)
for v in VALS:
    print(key, v, ' '.join('%.3f' % x for x in res[v]), ' median %.3f ms' % sorted(res[v])[len(res[v]) // 2])
