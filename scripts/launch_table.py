"""per-launch table of one timed ELBO step: class, duration, TFLOP/s of every GEMM / GRU launch in launch order
(HIP-event stamps through the avae_debug_timing hook).  usage: launch_table.py [f32|f32s|bf16]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import synth
from argsim_amd.model import VAE
dt = sys.argv[1] if len(sys.argv) > 1 else 'f32'
m = VAE('train', seed=0, dtype=dt, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
m.step = 20000
BB, SS = int(os.environ.get('B', '256')), int(os.environ.get('S', '64'))
ids = torch.as_tensor(synth.batch(BB, SS, 8192, ragged=bool(os.environ.get('RAGGED')), seed=0)).cuda()
print('tokens incl. eos:', int((ids != 1).sum()) + BB)
for kv in os.environ.get('OPTS', '').split(','):
    if kv: m.set_option(kv.split('=')[0], int(kv.split('=')[1]))
for i in range(3): m.train_step(ids, ids, seed=i)
m.set_option('timing', 1)
m.train_step(ids, ids, seed=9)
out = (C.c_double * (3 * 256))(); n = C.c_int(0)
assert m._l.avae_debug_timing(m._h, out, 256, C.byref(n)) == 0
m.set_option('timing', 0)
names = ['gemm', 'gru_fwd', 'gru_bwd']
tot = [0.0, 0.0, 0.0]
for i in range(n.value):
    c, ms, fl = int(out[3 * i]), out[3 * i + 1], out[3 * i + 2]
    tot[c] += ms
    print('%3d %-8s %8.1f us  %8.2f GFLOP  %7.1f TFLOP/s' % (i, names[c], ms * 1e3, fl / 1e9, fl / ms / 1e9 if ms > 0 else 0.0))
print('totals (ms):', dict(zip(names, [round(t, 2) for t in tot])))
