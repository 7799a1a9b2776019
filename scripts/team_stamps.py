"""phase stamps of the forward team kernel (diagnostic build: make -C argsim_amd/csrc clean && make DIAG=1)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import synth
from argsim_amd.model import VAE
m = VAE('train', seed=0, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
m.step = 20000
names = ['Aload+poll', 'mfma', 'barrier2', 'gates+store', 'turn-wait', 'loop', 'barrier1', 'write+prefetch']
for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '256,1024').split(',')]:
  ids = torch.as_tensor(synth.batch(B, 64, 8192, seed=0)).cuda()
  for turn in (0,):
    for ab in (128, 128 | 16):
      for i in range(2): m.encode(ids)
      m.set_option('gru_ablate', ab)
      out = (C.c_uint64 * 32)()
      m._l.avae_debug_stamps(m._h, out)
      for i in range(3): m.encode(ids)
      m._l.avae_debug_stamps(m._h, out)
      m.set_option('gru_ablate', 0)
      n, steps = out[10], out[8] / max(out[10], 1)
      per = [out[i] / max(n, 1) / max(steps, 1) * 0.01 for i in range(8)]
      print('B', B, 'turn', turn, 'ablate', ab, 'team-launches', n, 'steps %.1f' % steps, ' '.join('%s %.2f' % (a, b) for a, b in zip(names, per)), 'total %.2f us/step' % sum(per), flush=True)
