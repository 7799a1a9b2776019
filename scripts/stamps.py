"""GRU phase breakdown from in-kernel stamps (diagnostic build path: option gru_ablate bit 32)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argsim_amd import synth
from argsim_amd.model import VAE
m = VAE('train', seed=0, dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
m.step = 20000
ids = torch.as_tensor(synth.batch(256, 64, 8192, seed=0)).cuda()
names = ['prefetch', 'Aload+poll', 'mfma', 'barrier', 'gates', 'unused', 'tail', 'loop']
for extra in (0, 16):
    for slow in (0, 1):
        m.set_option('gru_force_slow', slow)
        m.set_option('gru_ablate', 0)
        for i in range(2):
            m.train_step(ids, ids, seed=i)
        m.losses()
        m.set_option('gru_ablate', 32 | extra)
        out = (C.c_uint64 * 32)()
        m._l.avae_debug_stamps(m._h, out)
        for i in range(3):
            m.train_step(ids, ids, seed=i)
        m._l.avae_debug_stamps(m._h, out)
        for k, nm in ((0, 'fwd'), (16, 'bwd')):
            wgs, fast = out[k + 10], out[k + 9]
            steps = out[k + 8] / max(wgs, 1)
            per = [out[k + i] / max(wgs, 1) / max(steps, 1) * 0.01 for i in range(8)]   # us per step per WG
            print('ablate', extra, 'force_slow', slow, nm, 'WG-launches', wgs, 'fast', fast, 'steps/launch %.1f' % steps,
                  ' '.join('%s %.2f' % (n, v) for n, v in zip(names, per)), 'total %.2f us/step' % sum(per), flush=True)
