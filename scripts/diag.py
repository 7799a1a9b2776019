import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
def log(*a):
    print(*a, flush=True)
p = torch.cuda.get_device_properties(0)
log('device', p.name, 'CUs', p.multi_processor_count, 'mem GB', p.total_memory / 2**30)
log('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads())
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try: log(f, open(f).read().strip())
    except Exception as e: log(f, 'n/a')
from argsim_amd import synth
from argsim_amd.model import VAE
CFG = dict(dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
m = VAE('train', seed=0, **CFG)
m.step = 20000
for B in (16, 64, 256):
    ids = torch.as_tensor(synth.batch(B, 64, 8192, seed=0)).cuda()
    for mode in (0, 1):
        m.set_option('persistent', mode)
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.train_step(ids, ids, seed=it)
            try:
                l = m.losses()
            except Exception as e:
                l = repr(e)
            log('B', B, 'persistent', mode, 'iter', it, 'ms %.2f' % (1e3 * (time.perf_counter() - t0)), l)
