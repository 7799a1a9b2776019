"""worker of tests/test_gpu_dp.py: one data-parallel rank computing on cuda:0, gradients all-reduced over gloo
(two ranks share the one GPU of the test box; RCCL refuses two ranks on one device, the code path above the
transport -- bucket hook, side stream, global-count scaling, Adam after the reduce -- is the production one; because
a persistent GRU launch needs every CU, the two processes take turns for those launches: option shared_device)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from argsim_amd.dist import DataParallel, shard_rows
from argsim_amd.model import VAE


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    case = np.load(sys.argv[1])
    out = sys.argv[2]
    dist.init_process_group('gloo', rank=rank, world_size=world)
    kw = dict(dim_tgt=int(case['V']), dim_emb=int(case['D']), dim_rep=int(case['R']), rnn_layers=3, seed=0)
    m = VAE('train', **kw)
    m.step = 20000
    m.set_option('shared_device', 1)      # two processes on ONE GPU: persistent launches take turns (a persistent launch needs every CU)
    dp = DataParallel(m)
    dp.broadcast_params(m.state)
    ids, keep, eps = case['ids'], case['keep'], case['eps']
    lo, hi = shard_rows(len(ids), rank, world)
    n_glob = float((ids != 1).sum() + len(ids))
    mine = ids[lo:hi]
    width = max(int((mine != 1).sum(1).max()), 1)      # VAE trims a shard to ITS longest row: ranks differ in width
    for i in range(int(case['steps'])):
        dp.train_step(mine, mine, n_glob, float(len(ids)), keep_mask=keep[:width, lo:hi], eps=eps[lo:hi])
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out, **{k.replace('/', '|'): v for k, v in m.get_params().items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
