"""data-parallel path on CPU: two gloo ranks, each with an oracle-backed stand-in for the device model,
shard a ragged global batch, scale by the GLOBAL token / row counts, all-reduce bucket by bucket through
argsim_amd.dist, and must reproduce the single-process gradient and one TF-Adam update."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleModel:
    """same surface as argsim_amd.model.VAE for DataParallel (grads, buckets, hook, forward_backward, adam_step)"""

    def __init__(self, cfg, P):
        from oracle import vae_torch as vt
        self.vt, self.cfg = vt, cfg
        self.names = list(P)
        self.sizes = [int(np.prod(P[k].shape)) for k in self.names]
        self.offs = np.concatenate([[0], np.cumsum(self.sizes)])
        n = int(self.offs[-1])
        self.params = torch.zeros(n, dtype=torch.float64)
        for k, o in zip(self.names, self.offs):
            self.params[o:o + P[k].size] = torch.tensor(P[k].ravel())
        self.grads = torch.zeros(n, dtype=torch.float64)
        self.m = torch.zeros(n, dtype=torch.float64)
        self.v = torch.zeros(n, dtype=torch.float64)
        self.step = 20000
        self.hook = None
        self.shapes = {k: P[k].shape for k in self.names}

    def buckets(self):
        # three uneven buckets covering the flat buffer
        n = len(self.grads)
        cuts = [0, n // 5, n // 2, n]
        return [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(3)]

    def set_grad_hook(self, fn):
        self.hook = fn

    def unflat(self, flat):
        return {k: flat[o:o + s].reshape(self.shapes[k]).numpy() for k, o, s in zip(self.names, self.offs, self.sizes)}

    def forward_backward(self, src, tgt, seed=None, keep_mask=None, eps=None, n_tok_global=0.0, b_global=0.0):
        vt, cfg = self.vt, self.cfg
        P = vt.to_torch(self.unflat(self.params))
        o = vt.forward(P, cfg, src, tgt, 'train', self.step, keep_mask, eps)
        n_loc = float(o['loss_gen_samp'].numel())
        b_loc = float(len(src))
        R = cfg['dim_rep']
        anneal = float(np.tanh(cfg['accelerate'] * self.step))
        # global-mean ELBO restricted to this shard: sums scaled by the GLOBAL denominators
        loss = o['loss_gen_samp'].sum() / (n_tok_global or n_loc) + anneal * o['loss_kld_samp'].sum() / ((b_global or b_loc) * R)
        loss.backward()
        g = torch.cat([(P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])).reshape(-1) for k in self.names])
        self.grads.copy_(g)
        if self.hook:
            for i, (off, cnt) in enumerate(self.buckets()):
                self.hook(-1, 0, 0)            # AVAE_HOOK_FENCE: what the library sends before a persistent launch
                self.hook(i, off, cnt)

    def adam_step(self):
        from oracle import vae_numpy as vn
        lr = vn.schedule(self.step, self.cfg['accelerate'], self.cfg['learn_rate'])[2]
        p, m, v = vn.adam_tf({'x': self.params.numpy()}, {'x': self.grads.numpy()}, {'x': self.m.numpy()}, {'x': self.v.numpy()}, self.step, lr)
        self.params, self.m, self.v = torch.tensor(p['x']), torch.tensor(m['x']), torch.tensor(v['x'])
        self.step += 1


def _case():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import make_case
    return make_case('mid')


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from argsim_amd.dist import DataParallel, global_token_count, shard_rows
    cfg, P, ids, keep, eps = _case()
    model = OracleModel(cfg, P)
    dp = DataParallel(model, overlap=(rank >= 0))
    lo, hi = shard_rows(len(ids), rank, world)
    n_loc = int((ids[lo:hi] != cfg['eos']).sum() + (hi - lo))
    n_glob = global_token_count(n_loc)
    # NB: each shard is trimmed to ITS longest row by the model; keep_mask rows beyond it are unused
    smax = int((ids[lo:hi] != cfg['eos']).sum(1).max())
    dp.train_step(ids[lo:hi], ids[lo:hi], n_glob, float(len(ids)), keep_mask=keep[:smax, lo:hi], eps=eps[lo:hi])
    if rank == 0:
        torch.save({'grads': model.grads, 'params': model.params, 'n_glob': n_glob}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(tmp_path):
    out = str(tmp_path / 'r0.pt')
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    cfg, P, ids, keep, eps = _case()
    ref = OracleModel(cfg, P)
    ref.forward_backward(ids, ids, keep_mask=keep, eps=eps)
    assert got['n_glob'] == float((ids != cfg['eos']).sum() + len(ids))
    d = (got['grads'] - ref.grads).norm() / ref.grads.norm()
    assert float(d) < 1e-12, float(d)
    ref.adam_step()
    assert float((got['params'] - ref.params).abs().max()) < 1e-12


def test_shard_rows_and_reducer_bookkeeping():
    from argsim_amd.dist import shard_rows
    assert [shard_rows(8192, r, 8) for r in (0, 7)] == [(0, 1024), (7168, 8192)]
    with pytest.raises(AssertionError):
        shard_rows(10, 0, 4)


def test_rccl_path_puts_every_collective_into_the_side_streams_order(monkeypatch):
    """The RCCL branch of GradReducer (no GPU or RCCL here: streams, events and the collective are stand-ins that log what
    is asked of them).  A collective runs on the process group's own stream; Work.wait() only makes the CURRENT stream wait
    for it.  It must therefore be taken at once, inside the side-stream context -- then the fence before a persistent GRU
    launch (compute stream waits for the side stream) really is ordered after every collective in flight."""
    import torch.distributed as tdist
    from argsim_amd import dist as D
    log = []

    class Stream:
        def __init__(self, name):
            self.name = name

        def wait_event(self, ev):
            log.append((self.name, 'wait_event'))

        def wait_stream(self, other):
            log.append((self.name, 'wait_stream', other.name))

    comm, main = Stream('comm'), Stream('main')
    current = [main]

    class Ctx:
        def __init__(self, s):
            self.s = s

        def __enter__(self):
            self.prev, current[0] = current[0], self.s

        def __exit__(self, *a):
            current[0] = self.prev

    class Event:
        def record(self, s):
            log.append((s.name, 'record'))

    class Work:
        def wait(self):
            log.append((current[0].name, 'work.wait'))

    def all_reduce(*a, **k):
        assert k.get('async_op') is True
        log.append((current[0].name, 'all_reduce'))
        return Work()

    monkeypatch.setattr(torch.cuda, 'Event', Event)
    monkeypatch.setattr(torch.cuda, 'stream', Ctx)
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda dev=None: current[0])
    monkeypatch.setattr(tdist, 'get_backend', lambda g=None: 'nccl')
    monkeypatch.setattr(tdist, 'all_reduce', all_reduce)
    r = D.GradReducer(torch.zeros(8), [(0, 4), (4, 4)], None, comm)
    assert r.stream_ordered
    r.reduce_bucket(0)
    assert log == [('main', 'record'), ('comm', 'wait_event'), ('comm', 'all_reduce'), ('comm', 'work.wait')]
    assert not r.pending and current[0] is main
    log.clear()
    r.fence()
    assert log == [('main', 'wait_stream', 'comm')]
    log.clear()
    r.reduce_bucket(1)
    r.wait()
    assert log[-1] == ('main', 'wait_stream', 'comm') and ('comm', 'work.wait') in log
    # gloo (host-completed collectives) keeps its Work objects and waits for them in fence() / wait()
    monkeypatch.setattr(tdist, 'get_backend', lambda g=None: 'gloo')
    g = D.GradReducer(torch.zeros(8), [(0, 8)], None, comm)
    log.clear()
    g.reduce_bucket(0)
    assert not g.stream_ordered and len(g.pending) == 1 and ('comm', 'work.wait') not in log
    g.fence()
    assert not g.pending and ('main', 'work.wait') in log and log[-1] == ('main', 'wait_stream', 'comm')


# ------------------------------------------------------------------------------------------ round 3: host-side scalars
class _EvalStub:
    """model.eval of the training driver on the host: per-token error / CE and per-element KL that depend only on the ids"""
    def eval(self, src, tgt):
        n = int((tgt != 1).sum() + len(tgt))
        r = np.random.default_rng(int(tgt.sum()))
        return (r.random(n) < 0.3).astype(np.float32), r.random(n).astype(np.float32), r.random((len(tgt), 4)).astype(np.float32)


def _host_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    host = dist.new_group(backend='gloo')
    from argsim_amd.dist import shard_rows
    from argsim_amd.train import pipe, summ, with_global_counts
    rng = np.random.default_rng(5)
    glob = []
    for _ in range(6):
        ids = np.ones((8, 10), np.int32)
        for b in range(8):
            n = int(rng.integers(1, 11)); ids[b, :n] = rng.integers(3, 50, n)
        glob.append(ids)
    lo, hi = shard_rows(8, rank, world)
    shards = ((g[lo:hi], g[lo:hi]) for g in glob)
    # the count rides with the batch and is computed inside the prefetch thread (as train.main wires it)
    got = [n for _, _, n in pipe(with_global_counts(shards, 1, host), 4)]
    want = [float((g != 1).sum() + len(g)) for g in glob]
    valid = np.concatenate(glob)
    mine = summ(_EvalStub(), valid, 5, rank, world, host)
    if rank == 0:
        torch.save({'got': got, 'want': want, 'summ': mine}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_host_side_token_counts_and_sharded_validation(tmp_path):
    """train.py under data parallelism: N_global is one host-side (gloo) all-reduce per batch issued from the prefetch
    thread -- no device scalar is read back in the training loop -- and validation chunks are dealt over the ranks with
    the sums added across them: the same three means as one process over the whole array (src/train.py:104-113)."""
    out = str(tmp_path / 'h.pt')
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_host_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    assert got['got'] == got['want']
    from argsim_amd.train import summ
    rng = np.random.default_rng(5)
    glob = []
    for _ in range(6):
        ids = np.ones((8, 10), np.int32)
        for b in range(8):
            n = int(rng.integers(1, 11)); ids[b, :n] = rng.integers(3, 50, n)
        glob.append(ids)
    one = summ(_EvalStub(), np.concatenate(glob), 5)
    assert np.allclose(got['summ'], one, rtol=1e-6, atol=0)


def _two_thread_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    count_group, valid_group = dist.new_group(backend='gloo'), dist.new_group(backend='gloo')      # as train.main creates them
    import itertools
    import time
    from argsim_amd.dist import shard_rows
    from argsim_amd.train import pipe, summ, with_global_counts
    rng = np.random.default_rng(9)
    glob = []
    for _ in range(40):
        ids = np.ones((8, 10), np.int32)
        for b in range(8):
            n = int(rng.integers(1, 11)); ids[b, :n] = rng.integers(3, 50, n)
        glob.append(ids)
    lo, hi = shard_rows(8, rank, world)

    def shards():
        for g in glob:
            if rank == 1:
                time.sleep(0.01)          # one rank's tokeniser is slow: its prefetch thread issues every count late
            yield g[lo:hi], g[lo:hi]
    stream = pipe(with_global_counts(shards(), 1, count_group), 2)       # shallow queue: the prefetch thread keeps issuing all the while
    valid = np.concatenate(glob[:5])
    got, sums = [], []
    for block in range(4):
        got += [n for _, _, n in itertools.islice(stream, 10)]
        # rank 0 reaches the validation all-reduce at once, rank 1 late; both prefetch threads are mid-stream
        if rank == 1:
            time.sleep(0.05)
        sums.append(summ(_EvalStub(), valid, 5, rank, world, valid_group))
    want = [float((g != 1).sum() + len(g)) for g in glob]
    if rank == 0:
        torch.save({'got': got, 'want': want, 'sums': sums}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_counts_from_the_prefetch_thread_and_validation_sums_interleave(tmp_path):
    """with_global_counts all-reduces from the prefetch thread, summ from the main thread; on ONE gloo group
    their per-group issue order could differ between ranks (a rank that reaches summ before its prefetch thread has issued the
    next count) and an 8-byte all-reduce be paired with a 40-byte one.  train.main now gives each thread its own group; here the
    count stream stays live while summ runs four times, one rank delayed on both paths: every count and every validation mean
    must still be the single-process value."""
    out = str(tmp_path / 't.pt')
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_two_thread_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    assert got['got'] == got['want']
    from argsim_amd.train import summ
    rng = np.random.default_rng(9)
    glob = []
    for _ in range(40):
        ids = np.ones((8, 10), np.int32)
        for b in range(8):
            n = int(rng.integers(1, 11)); ids[b, :n] = rng.integers(3, 50, n)
        glob.append(ids)
    one = summ(_EvalStub(), np.concatenate(glob[:5]), 5)
    for s in got['sums']:
        assert np.allclose(s, one, rtol=1e-6, atol=0)
