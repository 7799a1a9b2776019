#!/usr/bin/env python3
"""training driver (counterpart of reference src/train.py) on PyTorch-ROCm + libargsim_vae.so.

Same flags, same config.json schema (paths / model / train sections), same loop: 250 steps ->
validate over the whole validation array in chunks of ``batch_valid`` -> every 40 x 250 = 10 000
steps save a checkpoint named ``<trial><step // 10000>`` (src/train.py:115-121).

Differences: scalars go to ``<log>/<trial>.jsonl`` (names step_errt / step_loss_gen / step_loss_kld
as in src/train.py:98-102) instead of TensorBoard; ``--profile`` traces one warm run of the validation
loss on valid[:32] (src/train.py:76-82) in a child process under rocprofv3 and leaves the per-kernel
summary in ``<log>/<trial>_profile/``; with torchrun (WORLD_SIZE > 1) the batch is sharded over
data-parallel replicas (argsim_amd/dist.py), every rank tokenising only its own rows.
"""
import argparse
import json
import os
import queue
import sys
import threading
import time


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="trains a variational autoencoder on text (MI355X build).",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--trial', default="master", help="the trial name")
    p.add_argument('--config', default="config.json", help="the config file")
    p.add_argument('--ckpt', default=None, help="the checkpoint to resume")
    p.add_argument('--gpu', default="0", help="the gpu to use (single process)")
    p.add_argument('--seed', default=0, type=int, help="random seed")
    p.add_argument('--rounds', default=0, type=int, help="numbers of training rounds")
    p.add_argument('--prefetch', default=16, type=int, help="numbers of batches to prefetch")
    p.add_argument('--sample', action='store_true', help="train with sentencepiece sampling")
    p.add_argument('--profile', action='store_true', help="time the validation forward pass")
    # extensions
    p.add_argument('--steps-per-round', default=10000, type=int, help="reference: 40 x 250")
    p.add_argument('--valid-every', default=250, type=int)
    p.add_argument('--kl-beta', default=1.0, type=float, help="multiplies the tanh KL anneal (1 = reference)")
    p.add_argument('--free-bits', default=0.0, type=float, help="per-dimension KL floor (0 = reference)")
    p.add_argument('--save-tf', action='store_true', help="also write every checkpoint as TF V2 checkpoint files "
                   "(<ckpt>/<trial><round>.index/.data-*), the format the reference's eval scripts restore; CudnnGRU "
                   "names are unverified against a real TensorFlow (ckpt.py)")
    p.add_argument('--profile-run', action='store_true', help=argparse.SUPPRESS)    # the child that --profile traces
    return p.parse_args(argv)


def batch(size, path, vocab, seed, kudo, max_len, rank=0, world=1):
    """endless stream of (src, tgt) int32 arrays, rows padded with eos to the longest row (src/train.py:54-68):
    the corpus lines are visited in the order of util_np.sample and cut into consecutive groups of ``size``.
    Data parallel: every rank walks the same index stream but tokenises only its own rows
    [rank * size / world, (rank + 1) * size / world) of each group."""
    import itertools
    import numpy as np
    from .dist import shard_rows
    from .util_io import load_txt
    from .util_np import sample, vpack
    from .util_sp import encode_capped, encode_capped_sample_pair
    lines = tuple(load_txt(path))
    lo, hi = shard_rows(size, rank, world)
    pad = vocab.eos_id()

    def pack(rows):
        return vpack(rows, (len(rows), max(len(r) for r in rows)), pad, np.int32)

    picks = sample(len(lines), seed)
    while True:
        mine = list(itertools.islice(picks, size))[lo:hi]
        if kudo:      # two independent sampled segmentations of every line (--sample)
            pairs = [encode_capped_sample_pair(vocab, lines[i], cap=max_len) for i in mine]
            yield pack([a for a, _ in pairs]), pack([b for _, b in pairs])
        else:
            ids = pack([encode_capped(vocab, lines[i], cap=max_len) for i in mine])
            yield ids, ids


def pinned(gen):
    """host batches -> page-locked torch tensors, so the H2D copy of VAE._ids is asynchronous (the counterpart of the
    tf.data iterator handing device-ready tensors to the graph, util_tf.py:16-23)"""
    import torch
    for src, tgt, *rest in gen:        # (rest: the global token count of a data-parallel batch, passed through)
        a = torch.from_numpy(src).pin_memory()
        yield (a, (a if tgt is src else torch.from_numpy(tgt).pin_memory()), *rest)


def pipe(gen, prefetch=1):
    """background-thread prefetch of a generator (counterpart of util_tf.pipe, src/util_tf.py:16-23:
    Dataset.from_generator(...).repeat().prefetch(n)).  Yields whatever ``gen`` yields."""
    q = queue.Queue(maxsize=max(1, prefetch))
    stop = object()

    def work():
        try:
            for item in gen:
                q.put(item)
        finally:
            q.put(stop)

    threading.Thread(target=work, daemon=True).start()
    while True:
        item = q.get()
        if item is stop:
            return
        yield item


def summ(model, valid, batch_valid, rank=0, world=1, group=None):
    """means over ALL validation tokens / latent elements (src/train.py:104-113).
    Data parallel: the chunks of ``partition`` are dealt round-robin over the ranks and the six sums (errors, CE, tokens,
    KL, latent elements) are added across them on the host-side group, so every rank validates 1/world of the array and
    nobody waits in the next step's all-reduce while rank 0 walks the whole of it."""
    import numpy as np
    from .util_np import partition
    chunks = list(partition(len(valid), batch_valid, discard=False))[rank::world]
    parts = [model.eval(valid[i:j], valid[i:j]) for i, j in chunks]
    if world == 1:
        errt, lgen, lkld = (np.concatenate([p[k].ravel() for p in parts]) for k in range(3))
        return float(errt.mean()), float(lgen.mean()), float(lkld.mean())
    import torch
    import torch.distributed as dist
    t = torch.tensor([sum(float(p[0].sum(dtype=np.float64)) for p in parts), sum(float(p[1].sum(dtype=np.float64)) for p in parts),
                      float(sum(p[0].size for p in parts)), sum(float(p[2].sum(dtype=np.float64)) for p in parts),
                      float(sum(p[2].size for p in parts))], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    e, g, n, k, m = (float(x) for x in t)
    return e / n, g / n, k / m


def with_global_counts(gen, eos, group):
    """(src, tgt) -> (src, tgt, N_global): the tokens of the GLOBAL batch, sum over ranks of sum_b (len_b + 1)
    (model.py:181 is a mean over all of them).  One scalar all-reduce per batch on a HOST-side (gloo) group, issued from the
    prefetch thread that runs this generator: the training loop never blocks on it and never reads a device scalar back
    (a device all-reduce + .item() per step stalled the host behind the previous step's GPU work)."""
    import torch
    import torch.distributed as dist
    for src, tgt in gen:
        t = torch.tensor([float(int((tgt != eos).sum()) + len(tgt))], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        yield src, tgt, float(t[0])


def profile(A, argv):
    """--profile (src/train.py:76-82): the reference writes ONE fully traced run of the validation loss on
    valid[:32] to TensorBoard.  Here the same run happens in a child process under ``rocprofv3 --kernel-trace
    --stats`` (started before this process touches the GPU); the per-kernel summary lands in
    <log>/<trial>_profile/ and its top rows are printed.  Without rocprofv3 on PATH the child runs bare and only
    its HIP-event class timings are printed."""
    import glob
    import shutil
    import subprocess
    from .util_io import load_json, pform
    out = pform(load_json(A.config)['paths']['log'], A.trial, '_profile')
    os.makedirs(out, exist_ok=True)
    child = [sys.executable, '-m', 'argsim_amd.train', '--profile-run', '--trial', A.trial, '--config', A.config,
             '--gpu', str(A.gpu), '--seed', str(A.seed)] + (['--ckpt', A.ckpt] if A.ckpt else [])
    tool = shutil.which('rocprofv3')
    cmd = ([tool, '--kernel-trace', '--stats', '--output-format', 'csv', '-d', out, '--'] if tool else []) + child
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    text = res.stdout.decode(errors='replace')
    print('\n'.join(l for l in text.splitlines() if l.startswith('{')))
    if res.returncode:
        sys.exit("profile run failed:\n" + text[-2000:])
    for path in sorted(glob.glob(os.path.join(out, '**', '*kernel_stats.csv'), recursive=True))[-1:]:
        print("kernel summary:", path)
        with open(path) as f:
            for i, line in enumerate(f):
                if i > 12:
                    break
                print("   ", line.rstrip()[:160])


def main(argv=None):
    A = parse_args(argv)
    if not A.rounds and not A.profile and not A.profile_run:
        sys.exit("nothing to do")
    if A.profile and not A.profile_run:
        profile(A, argv)
        if not A.rounds:
            sys.exit("profiling done")
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', A.gpu if world == 1 else '0'))

    import numpy as np
    import torch
    from . import ckpt
    from .model import VAE
    from types import SimpleNamespace
    from .util_io import load_json, pform
    from .util_sp import load_spm

    config = load_json(A.config)
    P, T, Cm = SimpleNamespace(**config['paths']), SimpleNamespace(**config['train']), dict(config['model'])
    vocab = load_spm(P.vocab)
    valid = np.load(P.valid)
    torch.cuda.set_device(local)
    model = VAE('train', device=local, seed=A.seed, kl_beta=A.kl_beta, free_bits=A.free_bits, **Cm)

    if A.profile_run:
        # the traced child of --profile (src/train.py:76-82 via util_tf.profile, src/util_tf.py:9-13): three warm-up
        # runs, then ONE run of the validation loss on valid[:32] with per-class HIP-event stamps
        x = valid[:32]
        for _ in range(3):
            model.eval(x, x)
        model.set_option('timing', 1)
        t0 = time.perf_counter()
        model.eval(x, x)
        wall = 1e3 * (time.perf_counter() - t0)
        tm = model.timing_collect()
        print(json.dumps(dict(valid32_forward_ms=wall, classes={k: dict(ms=v[0], launches=v[1], tflops=(v[2] / (v[0] * 1e-3) / 1e12 if v[0] else 0.0))
                                                                  for k, v in tm.items()})))
        return
    if not A.rounds:
        sys.exit("profiling done")

    dp = count_group = valid_group = None
    if world > 1:
        import torch.distributed as dist
        from .dist import DataParallel
        dist.init_process_group('nccl')
        # host-side scalars never go on the GPU's queue.  ONE gloo group PER ISSUING THREAD: gloo pairs collectives by per-group
        # issue order, the token counts are issued from the prefetch thread and the validation sums from the main thread, and
        # nothing orders those two threads the same way on every rank (a rank with fewer validation chunks reaches summ's
        # all-reduce while its prefetch thread has not yet issued the next count) -- on one group an 8-byte all-reduce could be
        # paired with a 40-byte one
        count_group = dist.new_group(backend='gloo')     # prefetch thread: with_global_counts
        valid_group = dist.new_group(backend='gloo')     # main thread: summ
        dp = DataParallel(model)
    if A.ckpt:
        ckpt.restore(model, pform(P.ckpt, A.ckpt))
    if dp:
        dp.broadcast_params(model.state)

    eos = vocab.eos_id()
    batches = batch(T.batch_train, P.train, vocab, A.seed, A.sample, T.max_len, rank, world)
    if dp:      # the global token count rides with the batch (one host-side all-reduce inside the prefetch thread)
        batches = with_global_counts(batches, eos, count_group)
    stream = pipe(pinned(batches), A.prefetch)
    os.makedirs(P.log, exist_ok=True)
    os.makedirs(P.ckpt, exist_ok=True)
    log = open(pform(P.log, A.trial, '.jsonl'), 'a') if rank == 0 else None
    for _ in range(A.rounds):
        for _ in range(A.steps_per_round // A.valid_every):
            t0 = time.perf_counter()
            for _ in range(A.valid_every):
                if dp:      # this rank's shard; the ELBO means are over the GLOBAL token / row counts (model.py:181,184)
                    src, tgt, n_glob = next(stream)
                    dp.train_step(src, tgt, n_glob, float(T.batch_train))
                else:
                    src, tgt = next(stream)
                    model.train_step(src, tgt)
            lg, lk, lo = model.losses()
            dt = time.perf_counter() - t0
            step = model.step
            errt, vgen, vkld = summ(model, valid, T.batch_valid, rank, world, valid_group)      # every rank: its share of the chunks
            if rank == 0:
                rec = dict(step=step, step_errt=errt, step_loss_gen=vgen, step_loss_kld=vkld,
                           train_loss_gen=lg, train_loss_kld=lk, sentences_per_sec=A.valid_every * T.batch_train / dt)
                log.write(json.dumps(rec) + "\n")
                log.flush()
                print(rec)
        if rank == 0:
            ckpt.save(model, pform(P.ckpt, A.trial, model.step // 10000))
            if A.save_tf:
                ckpt.save_tf(model, pform(P.ckpt, A.trial, model.step // 10000))


if __name__ == '__main__':
    main()
