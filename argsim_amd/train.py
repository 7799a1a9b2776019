#!/usr/bin/env python3
"""training driver (counterpart of reference src/train.py) on PyTorch-ROCm + libargsim_vae.so.

Same flags, same config.json schema (paths / model / train sections), same loop: 250 steps ->
validate over the whole validation array in chunks of ``batch_valid`` -> every 40 x 250 = 10 000
steps save a checkpoint named ``<trial><step // 10000>`` (src/train.py:115-121).

Differences: scalars go to ``<log>/<trial>.jsonl`` (names step_errt / step_loss_gen / step_loss_kld
as in src/train.py:98-102) instead of TensorBoard; ``--profile`` times warm forward passes on
valid[:32] (src/train.py:76-82) and tells you the rocprofv3 command for a kernel trace; with
torchrun (WORLD_SIZE > 1) the batch is sharded over data-parallel replicas (argsim_amd/dist.py).
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
    return p.parse_args(argv)


def batch(size, path, vocab, seed, kudo, max_len):
    """endless (src, tgt) stream of eos-packed int32 batches (src/train.py:54-68)."""
    import numpy as np
    from .util_io import load_txt
    from .util_np import sample, vpack
    from .util_sp import encode_capped, encode_capped_sample_pair
    eos = vocab.eos_id()
    pac = lambda arrs: vpack(arrs, (size, max(map(len, arrs))), eos, np.int32)  # noqa: E731
    enc = encode_capped_sample_pair if kudo else encode_capped
    raw = tuple(load_txt(path))
    bat = []
    for i in sample(len(raw), seed):
        if size == len(bat):
            if kudo:
                src, tgt = map(pac, zip(*bat))
            else:
                src = tgt = pac(bat)
            yield src, tgt
            bat = []
        bat.append(enc(vocab, raw[i], cap=max_len))


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


def summ(model, valid, batch_valid):
    """means over ALL validation tokens / latent elements (src/train.py:104-113)"""
    import numpy as np
    from .util_np import partition
    parts = [model.eval(valid[i:j], valid[i:j]) for i, j in partition(len(valid), batch_valid, discard=False)]
    errt, lgen, lkld = (np.concatenate([p[k].ravel() for p in parts]) for k in range(3))
    return float(errt.mean()), float(lgen.mean()), float(lkld.mean())


def main(argv=None):
    A = parse_args(argv)
    if not A.rounds and not A.profile:
        sys.exit("nothing to do")
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', A.gpu if world == 1 else '0'))

    import numpy as np
    import torch
    from . import ckpt
    from .model import VAE
    from .util import Record
    from .util_io import load_json, pform
    from .util_sp import load_spm

    config = load_json(A.config)
    P, Cm, T = Record(config['paths']), Record(config['model']), Record(config['train'])
    vocab = load_spm(P.vocab)
    valid = np.load(P.valid)
    torch.cuda.set_device(local)
    model = VAE('train', device=local, seed=A.seed, kl_beta=A.kl_beta, free_bits=A.free_bits, **Cm)

    if A.profile:
        x = valid[:32]
        for _ in range(3):
            model.eval(x, x)
        t0 = time.perf_counter()
        model.eval(x, x)
        print("valid[:32] forward: %.3f ms;  kernel trace: rocprofv3 --kernel-trace --stats -- python -m argsim_amd.train --profile ..."
              % (1e3 * (time.perf_counter() - t0)))
    if not A.rounds:
        sys.exit("profiling done")

    dp = None
    if world > 1:
        import torch.distributed as dist
        from .dist import DataParallel
        dist.init_process_group('nccl')
        dp = DataParallel(model)
    if A.ckpt:
        ckpt.restore(model, pform(P.ckpt, A.ckpt))
    if dp:
        dp.broadcast_params(model.state)

    assert T.batch_train % world == 0
    per = T.batch_train // world
    stream = pipe(batch(T.batch_train, P.train, vocab, A.seed, A.sample, T.max_len), A.prefetch)
    os.makedirs(P.log, exist_ok=True)
    os.makedirs(P.ckpt, exist_ok=True)
    log = open(pform(P.log, A.trial, '.jsonl'), 'a') if rank == 0 else None
    eos = vocab.eos_id()
    for _ in range(A.rounds):
        for _ in range(A.steps_per_round // A.valid_every):
            t0 = time.perf_counter()
            for _ in range(A.valid_every):
                src, tgt = next(stream)
                if dp:
                    n_glob = float((tgt != eos).sum() + len(tgt))
                    sl = slice(rank * per, (rank + 1) * per)
                    dp.train_step(src[sl], tgt[sl], n_glob, float(len(tgt)))
                else:
                    model.train_step(src, tgt)
            lg, lk, lo = model.losses()
            dt = time.perf_counter() - t0
            step = model.step
            if rank == 0:
                errt, vgen, vkld = summ(model, valid, T.batch_valid)
                rec = dict(step=step, step_errt=errt, step_loss_gen=vgen, step_loss_kld=vkld,
                           train_loss_gen=lg, train_loss_kld=lk, sentences_per_sec=A.valid_every * T.batch_train / dt)
                log.write(json.dumps(rec) + "\n")
                log.flush()
                print(rec)
        if rank == 0:
            ckpt.save(model, pform(P.ckpt, A.trial, model.step // 10000))


if __name__ == '__main__':
    main()
