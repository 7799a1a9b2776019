#!/usr/bin/env python3
"""sentence / post embeddings for the downstream analysis scripts (counterpart of reference
src/eval_embed_reason.py and src/eval_embed_stance.py).

Writes the two arrays those scripts write, with the same shapes and dtype, so the reference's
untouched eval_classification*.py / eval_clustering*.py consume them:
    <out>.npy          (n, dim_rep) float32   z = mu of the deterministic segmentation, batches of 128
    <out>_sample.npy   (n, dim_rep) float32   mean z over `--samples` sampled segmentations (infer_avg)

    python -m argsim_amd.eval_embed --ckpt trial/ckpt/kudo396 --vocab trial/data/vocab.model \
           --data data/test_data.npz --key posts --out data/test_data_emb --config config.json
"""
import argparse

import numpy as np


def embed(model, vocab, text, batch=128):
    """eval_embed_reason.py:33-41: encode_capped -> vpack -> z per partition of 128 rows"""
    from . import util_sp as sp
    from .util_np import partition, vpack
    data = [sp.encode_capped(vocab, t) for t in text]
    data = vpack(data, (len(data), max(map(len, data))), vocab.eos_id(), np.int32)
    return np.concatenate([model.encode(data[i:j]) for i, j in partition(len(data), batch)], axis=0)


def infer_avg(model, vocab, sent, samples=128):
    """eval_embed_reason.py:47-51: mean z over `samples` sampled segmentations of one text"""
    from . import util_sp as sp
    from .util_np import vpack
    bat = [sp.encode_capped_sample(vocab, sent) for _ in range(samples)]
    bat = vpack(bat, (len(bat), max(map(len, bat))), vocab.eos_id(), np.int32)
    return model.encode(bat).mean(axis=0)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--ckpt', required=True)
    ap.add_argument('--vocab', required=True)
    ap.add_argument('--data', required=True, help=".npz with an array of strings, or a text file (one per line)")
    ap.add_argument('--key', default='posts')
    ap.add_argument('--out', required=True)
    ap.add_argument('--config', default='config.json')
    ap.add_argument('--samples', type=int, default=128)
    ap.add_argument('--no-sampled', action='store_true')
    A = ap.parse_args(argv)
    from . import ckpt
    from .model import VAE
    from .util_io import load_json, load_txt
    from .util_sp import load_spm
    text = list(np.load(A.data)[A.key]) if A.data.endswith('.npz') else list(load_txt(A.data))
    vocab = load_spm(A.vocab)
    model = VAE('infer', init=False, **load_json(A.config)['model'])
    ckpt.restore(model, A.ckpt, strict=True)
    np.save(A.out + '.npy', embed(model, vocab, text))
    if not A.no_sampled:
        np.save(A.out + '_sample.npy', np.stack([infer_avg(model, vocab, t, A.samples) for t in text], axis=0))


if __name__ == '__main__':
    main()
