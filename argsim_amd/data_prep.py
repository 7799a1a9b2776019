#!/usr/bin/env python3
"""corpus preparation (counterparts of reference src/data_ibm.py and src/data_iac.py).

Pure host code: both functions write what train.py reads -- ``vocab.model`` (SentencePiece, 8192 pieces,
unk=0 eos=1 bos=2), ``train.txt`` (one text per line) and ``valid.npy`` (int32, eos padded).

    python -m argsim_amd.data_prep ibm --src ../data/ibm_claim --out ../trial/data
    python -m argsim_amd.data_prep iac --src ../data/iac_v1.1/data/fourforums/discussions --val ../data/val.txt --out ../trial/data
"""
import argparse
import csv
import os

import numpy as np

from .util_io import clean, load_json, load_txt, pform, save_txt
from .util_np import vpack
from .util_sp import encode, encode_capped, spm

IBM_SPLITS = ("q_mc_heldout.csv", "q_mc_test.csv", "q_mc_train.csv", "test_set.csv")


def prep_ibm(path_csv, out, valid_size=4096, vocab_size=8192, splits=IBM_SPLITS):
    """data_ibm.py:18-48: column 3 of every split -> text file -> SentencePiece -> shuffle with the legacy
    numpy generator seeded 0 -> first ``valid_size`` sentences are the validation array, the rest train.txt"""
    def rows():
        for split in splits:
            r = csv.reader(load_txt(os.path.join(path_csv, split)))
            next(r)                                     # header
            for row in r:
                yield row[3]
    os.makedirs(out, exist_ok=True)
    path_txt = os.path.join(out, 'all.txt')
    save_txt(path_txt, rows())
    vocab = spm(os.path.join(out, 'vocab'), path_txt, size=vocab_size)
    sents = list(load_txt(path_txt))
    np.random.RandomState(0).shuffle(sents)             # == np.random.seed(0); np.random.shuffle(sents)
    save_txt(os.path.join(out, 'train.txt'), sents[valid_size:])
    np.save(os.path.join(out, 'valid.npy'), encode(vocab, sents[:valid_size]))
    return vocab


def prep_iac(path_raw, path_val, out, cap=512, vocab_size=8192):
    """data_iac.py:18-46: cleaned fourforums posts (field 3 of every post of every discussion json, sorted
    file order, empties dropped) -> SentencePiece -> every post capped to ``cap`` pieces at a sentence
    boundary and written back decoded -> validation posts capped and packed to (n, cap)"""
    posts = tuple(clean(post[3]) for fn in sorted(os.listdir(path_raw)) for post in load_json(pform(path_raw, fn))[0])
    posts = tuple(p for p in posts if 0 < len(p))
    os.makedirs(out, exist_ok=True)
    path_txt = os.path.join(out, 'all.txt')
    save_txt(path_txt, posts)
    vocab = spm(os.path.join(out, 'vocab'), path_txt, size=vocab_size)
    ids = [encode_capped(vocab, p, cap=cap) for p in posts]
    save_txt(os.path.join(out, 'train.txt'), map(vocab.decode_ids, ids))
    val = [encode_capped(vocab, clean(p), cap=cap) for p in load_txt(path_val)]
    np.save(os.path.join(out, 'valid.npy'), vpack(val, (len(val), cap), vocab.eos_id(), np.int32))
    return vocab


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('corpus', choices=('ibm', 'iac'))
    ap.add_argument('--src', required=True)
    ap.add_argument('--val', default=None)
    ap.add_argument('--out', required=True)
    ap.add_argument('--vocab-size', type=int, default=8192)
    A = ap.parse_args(argv)
    if A.corpus == 'ibm':
        prep_ibm(A.src, A.out, vocab_size=A.vocab_size)
    else:
        prep_iac(A.src, A.val, A.out, vocab_size=A.vocab_size)


if __name__ == '__main__':
    main()
