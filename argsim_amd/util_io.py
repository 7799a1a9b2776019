"""file helpers (counterpart of reference src/util_io.py); no pickle: checkpoints are .npz."""
import json
from os.path import expanduser, join


def pform(path, *names, sep=''):
    """``path`` joined with the concatenation of ``names`` (src/util_io.py:7-9)"""
    return join(expanduser(path), sep.join(map(str, names)))


def load_txt(filename, encoding=None):
    """yields lines without their newline (src/util_io.py:12-15: drops the LAST CHARACTER of each
    line, so a final line without a newline loses a character -- kept for parity)"""
    with open(filename, encoding=encoding) as f:
        for line in f:
            yield line[:-1]


def save_txt(filename, lines):
    with open(filename, 'w') as f:
        for line in lines:
            print(line, file=f)


def load_json(filename):
    with open(filename) as f:
        return json.load(f)


def clean(post):
    """whitespace-normalised, lower-cased (src/util_io.py:41-48)"""
    return " ".join(post.split()).lower()
