"""checkpoint save / restore (counterpart of tf.train.Saver in reference src/train.py:92-96,121 and
src/eval_embed_reason.py:24-27).

Native format: one ``.npz`` keyed by variable name, holding every model variable in its natural
(checkpoint) layout, ``global_step`` and, for training checkpoints, the Adam slots under
``train/<name>/Adam`` and ``train/<name>/Adam_1`` (TF's slot naming, model.py:188-189).  Restore is
name-based and partial like ``Saver().restore`` on an 'infer' graph: missing slots are left
untouched.

``to_tf_names`` / ``from_tf_names`` convert GRU variables to and from the canonical tensors that
TF 1.x's CudnnGRU saveable writes (names quoted from memory of
tf.contrib.cudnn_rnn, NOT verifiable offline):
    <scope>/cudnn_gru/rnn/multi_rnn_cell/cell_<k>/cudnn_compatible_gru_cell/
        gates/kernel (In+D, 2D) [r | u]      gates/bias (2D) = bW + bR
        candidate/input_projection/{kernel (In,D), bias (D)}
        candidate/hidden_projection/{kernel (D,D), bias (D)}
Reading/writing the TF bundle files themselves is SURVEY 8(f) item 4 (next)."""
import re

import numpy as np

from .model import ADAM_M, ADAM_V, PARAM


def state_dict(vae, slots=True):
    out = {'global_step': np.asarray(vae.step, np.int64)}
    for k, v in vae.get_params(PARAM).items():
        out[k] = v
    if slots:
        for k, v in vae.get_params(ADAM_M).items():
            out['train/%s/Adam' % k] = v
        for k, v in vae.get_params(ADAM_V).items():
            out['train/%s/Adam_1' % k] = v
    return out


def load_state_dict(vae, sd, strict=True):
    missing = [k for k in vae.names if k not in sd]
    if missing and strict:
        raise KeyError("checkpoint lacks variables: %s" % missing[:5])
    for k in vae.names:
        if k in sd:
            vae.set_tensor(k, sd[k], PARAM)
        if 'train/%s/Adam' % k in sd:
            vae.set_tensor(k, sd['train/%s/Adam' % k], ADAM_M)
        if 'train/%s/Adam_1' % k in sd:
            vae.set_tensor(k, sd['train/%s/Adam_1' % k], ADAM_V)
    if 'global_step' in sd:
        vae.step = int(sd['global_step'])


def save(vae, path, slots=True):
    """saver.save(sess, path, write_meta_graph=False)  (src/train.py:121)"""
    if not path.endswith('.npz'):
        path += '.npz'
    np.savez(path, **state_dict(vae, slots))
    return path


def restore(vae, path, strict=True):
    """saver.restore(sess, path)  (src/train.py:93-94)"""
    if not path.endswith('.npz'):
        path += '.npz'
    with np.load(path, allow_pickle=False) as f:
        load_state_dict(vae, {k: f[k] for k in f.files}, strict)


# ------------------------------------------------------------------ TF canonical names
_GRU = re.compile(r'^(encode/rnn\d+/(?:fwd|bwd)|decode/rnn/l(\d+))/(W|R|bW|bR)$')


def _tf_scope(name):
    m = _GRU.match(name)
    if m.group(2) is None:          # encoder: one single-layer CudnnGRU per direction (model.py:120-121)
        return m.group(1), 0
    return 'decode/rnn', int(m.group(2)) - 1     # decoder: one L-layer CudnnGRU (model.py:160)


def to_tf_names(sd):
    """native state dict -> dict keyed like a TF1 checkpoint of the reference graph"""
    out, groups = {}, {}
    for k, v in sd.items():
        if _GRU.match(k):
            scope, cell = _tf_scope(k)
            groups.setdefault((scope, cell), {})[k.rsplit('/', 1)[1]] = np.asarray(v)
        else:
            out[k] = v
    for (scope, cell), g in groups.items():
        D = g['R'].shape[1]
        W, R, bW, bR = g['W'], g['R'], g['bW'], g['bR']
        p = '%s/cudnn_gru/rnn/multi_rnn_cell/cell_%d/cudnn_compatible_gru_cell/' % (scope, cell)
        out[p + 'gates/kernel'] = np.concatenate([W[:2 * D].T, R[:2 * D].T], axis=0)
        out[p + 'gates/bias'] = bW[:2 * D] + bR[:2 * D]
        out[p + 'candidate/input_projection/kernel'] = W[2 * D:].T
        out[p + 'candidate/input_projection/bias'] = bW[2 * D:]
        out[p + 'candidate/hidden_projection/kernel'] = R[2 * D:].T
        out[p + 'candidate/hidden_projection/bias'] = bR[2 * D:]
    return out


def from_tf_names(sd, names):
    """inverse of to_tf_names for the variables in ``names``.  The r/u biases are stored summed in
    TF's canonical form; they are restored into bW with bR = 0 (the same function)."""
    out = {k: v for k, v in sd.items() if 'cudnn_compatible_gru_cell' not in k}
    for k in names:
        if not _GRU.match(k) or not k.endswith('/W'):
            continue
        scope, cell = _tf_scope(k)
        p = '%s/cudnn_gru/rnn/multi_rnn_cell/cell_%d/cudnn_compatible_gru_cell/' % (scope, cell)
        gk, gb = np.asarray(sd[p + 'gates/kernel']), np.asarray(sd[p + 'gates/bias'])
        D = gk.shape[1] // 2
        In = gk.shape[0] - D
        base = k[:-1]
        out[base + 'W'] = np.concatenate([gk[:In].T, np.asarray(sd[p + 'candidate/input_projection/kernel']).T], 0)
        out[base + 'R'] = np.concatenate([gk[In:].T, np.asarray(sd[p + 'candidate/hidden_projection/kernel']).T], 0)
        out[base + 'bW'] = np.concatenate([gb, np.asarray(sd[p + 'candidate/input_projection/bias'])])
        out[base + 'bR'] = np.concatenate([np.zeros_like(gb), np.asarray(sd[p + 'candidate/hidden_projection/bias'])])
    return out
