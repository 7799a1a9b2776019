"""torch-CPU autograd twin of ``oracle/vae_numpy.py`` (TEST INFRASTRUCTURE ONLY).

Second, independent restatement of ``/root/reference/src/model.py:75-189`` used
  * as the gradient oracle (float64 autograd) for the hand-written HIP backward,
  * as the timed "CPU restatement (PyTorch), not TensorFlow" baseline in bench.py
    (float32, ``kind: "port"``) -- the reference graph itself cannot run on any CPU
    (CudnnGRU is GPU-only, TensorFlow is absent).
Same parameter names as ``vae_numpy.param_shapes``.
"""
import numpy as np
import torch


def to_torch(P, dtype=torch.float64, requires_grad=True):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad) for k, v in P.items()}


def _gru(x, W, R, bW, bR, h0=None):
    """cuDNN reset-after GRU; input projection hoisted out of the time loop."""
    S, B, _ = x.shape
    D = R.shape[1]
    gi_all = torch.matmul(x, W.t()) + bW
    h = x.new_zeros((B, D)) if h0 is None else h0
    hs = []
    for t in range(S):
        gi = gi_all[t]
        gh = torch.matmul(h, R.t()) + bR
        r = torch.sigmoid(gi[:, :D] + gh[:, :D])
        u = torch.sigmoid(gi[:, D:2 * D] + gh[:, D:2 * D])
        n = torch.tanh(gi[:, 2 * D:] + r * gh[:, 2 * D:])
        h = (1.0 - u) * n + u * h
        hs.append(h)
    return torch.stack(hs), h


def _rev_index(S, lens):
    """index map of tf.reverse_sequence: pi_b(p) = len_b-1-p if p < len_b else p  -> (S,B) long"""
    p = torch.arange(S).unsqueeze(1)
    l = torch.as_tensor(lens, dtype=torch.long).unsqueeze(0)
    return torch.where(p < l, l - 1 - p, p.expand(S, l.shape[1]))


def _reverse_sequence(x, idx):
    return torch.gather(x, 0, idx.unsqueeze(-1).expand_as(x))


def forward(P, cfg, src, tgt, mode='train', step=0, keep_mask=None, eps=None, kl_beta=1.0, free_bits=0.0):
    """P: dict of torch tensors.  Other arguments as ``vae_numpy.forward``.
    returns dict with loss, loss_gen, loss_kld, mu, lv, z, pred, per-sample arrays (torch tensors)."""
    D, L, eos, bos = cfg['dim_emb'], cfg['rnn_layers'], cfg['eos'], cfg['bos']
    dt = P['embed/embedding'].dtype
    rate = cfg['accelerate'] * float(step)
    anneal = float(np.tanh(rate))
    src_tm = np.asarray(src).T
    tgt_tm = np.asarray(tgt).T
    len_src = (src_tm != eos).sum(0)
    len_tgt = (tgt_tm != eos).sum(0)
    src_tm = src_tm[:int(len_src.max())]
    tgt_tm = tgt_tm[:int(len_tgt.max())]
    S, B = src_tm.shape
    not_eos = tgt_tm != eos
    msk_tgt = np.concatenate([np.ones((1, B), bool), not_eos], 0)
    gold = np.concatenate([tgt_tm, np.full((1, B), eos, tgt_tm.dtype)], 0)
    lead = tgt_tm.copy()
    if mode == 'train':
        lead = lead * np.asarray(keep_mask).astype(lead.dtype)
    lead = np.concatenate([np.full((1, B), bos, lead.dtype), lead], 0)
    E = P['embed/embedding']
    emb_tgt = E[torch.as_tensor(lead, dtype=torch.long)]
    x = E[torch.as_tensor(src_tm, dtype=torch.long)]
    ridx = _rev_index(S, len_src)
    for i in range(1, L + 1):
        pf, pb = 'encode/rnn%d/fwd/' % i, 'encode/rnn%d/bwd/' % i
        fwd, _ = _gru(x, P[pf + 'W'], P[pf + 'R'], P[pf + 'bW'], P[pf + 'bR'])
        bwd, _ = _gru(_reverse_sequence(x, ridx), P[pb + 'W'], P[pb + 'R'], P[pb + 'bW'], P[pb + 'bR'])
        x = torch.cat([fwd, _reverse_sequence(bwd, ridx)], -1)
    h = x[torch.as_tensor(len_src - 1, dtype=torch.long), torch.arange(B)]
    mu = h @ P['latent/mu/kernel'] + P['latent/mu/bias']
    lv = h @ P['latent/lv/kernel'] + P['latent/lv/bias']
    z = mu
    if mode == 'train':
        z = z + torch.exp(0.5 * lv) * torch.as_tensor(np.asarray(eps), dtype=dt)
    h0 = z @ P['latent/ex/kernel'] + P['latent/ex/bias']
    xd = emb_tgt
    state_ex = []
    for i in range(1, L + 1):
        p = 'decode/rnn/l%d/' % i
        xd, hl = _gru(xd, P[p + 'W'], P[p + 'R'], P[p + 'bW'], P[p + 'bR'], h0)
        state_ex.append(hl)
    o = dict(mu=mu, lv=lv, z=z, state_ex=torch.stack(state_ex))
    m = torch.as_tensor(msk_tgt)
    hd = xd[m] if mode != 'infer' else xd.reshape(-1, D)
    hd = hd @ P['decode/out/kernel'] + P['decode/out/bias']
    logits = hd @ ((D ** -0.5) * E.t())
    o['logits'] = logits
    o['pred'] = logits.argmax(-1)
    if mode != 'infer':
        labels = torch.as_tensor(gold[msk_tgt], dtype=torch.long)
        o['loss_gen_samp'] = torch.nn.functional.cross_entropy(logits, labels, reduction='none')
        o['loss_gen'] = o['loss_gen_samp'].mean()
        o['loss_kld_samp'] = 0.5 * (mu * mu + torch.exp(lv) - lv - 1.0)
        # extensions of BASELINE configs[4] (not in the reference): per-dimension KL floor and a beta factor;
        # at (1, 0) this is exactly model.py:183-185
        o['loss_kld'] = torch.clamp(o['loss_kld_samp'], min=free_bits).mean() if free_bits > 0 else o['loss_kld_samp'].mean()
        o['loss'] = kl_beta * anneal * o['loss_kld'] + o['loss_gen']
        o['errt_samp'] = (labels != o['pred']).to(dt)
    return o


def loss_and_grads(P_np, cfg, src, tgt, step, keep_mask, eps, dtype=torch.float64, kl_beta=1.0, free_bits=0.0):
    """float64 gradient oracle: returns (outputs dict of numpy, grads dict of numpy)."""
    P = to_torch(P_np, dtype)
    o = forward(P, cfg, src, tgt, 'train', step, keep_mask, eps, kl_beta, free_bits)
    o['loss'].backward()
    grads = {k: (v.grad.numpy().copy() if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in P.items()}
    outs = {k: v.detach().numpy() for k, v in o.items()}
    return outs, grads


class CpuTrainer:
    """fwd + bwd + TF-style Adam in float32 on the host cores: the timed CPU baseline."""

    def __init__(self, P_np, cfg):
        self.cfg = cfg
        self.P = to_torch(P_np, torch.float32)
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.step = 0

    def train_step(self, src, tgt, keep_mask, eps):
        cfg = self.cfg
        for p in self.P.values():
            p.grad = None
        o = forward(self.P, cfg, src, tgt, 'train', self.step, keep_mask, eps)
        o['loss'].backward()
        rate = cfg['accelerate'] * float(self.step)
        lr = cfg['learn_rate'] / (np.sqrt(rate) + 1.0)
        t = self.step + 1
        b1, b2, e = 0.9, 0.999, 1e-8
        lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        with torch.no_grad():
            for k, p in self.P.items():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                self.m[k].mul_(b1).add_(g, alpha=1.0 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
                p.addcdiv_(self.m[k], self.v[k].sqrt().add_(e), value=-lr_t)
        self.step += 1
        return float(o['loss_gen']), float(o['loss_kld']), float(o['loss'])
