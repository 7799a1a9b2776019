"""host-side mirror of the reference's ``vAe()`` Record (reference src/model.py:48-191).

``VAE`` owns the flat device state (parameters, gradients, Adam slots) as PyTorch-ROCm tensors
and drives libargsim_vae.so through the C ABI; PyTorch is plumbing (device memory, streams,
``torch.distributed``), every kernel on the path is hand-written HIP (argsim_amd/csrc).

The reference's feed -> fetch pairs map to methods:
    sess.run(model.train_step)                         -> VAE.train_step(src, tgt)
    sess.run(model.step)                               -> VAE.step
    (errt_samp, loss_gen_samp, loss_kld_samp)          -> VAE.eval(src, tgt)
    model.z.eval({model.src: x})  / encode(sess,vae,x) -> VAE.encode(x) / encode(vae, x)
    decode(sess, vae, z, steps)                        -> VAE.decode(z, steps) / decode(vae, z, steps)
"""
import ctypes as C

import numpy as np
import torch

from . import lib as _lib

PARAM, GRAD, ADAM_M, ADAM_V = 0, 1, 2, 3


def _check_cfg(bidirectional, bidir_stacked, attentive, logit_use_embed):
    # reference src/model.py:124-131,136-145,167-168: alternate branches the paper did not train
    # (config.json:17-20); SURVEY section 2.1 marks them out of scope.
    if not (bidirectional and bidir_stacked):
        raise NotImplementedError("only the stacked bidirectional encoder (config.json:17-18) is implemented")
    if attentive:
        raise NotImplementedError("attentive=True is marked 'todo fixme' in the reference (model.py:136) and is not implemented")
    if not logit_use_embed:
        raise NotImplementedError("only tied logits (logit_use_embed=true, config.json:20) are implemented")


class VAE:
    def __init__(self, mode='train', device=0, dim_tgt=8192, dim_emb=512, dim_rep=1024, rnn_layers=3,
                 bidirectional=True, bidir_stacked=True, attentive=False, logit_use_embed=True,
                 accelerate=1e-4, learn_rate=1e-3, bos=2, eos=1, kl_beta=1.0, free_bits=0.0,
                 seed=0, init=True, dtype='f32'):
        assert mode in ('train', 'valid', 'infer')          # model.py:72
        _check_cfg(bidirectional, bidir_stacked, attentive, logit_use_embed)
        if not torch.cuda.is_available():
            raise RuntimeError("argsim_amd needs an MI355X (HIP device): there is no CPU fallback")
        self.mode, self.bos, self.eos = mode, bos, eos
        self.cfg = dict(dim_tgt=dim_tgt, dim_emb=dim_emb, dim_rep=dim_rep, rnn_layers=rnn_layers,
                        accelerate=accelerate, learn_rate=learn_rate, bos=bos, eos=eos)
        self._l = _lib.load()
        self.device = torch.device('cuda', device)
        assert dtype in ('f32', 'bf16', 'f32s'), \
            "dtype: 'f32' (fp32 MFMA, reference), 'f32s' (fp32 via split bf16 MFMA, fp32-accurate) or 'bf16' (bf16 operands in the GEMMs and the GRU recurrence, fp32 accumulate / state)"
        self.dtype = dtype
        c = _lib.AvaeConfig(dim_tgt, dim_emb, dim_rep, rnn_layers, accelerate, learn_rate, bos, eos, 0, 0, kl_beta, free_bits,
                            {'f32': 0, 'bf16': 1, 'f32s': 2}[dtype])
        h = C.c_void_p()
        if self._l.avae_create(C.byref(c), device, C.byref(h)):
            raise RuntimeError("avae_create: " + self._l.avae_last_error(None).decode())
        self._h = h
        n = self._l.avae_state_numel(h)
        with torch.cuda.device(self.device):
            self.state = torch.zeros((4, n), dtype=torch.float32, device=self.device)
        self.params, self.grads, self.adam_m, self.adam_v = (self.state[i] for i in range(4))
        self._ck(self._l.avae_bind_state(h, *(t.data_ptr() for t in (self.params, self.grads, self.adam_m, self.adam_v))))
        self.names = [self._l.avae_param_name(h, i).decode() for i in range(self._l.avae_param_count(h))]
        self.shapes, self.offsets = {}, {}
        for name in self.names:
            off, nd, shp = C.c_int64(), C.c_int32(), (C.c_int64 * 4)()
            self._ck(self._l.avae_param_info(h, name.encode(), C.byref(off), C.byref(nd), shp))
            self.shapes[name] = tuple(shp[i] for i in range(nd.value))
            self.offsets[name] = off.value
        self._hook_ref = None
        self._calls = 0
        self._seed = seed
        if init:
            self.init_params(seed)

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc):
        if rc:
            raise RuntimeError(self._l.avae_last_error(self._h).decode())

    def _stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        self._ck(self._l.avae_set_stream(self._h, C.c_void_p(s)))

    def close(self):
        if getattr(self, '_h', None):
            self._l.avae_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ids(self, x):
        """-> contiguous int32 cuda tensor (B, S)"""
        if isinstance(x, torch.Tensor):       # a pinned host tensor (train.pinned) crosses asynchronously
            return x.to(device=self.device, dtype=torch.int32, non_blocking=True).contiguous()
        return torch.as_tensor(np.ascontiguousarray(x, dtype=np.int32)).to(self.device, non_blocking=True)

    def trim(self, x):
        """host twin of util_tf.trim (src/util_tf.py:54-57) on a row-major (B, S') numpy batch:
        drops the all-eos tail columns.  Device tensors are passed through (padding columns are
        masked inside the kernels and cost only time)."""
        if isinstance(x, torch.Tensor):
            if x.is_cuda:
                return x
            m = int((x != self.eos).sum(1).max()) if x.numel() else 0
            return x[:, :max(m, 1)]
        x = np.asarray(x)
        m = int((x != self.eos).sum(1).max()) if x.size else 0
        return x[:, :max(m, 1)]

    # ------------------------------------------------------------------ variables
    def set_tensor(self, name, value, kind=PARAM):
        self._stream()
        t = torch.as_tensor(np.ascontiguousarray(value, dtype=np.float32)).to(self.device)
        assert tuple(t.shape) == self.shapes[name], (name, tuple(t.shape), self.shapes[name])
        self._ck(self._l.avae_set_tensor(self._h, name.encode(), kind, C.c_void_p(t.data_ptr())))
        torch.cuda.current_stream(self.device).synchronize()

    def get_tensor(self, name, kind=PARAM):
        self._stream()
        t = torch.empty(self.shapes[name], dtype=torch.float32, device=self.device)
        self._ck(self._l.avae_get_tensor(self._h, name.encode(), kind, C.c_void_p(t.data_ptr())))
        return t.cpu().numpy()

    def set_params(self, P):
        for k in self.names:
            self.set_tensor(k, P[k])

    def get_params(self, kind=PARAM):
        return {k: self.get_tensor(k, kind) for k in self.names}

    def get_grads(self):
        return self.get_params(GRAD)

    def init_params(self, seed=0):
        """reference initialisers: glorot-uniform kernels (variance_scaling(1.0, fan_avg, uniform),
        model.py:9), zero biases (model.py:8), embedding U(+-sqrt(6/(V/D+1))) (model.py:109-110).
        CudnnGRU draws each gate matrix separately with fan_in = input size, fan_out = num_units."""
        rng = np.random.default_rng(seed)
        V, D = self.cfg['dim_tgt'], self.cfg['dim_emb']
        for name in self.names:
            shp = self.shapes[name]
            if name == 'embed/embedding':
                lim = (6.0 / (V / D + 1.0)) ** 0.5
            elif len(shp) == 1:
                self.set_tensor(name, np.zeros(shp, np.float32))
                continue
            elif name.endswith('/W') or name.endswith('/R'):
                lim = (6.0 / (shp[1] + D)) ** 0.5
            else:
                lim = (6.0 / (shp[0] + shp[1])) ** 0.5
            self.set_tensor(name, rng.uniform(-lim, lim, shp).astype(np.float32))
        self.adam_m.zero_()
        self.adam_v.zero_()

    @property
    def step(self):
        s = C.c_int64()
        self._ck(self._l.avae_get_step(self._h, C.byref(s)))
        return s.value

    @step.setter
    def step(self, v):
        self._ck(self._l.avae_set_step(self._h, int(v)))

    def schedule(self):
        """(rate_keepwd, rate_anneal, rate_update) at the current step (model.py:78-80)"""
        out = (C.c_float * 3)()
        self._ck(self._l.avae_get_schedule(self._h, out))
        return tuple(out)

    def set_option(self, key, value):
        self._ck(self._l.avae_set_option(self._h, key.encode(), int(value)))

    def timing_collect(self):
        """{class: (ms, launches, flops)} from the HIP events recorded while option 'timing' was on"""
        out = (C.c_double * 9)()
        self._stream()
        self._ck(self._l.avae_timing_collect(self._h, out))
        return {k: (out[3 * i], int(out[3 * i + 1]), out[3 * i + 2]) for i, k in enumerate(('gemm', 'gru_fwd', 'gru_bwd'))}

    def present_ids(self):
        """(src, tgt): distinct ids in the last forward's encoder / decoder input where that layer was table-fed, else -1"""
        out = (C.c_int32 * 2)()
        self._stream()
        self._ck(self._l.avae_debug_present_ids(self._h, out))
        return int(out[0]), int(out[1])

    def train_ce(self, max_n=1 << 22):
        """per-token cross-entropy (loss_gen_samp, model.py:180) of the last forward_backward / train_step: the TRAIN forward,
        dropout and the latent draw live (test hook)"""
        buf = torch.empty(max_n, dtype=torch.float32, device=self.device)
        n = C.c_int32()
        self._stream()
        self._ck(self._l.avae_debug_train_ce(self._h, C.c_void_p(buf.data_ptr()), max_n, C.byref(n)))
        torch.cuda.synchronize(self.device)
        return buf[:n.value].cpu().numpy()

    def buckets(self):
        out = []
        for i in range(self._l.avae_bucket_count(self._h)):
            o, c = C.c_int64(), C.c_int64()
            self._l.avae_bucket_info(self._h, i, C.byref(o), C.byref(c))
            out.append((o.value, c.value))
        return out

    def set_grad_hook(self, fn):
        """fn(bucket, offset, count) is called while backward is being enqueued (data parallel)."""
        if fn is None:
            self._hook_ref = None
            self._ck(self._l.avae_set_grad_hook(self._h, _lib.GRAD_HOOK(), None))
            return
        self._hook_ref = _lib.GRAD_HOOK(lambda user, b, o, c: fn(b, o, c))
        self._ck(self._l.avae_set_grad_hook(self._h, self._hook_ref, None))

    # ------------------------------------------------------------------ training
    def next_seed(self):
        """the RNG key the next step would use when none is passed: a function of (seed, step, call count)"""
        return (self._seed * 0x9E3779B1 + self.step * 1000003 + self._calls) & 0xFFFFFFFFFFFFFFFF

    def _rng_args(self, seed, keep_mask, eps):
        if seed is None:
            seed = self.next_seed()
        self._calls += 1
        km = ep = None
        if keep_mask is not None:
            km = torch.as_tensor(np.ascontiguousarray(keep_mask)).to(torch.uint8).to(self.device).contiguous()
        if eps is not None:
            ep = torch.as_tensor(np.ascontiguousarray(eps, dtype=np.float32)).to(self.device).contiguous()
        return seed, km, ep

    def forward_backward(self, src, tgt, seed=None, keep_mask=None, eps=None, n_tok_global=0.0, b_global=0.0):
        """gradients of the ELBO into ``self.grads`` (model.py:75-185 + autodiff of minimize())."""
        src, tgt = self._ids(self.trim(src)), self._ids(self.trim(tgt))
        seed, km, ep = self._rng_args(seed, keep_mask, eps)
        if km is not None:
            assert tuple(km.shape) == (tgt.shape[1], tgt.shape[0]), "keep_mask is (S_tgt, B) time-major"
        self._stream()
        self._keep = (src, tgt, km, ep)
        self._ck(self._l.avae_forward_backward(
            self._h, C.c_void_p(src.data_ptr()), C.c_void_p(tgt.data_ptr()), src.shape[0], src.shape[1], tgt.shape[1],
            C.c_uint64(seed), C.c_void_p(km.data_ptr()) if km is not None else None,
            C.c_void_p(ep.data_ptr()) if ep is not None else None, float(n_tok_global), float(b_global)))

    def adam_step(self):
        """tf.train.AdamOptimizer(rate_update) apply + global_step += 1 (model.py:189)"""
        self._stream()
        self._ck(self._l.avae_adam_step(self._h))

    def train_step(self, src, tgt, seed=None, keep_mask=None, eps=None):
        """sess.run(model_train.train_step)  (src/train.py:118)"""
        self.forward_backward(src, tgt, seed, keep_mask, eps)
        self.adam_step()

    def losses(self):
        """(loss_gen, loss_kld, loss) of the last forward; synchronises"""
        self._stream()
        out = (C.c_float * 3)()
        self._ck(self._l.avae_get_losses(self._h, out))
        return tuple(out)

    # ------------------------------------------------------------------ validation / inference
    def eval(self, src, tgt):
        """(errt_samp (N,), loss_gen_samp (N,), loss_kld_samp (B,R)) in 'valid' mode (src/train.py:109-110)"""
        src, tgt = self._ids(self.trim(src)), self._ids(self.trim(tgt))
        B, R = src.shape[0], self.cfg['dim_rep']
        rt = B * (tgt.shape[1] + 1)
        errt = torch.empty(rt, dtype=torch.float32, device=self.device)
        lgen = torch.empty(rt, dtype=torch.float32, device=self.device)
        lkld = torch.empty((B, R), dtype=torch.float32, device=self.device)
        n = C.c_int32()
        self._stream()
        self._ck(self._l.avae_eval(self._h, C.c_void_p(src.data_ptr()), C.c_void_p(tgt.data_ptr()), B, src.shape[1], tgt.shape[1],
                                   C.c_void_p(errt.data_ptr()), C.c_void_p(lgen.data_ptr()), C.c_void_p(lkld.data_ptr()), C.byref(n)))
        return errt[:n.value].cpu().numpy(), lgen[:n.value].cpu().numpy(), lkld.cpu().numpy()

    def encode(self, src, return_lv=False):
        """latent states z = mu (b, dim_rep) float32 (model.py:194-201)"""
        src = self._ids(self.trim(src))
        b, R = src.shape[0], self.cfg['dim_rep']
        z = torch.empty((b, R), dtype=torch.float32, device=self.device)
        lv = torch.empty((b, R), dtype=torch.float32, device=self.device) if return_lv else None
        self._stream()
        self._ck(self._l.avae_encode(self._h, C.c_void_p(src.data_ptr()), b, src.shape[1], C.c_void_p(z.data_ptr()),
                                     C.c_void_p(lv.data_ptr()) if lv is not None else None))
        z = z.cpu().numpy()
        return (z, lv.cpu().numpy()) if return_lv else z

    def decode_init(self, z):
        """state_in (L, b, D) from z (model.py:156,159,213)"""
        z = torch.as_tensor(np.ascontiguousarray(z, dtype=np.float32)).to(self.device)
        b = z.shape[0]
        s = torch.empty((self.cfg['rnn_layers'], b, self.cfg['dim_emb']), dtype=torch.float32, device=self.device)
        self._stream()
        self._ck(self._l.avae_decode_init(self._h, C.c_void_p(z.data_ptr()), b, C.c_void_p(s.data_ptr())))
        return s

    def decode_step(self, lead, state_in):
        """(pred (1,b) int32, state_ex (L,b,D)) for lead (1,b), one GRU step (model.py:216)"""
        lead = torch.as_tensor(lead).to(device=self.device, dtype=torch.int32).contiguous().view(-1)
        b = lead.shape[0]
        pred = torch.empty(b, dtype=torch.int32, device=self.device)
        out = torch.empty_like(state_in)
        self._stream()
        self._ck(self._l.avae_decode_step(self._h, C.c_void_p(lead.data_ptr()), C.c_void_p(state_in.data_ptr()), b,
                                          C.c_void_p(pred.data_ptr()), C.c_void_p(out.data_ptr())))
        return pred.view(1, b), out

    def decode(self, z, steps=256):
        """greedy decoding, array i32 (b, t<=steps) (model.py:204-219)"""
        z = torch.as_tensor(np.ascontiguousarray(z, dtype=np.float32)).to(self.device)
        b = z.shape[0]
        out = torch.empty((b, steps), dtype=torch.int32, device=self.device)
        n = C.c_int32()
        self._stream()
        self._ck(self._l.avae_decode_greedy(self._h, C.c_void_p(z.data_ptr()), b, steps, C.c_void_p(out.data_ptr()), C.byref(n)))
        return out[:, :n.value].cpu().numpy()


def vAe(mode, src=None, tgt=None, **cfg):
    """reference-shaped constructor (src/model.py:48): returns the VAE object in place of the Record.
    ``src``/``tgt`` pipeline tensors have no counterpart: batches are passed to the methods."""
    return VAE(mode, **cfg)


def encode(vae, src):
    """src/model.py:194-201 without the session argument"""
    return vae.encode(src)


def decode(vae, z, steps=256):
    """src/model.py:204-219 without the session argument"""
    return vae.decode(z, steps)
