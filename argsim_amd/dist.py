"""data-parallel replicas: one process per GPU, gradients summed with RCCL over xGMI.

The reference is single-GPU (src/train.py:14,26); this is the extension BASELINE.json asks for
(SURVEY.md section 8e).  Sentences are independent inside a step and parameters are replicated,
so rank r takes rows [r*B/P, (r+1)*B/P) of the global batch.  There is ONE data-path collective:
the gradient all-reduce, issued per bucket of the flat gradient buffer in backward-completion
order from a hook that fires while backward is still being enqueued, on a side HIP stream.  The
library announces a finished bucket right after it has enqueued the NEXT persistent GRU launch and
asks for a fence (bucket -1) before every such launch: the collectives therefore run beside the
GEMM phases of backward and never beside a persistent launch, which needs every CU resident
(include/argsim_vae.h, avae_grad_hook).  RCCL with world > 1 has not run on hardware yet.

Exactness: loss_gen is a mean over the GLOBAL token count N (model.py:181) and loss_kld a mean
over the global (B, R) (model.py:184), so every rank scales its local gradients by 1/N_global and
1/(B_global R) (arguments of avae_forward_backward) and the all-reduce is a plain SUM.

``GradReducer`` only needs flat tensors + bucket ranges, so the same code runs over gloo on CPU
tensors in the tests (world_size 2) and over RCCL on the device."""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat_grads, buckets, group=None, comm_stream=None):
        self.g = flat_grads
        self.buckets = list(buckets)
        self.group = group
        self.comm_stream = comm_stream
        self.pending = []
        # RCCL runs a collective on the process group's own internal stream, ordered after the stream that was current at
        # the call; Work.wait() then only makes the CURRENT stream wait for it (no host block).  Taken at once, inside the
        # side-stream context, it puts the collective's completion INTO the side stream's order, so that fence() and
        # wait() -- which order the compute stream after the side stream -- really are ordered after the collectives.
        # (gloo completes on the host: its Work objects are kept and waited for where the results are needed.)
        self.stream_ordered = comm_stream is not None and dist.get_backend(group) == 'nccl'

    def reduce_bucket(self, i):
        """all-reduce(sum) of bucket i; on a device, ordered after the work already enqueued on the
        current stream and executed on the side stream."""
        off, cnt = self.buckets[i]
        view = self.g[off:off + cnt]
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.g.device))
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.stream_ordered:
                    w.wait()
                else:
                    self.pending.append(w)
        else:
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def fence(self):
        """the compute stream waits for every collective in flight (called before a persistent GRU launch)"""
        if self.stream_ordered:
            torch.cuda.current_stream(self.g.device).wait_stream(self.comm_stream)
        else:
            for w in self.pending:
                w.wait()
            self.pending = []
            if self.comm_stream is not None:
                torch.cuda.current_stream(self.g.device).wait_stream(self.comm_stream)

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.g.device).wait_stream(self.comm_stream)


def rank_seed(seed, rank):
    """a 64-bit RNG key per rank from one shared key (splitmix64 finaliser over seed + golden-ratio * (rank + 1))"""
    m = (1 << 64) - 1
    x = (seed + 0x9E3779B97F4A7C15 * (rank + 1)) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    return x ^ (x >> 31)


def shard_rows(B, rank, world):
    """rows of the global batch owned by ``rank`` (contiguous, equal shares; B % world == 0)"""
    assert B % world == 0, "global batch must divide evenly over the ranks"
    per = B // world
    return rank * per, (rank + 1) * per


def global_token_count(n_local, group=None, device='cpu'):
    """N_global = sum over ranks of sum_b (len_b + 1): one scalar all-reduce (skippable for FULL batches)"""
    t = torch.tensor([float(n_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())


class DataParallel:
    """wraps a model exposing: grads (flat tensor), buckets(), set_grad_hook(fn), forward_backward(src,
    tgt, seed=, keep_mask=, eps=, n_tok_global=, b_global=), adam_step().  ``argsim_amd.model.VAE`` does."""

    def __init__(self, model, group=None, overlap=True):
        self.m = model
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        on_gpu = model.grads.is_cuda
        self.comm_stream = torch.cuda.Stream(model.grads.device) if (on_gpu and overlap) else None
        self.reducer = GradReducer(model.grads, model.buckets(), group, self.comm_stream)
        self.overlap = overlap
        if overlap:
            model.set_grad_hook(lambda b, off, cnt: self.reducer.fence() if b < 0 else self.reducer.reduce_bucket(b))

    def broadcast_params(self, flat_params):
        dist.broadcast(flat_params, src=0, group=self.group)

    def train_step(self, src_local, tgt_local, n_tok_global, b_global, seed=None, keep_mask=None, eps=None):
        """one ELBO step on this rank's shard.  n_tok_global: tokens of the global batch.
        seed=None: the model's own (seed, step, call) key, decorrelated per rank -- the reference draws word dropout
        and eps i.i.d. over the whole batch (model.py:93,153), so two shards must not share one stream."""
        if seed is None and hasattr(self.m, 'next_seed'):
            seed = rank_seed(self.m.next_seed(), self.rank)
        self.m.forward_backward(src_local, tgt_local, seed=seed, keep_mask=keep_mask, eps=eps,
                                n_tok_global=n_tok_global, b_global=b_global)
        if not self.overlap:
            for i in range(len(self.reducer.buckets)):
                self.reducer.reduce_bucket(i)
        self.reducer.wait()
        self.m.adam_step()
