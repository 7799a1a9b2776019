#!/usr/bin/env python3
"""bench.py -- sentences/sec per ELBO step (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one full ELBO training step (forward + hand-written backward + TF-style Adam, plus the
RCCL gradient all-reduce when N > 1) over one synthetic FULL batch already resident in HBM.
Workload at every N: BASELINE.json configs[1] per GPU -- fp32, batch 256 x seq 64, vocab 8192,
dim_emb 512, latent 128, 3 layers (weak scaling: the per-GPU batch is fixed, global batch 256 N).

Extra objects on the JSON line (N = 1, rank 0):
  roofline      the kernel class with the largest share of the step, its EXECUTED FLOPs per launch (2MNK, a GEMM
                whose row count is a device-side count priced at that count) divided by its mean launch duration from HIP events recorded on the launch stream
                during the timed steps, against the fp32 MFMA peak (157.3 TFLOP/s).
  cpu_baseline  the torch-CPU restatement (oracle/vae_torch.py, "port": the reference's TF graph has
                a GPU-only CudnnGRU and TensorFlow is absent) timed on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(dim_tgt=8192, dim_emb=512, dim_rep=128, rnn_layers=3)
B, S = 256, 64
PEAK_F32_MFMA = 157.3      # TFLOP/s, MI355X_MICROARCH.md chip table
PEAK_BF16_MFMA = 2500.0    # TFLOP/s dense (same table)
STAMP_EVERY = 8            # every n-th timed step carries the per-launch HIP-event stamps of the roofline leg
PMC_TAG = 'r04'            # the committed rocprofv3 --pmc passes `traffic` / `mfma_busy` are read from (quoted only for the same sources: pmc_summary)


def pmc_summary(name):
    """the committed counter summary profiles/<name>_pmc_summary.json, or (None, why): its counters describe the kernels of the
    sources it was taken on (source_digest recorded by scripts/summarize_profile.py), so a summary of other sources is NOT quoted"""
    from argsim_amd.lib import source_digest
    path = os.path.join(ROOT, 'profiles', name + '_pmc_summary.json')
    try:
        pm = json.load(open(path))
    except Exception:
        return None, "no committed counter passes (%s)" % os.path.relpath(path, ROOT)
    if pm.get('source_digest') != source_digest():
        return None, "the committed counter passes %s were taken on other kernel sources (digest %s, this build %s): not quoted" % (
            os.path.relpath(path, ROOT), pm.get('source_digest'), source_digest())
    return pm, None


def host_cores():
    """cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU
    box shows 256 logical CPUs but grants 16; oversubscribing them makes torch-CPU crawl)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(steps=5, warmups=2):
    """the torch-CPU restatement on the SAME full batch (256 x 64, FULL), fwd + bwd + Adam, 2 warm-ups then
    `steps` timed steps on the cores this process may use (~35 s on the GPU box's 16-core share)"""
    import numpy as np
    import torch
    from argsim_amd import synth
    from oracle import vae_numpy as vn
    from oracle import vae_torch as vt
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = vn.make_cfg(**CFG)
    P = vn.init_params(cfg, 0)
    tr = vt.CpuTrainer(P, cfg)
    tr.step = 20000
    b, s = 256, 64
    ids = synth.batch(b, s, CFG['dim_tgt'], seed=0)
    rng = np.random.default_rng(0)
    keep = (rng.random((s, b)) < 0.88).astype(np.int32)
    eps = rng.standard_normal((b, CFG['dim_rep'])).astype(np.float32)
    for _ in range(warmups):
        tr.train_step(ids, ids, keep, eps)
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(ids, ids, keep, eps)
    dt = time.perf_counter() - t0
    return dict(value=b * steps / dt, unit="sentences/sec", cores=cores, kind="port",
                sample="%d timed ELBO steps after %d warm-ups (fwd+bwd+Adam, fp32 torch-CPU restatement of model.py; TensorFlow is "
                       "absent and CudnnGRU is GPU-only) on the full %d x %d FULL batch of the headline workload" % (steps, warmups, b, s))


def side_config(label, dtype, b, s, steps=3, warmup=2, pmc=None, ragged=None, **model_kw):
    """one of the other BASELINE configurations on this GPU, a few steps, with its kernel-class split (HIP events)"""
    import torch
    from argsim_amd import synth
    from argsim_amd.model import VAE
    m = VAE('train', device=torch.cuda.current_device(), seed=0, dtype=dtype, **dict(CFG, **model_kw))
    m.step = 20000
    # (ragged = (median, sigma): LogNormal row lengths clipped to [2, s], eos-padded to s and handed over untrimmed)
    ids = torch.as_tensor(synth.batch(b, s, CFG['dim_tgt'], seed=0, **(dict(ragged=True, len_median=ragged[0], len_sigma=ragged[1]) if ragged else {}))).to(m.device)
    n_real = int((ids != CFG.get('eos', 1)).sum()) if ragged else b * s
    for i in range(warmup):
        m.train_step(ids, ids, seed=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        m.train_step(ids, ids, seed=warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    m.set_option('timing', 1)                      # one more, stamped step for the class split
    m.train_step(ids, ids, seed=99)
    tm = m.timing_collect()
    m.set_option('timing', 0)
    losses = m.losses()
    assert all(x == x and abs(x) < 1e6 for x in losses), losses
    peak = {'f32': PEAK_F32_MFMA, 'bf16': PEAK_BF16_MFMA, 'f32s': PEAK_BF16_MFMA / 6.0}[dtype]
    out = {"workload": label, "dtype": dtype, "batch": b, "seq_len": s, "steps": steps, "warmup": warmup,
           "value": b / dt, "unit": "sentences/sec", "ms_per_step": 1e3 * dt, "loss": losses[2], "real_tokens_per_batch": n_real,
           "classes": {k: {"ms_per_step": v[0], "launches_per_step": v[1], "tflops": (v[2] / (v[0] * 1e-3) / 1e12 if v[0] > 0 else 0.0)}
                       for k, v in tm.items()},
           "gemm_frac_of_peak": (tm['gemm'][2] / (tm['gemm'][0] * 1e-3) / 1e12 / peak) if tm['gemm'][0] > 0 else None,
           "gemm_peak_tflops": peak}
    if pmc:        # the committed rocprofv3 --pmc passes of THIS configuration (scripts/profile_round.sh), per GEMM-class launch
        pm, why = pmc_summary('%s_%s' % (PMC_TAG, pmc))
        out["traffic"] = pm['gemm_class']['hbm_bytes_per_dispatch'] if pm else None
        out["mfma_busy"] = pm['gemm_class'].get('mfma_busy') if pm else None
        out["traffic_unit"] = ("HBM bytes per GEMM-class launch, from the committed PMC passes profiles/%s_%s_pmc_summary.json of these sources (not measured by this run)" % (PMC_TAG, pmc)) if pm else why
    m.close()
    del m, ids
    torch.cuda.empty_cache()
    return out


def main():
    global B, S
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dtype', default='f32', choices=('f32', 'f32s', 'bf16'),
                    help="f32 = exact-fp32 MFMA (headline); f32s = fp32 GEMMs on the bf16 matrix cores by 3-way operand "
                         "splitting (fp32-accurate); bf16 = bf16 operands in every contraction (GEMMs and the GRU recurrence), fp32 accumulate / state / Adam")
    ap.add_argument('--no-timing', action='store_true', help="skip the per-kernel HIP-event stamps (roofline leg)")
    ap.add_argument('--batch', type=int, default=B, help="rows per GPU (default: BASELINE configs[1]; 1024 = configs[2]/[3] per-GPU load)")
    ap.add_argument('--seq', type=int, default=S, help="sequence length (default 64; 128 = configs[2])")
    ap.add_argument('--ragged', action='store_true', help="RAGGED synthetic set (LogNormal lengths, eos padded) instead of FULL; not the headline")
    ap.add_argument('--no-alt', action='store_true', help="skip the extra f32s leg (same workload with the split-bf16 fp32 GEMMs)")
    ap.add_argument('--no-configs', action='store_true', help="skip the 'configs' object (BASELINE configs[2] and configs[3]'s per-GPU load, a few steps each)")
    ap.add_argument('--gru-stagger', type=int, default=0)
    ap.add_argument('--gru-item', type=int, default=-1, help="forward GRU kernel: 0 generic, 1 item pipeline, 2 four-team LDS-weight kernel")
    ap.add_argument('--gru-force-slow', action='store_true', help="never use the same-XCD L2 exchange path")
    ap.add_argument('--gru-ablate', type=int, default=0, help="timing experiments only (results are wrong)")
    ap.add_argument('--stepwise', action='store_true', help="one GRU launch per time step instead of the persistent kernels")
    A = ap.parse_args()
    headline = (A.batch, A.seq) == (B, S) and not A.ragged
    B, S = A.batch, A.seq

    import torch
    import torch.distributed as dist
    from argsim_amd import synth
    from argsim_amd.model import VAE

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == A.gpus, "launch with torch.distributed.run --nproc-per-node %d" % A.gpus
    if os.environ.get('AVAE_FORCE_DEVICE'):          # test hook: several ranks on one GPU (with AVAE_DIST_BACKEND=gloo)
        local = int(os.environ['AVAE_FORCE_DEVICE'])
    torch.cuda.set_device(local)
    dp = None
    model = VAE('train', device=local, seed=0, dtype=A.dtype, **CFG)
    if world > 1 or os.environ.get('AVAE_FORCE_DP') == '1':     # the env knob exercises the DP path on one GPU
        from argsim_amd.dist import DataParallel
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            backend = os.environ.get('AVAE_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm; gloo only for the one-GPU rehearsal
            if backend == 'nccl':
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        dp = DataParallel(model)
        dp.broadcast_params(model.state)
    if A.stepwise:
        model.set_option('persistent', 0)
    model.set_option('gru_stagger', A.gru_stagger)
    if A.gru_item >= 0:
        model.set_option('gru_item', A.gru_item)
    if A.gru_force_slow:
        model.set_option('gru_force_slow', 1)
    if A.gru_ablate:
        model.set_option('gru_ablate', A.gru_ablate)
    model.step = 20000                                  # anneal = tanh(2): the KL backward is live
    ids_np = synth.batch(B, S, CFG['dim_tgt'], ragged=A.ragged, seed=rank)
    ids = torch.as_tensor(ids_np).to(model.device)      # FULL batch (default), resident in HBM
    n_glob, b_glob = float(world * B * (S + 1)), float(world * B)
    if A.ragged:                                        # tokens of the global batch (same length law on every rank; exact for world 1)
        n_glob = float(world * int((ids_np != 1).sum() + B))

    def one(i):
        if dp:
            dp.train_step(ids, ids, n_glob, b_glob, seed=1000 * rank + i)
        else:
            model.train_step(ids, ids, seed=i)

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(A.warmup):
        one(i)
    fence()
    timing = world == 1 and not A.no_timing
    # HIP-event stamps around every GEMM / GRU launch cost ~8 us of stream gap each (0.5 ms per step when
    # every step is stamped), so only every 8th step of the timed region is stamped (3 of the default 20): the
    # roofline figures are live measurements over the timed region, `value` is perturbed by < 0.4 %
    stamped = [i for i in range(A.steps) if i % STAMP_EVERY == 0] if timing else []
    if timing:
        model.set_option('timing', 1)          # resets the stamp list
        model.set_option('timing_pause', 1)
    # one HIP event per step boundary on the launch stream (no synchronisation inside the timed region): the spread of the
    # individual steps -- SURVEY 8(d) asks for a median -- beside the mean the contract's `value` is
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(A.steps + 1)]
    t0 = time.perf_counter()
    for i in range(A.steps):
        marks[i].record()
        if timing and i % STAMP_EVERY == 0:
            model.set_option('timing_pause', 0)
        one(A.warmup + i)
        if timing and i % STAMP_EVERY == 0:
            model.set_option('timing_pause', 1)
    marks[A.steps].record()
    fence()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(A.steps))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=model.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = model.losses()
    assert A.gru_ablate or all(x == x and abs(x) < 1e6 for x in losses), losses

    if rank == 0:
        out = {
            "metric": "sentences/sec per ELBO step", "value": world * B * A.steps / dt, "unit": "sentences/sec",
            "n_gpus": world, "steps": A.steps, "warmup": A.warmup, "ms_per_step": 1e3 * dt / A.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "f32s": "f32 (GEMM operands split into 3 x bf16 in registers, 6 partial products, f32 accumulate)",
                      "bf16": "bf16 operands in every contraction (GEMMs and the GRU recurrence), f32 accumulate/state/gate math/Adam"}[A.dtype], "data": "synthetic",
            "config": {"workload": ("" if headline else "NOT the headline workload (batch %d x seq %d per GPU%s) -- " % (B, S, ", RAGGED lengths" if A.ragged else "")) +
                                   "BASELINE configs[1]: 1xMI355X %s, latent_dim 128, vocab 8k, seq_len 64, batch 256 per GPU; "
                                   "FULL synthetic Zipf batches, dim_emb 512, 3 layers, step 20000"
                                   % {"f32": "fp32", "f32s": "fp32 (split-bf16 MFMA GEMMs, fp32-accurate)",
                                      "bf16": "(bf16 operands: the opt-in mode of configs[2], same shapes)"}[A.dtype],
                       "global_batch": world * B, "seq_len": S, "parallelism": "dp%d" % world,
                       "gru": "stepwise" if A.stepwise else "persistent"},
            "loss": losses[2],
            "step_ms": {"median": per_step[len(per_step) // 2], "min": per_step[0], "max": per_step[-1],
                        "what": "device time between consecutive step boundaries of the timed region (HIP events on the launch stream, rank 0; "
                                "stamped steps carry ~0.5 ms of event gaps)"},
        }
        if dp:
            # what the collective layer really did, so that a scaling record can be checked: the world size the process group
            # reports (RCCL's, backend nccl), the buckets announced per step and the bytes summed per rank and step
            out["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "buckets_per_step": len(model.buckets()),
                           "bytes_reduced_per_rank_per_step": 4 * sum(c for _, c in model.buckets()),
                           "overlap": "bucketed all-reduce from the backward hook on a side HIP stream, fenced before every persistent GRU launch",
                           "note": "no multi-GPU scaling curve has been measured for this build yet (RCCL with world > 1 only runs in the driver's scaling bench)" if world > 1 else "single-GPU rehearsal of the data-parallel path"}
        if timing:
            tm = model.timing_collect()
            model.set_option('timing', 0)
            nst = max(len(stamped), 1)
            total_ms = 1e3 * dt * nst / A.steps            # wall time of the stamped steps
            name, (ms, n, fl) = max(tm.items(), key=lambda kv: kv[1][0])
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            traffic = mfma_busy = None
            pmc_file = 'profiles/%s%s_pmc_summary.json' % (PMC_TAG, '' if A.dtype == 'f32' else '_' + A.dtype)
            # per launch of this kernel class, from the COMMITTED rocprofv3 --pmc passes of this same command (scripts/profile_round.sh
            # + summarize_profile.py; FETCH_SIZE doubled per the gfx950 rule): not a measurement of this run, and quoted only while
            # the kernel sources are the ones the passes were taken on
            pm, why = pmc_summary('%s%s' % (PMC_TAG, '' if A.dtype == 'f32' else '_' + A.dtype))
            if pm and (name + '_class') in pm:
                traffic = pm[name + '_class']['hbm_bytes_per_dispatch']
                mfma_busy = pm[name + '_class'].get('mfma_busy')
            # (the matrix instruction the class runs on: fp32 MFMA; bf16 mode: bf16 MFMA in the GEMMs AND the recurrence; f32s: six bf16 products per fp32 one in the GEMMs only)
            peak = PEAK_F32_MFMA if A.dtype == 'f32' else (PEAK_BF16_MFMA if A.dtype == 'bf16' else (PEAK_BF16_MFMA / 6.0 if name == 'gemm' else PEAK_F32_MFMA))
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": traffic,
                               "traffic_unit": ("HBM bytes per launch, from the committed PMC passes %s of these sources (not measured by this run)" % pmc_file) if pm else why,
                               "mfma_busy": mfma_busy,
                               "flops_per_launch": fl / max(n, 1),
                               "launches_per_step": n / nst, "avg_launch_ms": ms / max(n, 1), "stamped_steps": nst,
                               "share_of_step": ms / total_ms,
                               "classes": {k: {"ms_per_step": v[0] / nst, "launches_per_step": v[1] / nst,
                                               "tflops": (v[2] / (v[0] * 1e-3) / 1e12 if v[0] > 0 else 0.0)} for k, v in tm.items()}}
            us, ut = model.present_ids()
            if (us > 0 or ut > 0) and not A.no_alt:
                # The speed of this line rests on two exact eliminations of work the reference's graph executes without using
                # (parity-tested against the oracle): the first encoder / decoder layer project only the ids
                # PRESENT in the batch (depends on id multiplicity), and the top encoder layer's backward direction runs its one
                # live step instead of S.  Stated here with the rate without each, and without both ("graph_as_written").
                out["config"]["present_ids"] = {"src": us, "tgt": ut, "tokens_src": B * S, "tokens_tgt": B * (S + 1), "vocab": CFG['dim_tgt']}
                out["config"]["eliminated"] = "table_l1 (present-id projection of the embedding-fed layers) + enc_top1 (top encoder layer, backward direction: 1 live step of %d)" % S

                def leg(opts, what):
                    for k, v in opts.items():
                        model.set_option(k, v)
                    for i in range(2):
                        one(10 ** 6 + i)
                    fence()
                    t1 = time.perf_counter()
                    for i in range(A.steps):
                        one(10 ** 6 + 2 + i)
                    fence()
                    dt1 = time.perf_counter() - t1
                    for k in opts:
                        model.set_option(k, 1)
                    return {"value": B * A.steps / dt1, "unit": "sentences/sec", "ms_per_step": 1e3 * dt1 / A.steps, "what": what}

                out["table_l1_0"] = leg({'table_l1': 0}, "same workload with a per-token input projection in the first encoder / decoder layer (option table_l1 = 0)")
                out["enc_top1_0"] = leg({'enc_top1': 0}, "same workload with all %d steps of the top encoder layer's backward direction (option enc_top1 = 0)" % S)
                # SURVEY 8(d)'s second synthetic set: RAGGED rows (len = clip(round(LogNormal(ln 24, 0.5)), 2, S), eos padded)
                ids_r = synth.batch(B, S, CFG['dim_tgt'], ragged=True, seed=0)
                tok_r = int((ids_r != 1).sum()) + B                      # decoder positions incl. the eos step, model.py:91-95
                ids_rd = torch.as_tensor(ids_r).to(model.device)
                for i in range(3):
                    model.train_step(ids_rd, ids_rd, seed=i)
                fence()
                t1 = time.perf_counter()
                for i in range(A.steps):
                    model.train_step(ids_rd, ids_rd, seed=50 + i)
                fence()
                dtr = (time.perf_counter() - t1) / A.steps
                model.set_option('skip_pad', 0)          # the same batch with every step of every row run on the padded layout
                model.set_option('compact', 0)           # (padding neither skipped in the recurrence nor dropped between the GEMMs and the GRU launches)
                for i in range(2):
                    model.train_step(ids_rd, ids_rd, seed=i)
                fence()
                t1 = time.perf_counter()
                for i in range(A.steps):
                    model.train_step(ids_rd, ids_rd, seed=50 + i)
                fence()
                dtr0 = (time.perf_counter() - t1) / A.steps
                model.set_option('skip_pad', 1)
                model.set_option('compact', 2)
                out["ragged"] = {"ms_per_step_every_step_run": 1e3 * dtr0,"value": B / dtr, "unit": "sentences/sec", "tokens_per_sec": tok_r / dtr, "ms_per_step": 1e3 * dtr,
                                 "mean_len": float((ids_r != 1).sum(1).mean()), "max_len": int((ids_r != 1).sum(1).max()),
                                 "padded_step_share": 1.0 - float((ids_r != 1).sum()) / float(B * int((ids_r != 1).sum(1).max())),
                                 "what": "same model, RAGGED synthetic batch %d x %d (LogNormal lengths); not the headline" % (B, S)}
                out["graph_as_written"] = leg({'table_l1': 0, 'enc_top1': 0}, "both off: every FLOP of the reference's graph executed (2 153 GFLOP per step)")
                # Data-parallel overlap rehearsed on ONE GPU: the bucket hook live, and for every announced bucket a
                # stand-in for its all-reduce on a side stream -- device-local copies moving 2 x the bucket's bytes in and out, what a
                # reduce-scatter + all-gather touches in HBM -- fenced before every persistent GRU launch exactly as GradReducer fences
                # RCCL.  It prices what side-stream traffic beside backward costs the step on this GPU (the persistent GEMM's static
                # schedule assumes an otherwise idle chip); it says nothing about xGMI time: no peer, no RCCL kernel.
                side = torch.cuda.Stream(model.device)
                scratch = torch.empty(max(c for _, c in model.buckets()), dtype=torch.float32, device=model.device)

                def rehearsal_hook(bucket, off, cnt):
                    cur = torch.cuda.current_stream(model.device)
                    if bucket < 0:
                        cur.wait_stream(side)
                        return
                    ev = torch.cuda.Event()
                    ev.record(cur)
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        scratch[:cnt].copy_(model.grads[off:off + cnt])
                        model.grads[off:off + cnt].copy_(scratch[:cnt])

                def rehearse(n):
                    for i in range(n):
                        model.forward_backward(ids, ids, seed=2 * 10 ** 6 + i)
                        torch.cuda.current_stream(model.device).wait_stream(side)
                        model.adam_step()
                model.set_grad_hook(rehearsal_hook)
                rehearse(2)
                fence()
                t1 = time.perf_counter()
                rehearse(A.steps)
                fence()
                dth = (time.perf_counter() - t1) / A.steps
                model.set_grad_hook(None)
                out["dp_overlap_rehearsal"] = {
                    "ms_per_step": 1e3 * dth, "ms_cost_vs_headline": 1e3 * dth - out["ms_per_step"], "buckets_per_step": len(model.buckets()),
                    "bytes_per_step": 4 * sum(c for _, c in model.buckets()),
                    "what": "one GPU, bucket hook live, a device-local stand-in for every bucket's all-reduce on a side stream (2 x bucket bytes read and "
                            "written), fenced before each persistent GRU launch: the cost of side-stream traffic beside backward, NOT a multi-GPU measurement"}
            if not A.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
        if world == 1 and A.dtype == 'f32' and not A.no_alt and headline:
            # same workload, same steps, GEMMs on the bf16 matrix cores with fp32-accurate operand splitting
            # (opt-in mode, held to the fp32 tolerances by the parity tests): reported beside the headline, never as it
            del model
            torch.cuda.empty_cache()
            m2 = VAE('train', device=local, seed=0, dtype='f32s', **CFG)
            m2.step = 20000
            for i in range(A.warmup):
                m2.train_step(ids, ids, seed=i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(A.steps):
                m2.train_step(ids, ids, seed=A.warmup + i)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            l2 = m2.losses()
            assert all(x == x and abs(x) < 1e6 for x in l2), l2
            out["f32s"] = {"value": B * A.steps / dt2, "unit": "sentences/sec", "ms_per_step": 1e3 * dt2 / A.steps, "loss": l2[2],
                           "dtype": "f32 in / f32 accumulate / f32 out; GEMM operands split into 3 x bf16 in registers, 6 partial "
                                    "products on v_mfma_f32_32x32x16_bf16 (VAE(dtype='f32s'), compute_dtype 2)"}
        if world == 1 and headline and A.dtype == 'f32' and not A.no_configs:
            # the other single-GPU BASELINE configurations, labelled; never the headline value
            try:
                del m2
            except NameError:
                pass
            torch.cuda.empty_cache()
            out["configs"] = {
                "configs[2]": side_config("BASELINE configs[2]: 1xMI355X bf16 operands in the GEMMs and the GRU recurrence (fp32 accumulate/state), seq_len 128, batch 1024", 'bf16', 1024, 128, pmc='configs2'),
                "configs[4]": side_config("BASELINE configs[4]: 1xMI355X fp32, latent_dim 512 with the beta / free-bits extension live (kl_beta 0.5, free_bits 0.02), seq_len 64, batch 256", 'f32', 256, 64, steps=10, dim_rep=512, kl_beta=0.5, free_bits=0.02),
                "reference_training_geometry": side_config("the reference's own training geometry (src/config.json: batch_train 100, max_len 512, dim_rep 1024), FULL 512-piece rows, fp32 -- context for BASELINE.md section 1 (paper: ~1.14 s / step on an unstated NVIDIA GPU), not a BASELINE config", 'f32', 100, 512, steps=3, dim_rep=1024),
                "reference_training_geometry_ragged": side_config("the same geometry (batch 100, rows padded to 512, dim_rep 1024) with RAGGED rows: lengths LogNormal(median 120, sigma 0.7) clipped to [2, 512] -- an ASSUMED post-length distribution (the corpora are not in the container); batch 100 has no team-kernel geometry of its own: 128 slots, 28 phantom rows (DESIGN 4.2f)", 'f32', 100, 512, steps=3, dim_rep=1024, ragged=(120.0, 0.7)),
                "configs[3]/gpu": side_config("BASELINE configs[3] per-GPU load: fp32, batch 1024 (global 8192 over 8 GPUs), seq_len 64; the all-reduce is not part of it", 'f32', 1024, 64),
            }
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
