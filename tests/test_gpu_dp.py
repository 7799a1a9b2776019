"""two REAL data-parallel ranks (both computing on the one test GPU, gloo transport) against one process on the whole
batch: the bucket hook, the side-stream reduce, the 1/N_global and 1/(B_global R) scaling on a RAGGED batch and the
update after the reduce must reproduce the single-process parameters."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("V,D,R,B,S,steps,tol", [(512, 64, 32, 16, 12, 2, 2e-6),
                                                 # benchmark dims: persistent LDS-weight GRU kernels, split-K weight-gradient
                                                 # GEMMs (float atomics: an element whose gradient is ~eps moves by ~lr)
                                                 (8192, 512, 128, 128, 32, 1, 5e-5),
                                                 # 100 rows per rank (the reference's batch_train): no team-kernel geometry of its own --
                                                 # phantom rows, compact layout, table-fed first layers under the bucket hook;
                                                 # the single process runs 200 rows on 256 slots
                                                 (1024, 512, 64, 200, 24, 2, 5e-5)])
def test_two_ranks_on_one_gpu_match_single_process(tmp_path, V, D, R, B, S, steps, tol):
    from argsim_amd.model import VAE
    rng = np.random.default_rng(11)
    lens = rng.integers(2, S + 1, B); lens[0] = S
    ids = np.ones((B, S), np.int32)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(3, V, n)
    keep = (rng.random((S, B)) < 0.8).astype(np.uint8)
    eps = rng.standard_normal((B, R)).astype(np.float32)
    case = str(tmp_path / 'case.npz')
    np.savez(case, V=V, D=D, R=R, ids=ids, keep=keep, eps=eps, steps=steps)
    out = str(tmp_path / 'dp.npz')
    port = 29500 + (os.getpid() % 400)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), case, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    m = VAE('train', dim_tgt=V, dim_emb=D, dim_rep=R, rnn_layers=3, seed=0)
    m.step = 20000
    for i in range(steps):
        m.train_step(ids, ids, keep_mask=keep, eps=eps)
    ref = m.get_params()
    got = {k.replace('|', '/'): v for k, v in np.load(out).items()}
    assert sorted(got) == sorted(ref)
    worst = max(float(np.abs(got[k] - ref[k]).max()) for k in ref)
    assert worst <= tol, worst
    mean = float(np.mean([np.abs(got[k] - ref[k]).mean() for k in ref]))
    assert mean <= 1e-7, mean
