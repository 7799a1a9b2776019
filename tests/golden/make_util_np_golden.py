"""captures golden vectors of the reference's pure numpy helpers (src/util_np.py:5-33) by importing
the reference module IN THE BUILD CONTAINER (it needs only numpy).  The reference never travels to
the GPU box: only the JSON written here is committed.

    python tests/golden/make_util_np_golden.py
"""
import json
import os
import sys
from itertools import islice

sys.path.insert(0, '/root/reference/src')
import numpy as np  # noqa: E402
import util_np as ref  # noqa: E402

out = {'vpack': [], 'partition': [], 'sample': []}
rng = np.random.default_rng(0)
for case in range(6):
    rows = int(rng.integers(1, 7))
    lens = [int(rng.integers(0, 9)) for _ in range(rows)]
    arrs = [[int(x) for x in rng.integers(3, 100, n)] for n in lens]
    shape = (rows + int(rng.integers(0, 2)), max(lens + [1]) + int(rng.integers(0, 3)))
    fill = 1
    got = ref.vpack(arrs, shape, fill, np.int32)
    out['vpack'].append(dict(arrays=arrs, shape=list(shape), fill=fill, result=got.tolist()))
for n, m, discard in [(10, 4, False), (10, 4, True), (12, 4, False), (3, 5, False), (3, 5, True), (0, 3, False), (4096, 200, False), (7, 1, False)]:
    out['partition'].append(dict(n=n, m=m, discard=discard, result=[list(p) for p in ref.partition(n, m, discard)]))
for n, seed, k in [(5, 0, 12), (7, 3, 30), (1, 0, 4), (10, 25, 45)]:
    out['sample'].append(dict(n=n, seed=seed, result=[int(x) for x in islice(ref.sample(n, seed), k)]))
with open(os.path.join(os.path.dirname(__file__), 'util_np_golden.json'), 'w') as f:
    json.dump(out, f)
print({k: len(v) for k, v in out.items()})
