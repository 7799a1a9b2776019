"""turns the rocprofv3 outputs under gpurun_out/<tag>_{trace,pmc_fetch,pmc_write,pmc_mfma} into the committed
summaries profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_summary.json.

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC
slot budget), values are KiB, and on gfx950 FETCH_SIZE reports one half of the bytes of wide coalesced reads, so it is
doubled; WRITE_SIZE is taken as is.
Matrix-pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over every SIMD) over the SIMD-cycles of the
dispatch = GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs.  The clock the chip held = GRBM_GUI_ACTIVE / 8 / duration."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
suffix = sys.argv[2] if len(sys.argv) > 2 else ''          # e.g. _f32s
os.makedirs('profiles', exist_ok=True)
ks = glob.glob('gpurun_out/%s_trace/**/*kernel_stats.csv' % tag, recursive=True)[0]
shutil.copy(ks, 'profiles/%s%s_kernel_stats.csv' % (tag, suffix))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argsim_amd.lib import source_digest  # noqa: E402
out = {'source_digest': source_digest(), 'units': 'bytes per dispatch (mean); fetch doubled per the gfx950 correction; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / '
                '(GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); clock_ghz = GRBM_GUI_ACTIVE / 8 / dispatch duration (profiled pass)',
       'kernels': {}}


def rows(nm):
    fs = glob.glob('gpurun_out/%s_pmc_%s/**/*counter_collection.csv' % (tag, nm), recursive=True)
    return csv.DictReader(open(fs[0])) if fs else []


for nm in ('fetch', 'write'):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows(nm):
        k = r['Kernel_Name']
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
    for k, (n, v) in agg.items():
        e = out['kernels'].setdefault(k, {})
        e['dispatches_' + nm] = n
        e[nm + '_bytes'] = v / n * 1024.0 * (2.0 if nm == 'fetch' else 1.0)
mf = collections.defaultdict(lambda: collections.defaultdict(float))
seen = set()
for r in rows('mfma'):
    k = r['Kernel_Name']
    mf[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (k, r['Dispatch_Id']) not in seen:
        seen.add((k, r['Dispatch_Id']))
        mf[k]['ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        mf[k]['n'] += 1
for k, d in mf.items():
    simd_cycles = d['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0
    e = out['kernels'].setdefault(k, {})
    e['mfma_busy'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles if simd_cycles else 0.0
    e['clock_ghz'] = d['GRBM_GUI_ACTIVE'] / 8.0 / d['ns'] if d['ns'] else 0.0
    e['dispatches_mfma'] = int(d['n'])


def klass(pred):
    g = {'fetch': 0.0, 'write': 0.0, 'n': 0, 'busy': 0.0, 'cyc': 0.0, 'act': 0.0, 'ns': 0.0}
    for k, e in out['kernels'].items():
        if not pred(k):
            continue
        g['fetch'] += e.get('fetch_bytes', 0.0) * e.get('dispatches_fetch', 0)
        g['write'] += e.get('write_bytes', 0.0) * e.get('dispatches_write', 0)
        g['n'] += e.get('dispatches_fetch', 0)
        d = mf.get(k)
        if d:
            g['busy'] += d['SQ_VALU_MFMA_BUSY_CYCLES']; g['act'] += d['GRBM_GUI_ACTIVE']; g['ns'] += d['ns']
    return {'dispatches': g['n'], 'hbm_bytes_per_dispatch': (g['fetch'] + g['write']) / max(g['n'], 1),
            'mfma_busy': g['busy'] / (g['act'] / 8.0 * 1024.0) if g['act'] else None,
            'clock_ghz': g['act'] / 8.0 / g['ns'] if g['ns'] else None}


# class views used by bench.py
out['gemm_class'] = klass(lambda k: 'gemm_f32' in k or 'gemm_bf16' in k or 'logits_ce' in k)
out['gru_fwd_class'] = klass(lambda k: 'gru_fwd' in k)
out['gru_bwd_class'] = klass(lambda k: 'gru_bwd' in k)
json.dump(out, open('profiles/%s%s_pmc_summary.json' % (tag, suffix), 'w'), indent=1)
print(json.dumps({k: out[k] for k in ('gemm_class', 'gru_fwd_class', 'gru_bwd_class')}))
