"""turns the rocprofv3 outputs under gpurun_out/<tag>_{trace,pmc_fetch,pmc_write} into the committed
summaries profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_summary.json.

PMC handling follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected in
SEPARATE passes (TCC slot budget), values are KiB, and on gfx950 FETCH_SIZE reports one half of the
bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken as is."""
import collections, csv, glob, json, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
ks = glob.glob('gpurun_out/%s_trace/**/*kernel_stats.csv' % tag, recursive=True)[0]
shutil.copy(ks, 'profiles/%s_kernel_stats.csv' % tag)
out = {'units': 'bytes per dispatch (mean); fetch doubled per the gfx950 correction', 'kernels': {}}
for nm in ('fetch', 'write'):
    f = glob.glob('gpurun_out/%s_pmc_%s/**/*counter_collection.csv' % (tag, nm), recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
    for k, (n, v) in agg.items():
        e = out['kernels'].setdefault(k, {})
        e['dispatches_' + nm] = n
        e[nm + '_bytes'] = v / n * 1024.0 * (2.0 if nm == 'fetch' else 1.0)
# class view used by bench.py: every gemm_f32_kernel instantiation together
g = {'fetch': 0.0, 'write': 0.0, 'n': 0}
for k, e in out['kernels'].items():
    if 'gemm_f32' in k:
        g['fetch'] += e.get('fetch_bytes', 0.0) * e.get('dispatches_fetch', 0)
        g['write'] += e.get('write_bytes', 0.0) * e.get('dispatches_write', 0)
        g['n'] += e.get('dispatches_fetch', 0)
out['gemm_class'] = {'dispatches': g['n'], 'hbm_bytes_per_dispatch': (g['fetch'] + g['write']) / max(g['n'], 1)}
json.dump(out, open('profiles/%s_pmc_summary.json' % tag, 'w'), indent=1)
print(json.dumps(out['gemm_class']))
