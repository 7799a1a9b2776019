#!/bin/bash
# PMC passes over one GEMM shape: L2 hit/miss and fabric bytes.  usage: pmc_gemm.sh <tag> <dtype> a_mc b_nc M N K
tag=$1; export DTYPE=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace -d $d -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gemm_one.py "$@" > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/*/*counter_collection.csv') + glob.glob('$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'gemm' in r['Kernel_Name']: agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print('   %-28s %14.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
