#!/bin/bash
# per-kernel durations of scripts/gru_steps.py under rocprofv3 (kernel trace only).  usage: kstats.sh <tag> [B S]; OPTS / DTYPE / RAGGED from the environment
tag=$1; shift
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/${tag}_ks -o runc --output-format csv -- python3 $root/scripts/gru_steps.py "$@" > $out/${tag}_ks.log 2>&1 || { tail -5 $out/${tag}_ks.log; exit 1; }
f=$(find $out/${tag}_ks -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gru' in r['Name']:
        print('%-90s calls %4s  avg %9.1f us  min %9.1f' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
