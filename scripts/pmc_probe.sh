#!/bin/bash
# exploratory counter passes over the bench command (one small counter set per pass: at most two counters of one
# hardware block, or the profiler refuses the set); usage: scripts/pmc_probe.sh <tag> "<SET1>" "<SET2>" ...
tag=$1; shift
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  echo "pass $i: $set" >> $out/${tag}.progress
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace -d $out/${tag}_p$i -o runc --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-configs --no-timing > $out/${tag}_p$i.log 2>&1 || { echo "pass $i ($set) failed"; grep -m1 "error code" $out/${tag}_p$i.log; }
done
echo probe passes done
