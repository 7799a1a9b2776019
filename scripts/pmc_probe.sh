#!/bin/bash
# exploratory counter passes over the bench command (one small counter set per pass); usage: scripts/pmc_probe.sh <tag> "<SET1>" "<SET2>" ...
tag=$1; shift
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $out/${tag}_avail.txt 2>&1
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/${tag}_p$i -o runc --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-configs --no-timing > $out/${tag}_p$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $out/${tag}_p$i.log; }
done
echo probe passes done
