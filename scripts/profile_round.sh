#!/bin/bash
# round profile on the GPU box: bench line, rocprofv3 kernel stats of the same command, HBM and MFMA PMC passes.
# usage (from the repo root, via gpurun): scripts/profile_round.sh <tag> [extra bench args]
# Counter passes carry --kernel-trace only (never a sys/hip/hsa trace) and the program sits directly behind `--`.
tag=${1:-r02}; shift
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
python3 $root/bench.py --steps 20 --warmup 5 "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
tail -1 $out/${tag}_bench.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
quick="--no-cpu-baseline --no-alt --no-configs"
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o runc --output-format csv -- python3 $root/bench.py --steps 12 --warmup 1 $quick "$@" > $out/${tag}_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/${tag}_pmc_fetch -o runc --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 $quick --no-timing "$@" > $out/${tag}_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/${tag}_pmc_write -o runc --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 $quick --no-timing "$@" > $out/${tag}_pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/${tag}_pmc_mfma -o runc --output-format csv -- python3 $root/bench.py --steps 3 --warmup 1 $quick --no-timing "$@" > $out/${tag}_pmc_mfma.log 2>&1 || exit 1
cd $root && python3 scripts/summarize_profile.py $tag
echo profile passes done
