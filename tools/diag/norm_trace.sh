#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of the norm microbenchmark, one shape per run: bash tools/diag/norm_trace.sh "33280 192" "50432 768"
set -e
root=$(pwd)
out=$root/gpurun_out/norm_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for shape in "$@"; do
	tag=${shape// /x}
	timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/new_$tag -o t -- python3 $root/tools/bench_norm_small.py $shape > $out/new_$tag.log 2>&1
done
