#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of the narrow-row norm microbenchmark, several rows per wavefront against one
set -e
root=$(pwd)
out=$root/gpurun_out/norm_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for shape in "33280 192" "25088 384" "100352 96" "16384 64"; do
	tag=${shape// /x}
	unset VITK_NO_NORM_SMALL_ROWS
	timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/new_$tag -o t -- python3 $root/tools/bench_norm_small.py $shape > $out/new_$tag.log 2>&1
	export VITK_NO_NORM_SMALL_ROWS=1
	timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/old_$tag -o t -- python3 $root/tools/bench_norm_small.py $shape > $out/old_$tag.log 2>&1
done
