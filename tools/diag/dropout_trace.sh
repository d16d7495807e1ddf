#!/bin/bash
# headline step with the reference's default dropout (0.3): bench line and kernel trace
set -e
root=$(pwd)
out=$root/gpurun_out/drop
mkdir -p $out
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --dropout 0.3 > $out/bench_drop03.json
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_drop0.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 $root/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --dropout 0.3 > $out/trace.log 2>&1
