#!/bin/bash
# The measurements behind profiles/<tag>_*: bench line, kernel-trace summary and the PMC passes (each in a run of its own)
# of the headline configuration.  Run on the GPU box from the repository root: bash tools/profile_round.sh r2
set -e
tag=${1:-rN}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 300 python3 bench.py --steps 25 --warmup 5 > $out/bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 $root/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $out/trace.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/write.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/sq -o s -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $out/sq.log 2>&1
cd $root
find $out -name "*counter_collection.csv" | head
