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
# kernel trace of the bench command as it runs by default (weight-gradient GEMMs on the second stream share the CUs with the
# main stream's kernels: their durations overlap), and of the same command with every kernel on one stream (each kernel's own
# duration: what the bench line's roofline object, timed in its all-on-one-stream pass, is reconciled with; the optimizer
# step not pipelined under the next forward either)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace_overlap -o t -- python3 $root/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-timing > $out/trace_overlap.log 2>&1
export VITK_WGRAD_OVERLAP=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 $root/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-pipelined-step > $out/trace.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing --no-pipelined-step > $out/fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing --no-pipelined-step > $out/write.log 2>&1
timeout -s KILL 600 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/sq -o s -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing --no-pipelined-step > $out/sq.log 2>&1
cd $root
unset VITK_WGRAD_OVERLAP
find $out -name "*counter_collection.csv" | head
# summaries (per training step): the one-stream trace holds 2 + 8 + 8 steps (warm-up, timed, event-timed), the default one 2 + 8
python3 tools/rocpd_stats.py $(find $out/trace -name "*.db" | head -1) 18 $out/kernel_stats.csv > $out/kernel_stats.txt
python3 tools/rocpd_stats.py $(find $out/trace_overlap -name "*.db" | head -1) 10 $out/kernel_stats_overlap.csv > $out/kernel_stats_overlap.txt
python3 tools/diag/trace_gaps.py $(find $out/trace_overlap -name "*.db" | head -1) 0.4 > $out/timeline_overlap.txt
python3 tools/pmc_traffic.py $(find $out/fetch -name "*counter_collection.csv" | head -1) $(find $out/write -name "*counter_collection.csv" | head -1) $out/pmc_hbm_traffic.json 4
python3 tools/pmc_sq.py $(find $out/sq -name "*counter_collection.csv" | head -1) $out/pmc_sq_counters.json "bench.py --steps 3 --warmup 1 (whole training step, all kernels on one stream)"

