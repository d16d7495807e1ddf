#!/bin/bash
# kernel trace of one configuration: CFG="coatnet" STEPS=6 bash tools/diag/prof_one.sh  (summary in gpurun_out/prof_one/<tag>.txt)
set -e
root=$(pwd)
c=${CFG:-coatnet}; steps=${STEPS:-6}; tag=${TAG:-$(echo $c | tr ' -.' '___')}
out=$root/gpurun_out/prof_one
rm -rf $out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/$tag -o t -- python3 $root/bench.py --config $c --steps $steps --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary > $out/$tag.log 2>&1
python3 $root/tools/rocpd_stats.py $(find $out/$tag -name "*.db" | head -1) $((steps + 2)) $out/$tag.csv > $out/$tag.txt
python3 $root/tools/diag/rocpd_by_grid.py $(find $out/$tag -name "*.db" | head -1) gemm_bf16 $((steps + 2)) > $out/$tag.grids.txt
rm -rf $out/$tag
cat $out/$tag.txt
