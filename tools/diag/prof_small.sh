#!/bin/bash
# kernel traces of the small configurations: bash tools/diag/prof_small.sh  (summaries in gpurun_out/prof_small/<config>.txt)
set -e
root=$(pwd)
out=$root/gpurun_out/prof_small
rm -rf $out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in vit_tiny cvt deit_tiny coatnet; do
	timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/$c -o t -- python3 $root/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary > $out/$c.log 2>&1
	python3 $root/tools/rocpd_stats.py $(find $out/$c -name "*.db" | head -1) 12 $out/$c.csv > $out/$c.txt
	python3 $root/tools/diag/rocpd_by_grid.py $(find $out/$c -name "*.db" | head -1) gemm_bf16_kernel 12 > $out/$c.grids.txt
	rm -rf $out/$c
done
head -32 $out/vit_tiny.txt $out/cvt.txt
