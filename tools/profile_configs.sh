#!/bin/bash
# rocprofv3 kernel traces of the secondary configurations (eager step, 12 steps each): bash tools/profile_configs.sh
# then, here: python tools/rocpd_stats.py gpurun_out/prof_cfg/<config>/t_results.db 12 profiles/<tag>_bench_<config>_kernel_stats.csv
set -e
root=$(pwd)
out=$root/gpurun_out/prof_cfg
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in vit_tiny deit_tiny cvt coatnet; do
	timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/$c -o t -- python3 $root/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing > $out/$c.log 2>&1
done
