#!/bin/bash
# A/B of one environment switch on another configuration, alternating, one box: CFG="vit_tiny --graph" bash tools/diag/ab_cfg_env.sh VAR [n_pairs]
var=$1; n=${2:-2}
for i in $(seq 1 $n); do for v in 0 1; do
	echo -n "$CFG $var=$v: "
	env $var=$v timeout -k 10 150 python bench.py --config $CFG --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" || exit 1
done; done
