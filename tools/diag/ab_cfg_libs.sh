#!/bin/bash
# A/B of whole-library builds (diag_bin/libvitk_<name>.so) on another configuration: CFG="vit_tiny --graph" bash tools/diag/ab_cfg_libs.sh prev new
rounds=${ROUNDS:-3}
for round in $(seq $rounds); do
	for n in "$@"; do
		echo -n "$CFG $n: "
		VITK_LIB=$PWD/diag_bin/libvitk_$n.so timeout -k 10 150 python bench.py --config $CFG --steps 30 --warmup 8 --no-secondary --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" || exit 1
	done
done
