#!/bin/bash
# A/B of whole-library builds (diag_bin/libvitk_<name>.so) on the headline step: alternating rounds of bench.py
rounds=${ROUNDS:-3}
for round in $(seq $rounds); do
	for n in "$@"; do
		echo -n "$n: "
		VITK_LIB=$PWD/diag_bin/libvitk_$n.so timeout -k 10 150 python bench.py --steps 20 --warmup 3 --no-secondary 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'], {k: v['ms_per_step'] for k, v in d['roofline']['all_gemm_kernels'].items()})" || exit 1
	done
done
