#!/bin/bash
# A/B of builds of the one-pass attention backward (diag_bin/libvitk_<name>.so): two alternating rounds of tools/bench_attn_bwd.py each
for round in 1 2; do
	for n in "$@"; do
		echo -n "$n: "
		VITK_LIB=$PWD/diag_bin/libvitk_$n.so timeout -k 10 100 python tools/bench_attn_bwd.py 2>&1 | grep "one pass" | tail -1
	done
done
