"""Matrix-pipe busy fraction per kernel family and for the whole run from one rocprofv3 PMC pass
(`--pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv`).

busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): matrix-pipe cycles of the four SIMDs of a CU over the
cycles the CU was busy, i.e. against the clock the chip actually ran at.

usage: python tools/pmc_sq.py counter_collection.csv out.json "<what was run>"
"""
import collections
import csv
import json
import re
import sys


def family(name):
	m = re.search(r"(gemm_bf16_p8w|gemm_bf16_p8|gemm_bf16_t128|gemm_bf16_256p?|gemm_bf16|gemm_generic|attn_\w+?|norm_fwd|norm_bwd|split_reduce\w*|adamw)_kernel", name)
	return m.group(0) if m else "other"


def main():
	tot = collections.defaultdict(lambda: collections.defaultdict(float))
	for r in csv.DictReader(open(sys.argv[1])):
		v = float(r["Counter_Value"])
		tot[family(r["Kernel_Name"])][r["Counter_Name"]] += v
		tot["whole run"][r["Counter_Name"]] += v
	out = {"what": sys.argv[3] if len(sys.argv) > 3 else ""}
	for fam, c in sorted(tot.items()):
		rec = dict(c)
		if c.get("SQ_BUSY_CU_CYCLES"):
			rec["derived_mfma_pipe_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * c["SQ_BUSY_CU_CYCLES"])
		if c.get("SQ_WAVE_CYCLES"):
			rec["derived_wait_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
		out[fam] = rec
		if "derived_mfma_pipe_busy_frac" in rec:
			print(f"{fam:28s} mfma pipe busy {rec['derived_mfma_pipe_busy_frac']:.3f}")
	json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
	main()
