"""HBM traffic per launch of the GEMM kernels from two rocprofv3 PMC passes over bench.py
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate runs, csv output).

FETCH_SIZE / WRITE_SIZE are in KiB at the L2's memory-side interface.  On gfx950 FETCH_SIZE tallies the 128-byte
requests of wide (16 B per lane) streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane
stores.  Writes profiles/<out>.json: {label: {"kernel_launches_per_step", "fetch_bytes_per_step", "write_bytes_per_step",
"traffic_bytes_per_step"}} (one vitk_gemm call can be two kernel launches, so the totals are kept per training step and
bench.py divides by its own count of calls per step).

usage: python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json steps_in_the_profiled_run
"""
import collections
import csv
import json
import re
import sys


def label_of(kernel_name):
	"""bench.py's GEMM label (the kernel family vitk_gemm_last_kernel() reports) of a kernel instantiation, or None."""
	m = re.search(r"gemm_(bf16_p8w|bf16_p8|bf16_t128|bf16_256p?|bf16|generic)_kernel", kernel_name)
	if m:
		return "gemm_" + m.group(1).replace("256p", "256") + "_kernel"
	return None


def per_label(path, counter):
	tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
	for r in csv.DictReader(open(path)):
		if r["Counter_Name"] != counter:
			continue
		lab = label_of(r["Kernel_Name"])
		if lab is None:
			continue
		tot[lab] += float(r["Counter_Value"])
		cnt[lab] += 1
	return tot, cnt


def main():
	fetch, fc = per_label(sys.argv[1], "FETCH_SIZE")
	write, wc = per_label(sys.argv[2], "WRITE_SIZE")
	steps = float(sys.argv[4])
	out = {}
	for lab in fetch:
		f = 2.0 * 1024.0 * fetch[lab] / steps
		w = 1024.0 * write.get(lab, 0.0) / steps
		out[lab] = {"kernel_launches_per_step": fc[lab] / steps, "fetch_bytes_per_step": f, "write_bytes_per_step": w, "traffic_bytes_per_step": f + w}
		print(f"{lab}: {fc[lab] / steps:.1f} kernel launches per step, fetch {f/1e9:.2f} GB (x2 corrected), write {w/1e9:.2f} GB per step")
	json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
	main()
