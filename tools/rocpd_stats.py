"""Per-kernel summary of a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`).

usage: python tools/rocpd_stats.py results.db [divide_by_steps] [out.csv]
"""
import csv
import sqlite3
import sys


def main():
	db = sys.argv[1]
	steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
	out = sys.argv[3] if len(sys.argv) > 3 else None
	c = sqlite3.connect(db)
	cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
	namecol = "name" if "name" in cols else "kernel_name"
	rows = c.execute(
		f"select {namecol}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by {namecol} order by 3 desc"
	).fetchall()
	total = sum(r[2] for r in rows)
	table = [("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "MsPerStep")]
	for n, calls, tot, avg, mn, mx in rows:
		table.append((n, calls, tot, round(avg, 1), mn, mx, round(100.0 * tot / total, 2), round(tot / steps / 1e6, 4)))
	if out:
		with open(out, "w", newline="") as f:
			csv.writer(f).writerows(table)
	for r in table[1:26]:
		print(f"{r[0][:100]:100s} {r[1]:6d} {r[7]:9.3f} ms/step {r[3]/1e3:9.1f} us avg {r[6]:6.2f}%")
	print("total ms/step", total / steps / 1e6)


if __name__ == "__main__":
	main()
