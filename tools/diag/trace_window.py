"""Prints the kernels of a rocprofv3 kernel trace (rocpd database) around the n-th optimizer launch: start / end in us
relative to it, duration, queue.  usage: python tools/diag/trace_window.py results.db [which_adamw=-3] [count=80]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = c.execute(f"select start, end, {namecol}, {qcol} from kernels order by start").fetchall()
ad = [i for i, r in enumerate(rows) if "adamw_advance" in r[2]]
i0 = ad[which]
t0 = rows[i0][0]
for s, e, n, q in rows[max(0, i0 - 6):i0 + count]:
	short = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:48]
	print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{q}  {short}")
