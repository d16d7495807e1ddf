"""Timeline summary of a rocprofv3 kernel trace (rocpd database): busy time (union of kernel intervals), idle gaps,
time with two or more kernels in flight, per-stream totals.  usage: python tools/diag/trace_gaps.py results.db [skip_fraction]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute(f"select start, end, {namecol}" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start").fetchall()
# steady state: from the end of one optimizer kernel to the end of the last one (whole training steps), skipping the first ones
ad = [r[1] for r in rows if "adamw_advance" in r[2]]  # one per optimizer step
if len(ad) >= 4:
	first = ad[max(1, int(len(ad) * skip))]
	rows = [r for r in rows if r[0] >= first and r[1] <= ad[-1]]
	print("steps in window:", sum(1 for t in ad if first < t <= ad[-1]))
else:
	t_begin = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip
	rows = [r for r in rows if r[0] >= t_begin]
span = rows[-1][1] - rows[0][0]
busy = 0
multi = 0
cur_s, cur_e = rows[0][0], rows[0][1]
gaps = []
for s, e, n, q in rows[1:]:
	if s <= cur_e:
		multi += min(e, cur_e) - s
		cur_e = max(cur_e, e)
	else:
		busy += cur_e - cur_s
		gaps.append((s - cur_e, n))
		cur_s, cur_e = s, e
busy += cur_e - cur_s
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {100 * (span - busy) / span:.2f} %  >=2 kernels in flight {100 * multi / span:.2f} % of span")
gaps.sort(reverse=True)
print("gaps: n", len(gaps), "sum ms", sum(g for g, _ in gaps) / 1e6, "median us", sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3)
by = {}
for g, n in gaps:
	a = by.setdefault(n[:70], [0, 0])
	a[0] += g
	a[1] += 1
for n, (g, k) in sorted(by.items(), key=lambda kv: -kv[1][0])[:12]:
	print(f"   gap before {n:70s} {g / 1e6:8.3f} ms in {k} gaps ({g / k / 1e3:.1f} us each)")
per_q = {}
for s, e, n, q in rows:
	per_q[q] = per_q.get(q, 0) + (e - s)
print("kernel time per queue (ms):", {k: round(v / 1e6, 2) for k, v in per_q.items()})
