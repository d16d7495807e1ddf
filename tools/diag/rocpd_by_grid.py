"""Launches of the kernels whose name contains a pattern, grouped by grid size (a rocprofv3 rocpd database).

usage: python tools/diag/rocpd_by_grid.py results.db pattern [divide_by_steps]
"""
import sqlite3
import sys

db, pat = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
gcols = [x for x in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z") if x in cols][:3]
wcols = [x for x in ("workgroup_x", "workgroup_size_x") if x in cols][:1]
lcols = [x for x in ("lds_size", "lds_block_size", "group_segment_size") if x in cols][:1]
sel = ", ".join(gcols + wcols + lcols)
rows = c.execute(f"select {namecol}, {sel}, count(*), avg(end - start), min(end - start) from kernels where {namecol} like ? group by {namecol}, {sel} order by 1, count(*) * avg(end - start) desc", (f"%{pat}%",)).fetchall()
for r in rows:
	n = len(gcols + wcols + lcols)
	print(f"{r[0][:60]:60s} grid/wg/lds {r[1:1 + n]}  {r[1 + n] / steps:6.1f}/step  {r[2 + n] / 1e3:7.1f} us avg {r[3 + n] / 1e3:7.1f} min  {r[1 + n] * r[2 + n] / steps / 1e6:7.3f} ms/step")
