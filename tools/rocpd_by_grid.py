"""Average kernel duration grouped by (name, grid, workgroup) from a rocprofv3 rocpd database.
    python3 tools/rocpd_by_grid.py <results.db> [name-substring]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size")]
gy = [c for c in cols if c.lower() in ("grid_y", "grid_size_y")]
wx = [c for c in cols if c.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size")]
sel = ", ".join(["name"] + gx[:1] + gy[:1] + wx[:1])
q = f"select {sel}, count(*), avg(end - start), min(end - start) from kernels where name like ? group by {sel} order by 1, 2"
for row in db.execute(q, (f"%{pat}%",)):
    name = re.sub(r"\(.*", "", row[0].replace("void (anonymous namespace)::", ""))[:60]
    print(f"{name:60s} grid/wg {row[1:-3]}  n {row[-3]:5d}  avg {row[-2] / 1e3:8.2f} us  min {row[-1] / 1e3:8.2f} us")
