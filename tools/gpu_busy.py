"""GPU busy fraction from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`):

    python tools/gpu_busy.py <dir> [skip_fraction]

Union of the kernel intervals over the trace's last (1 - skip_fraction) of wall time (the front holds start-up and warm-up),
the idle gaps between consecutive kernels by size class, and the kernels that most often precede a gap.
"""
import csv
import glob
import sys
from collections import Counter

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + skip * (t1 - t0)
rows = [r for r in rows if r[0] >= cut]
busy, end = 0, rows[0][0]
gaps, before = [], Counter()
prev = None
for s, e, n in rows:
    if s > end:
        gaps.append(s - end)
        if prev and s - end > 5000:
            before[prev[:60]] += s - end
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    if e > end:
        end = e
        prev = n
wall = end - rows[0][0]
print(f"window {wall / 1e6:.2f} ms, {len(rows)} kernels, busy {busy / 1e6:.2f} ms = {busy / wall:.4f}")
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12)):
    g = [x for x in gaps if lo <= x < hi]
    print(f"  gaps {lo / 1e3:6.0f}..{hi / 1e3:8.0f} us: {len(g):6d}  total {sum(g) / 1e6:8.3f} ms")
print("largest idle time after:")
for n, t in before.most_common(8):
    print(f"  {t / 1e6:8.3f} ms  {n}")
