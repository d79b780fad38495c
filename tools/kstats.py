"""Print the top rows of a rocprofv3 `--kernel-trace --stats --output-format csv` run:  python tools/kstats.py <dir> [n]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in list(csv.DictReader(open(f)))[:n]:
    print(f"{r['Name'][:78]:78s} calls {r['Calls']:>6s}  avg {float(r['AverageNs']) / 1e3:9.2f} us  {r['Percentage']:>6s} %")
