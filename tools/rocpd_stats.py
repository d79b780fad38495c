"""Per-kernel summary of a rocprofv3 run stored as a rocpd SQLite database (rocprofv3's default output on ROCm 7.2):
the same columns as `--stats`' kernel_stats.csv, for committing under profiles/.

    python3 tools/rocpd_stats.py gpurun_out/prof_rN/rN_results.db [steps] > profiles/rNN_kernel_stats.csv
With `steps` given, a trailer line reports the kernel time of OUR kernels per bench step (recon kernels only)."""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", name)[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = list(db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    recon = 0
    for n, c, t, a, mn, mx in rows:
        print(f'"{short(n)}",{c},{t},{a:.1f},{100.0 * t / tot:.2f},{mn},{mx}')
        ours = "anonymous namespace" in n and "at::" not in n
        if ours and not any(k in n for k in ("gemv", "decode_", "argmax", "swiglu_bf16_kernel", "embed_rows")):
            recon += t
    if steps:
        print(f'"# recon kernels of this library: {recon / steps / 1e6:.2f} ms per step over {steps} steps (warm-up included)"')


if __name__ == "__main__":
    main()
