"""Diagnostic: per-step GPU times of the C3 bench step over many steps, with the caching allocator's device-malloc count around the loop
(a hipMalloc in steady state is a device-wide stall).      python3 tools/step_outliers.py [steps]"""
import os
import subprocess
import sys

if __name__ == "__main__":
    steps = sys.argv[1] if len(sys.argv) > 1 else "40"
    env = dict(os.environ, G2V_STEP_DIAG="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--decode-tokens", "0", "--overlap", "1", "--steps", steps],
                       capture_output=True, text=True, env=env)
    import json
    d = json.loads(r.stdout.strip().splitlines()[-1])
    ms = d["step_ms"]
    print("steps", len(ms), "median", sorted(ms)[len(ms) // 2], "max", max(ms), "mean", round(sum(ms) / len(ms), 2))
    print("first five steps:", ms[:5])
    print("outliers (> 1.1 x median):", [(i, v) for i, v in enumerate(ms) if v > 1.1 * sorted(ms)[len(ms) // 2]])
    for ln in r.stderr.splitlines():
        if "G2V_STEP_DIAG" in ln:
            print(ln)
