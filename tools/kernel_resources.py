"""Print VGPR / scratch / LDS / occupancy per kernel of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

src = sys.argv[1]
root = __file__.rsplit("/tools/", 1)[0]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{root}/include", f"-I{root}/g2vlm_amd/csrc",
                      "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + (["-fno-slp-vectorize"] if src.endswith("attn.hip") else []), capture_output=True, text=True).stderr
cur = None
rows = []
for ln in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", ln)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    print(f"{r['name'][:70]:70s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} "
          f"lds {r.get('LDS Size [bytes/block]','?'):>6s} occ {r.get('Occupancy [waves/SIMD]','?')}")
