"""Batch-1 decode step in isolation: random und-expert weights at G2VLM-2B-MoT widths, a KV cache of --kv rows, the captured
step replayed --steps times.  Prints ms per token / tokens per second / achieved HBM GB/s for each requested variant:

    python tools/decode_bench.py --variants gen1,gen2 [--kv 10976] [--steps 300] [--layers 28]

(variant = decode kernel generation).  Under
`rocprofv3 --kernel-trace --stats` the same command gives the per-kernel table of profiles/r02*_decode_kernels.csv.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd import hip  # noqa: E402
from g2vlm_amd.engine import Engine, KVCache  # noqa: E402
from g2vlm_amd.synthetic import REAL_DIMS  # noqa: E402
from g2vlm_amd.weights import interleave_gate_up  # noqa: E402


class FakeWeights:
    """Just the tensors the decode step touches, drawn on the device."""

    def __init__(self, dims, device, layers):
        L = dims["llm"]
        H, Hq, Hkv, Fd, V = L["hidden"], L["heads"], L["kv_heads"], L["ffn"], L["vocab"]
        g = torch.Generator(device=device); g.manual_seed(0)
        r = lambda *s, sc=0.02: (torch.randn(s, generator=g, device=device) * sc)  # noqa: E731
        self.device, self.t = device, {}
        t = self.t
        t["embed"] = r(V, H, sc=1.0)
        t["lm_head"] = r(V, H).bfloat16()
        t["norm.und"] = torch.ones(H, device=device)
        t["inv_freq"] = (1.0 / (L["theta"] ** (torch.arange(0, 128, 2, dtype=torch.int64).float() / 128))).to(device)
        for i in range(layers):
            p = f"L{i}.und."
            t[p + "qkv.w"] = r((Hq + 2 * Hkv) * 128, H, sc=H ** -0.5).bfloat16()
            t[p + "qkv.b"] = r((Hq + 2 * Hkv) * 128).bfloat16()
            t[p + "o.w"] = r(H, Hq * 128, sc=H ** -0.5).bfloat16()
            t[p + "qn"] = torch.ones(128, device=device); t[p + "kn"] = torch.ones(128, device=device)
            t[p + "gu.w"] = interleave_gate_up(r(Fd, H, sc=H ** -0.5), r(Fd, H, sc=H ** -0.5)).bfloat16()
            t[p + "down.w"] = r(H, Fd, sc=Fd ** -0.5).bfloat16()
            t[p + "ln1"] = torch.ones(H, device=device); t[p + "ln2"] = torch.ones(H, device=device)

    def __getitem__(self, k):
        return self.t[k]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="gen1,gen2")
    ap.add_argument("--kv", type=int, default=10976)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", default="", help="comma list of B: time the BATCHED step (B copies of the cache) with the persistent-grid "
                                                "GEMVs (pgb) and with the skinny-GEMM body (gemm) instead of the batch-1 variants")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    import copy
    dims = copy.deepcopy(REAL_DIMS)
    dims["llm"]["layers"] = a.layers
    L = dims["llm"]
    w = FakeWeights(dims, dev, a.layers)
    eng = Engine(w, dims)
    cache = KVCache(a.layers, L["kv_heads"], dev, capacity=a.kv + a.steps * (a.rounds + 1) * 4 + 64)
    g = torch.Generator(device=dev); g.manual_seed(1)
    for i in range(a.layers):
        cache.k[i][:a.kv] = torch.randn((a.kv, L["kv_heads"], 128), generator=g, device=dev).bfloat16()
        cache.v[i][:a.kv] = torch.randn((a.kv, L["kv_heads"], 128), generator=g, device=dev).bfloat16()
    cache.length = a.kv
    wbytes = a.layers * 2 * (L["hidden"] * (L["heads"] + 2 * L["kv_heads"]) * 128 + L["hidden"] * L["heads"] * 128 + 3 * L["hidden"] * L["ffn"]) \
        + 2 * L["vocab"] * L["hidden"]
    kvbytes = a.layers * 2 * 2 * L["kv_heads"] * 128 * a.kv
    out = {"kv_len": a.kv, "layers": a.layers, "bytes_per_token": wbytes + kvbytes}
    if a.batch:
        for B in (int(v) for v in a.batch.split(",")):
            rec = {}
            for name, gen in (("gemm", 1), ("pgb", 2)):
                eng.decode_gen = gen                      # gen 1: skinny-GEMM Linears + separate norms; gen 2: gemv_pg_batch (B <= 8)
                cache.length = a.kv
                st = eng.decode_begin_batch([cache] * B, [5] * B, [a.kv] * B, a.steps * (a.rounds + 1) + 8, use_graph=True)
                ts = []
                for rd in range(a.rounds + 1):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(a.steps):
                        eng.decode_step_batch(st)
                    torch.cuda.synchronize()
                    if rd:
                        ts.append((time.perf_counter() - t0) / a.steps)
                best = min(ts)
                rec[name] = dict(ms_per_step=round(best * 1e3, 4), tokens_per_s=round(B / best, 1),
                                 hbm_gb_per_s=round((wbytes + B * kvbytes) / best / 1e9, 1))
                del st
            out[f"B{B}"] = rec
        print(json.dumps(out))
        return
    variants = a.variants.split(",")
    states = {}
    for v in variants:
        gen = int(v[3]) if v[:3] == "gen" and v[3:4].isdigit() else 2
        eng.decode_gen = gen
        eng._decode_cached.clear()
        cache.length = a.kv
        st = eng.decode_begin(cache, 5, a.kv, a.steps * (a.rounds + 1) + 8, use_graph=True)
        states[v] = (st, gen)
    res = {v: [] for v in variants}
    for rd in range(a.rounds + 1):
        for v in variants:
            st, gen = states[v]
            eng.decode_gen = gen
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                eng.decode_step(st)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            if rd > 0:
                res[v].append(dt)
    for v in variants:
        best = min(res[v])
        out[v] = dict(ms_per_token=round(best * 1e3, 4), tokens_per_s=round(1 / best, 1), hbm_gb_per_s=round((wbytes + kvbytes) / best / 1e9, 1),
                      frac_of_8TBps=round((wbytes + kvbytes) / best / 8e12, 4), all_ms=[round(x * 1e3, 4) for x in res[v]])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
