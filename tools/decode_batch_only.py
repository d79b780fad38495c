"""Batched decode steps on synthetic caches at full width/depth (target of rocprofv3 --kernel-trace, and a wall-clock
timer of the graph-replayed step without the cache packing / capture that bench.py's decode_batch includes).
    python3 tools/decode_batch_only.py [B] [kv_len] [steps] [eager|graph]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd.engine import KVCache  # noqa: E402
from g2vlm_amd.g2vlm_utils import build_model, configs_from_dims  # noqa: E402
from g2vlm_amd.synthetic import REAL_DIMS, SyntheticStateDict  # noqa: E402

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    kv = int(sys.argv[2]) if len(sys.argv) > 2 else 10976
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    mode = sys.argv[4] if len(sys.argv) > 4 else "graph"
    dims, dev = REAL_DIMS, torch.device("cuda", 0)
    model = build_model(*configs_from_dims(dims), SyntheticStateDict(dims, dev, seed=0), dev)
    eng = model.engine
    L = dims["llm"]
    cache = KVCache(L["layers"], L["kv_heads"], dev, capacity=kv + 8)
    g = torch.Generator(device=dev); g.manual_seed(0)
    for i in range(L["layers"]):
        cache.k[i].copy_(torch.randn(cache.k[i].shape, device=dev, generator=g).bfloat16())
        cache.v[i].copy_(torch.randn(cache.v[i].shape, device=dev, generator=g).bfloat16())
    cache.length = kv
    if B == 1:
        st = eng.decode_begin(cache, 5, kv, steps + 4, use_graph=mode == "graph")
        step = lambda: eng.decode_step(st)
    else:
        st = eng.decode_begin_batch([cache] * B, [5] * B, [kv] * B, steps + 4, use_graph=mode == "graph")
        step = lambda: eng.decode_step_batch(st)
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    w_bytes = L["layers"] * 2 * (L["hidden"] * (L["heads"] + 2 * L["kv_heads"]) * 128 + L["heads"] * 128 * L["hidden"]
                                 + 3 * L["hidden"] * L["ffn"]) + 2 * L["vocab"] * L["hidden"]
    kv_bytes = L["layers"] * 2 * L["kv_heads"] * 128 * 2 * kv
    print(f"B {B} kv {kv} {mode}: {dt * 1e3:.3f} ms/step, {B / dt:.0f} tok/s, {(w_bytes + B * kv_bytes) / dt / 1e9:.0f} GB/s algorithmic")
