"""DINOv3 ViT-L/16 encoder throughput on one MI355X: N views of HxW through 24 layers, random-init weights.
    python3 tools/bench_dinov3.py [views] [H] [W] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g2vlm_amd.modeling.dinov3 import DINOv3ViTConfig, DINOv3ViTModel  # noqa: E402


def synth_sd(c, seed=0):
    g = torch.Generator(); g.manual_seed(seed)
    C, I, ps, R = c.hidden_size, c.intermediate_size, c.patch_size, c.num_register_tokens
    n = lambda *s, sc=0.02: torch.randn(*s, generator=g) * sc
    sd = {"embeddings.cls_token": n(1, 1, C), "embeddings.register_tokens": n(1, R, C), "embeddings.patch_embeddings.weight": n(C, 3, ps, ps),
          "embeddings.patch_embeddings.bias": n(C), "norm.weight": torch.ones(C), "norm.bias": torch.zeros(C)}
    for i in range(c.num_hidden_layers):
        p = f"layer.{i}."
        for nm in ("norm1", "norm2"):
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = torch.ones(C), torch.zeros(C)
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[p + f"attention.{nm}.weight"] = n(C, C, sc=C ** -0.5)
            if nm != "k_proj":
                sd[p + f"attention.{nm}.bias"] = n(C)
        sd[p + "layer_scale1.lambda1"], sd[p + "layer_scale2.lambda1"] = torch.ones(C), torch.ones(C)
        sd[p + "mlp.up_proj.weight"], sd[p + "mlp.up_proj.bias"] = n(I, C, sc=C ** -0.5), n(I)
        sd[p + "mlp.down_proj.weight"], sd[p + "mlp.down_proj.bias"] = n(C, I, sc=I ** -0.5), n(C)
    return sd


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    c = DINOv3ViTConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, num_register_tokens=4)
    model = DINOv3ViTModel(c).load_state_dict(synth_sd(c), "cuda")
    imgs = torch.randn((N, 3, H, W), device="cuda")
    P = (H // 16) * (W // 16)
    cu = [i * (P + 5) for i in range(N + 1)]
    for _ in range(3):
        model(imgs, cu)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        model(imgs, cu)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    T = N * (P + 5)
    flops = 24 * (2 * T * 1024 * (3 * 1024 + 1024 + 2 * 4096) + 4 * N * (P + 5) ** 2 * 1024) + 2 * N * P * 768 * 1024
    print(f"DINOv3 ViT-L/16, {N} views {H}x{W} ({P} patches): {dt * 1e3:.2f} ms, {N / dt:.1f} views/s, {flops / dt / 1e12:.0f} TFLOP/s")
