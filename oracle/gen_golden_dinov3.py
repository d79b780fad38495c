"""Golden vectors for the DINOv3 encoder, produced by the REFERENCE module itself.

TEST INFRASTRUCTURE ONLY; runs in the build container (needs /root/reference), never on the GPU box.

    python -m oracle.gen_golden_dinov3            # check the oracle against the reference, write tests/golden/dinov3_*.{safetensors,json}

`modeling/dinov3/dinov3_model.py::DINOv3ViTModel` is imported through oracle/ref_shim.py (bf16 CPU autocast standing in for
CUDA autocast, flash-attn replaced by ref_shim's fp32-softmax varlen restatement), loaded with the seeded weights of
oracle/dinov3_oracle.py::synth_state_dict and run on seeded images.  cu_seqlens are built as the reference's callers build
them (g2vlm.py:348-350, 988-990: cumulative PATCH counts, although every view has 1 + R more tokens - hazard H1) and, for a
second fixture, as clean per-view windows.  Fixtures hold inputs by seed, the reference's patch tokens and a few
intermediate tensors; the oracle must reproduce them bit for bit before anything is written.
"""
import importlib
import json
import os
import sys

import torch
from safetensors.torch import save_file

from oracle import dinov3_oracle as O3
from oracle import ref_shim

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

FIXTURES = {
    # name: (config overrides, n, h, w, seed, cu mode)
    "dinov3_tiny": (dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_register_tokens=4),
                    3, 48, 64, 11, "h1"),
    "dinov3_tiny_clean": (dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_register_tokens=4),
                          2, 64, 32, 12, "per_view"),
    # no register tokens (the config default), odd grid
    "dinov3_tiny_r0": (dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_register_tokens=0),
                       2, 80, 48, 15, "h1"),
    # ViT-H+-style gated MLP (SiLU gate, no MLP biases)
    "dinov3_tiny_gated": (dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_register_tokens=4,
                               use_gated_mlp=True, hidden_act="silu", mlp_bias=False), 2, 48, 48, 14, "h1"),
    # ViT-L/16 widths (hidden 1024, 16 heads, MLP 4096, 4 registers), 2 layers
    "dinov3_real2": (dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=2, num_attention_heads=16, num_register_tokens=4),
                     2, 224, 224, 13, "h1"),
}


def cu_for(mode, n, P, S):
    if mode == "h1":                                       # what G2VLM passes: cumsum of per-view PATCH counts
        return [i * P for i in range(n + 1)]
    return [i * S for i in range(n + 1)]


def run_reference(cfg, sd, images, cu):
    ref_shim.install()
    m3 = importlib.import_module("modeling.dinov3.dinov3_model")
    conf = m3.DINOv3ViTConfig(**{k: v for k, v in cfg.items()}, image_size=224)
    model = m3.DINOv3ViTModel(conf).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("inv_freq" in k for k in missing), (missing, unexpected)
    with torch.no_grad(), torch.amp.autocast("cpu", dtype=torch.bfloat16):
        return model(pixel_values=images, cu_seqlens=torch.tensor(cu, dtype=torch.int32), max_seqlen=max(b - a for a, b in zip(cu, cu[1:])))


def main():
    os.makedirs(GOLD, exist_ok=True)
    for name, (over, n, h, w, seed, mode) in FIXTURES.items():
        cfg = O3.default_config(**over)
        sd = O3.synth_state_dict(cfg, seed)
        images = O3.synth_images(n, h, w, seed)
        P = (h // cfg["patch_size"]) * (w // cfg["patch_size"])
        S = P + 1 + cfg["num_register_tokens"]
        cu = cu_for(mode, n, P, S)
        ref = run_reference(cfg, sd, images, cu).float()
        mine = O3.forward(sd, cfg, images, cu)
        exact = torch.equal(ref, mine)
        err = float((ref - mine).norm() / ref.norm())
        print(f"{name}: ref {tuple(ref.shape)}  oracle == reference: {exact}  rel {err:.3e}")
        assert exact, "oracle restatement differs from the reference module"
        save_file({"ref.patch_tokens": ref.contiguous()}, os.path.join(GOLD, name + ".safetensors"))
        json.dump(dict(cfg=cfg, n=n, h=h, w=w, seed=seed, cu=cu, cu_mode=mode,
                       note="reference DINOv3ViTModel.forward (modeling/dinov3/dinov3_model.py) on CPU, bf16 autocast, ref_shim flash-attn; "
                            "weights oracle.dinov3_oracle.synth_state_dict(cfg, seed), images synth_images(n, h, w, seed)"),
                  open(os.path.join(GOLD, name + ".json"), "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
