"""GPU: the DINOv3 encoder (g2vlm_amd/modeling/dinov3) against the patch tokens the reference's own
DINOv3ViTModel produced (tests/golden/dinov3_*, made by oracle/gen_golden_dinov3.py) and against the CPU oracle.

Tolerance: the golden vectors are a bf16-autocast run; an engine that differs only in fp32 accumulation order and in
rounding the rotated q/k to bf16 for the attention kernel (the reference hands them over in fp32, which a real flash
kernel does not accept) disagrees at the bf16 rounding level.  Checked two ways, as for DINOv2 (tests/test_e2e_gpu.py):
rel-L2 against the golden tokens < 2e-2, and error against a full-precision evaluation of the same network no larger
than 1.5 x the bf16 reference's own error against it."""
import json
import os

import pytest
import torch
import torch.nn.functional as F
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu

from oracle import dinov3_oracle as O3  # noqa: E402  (checker only)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def precise_forward(sd, cfg, images, cu):
    """the same network in fp32 weights / activations and fp64 attention: the yardstick both bf16 runs are measured against"""
    C, nh, R, ps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_register_tokens"], cfg["patch_size"]
    B = images.shape[0]
    pe = F.conv2d(images.float(), sd["embeddings.patch_embeddings.weight"], sd["embeddings.patch_embeddings.bias"], stride=ps).flatten(2).transpose(1, 2)
    x = torch.cat([sd["embeddings.cls_token"].expand(B, -1, -1), sd["embeddings.register_tokens"].expand(B, -1, -1), pe], 1)
    S = x.shape[1]
    x = x.reshape(B * S, C)
    cos, sin = O3.rope_cos_sin(images.shape[2] // ps, images.shape[3] // ps, C // nh, cfg["rope_theta"])
    lin = lambda t, n: F.linear(t, sd[n + ".weight"], sd.get(n + ".bias"))
    from oracle.g2vlm_oracle import varlen_attention
    for i in range(cfg["num_hidden_layers"]):
        p = f"layer.{i}."
        h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg["layer_norm_eps"])
        q, k, v = (lin(h, p + f"attention.{n}_proj").view(B, S, nh, -1).transpose(1, 2) for n in "qkv")
        q, k = O3.rope_patches(q, cos, sin, 1 + R), O3.rope_patches(k, cos, sin, 1 + R)
        q, k, v = (t.transpose(1, 2).reshape(B * S, nh, -1) for t in (q, k, v))
        ctx = varlen_attention(q, k, v, cu, cu, False, precise=True).reshape(B * S, -1)
        x = lin(ctx, p + "attention.o_proj") * sd[p + "layer_scale1.lambda1"] + x
        h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg["layer_norm_eps"])
        if cfg["use_gated_mlp"]:
            m = lin(F.silu(lin(h, p + "mlp.gate_proj")) * lin(h, p + "mlp.up_proj"), p + "mlp.down_proj")
        else:
            m = lin(F.gelu(lin(h, p + "mlp.up_proj")), p + "mlp.down_proj")
        x = m * sd[p + "layer_scale2.lambda1"] + x
    return F.layer_norm(x, (C,), sd["norm.weight"], sd["norm.bias"], cfg["layer_norm_eps"]).reshape(B, S, C)[:, 1 + R:]


@pytest.mark.parametrize("name", ["dinov3_tiny", "dinov3_tiny_clean", "dinov3_tiny_r0", "dinov3_tiny_gated", "dinov3_real2"])
def test_dinov3_matches_reference_golden(golden_dir, name):
    from g2vlm_amd.modeling.dinov3 import DINOv3ViTConfig, DINOv3ViTModel
    with open(os.path.join(golden_dir, name + ".json")) as f:
        meta = json.load(f)
    ref = load_file(os.path.join(golden_dir, name + ".safetensors"))["ref.patch_tokens"]
    cfg = meta["cfg"]
    sd = O3.synth_state_dict(cfg, meta["seed"])
    images = O3.synth_images(meta["n"], meta["h"], meta["w"], meta["seed"])
    model = DINOv3ViTModel(DINOv3ViTConfig(**cfg)).load_state_dict(sd, "cuda")
    cu = torch.tensor(meta["cu"], dtype=torch.int32)
    out = model(pixel_values=images, cu_seqlens=cu, max_seqlen=int((cu[1:] - cu[:-1]).max()))
    assert out.shape == ref.shape and out.dtype == torch.float32 and torch.isfinite(out).all()
    r = rel(out, ref)
    prec = precise_forward(sd, cfg, images, meta["cu"])
    e_mine, e_ref = rel(out, prec), rel(ref, prec)
    print(name, f"rel vs golden {r:.3e}; vs full precision: engine {e_mine:.3e}, reference {e_ref:.3e}")
    assert r < 2e-2
    assert e_mine <= 1.5 * e_ref + 1e-6
    # deterministic, and a second call (cached rope rows / attention plan) is bit-identical
    assert torch.equal(out, model(pixel_values=images, cu_seqlens=cu, max_seqlen=0))


def test_dinov3_front_end_kernels():
    """g2v_im2col_patch + g2v_vit_assemble against torch: the patch gather is exact (bf16 rounding of fp32 pixels), the
    assembled token matrix is [cls | registers | conv rows] per view."""
    from g2vlm_amd import hip
    g = torch.Generator(); g.manual_seed(5)
    N, H, W, ps, C, R = 3, 32, 48, 16, 64, 4
    img = torch.randn((N, 3, H, W), generator=g)
    cols = hip.im2col_patch(img.cuda(), ps, 768)
    want = F.unfold(img, ps, stride=ps).transpose(1, 2).reshape(-1, 3 * ps * ps).bfloat16()
    assert torch.equal(cols.cpu(), want)
    cols2 = hip.im2col_patch(img.cuda(), ps, 832)                        # zero-padded K
    assert torch.equal(cols2[:, :768].cpu(), want) and float(cols2[:, 768:].abs().max()) == 0
    P = (H // ps) * (W // ps)
    patch = torch.randn((N * P, C), generator=g).bfloat16()
    cls, regs = torch.randn(C, generator=g), torch.randn((R, C), generator=g)
    x = hip.vit_assemble(patch.cuda(), cls.cuda(), regs.cuda(), N, P, R).cpu().view(N, P + 1 + R, C)
    for n in range(N):
        assert torch.equal(x[n, 0], cls) and torch.equal(x[n, 1:1 + R], regs)
        assert torch.equal(x[n, 1 + R:], patch[n * P:(n + 1) * P].float())
    x0 = hip.vit_assemble(patch.cuda(), cls.cuda(), None, N, P, 0).cpu().view(N, P + 1, C)     # no registers
    assert torch.equal(x0[:, 0], cls.expand(N, -1)) and torch.equal(x0[:, 1:].reshape(-1, C), patch.float())


def test_dinov3_full_width_properties():
    """ViT-L/16 widths at full depth (24 layers) on 4 views of 512x512 (1024 patches each): finite fp32 tokens of the right
    shape, per-view independence under clean per-view windows (a view's tokens do not change when other views change),
    and the H1 layout's signature: the rows past n*P of the token axis get no attention."""
    from g2vlm_amd.modeling.dinov3 import DINOv3ViTConfig, DINOv3ViTModel
    cfg = O3.default_config(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, num_register_tokens=4)
    sd = O3.synth_state_dict(cfg, 3)
    model = DINOv3ViTModel(DINOv3ViTConfig(**cfg)).load_state_dict(sd, "cuda")
    imgs = O3.synth_images(4, 512, 512, 9)
    P, S = 1024, 1029
    clean = [i * S for i in range(5)]
    a = model(imgs, clean)
    assert a.shape == (4, P, 1024) and torch.isfinite(a).all()
    imgs2 = imgs.clone(); imgs2[2:] = O3.synth_images(2, 512, 512, 10)
    b = model(imgs2, clean)
    assert torch.equal(a[:2], b[:2]) and not torch.equal(a[2:], b[2:])
    h1 = model(imgs, [i * P for i in range(5)])
    assert torch.isfinite(h1).all() and not torch.equal(h1, a)
