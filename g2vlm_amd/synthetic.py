"""State-dict key contract of the reference checkpoint and a lazy random-init state dict.

No checkpoint is reachable offline, so benchmarks and smoke tests run on random-init weights of
the exact G2VLM-2B-MoT architecture.  `param_shapes` enumerates every key the
reference's `model.state_dict()` holds (verified against the reference classes in
tests/ via the golden key list); `SyntheticStateDict` materialises each tensor on demand directly
on the target device, so the 4.5 B-parameter model never exists as an fp32 copy on the host.
"""
import zlib

import torch

REAL_DIMS = {
    "llm": dict(hidden=1536, layers=28, heads=12, kv_heads=2, ffn=8960, vocab=151936, eps=1e-6, theta=1e6),
    "dino": dict(hidden=1024, layers=24, heads=16),
    "vit": dict(embed=1280, depth=32, heads=16, mlp_ratio=4, out=1536),
    "dec": dict(depth=5, heads=16),
}


def dinov3_shapes(cfg, prefix="dino_model."):
    """Parameter names / shapes of the reference's DINOv3ViTModel module tree (modeling/dinov3/dinov3_model.py:491-500) for
    a config dict with DINOv3ViTConfig's fields (the persistent `rope_embeddings.inv_freq` buffer is derived, not stored)."""
    C, I, ps, R = cfg["hidden_size"], cfg["intermediate_size"], cfg["patch_size"], cfg["num_register_tokens"]
    s = {}
    e = prefix + "embeddings."
    s[e + "cls_token"] = (1, 1, C); s[e + "mask_token"] = (1, 1, C); s[e + "register_tokens"] = (1, R, C)
    s[e + "patch_embeddings.weight"] = (C, 3, ps, ps); s[e + "patch_embeddings.bias"] = (C,)
    for i in range(cfg["num_hidden_layers"]):
        q = f"{prefix}layer.{i}."
        for n in ("norm1", "norm2"):
            s[q + n + ".weight"] = (C,); s[q + n + ".bias"] = (C,)
        for n, has_b in (("q_proj", cfg["query_bias"]), ("k_proj", cfg["key_bias"]), ("v_proj", cfg["value_bias"]),
                         ("o_proj", cfg["proj_bias"])):
            s[f"{q}attention.{n}.weight"] = (C, C)
            if has_b:
                s[f"{q}attention.{n}.bias"] = (C,)
        s[q + "layer_scale1.lambda1"] = (C,); s[q + "layer_scale2.lambda1"] = (C,)
        for n in (("gate_proj", "up_proj") if cfg["use_gated_mlp"] else ("up_proj",)):
            s[f"{q}mlp.{n}.weight"] = (I, C)
            if cfg["mlp_bias"]:
                s[f"{q}mlp.{n}.bias"] = (I,)
        s[q + "mlp.down_proj.weight"] = (C, I)
        if cfg["mlp_bias"]:
            s[q + "mlp.down_proj.bias"] = (C,)
    s[prefix + "norm.weight"] = (C,); s[prefix + "norm.bias"] = (C,)
    return s


def param_shapes(dims, conf=False):
    """conf=True: plus the confidence branch of a `train_conf_pi3` checkpoint (reference g2vlm.py:209-219)."""
    L, D, V, K = dims["llm"], dims["dino"], dims["vit"], dims["dec"]
    H, hd = L["hidden"], 128
    s = {}
    p = "language_model.model."
    s[p + "embed_tokens.weight"] = (L["vocab"], H)
    for i in range(L["layers"]):
        q = f"{p}layers.{i}."
        s[q + "ls1.gamma"] = (H,); s[q + "ls2.gamma"] = (H,)
        for sfx in ("", "_moe_geo"):
            a = q + "self_attn."
            for n, rows in (("q", L["heads"] * hd), ("k", L["kv_heads"] * hd), ("v", L["kv_heads"] * hd)):
                s[f"{a}{n}_proj{sfx}.weight"] = (rows, H); s[f"{a}{n}_proj{sfx}.bias"] = (rows,)
            s[f"{a}o_proj{sfx}.weight"] = (H, L["heads"] * hd)
            s[f"{a}q_norm{sfx}.weight"] = (hd,); s[f"{a}k_norm{sfx}.weight"] = (hd,)
            m = f"{q}mlp{sfx}."
            s[m + "gate_proj.weight"] = (L["ffn"], H); s[m + "up_proj.weight"] = (L["ffn"], H); s[m + "down_proj.weight"] = (H, L["ffn"])
            s[f"{q}input_layernorm{sfx}.weight"] = (H,); s[f"{q}post_attention_layernorm{sfx}.weight"] = (H,)
    s[p + "norm.weight"] = (H,); s[p + "norm_moe_geo.weight"] = (H,)
    s["language_model.lm_head.weight"] = (L["vocab"], H)
    pp = D.get("patch", 14) ** 2                          # Pi3LinearPts3d(patch_size=14 | 16), g2vlm.py:169-172
    dh = D["hidden"]
    if D.get("v3"):                                        # use_dinov3 (reference g2vlm.py:134, 169-172)
        s.update(dinov3_shapes(D["v3"]))
    else:
        e = "dino_model.embeddings."
        s[e + "cls_token"] = (1, 1, dh); s[e + "mask_token"] = (1, dh); s[e + "register_tokens"] = (1, 4, dh)
        s[e + "position_embeddings"] = (1, 37 * 37 + 1, dh)
        s[e + "patch_embeddings.projection.weight"] = (dh, 3, 14, 14); s[e + "patch_embeddings.projection.bias"] = (dh,)
        for i in range(D["layers"]):
            q = f"dino_model.encoder.layer.{i}."
            for n in ("norm1", "norm2"):
                s[q + n + ".weight"] = (dh,); s[q + n + ".bias"] = (dh,)
            for n in ("query", "key", "value"):
                s[f"{q}attention.attention.{n}.weight"] = (dh, dh); s[f"{q}attention.attention.{n}.bias"] = (dh,)
            s[q + "attention.output.dense.weight"] = (dh, dh); s[q + "attention.output.dense.bias"] = (dh,)
            s[q + "layer_scale1.lambda1"] = (dh,); s[q + "layer_scale2.lambda1"] = (dh,)
            s[q + "mlp.fc1.weight"] = (4 * dh, dh); s[q + "mlp.fc1.bias"] = (4 * dh,)
            s[q + "mlp.fc2.weight"] = (dh, 4 * dh); s[q + "mlp.fc2.bias"] = (dh,)
        s["dino_model.layernorm.weight"] = (dh,); s["dino_model.layernorm.bias"] = (dh,)
    s["dino2llm.weight"] = (H, dh); s["dino2llm.bias"] = (H,)
    decs = [("point_decoder", 1024, False), ("camera_decoder", 512, False), ("global_points_decoder", 1024, True)]
    if conf:
        decs.append(("conf_decoder", 1024, False))
        s["conf_head.proj.weight"] = (pp, 1024); s["conf_head.proj.bias"] = (pp,)
    for name, out, cross in decs:
        for i in range(K["depth"]):
            q = f"{name}.blocks.{i}."
            for n in ["norm1", "norm2"] + (["norm_y", "norm3"] if cross else []):
                s[f"{q}{n}.weight"] = (H,); s[f"{q}{n}.bias"] = (H,)
            s[q + "attn.qkv.weight"] = (3 * H, H); s[q + "attn.qkv.bias"] = (3 * H,)
            s[q + "attn.proj.weight"] = (H, H); s[q + "attn.proj.bias"] = (H,)
            if cross:
                for n in ("q_proj", "k_proj", "v_proj", "proj"):
                    s[f"{q}cross_attn.{n}.weight"] = (H, H); s[f"{q}cross_attn.{n}.bias"] = (H,)
            s[q + "mlp.fc1.weight"] = (4 * H, H); s[q + "mlp.fc1.bias"] = (4 * H,)
            s[q + "mlp.fc2.weight"] = (H, 4 * H); s[q + "mlp.fc2.bias"] = (H,)
        s[f"{name}.linear_out.weight"] = (out, H); s[f"{name}.linear_out.bias"] = (out,)
    for n in ("point_head", "global_point_head"):
        s[n + ".proj.weight"] = (3 * pp, 1024); s[n + ".proj.bias"] = (3 * pp,)
    for i in range(2):
        for j in (1, 2, 3):
            s[f"camera_head.res_conv.{i}.res_conv{j}.weight"] = (512, 512); s[f"camera_head.res_conv.{i}.res_conv{j}.bias"] = (512,)
    for j in (0, 2):
        s[f"camera_head.more_mlps.{j}.weight"] = (512, 512); s[f"camera_head.more_mlps.{j}.bias"] = (512,)
    s["camera_head.fc_t.weight"] = (3, 512); s["camera_head.fc_t.bias"] = (3,)
    s["camera_head.fc_rot.weight"] = (9, 512); s["camera_head.fc_rot.bias"] = (9,)
    ve = V["embed"]
    if V["depth"] > 0:
        s["vit_model.patch_embed.proj.weight"] = (ve, 3, 2, 14, 14)
        for i in range(V["depth"]):
            q = f"vit_model.blocks.{i}."
            for n in ("norm1", "norm2"):
                s[q + n + ".weight"] = (ve,); s[q + n + ".bias"] = (ve,)
            s[q + "attn.qkv.weight"] = (3 * ve, ve); s[q + "attn.qkv.bias"] = (3 * ve,)
            s[q + "attn.proj.weight"] = (ve, ve); s[q + "attn.proj.bias"] = (ve,)
            hdim = int(ve * V["mlp_ratio"])
            s[q + "mlp.fc1.weight"] = (hdim, ve); s[q + "mlp.fc1.bias"] = (hdim,)
            s[q + "mlp.fc2.weight"] = (ve, hdim); s[q + "mlp.fc2.bias"] = (ve,)
        s["vit_model.merger.ln_q.weight"] = (ve,); s["vit_model.merger.ln_q.bias"] = (ve,)
        s["vit_model.merger.mlp.0.weight"] = (4 * ve, 4 * ve); s["vit_model.merger.mlp.0.bias"] = (4 * ve,)
        s["vit_model.merger.mlp.2.weight"] = (V["out"], 4 * ve); s["vit_model.merger.mlp.2.bias"] = (V["out"],)
    return s


class SyntheticStateDict:
    """dict-like: sd[key] -> fp32 tensor on `device`, generated at access time (SURVEY §8d init:
    Linear/conv N(0,0.02), norms weight 1 / bias 0, MoT ls gamma 0.01, DINO lambda 1, embeddings and
    position tables N(0,0.02); dino2llm N(0,0.02) instead of the reference's zero init, which would
    null the geo path)."""

    def __init__(self, dims, device, seed=0):
        self.shapes = param_shapes(dims)
        self.device, self.seed = device, seed

    def __contains__(self, k):
        return k in self.shapes

    def keys(self):
        return self.shapes.keys()

    def items(self):
        return ((k, self[k]) for k in self.shapes)

    def __getitem__(self, key):
        shape = self.shapes[key]
        leaf = key.rsplit(".", 1)[-1]
        parent = key.split(".")[-2] if "." in key else ""
        g = torch.Generator(device=self.device)
        g.manual_seed((self.seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 63))
        if leaf == "gamma":
            return torch.full(shape, 0.01, device=self.device)
        if leaf == "lambda1":
            return torch.ones(shape, device=self.device)
        if ("norm" in parent or parent in ("ln_q", "layernorm")) and leaf in ("weight", "bias"):
            return torch.ones(shape, device=self.device) if leaf == "weight" else torch.zeros(shape, device=self.device)
        if leaf == "bias":
            return torch.zeros(shape, device=self.device)
        return torch.randn(shape, generator=g, device=self.device) * 0.02
