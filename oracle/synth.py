"""Seeded synthetic weights / inputs shared by the oracle, the fixture generator and tests.

TEST INFRASTRUCTURE (oracle/).  No checkpoint is available offline, so every parity vector is
made from these seeded tensors (SURVEY.md §8d "Synthetic inputs").  ``param_shapes`` is the
state-dict key contract of SURVEY.md §8(b): it is asserted equal to the reference model's own
``state_dict()`` in ``oracle/gen_golden.py``.

Each tensor is drawn from its own generator seeded by (seed, crc32(key)), so values do not
depend on enumeration order or on which model (reference / oracle / HIP engine) asks.
"""
import zlib

import torch


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
    """key -> shape for the full G2VLM state dict (modeling/g2vlm/g2vlm.py:123-243 et al.).
    conf=True adds the confidence branch a `train_conf_pi3` checkpoint ships (g2vlm.py:209-219): `conf_decoder` =
    a copy of `point_decoder`'s architecture, `conf_head` = Pi3LinearPts3d(output_dim=1)."""
    L, D, V, K = dims["llm"], dims["dino"], dims["vit"], dims["dec"]
    H, hd = L["hidden"], 128
    assert H // L["heads"] == hd, "LLM head_dim must be 128 (hard-coded mrope sections)"
    s = {}
    p = "language_model.model."
    s[p + "embed_tokens.weight"] = (L["vocab"], H)
    for i in range(L["layers"]):
        q = f"{p}layers.{i}."
        s[q + "ls1.gamma"] = (H,)
        s[q + "ls2.gamma"] = (H,)
        for sfx in ("", "_moe_geo"):
            a = q + "self_attn."
            s[f"{a}q_proj{sfx}.weight"] = (L["heads"] * hd, H); s[f"{a}q_proj{sfx}.bias"] = (L["heads"] * hd,)
            s[f"{a}k_proj{sfx}.weight"] = (L["kv_heads"] * hd, H); s[f"{a}k_proj{sfx}.bias"] = (L["kv_heads"] * hd,)
            s[f"{a}v_proj{sfx}.weight"] = (L["kv_heads"] * hd, H); s[f"{a}v_proj{sfx}.bias"] = (L["kv_heads"] * hd,)
            s[f"{a}o_proj{sfx}.weight"] = (H, L["heads"] * hd)
            s[f"{a}q_norm{sfx}.weight"] = (hd,); s[f"{a}k_norm{sfx}.weight"] = (hd,)
            m = f"{q}mlp{sfx}."
            s[m + "gate_proj.weight"] = (L["ffn"], H); s[m + "up_proj.weight"] = (L["ffn"], H)
            s[m + "down_proj.weight"] = (H, L["ffn"])
            s[f"{q}input_layernorm{sfx}.weight"] = (H,)
            s[f"{q}post_attention_layernorm{sfx}.weight"] = (H,)
    s[p + "norm.weight"] = (H,)
    s[p + "norm_moe_geo.weight"] = (H,)
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

    def block(q, cross):
        names = ["norm1", "norm2"] + (["norm_y", "norm3"] if cross else [])
        for n in names:
            s[f"{q}{n}.weight"] = (H,); s[f"{q}{n}.bias"] = (H,)
        s[q + "attn.qkv.weight"] = (3 * H, H); s[q + "attn.qkv.bias"] = (3 * H,)
        s[q + "attn.proj.weight"] = (H, H); s[q + "attn.proj.bias"] = (H,)
        if cross:
            for n in ("q_proj", "k_proj", "v_proj", "proj"):
                s[f"{q}cross_attn.{n}.weight"] = (H, H); s[f"{q}cross_attn.{n}.bias"] = (H,)
        s[q + "mlp.fc1.weight"] = (4 * H, H); s[q + "mlp.fc1.bias"] = (4 * H,)
        s[q + "mlp.fc2.weight"] = (H, 4 * H); s[q + "mlp.fc2.bias"] = (H,)

    for name, out, cross in (("point_decoder", 1024, False), ("camera_decoder", 512, False),
                             ("global_points_decoder", 1024, True)):
        for i in range(K["depth"]):
            block(f"{name}.blocks.{i}.", cross)
        s[f"{name}.linear_out.weight"] = (out, H); s[f"{name}.linear_out.bias"] = (out,)
    for n in ("point_head", "global_point_head"):
        s[n + ".proj.weight"] = (3 * pp, 1024); s[n + ".proj.bias"] = (3 * pp,)
    if conf:
        for i in range(K["depth"]):
            block(f"conf_decoder.blocks.{i}.", False)
        s["conf_decoder.linear_out.weight"] = (1024, H); s["conf_decoder.linear_out.bias"] = (1024,)
        s["conf_head.proj.weight"] = (pp, 1024); s["conf_head.proj.bias"] = (pp,)
    for i in range(2):
        for j in (1, 2, 3):
            s[f"camera_head.res_conv.{i}.res_conv{j}.weight"] = (512, 512)
            s[f"camera_head.res_conv.{i}.res_conv{j}.bias"] = (512,)
    for j in (0, 2):
        s[f"camera_head.more_mlps.{j}.weight"] = (512, 512); s[f"camera_head.more_mlps.{j}.bias"] = (512,)
    s["camera_head.fc_t.weight"] = (3, 512); s["camera_head.fc_t.bias"] = (3,)
    s["camera_head.fc_rot.weight"] = (9, 512); s["camera_head.fc_rot.bias"] = (9,)

    ve = V["embed"]
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


def _gen(seed, key):
    g = torch.Generator()
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 63))
    return g


def synth_tensor(key, shape, seed=0, jitter=True):
    """Value rule by key name.  ``jitter=False`` is the plain bench init of SURVEY §8d."""
    g = _gen(seed, key)
    r = lambda std: torch.randn(shape, generator=g) * std  # noqa: E731
    leaf = key.rsplit(".", 1)[-1]
    if key.endswith("ls1.gamma") or key.endswith("ls2.gamma"):
        # reference init 0.01 (qwen2vl.py:765-766); jittered so a wrong-row routing shows up
        return 0.25 * (1 + 0.2 * r(1.0)) if jitter else torch.full(shape, 0.01)
    if leaf == "lambda1":
        return 1.0 + (0.1 * r(1.0) if jitter else torch.zeros(shape))
    if leaf in ("cls_token", "register_tokens", "position_embeddings", "mask_token"):
        return r(0.02 if not jitter else 0.5)
    is_norm = ("norm" in key.split(".")[-2]) or key.split(".")[-2] in ("ln_q", "layernorm")
    if is_norm:
        if leaf == "weight":
            return 1.0 + (0.1 * r(1.0) if jitter else torch.zeros(shape))
        return 0.05 * r(1.0) if jitter else torch.zeros(shape)
    if leaf == "bias":
        return r(0.02)
    if key.endswith("embed_tokens.weight"):
        return r(0.02 if not jitter else 1.0)
    if leaf == "weight":
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        if jitter:
            # ~unit-gain layers keep activations O(1) through the stack, so softmaxes are not flat
            return r(1.0 / fan_in ** 0.5)
        return r(0.02)
    raise KeyError(key)


def synth_state_dict(dims, seed=0, jitter=True, shapes=None, threads=1):
    """threads > 1: tensors are drawn concurrently (each has its own generator seeded by (seed, key), so the values do not
    depend on the thread count); the full-depth dict is 3.4 G values, 80 s single-threaded."""
    shapes = shapes if shapes is not None else param_shapes(dims)
    make = lambda kv: (kv[0], synth_tensor(kv[0], tuple(kv[1]), seed, jitter).float().contiguous())  # noqa: E731
    if threads <= 1:
        return dict(map(make, shapes.items()))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return dict(ex.map(make, shapes.items()))


def peaked_lm_head(sd, sigma, seed):
    """Give the lm_head rows log-normal scales exp(sigma n_i) (in place; returns sd).  Random-init logits are flat: over a
    2048-row vocabulary the top-2 gap is below 4 bf16 ulp at one step in five, so a greedy decode cannot be compared token
    for token (hazard H2).  With heavy-tailed row scales the winner at each step comes from a few dozen tokens with a
    clear gap (2 % of steps below 4 ulp at sigma 2), which lets a fixture demand exact ids for its whole length."""
    g = torch.Generator(); g.manual_seed(977 * seed + 13)
    w = sd["language_model.lm_head.weight"]
    sd["language_model.lm_head.weight"] = (w * torch.exp(sigma * torch.randn(w.shape[0], generator=g)).unsqueeze(1)).contiguous()
    return sd


def synth_images(n, h, w, seed=0):
    """U[0,1) fp32 [n,3,h,w]; generator seeded per view."""
    out = []
    for v in range(n):
        g = torch.Generator(); g.manual_seed(seed * 7919 + v)
        out.append(torch.rand((3, h, w), generator=g))
    return torch.stack(out, 0)


class FakeTokenizer:
    """Deterministic stand-in for Qwen2Tokenizer (vocab files are not available offline).

    encode(): one id per UTF-8 byte, offset past the 4 special ids; the specials map to fixed ids.
    """
    SPECIALS = {"<|im_start|>": 1, "<|im_end|>": 2, "<|vision_start|>": 3, "<|vision_end|>": 4}

    def __init__(self, vocab):
        self.vocab = vocab
        self.eos_token_id = 2

    def encode(self, text, add_special_tokens=False):
        ids, i = [], 0
        while i < len(text):
            for tok, tid in self.SPECIALS.items():
                if text.startswith(tok, i):
                    ids.append(tid); i += len(tok); break
            else:
                for b in text[i].encode("utf-8"):
                    ids.append(5 + (b % (self.vocab - 5)))
                i += 1
        return ids

    def decode(self, ids):
        return " ".join(str(int(x)) for x in ids)

    @property
    def new_token_ids(self):
        return dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)
