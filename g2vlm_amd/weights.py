"""Device-side weight store: takes the reference state dict (keys of SURVEY.md §8b, fp32 master
weights as `model.safetensors` holds them) and lays it out for the gfx950 kernels.

The reference rounds every Linear/conv weight AND bias to bf16 at each autocast call
(SURVEY App. C); doing that rounding once at load time is bit-identical.  Layout decisions:
  * q/k/v projections of one expert are concatenated row-wise -> one GEMM, N = (Hq+2Hkv)*128;
  * gate_proj / up_proj are interleaved in blocks of 16 rows so the SwiGLU epilogue finds the
    matching gate and up accumulators in the same lane (csrc/gemm.hip, G2V_EPI_SWIGLU);
  * the 14x14 patch convs become [C_out, K] matrices zero-padded to K % 64 == 0 (588->640,
    1176->1216) for the im2col GEMM;
  * norm weights, layer-scale gammas, position tables and the fp32-island heads stay fp32.
"""
import torch


def interleave_gate_up(wg, wu):
    """[F,K],[F,K] -> [2F,K]: rows 32b..32b+15 = gate rows 16b.., rows 32b+16..32b+31 = up rows 16b.."""
    Fd, K = wg.shape
    assert Fd % 16 == 0
    return torch.stack([wg.view(Fd // 16, 16, K), wu.view(Fd // 16, 16, K)], dim=1).reshape(2 * Fd, K).contiguous()


def pad_k(w2d, mult=64):
    n, k = w2d.shape
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w2d.contiguous()
    out = torch.zeros((n, kp), dtype=w2d.dtype, device=w2d.device)
    out[:, :k] = w2d
    return out


class _Prefixed:
    """sd[prefix + k] as sd[k]: hands one sub-module's tensors of a (possibly lazy) state dict to its own loader"""

    def __init__(self, sd, prefix):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def __contains__(self, k):
        return (self.prefix + k) in self.sd


class Weights:
    """Flat name -> device tensor store built from a CPU fp32 state dict."""

    def __init__(self, sd, dims, device):
        self.dims = dims
        self.device = device
        self.t = {}
        self._build(sd)

    def __getitem__(self, k):
        return self.t[k]

    def get(self, k, default=None):
        return self.t.get(k, default)

    def _bf(self, name, x):
        self.t[name] = x.to(torch.bfloat16).contiguous().to(self.device)

    def _f32(self, name, x):
        self.t[name] = x.to(torch.float32).contiguous().to(self.device)

    def _build(self, sd):
        L, Dn, V, K = self.dims["llm"], self.dims["dino"], self.dims["vit"], self.dims["dec"]
        g = lambda k: sd[k]                                                          # noqa: E731
        p = "language_model.model."
        self._f32("embed", g(p + "embed_tokens.weight"))
        self._bf("lm_head", g("language_model.lm_head.weight"))
        self._f32("norm.und", g(p + "norm.weight"))
        self._f32("norm.geo", g(p + "norm_moe_geo.weight"))
        for i in range(L["layers"]):
            q = f"{p}layers.{i}."
            for tag, sfx in (("und", ""), ("geo", "_moe_geo")):
                a = q + "self_attn."
                o = f"L{i}.{tag}."
                self._bf(o + "qkv.w", torch.cat([g(f"{a}q_proj{sfx}.weight"), g(f"{a}k_proj{sfx}.weight"),
                                                 g(f"{a}v_proj{sfx}.weight")], 0))
                self._bf(o + "qkv.b", torch.cat([g(f"{a}q_proj{sfx}.bias"), g(f"{a}k_proj{sfx}.bias"),
                                                 g(f"{a}v_proj{sfx}.bias")], 0))
                self._bf(o + "o.w", g(f"{a}o_proj{sfx}.weight"))
                self._f32(o + "qn", g(f"{a}q_norm{sfx}.weight"))
                self._f32(o + "kn", g(f"{a}k_norm{sfx}.weight"))
                self._bf(o + "gu.w", interleave_gate_up(g(f"{q}mlp{sfx}.gate_proj.weight"), g(f"{q}mlp{sfx}.up_proj.weight")))
                self._bf(o + "down.w", g(f"{q}mlp{sfx}.down_proj.weight"))
                self._f32(o + "ln1", g(f"{q}input_layernorm{sfx}.weight"))
                self._f32(o + "ln2", g(f"{q}post_attention_layernorm{sfx}.weight"))
            self._f32(f"L{i}.ls1", g(q + "ls1.gamma"))
            self._f32(f"L{i}.ls2", g(q + "ls2.gamma"))
        theta = L["theta"]
        self._f32("inv_freq", 1.0 / (theta ** (torch.arange(0, 128, 2, dtype=torch.int64).float() / 128)))

        self.dinov3 = None
        if Dn.get("v3"):
            # use_dinov3 (reference g2vlm.py:134): the encoder is the DINOv3 module with its own layouts
            from .modeling.dinov3.dinov3_model import DINOv3ViTConfig, DINOv3ViTModel
            self.dinov3 = DINOv3ViTModel(DINOv3ViTConfig(**Dn["v3"])).load_state_dict(_Prefixed(sd, "dino_model."), self.device)
        else:
            e = "dino_model.embeddings."
            w = g(e + "patch_embeddings.projection.weight")
            self._bf("dino.patch.w", pad_k(w.reshape(w.shape[0], -1)))
            self._bf("dino.patch.b", g(e + "patch_embeddings.projection.bias"))
            self._f32("dino.cls", g(e + "cls_token").reshape(-1))
            self._f32("dino.regs", g(e + "register_tokens").reshape(4, -1))
            self.dino_pos_cpu = g(e + "position_embeddings").float().cpu().clone()              # [1, 1+37*37, C], resampled on host
            for i in range(Dn["layers"]):
                q = f"dino_model.encoder.layer.{i}."
                o = f"D{i}."
                a = q + "attention.attention."
                self._bf(o + "qkv.w", torch.cat([g(a + "query.weight"), g(a + "key.weight"), g(a + "value.weight")], 0))
                self._bf(o + "qkv.b", torch.cat([g(a + "query.bias"), g(a + "key.bias"), g(a + "value.bias")], 0))
                self._bf(o + "dense.w", g(q + "attention.output.dense.weight")); self._bf(o + "dense.b", g(q + "attention.output.dense.bias"))
                self._bf(o + "fc1.w", g(q + "mlp.fc1.weight")); self._bf(o + "fc1.b", g(q + "mlp.fc1.bias"))
                self._bf(o + "fc2.w", g(q + "mlp.fc2.weight")); self._bf(o + "fc2.b", g(q + "mlp.fc2.bias"))
                for n in ("norm1", "norm2"):
                    self._f32(o + n + ".w", g(q + n + ".weight")); self._f32(o + n + ".b", g(q + n + ".bias"))
                self._f32(o + "ls1", g(q + "layer_scale1.lambda1")); self._f32(o + "ls2", g(q + "layer_scale2.lambda1"))
            self._f32("dino.ln.w", g("dino_model.layernorm.weight")); self._f32("dino.ln.b", g("dino_model.layernorm.bias"))
        self._bf("dino2llm.w", g("dino2llm.weight")); self._bf("dino2llm.b", g("dino2llm.bias"))

        # the confidence branch exists only in `train_conf_pi3` checkpoints (reference g2vlm.py:209-219)
        self.has_conf = "conf_head.proj.weight" in sd
        decs = [("point_decoder", False), ("camera_decoder", False), ("global_points_decoder", True)]
        if self.has_conf:
            decs.append(("conf_decoder", False))
            self._f32("conf_head.w", g("conf_head.proj.weight")); self._f32("conf_head.b", g("conf_head.proj.bias"))
        for name, cross in decs:
            for i in range(K["depth"]):
                q = f"{name}.blocks.{i}."
                o = f"{name}.{i}."
                for n in ["norm1", "norm2"] + (["norm_y", "norm3"] if cross else []):
                    self._f32(o + n + ".w", g(q + n + ".weight")); self._f32(o + n + ".b", g(q + n + ".bias"))
                for n in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
                    self._bf(o + n + ".w", g(q + n + ".weight")); self._bf(o + n + ".b", g(q + n + ".bias"))
                if cross:
                    c = q + "cross_attn."
                    self._bf(o + "cq.w", g(c + "q_proj.weight")); self._bf(o + "cq.b", g(c + "q_proj.bias"))
                    self._bf(o + "ckv.w", torch.cat([g(c + "k_proj.weight"), g(c + "v_proj.weight")], 0))
                    self._bf(o + "ckv.b", torch.cat([g(c + "k_proj.bias"), g(c + "v_proj.bias")], 0))
                    self._bf(o + "cproj.w", g(c + "proj.weight")); self._bf(o + "cproj.b", g(c + "proj.bias"))
            self._bf(name + ".out.w", g(name + ".linear_out.weight")); self._bf(name + ".out.b", g(name + ".linear_out.bias"))
        for n in ("point_head", "global_point_head"):
            self._f32(n + ".w", g(n + ".proj.weight")); self._f32(n + ".b", g(n + ".proj.bias"))
        for i in range(2):
            for j in (1, 2, 3):
                self._f32(f"cam.res{i}.{j}.w", g(f"camera_head.res_conv.{i}.res_conv{j}.weight"))
                self._f32(f"cam.res{i}.{j}.b", g(f"camera_head.res_conv.{i}.res_conv{j}.bias"))
        for j, n in ((0, "mlp0"), (2, "mlp1")):
            self._f32(f"cam.{n}.w", g(f"camera_head.more_mlps.{j}.weight")); self._f32(f"cam.{n}.b", g(f"camera_head.more_mlps.{j}.bias"))
        for n in ("fc_t", "fc_rot"):
            self._f32(f"cam.{n}.w", g(f"camera_head.{n}.weight")); self._f32(f"cam.{n}.b", g(f"camera_head.{n}.bias"))

        if "vit_model.patch_embed.proj.weight" in sd:
            w = g("vit_model.patch_embed.proj.weight")
            self._bf("vit.patch.w", pad_k(w.reshape(w.shape[0], -1)))
            for i in range(V["depth"]):
                q = f"vit_model.blocks.{i}."
                o = f"V{i}."
                for n in ("norm1", "norm2"):
                    self._f32(o + n + ".w", g(q + n + ".weight")); self._f32(o + n + ".b", g(q + n + ".bias"))
                for n in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
                    self._bf(o + n + ".w", g(q + n + ".weight")); self._bf(o + n + ".b", g(q + n + ".bias"))
            m = "vit_model.merger."
            self._f32("vit.ln_q.w", g(m + "ln_q.weight")); self._f32("vit.ln_q.b", g(m + "ln_q.bias"))
            self._bf("vit.m0.w", g(m + "mlp.0.weight")); self._bf("vit.m0.b", g(m + "mlp.0.bias"))
            self._bf("vit.m2.w", g(m + "mlp.2.weight")); self._bf("vit.m2.b", g(m + "mlp.2.bias"))
