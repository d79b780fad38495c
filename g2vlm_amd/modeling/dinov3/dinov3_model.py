"""DINOv3 ViT encoder on libg2vlm_hip.so - the `use_dinov3` variant of the geometry encoder.

Mirrors the reference module `modeling/dinov3/dinov3_model.py` at its own boundary:
`DINOv3ViTModel.forward(pixel_values, cu_seqlens, max_seqlen) -> patch tokens [B, P, C]` (:491-543), state-dict keys as
the reference's module tree names them.  Differences from the DINOv2 encoder that matter to the kernels: Conv2d 16/16
(`g2v_im2col_patch`), no position table - RoPE on the patch tokens of q and k from patch-centre coordinates in
[-1, 1] (:72-97, 129-176, 216-246), `R` register tokens from the config, LayerNorm eps 1e-5, separate q/k/v Linears
(k without bias) which are concatenated into one qkv GEMM here, plain or gated MLP.

Numerics (reference under autocast(bf16)): Linears / conv in bf16, LayerNorm and LayerScale in fp32, residual stream
fp32.  RoPE is evaluated in fp32 on the bf16 q/k; the reference hands the fp32 result to flash_attn_varlen_func, which
only takes fp16/bf16 - this engine rounds the rotated q/k to bf16 once (the only deviation, documented in DESIGN.md).
The reference's inference entry (`G2VLM.forward_cache_update_dino`, g2vlm.py:997-1001) calls its dino model with the
DINOv2 keyword `packed_pixel_values=` and cannot reach this module; the module boundary above is therefore the drop-in.
"""
import math

import torch

from ... import hip


class DINOv3ViTConfig:
    """The fields of the reference's DINOv3ViTConfig (configuration_dinov3_vit.py:108-160) that the forward reads."""

    def __init__(self, patch_size=16, hidden_size=384, intermediate_size=1536, num_hidden_layers=12, num_attention_heads=6,
                 hidden_act="gelu", layer_norm_eps=1e-5, rope_theta=100.0, query_bias=True, key_bias=False, value_bias=True,
                 proj_bias=True, mlp_bias=True, layerscale_value=1.0, use_gated_mlp=False, num_register_tokens=0, **_):
        self.patch_size, self.hidden_size, self.intermediate_size = patch_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.hidden_act, self.layer_norm_eps, self.rope_theta = hidden_act, layer_norm_eps, rope_theta
        self.query_bias, self.key_bias, self.value_bias, self.proj_bias, self.mlp_bias = query_bias, key_bias, value_bias, proj_bias, mlp_bias
        self.layerscale_value, self.use_gated_mlp, self.num_register_tokens = layerscale_value, use_gated_mlp, num_register_tokens


def rope_tables(gh, gw, head_dim, base):
    """DINOv3ViTRopePositionEmbedding.forward in eval mode (dinov3_model.py:144-176): fp32 cos / sin [gh*gw, head_dim]."""
    ch = torch.arange(0.5, gh, dtype=torch.float32) / gh
    cw = torch.arange(0.5, gw, dtype=torch.float32) / gw
    coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), dim=-1).flatten(0, 1)
    coords = 2.0 * coords - 1.0
    inv_freq = 1 / base ** torch.arange(0, 1, 4 / head_dim, dtype=torch.float32)
    angles = 2 * math.pi * coords[:, :, None] * inv_freq[None, None, :]
    angles = angles.flatten(1, 2).tile(2)
    return torch.cos(angles), torch.sin(angles)


class DINOv3ViTModel:
    def __init__(self, config):
        self.config = config
        if config.hidden_act != "gelu" and not config.use_gated_mlp:
            raise NotImplementedError("plain MLP: hidden_act gelu only")
        if config.use_gated_mlp and (config.hidden_act != "silu" or config.mlp_bias):
            raise NotImplementedError("gated MLP: SiLU gate without biases only (the fused SwiGLU epilogue)")
        self.w = {}
        self.device = None
        self._rope, self._plans = {}, {}

    # ---- weights: reference key names -> device tensors laid out for the kernels
    def load_state_dict(self, sd, device="cuda"):
        c, dev = self.config, torch.device(device)
        self.device = dev
        bf = lambda t: t.detach().to(torch.bfloat16).contiguous().to(dev)
        f32 = lambda t: t.detach().float().contiguous().to(dev)
        C = c.hidden_size
        zeros = torch.zeros(C)
        w = self.w
        pw = sd["embeddings.patch_embeddings.weight"]
        k = pw[0].numel()
        self.kpad = (k + 63) // 64 * 64
        w["patch.w"] = bf(torch.nn.functional.pad(pw.reshape(C, k), (0, self.kpad - k)))
        w["patch.b"] = bf(sd["embeddings.patch_embeddings.bias"])
        w["cls"] = f32(sd["embeddings.cls_token"].reshape(-1))
        R = c.num_register_tokens
        w["regs"] = f32(sd["embeddings.register_tokens"].reshape(R, C)) if R else None
        for i in range(c.num_hidden_layers):
            p, a = f"layer.{i}.", f"layer.{i}.attention."
            qb = sd[a + "q_proj.bias"] if c.query_bias else zeros
            kb = sd[a + "k_proj.bias"] if c.key_bias else zeros
            vb = sd[a + "v_proj.bias"] if c.value_bias else zeros
            w[f"{i}.qkv.w"] = bf(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0))
            w[f"{i}.qkv.b"] = bf(torch.cat([t.detach().float().to(dev) for t in (qb, kb, vb)], 0))
            w[f"{i}.o.w"] = bf(sd[a + "o_proj.weight"])
            w[f"{i}.o.b"] = bf(sd[a + "o_proj.bias"]) if c.proj_bias else None
            for n in ("norm1", "norm2"):
                w[f"{i}.{n}.w"], w[f"{i}.{n}.b"] = f32(sd[p + n + ".weight"]), f32(sd[p + n + ".bias"])
            w[f"{i}.ls1"], w[f"{i}.ls2"] = f32(sd[p + "layer_scale1.lambda1"]), f32(sd[p + "layer_scale2.lambda1"])
            if c.use_gated_mlp:
                from ...weights import interleave_gate_up
                w[f"{i}.gu.w"] = bf(interleave_gate_up(sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]))
            else:
                w[f"{i}.up.w"] = bf(sd[p + "mlp.up_proj.weight"])
                w[f"{i}.up.b"] = bf(sd[p + "mlp.up_proj.bias"]) if c.mlp_bias else None
            w[f"{i}.down.w"] = bf(sd[p + "mlp.down_proj.weight"])
            w[f"{i}.down.b"] = bf(sd[p + "mlp.down_proj.bias"]) if (c.mlp_bias and not c.use_gated_mlp) else None
        w["norm.w"], w["norm.b"] = f32(sd["norm.weight"]), f32(sd["norm.bias"])
        return self

    def _rope_rows(self, N, gh, gw):
        """cos / sin fp32 [N*(1+R+P), D]: identity rows for cls + registers (rope only touches patch tokens, :216-246)"""
        key = (N, gh, gw)
        if key not in self._rope:
            c = self.config
            D = c.hidden_size // c.num_attention_heads
            cos, sin = rope_tables(gh, gw, D, c.rope_theta)
            pre = 1 + c.num_register_tokens
            cos = torch.cat([torch.ones(pre, D), cos], 0).repeat(N, 1)
            sin = torch.cat([torch.zeros(pre, D), sin], 0).repeat(N, 1)
            self._rope[key] = (hip.h2d(cos, self.device, resident=True), hip.h2d(sin, self.device, resident=True))
        return self._rope[key]

    def _plan(self, cu, nh):
        key = tuple(cu)
        if key not in self._plans:
            wins = tuple((cu[i], cu[i + 1] - cu[i], cu[i], cu[i + 1] - cu[i], False) for i in range(len(cu) - 1) if cu[i + 1] > cu[i])
            self._plans[key] = hip.make_attn_plan(wins, nh, self.device, tile_rows=128) if wins else None
        return self._plans[key]

    @torch.no_grad()
    def embed(self, pixel_values):
        """DINOv3ViTEmbeddings (dinov3_model.py:72-127): Conv2d 16/16 as a bf16 GEMM, [cls | R registers | patches] per view.
        Returns (x fp32 [B*(1+R+P), C], gh, gw)."""
        c, w, hp = self.config, self.w, hip
        img = hp.h2d(pixel_values, self.device, torch.float32).contiguous()
        B, _, H, W = img.shape
        gh, gw = H // c.patch_size, W // c.patch_size
        cols = hp.im2col_patch(img, c.patch_size, self.kpad)
        emb = hp.linear(cols, w["patch.w"], w["patch.b"])
        return hp.vit_assemble(emb, w["cls"], w["regs"], B, gh * gw, c.num_register_tokens), gh, gw

    @torch.no_grad()
    def run_layers(self, x, cos, sin, cu, num_layers=None):
        """The encoder layers + final LayerNorm (dinov3_model.py:304-314, 536-541) on token rows x fp32 [T, C] (updated in
        place) with their RoPE rows cos / sin [T, D] and attention windows [cu[i], cu[i+1]) of THIS row axis; rows outside
        every window get a zero attention output.  Token rows only interact inside a window, so any union of whole windows
        (plus uncovered rows) can be run on its own - what the view-sharded prefill does (g2vlm_amd/sharded.py).
        Returns the normalised tokens fp32 [T, C]."""
        c, w, hp = self.config, self.w, hip
        C, nh = c.hidden_size, c.num_attention_heads
        D = C // nh
        T = x.shape[0]
        cu = [int(v) for v in (cu.tolist() if torch.is_tensor(cu) else cu)]
        if cu[-1] > T:
            raise ValueError("cu_seqlens runs past the token axis")
        plan = self._plan(cu, nh)
        h = torch.empty((T, C), dtype=torch.bfloat16, device=self.device)
        qkv = torch.empty((T, 3 * C), dtype=torch.bfloat16, device=self.device)
        ao = torch.zeros((T, C), dtype=torch.bfloat16, device=self.device)      # rows outside every window stay 0
        mid = torch.empty((T, c.intermediate_size), dtype=torch.bfloat16, device=self.device)
        for i in range(c.num_hidden_layers if num_layers is None else num_layers):
            hp.layernorm(x, w[f"{i}.norm1.w"], w[f"{i}.norm1.b"], c.layer_norm_eps, out=h)
            hp.linear(h, w[f"{i}.qkv.w"], w[f"{i}.qkv.b"], out=qkv)
            hp.rope_vision(qkv, 2 * nh, D, cos, sin)
            if plan is not None:
                hp.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], ao, plan, nh, nh, D)
            hp.linear(ao, w[f"{i}.o.w"], w[f"{i}.o.b"], hp.EPI_RES_F32, out=x, res=x, gamma=w[f"{i}.ls1"])
            hp.layernorm(x, w[f"{i}.norm2.w"], w[f"{i}.norm2.b"], c.layer_norm_eps, out=h)
            if c.use_gated_mlp:
                hp.linear(h, w[f"{i}.gu.w"], None, hp.EPI_SWIGLU, out=mid)
            else:
                hp.linear(h, w[f"{i}.up.w"], w[f"{i}.up.b"], hp.EPI_GELU, out=mid)
            hp.linear(mid, w[f"{i}.down.w"], w[f"{i}.down.b"], hp.EPI_RES_F32, out=x, res=x, gamma=w[f"{i}.ls2"])
        return hp.layernorm(x, w["norm.w"], w["norm.b"], c.layer_norm_eps, out_dtype=torch.float32)

    @torch.no_grad()
    def forward(self, pixel_values, cu_seqlens, max_seqlen=None, num_layers=None):
        """pixel_values fp32 [B,3,H,W]; cu_seqlens: window boundaries on the flat [B*(1+R+P)] token axis exactly as the
        caller built them (the reference's callers pass multiples of P - hazard H1 - so windows straddle views and the
        last rows attend to nothing: their attention output is zero, as with the varlen kernel).  Returns fp32 [B, P, C]."""
        R = self.config.num_register_tokens
        B = pixel_values.shape[0]
        x, gh, gw = self.embed(pixel_values)
        cos, sin = self._rope_rows(B, gh, gw)
        out = self.run_layers(x, cos, sin, cu_seqlens, num_layers)
        return out.view(B, gh * gw + 1 + R, -1)[:, 1 + R:]

    __call__ = forward
