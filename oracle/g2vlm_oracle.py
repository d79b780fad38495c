"""CPU oracle: a plain-PyTorch restatement of the G2VLM inference hot path.

TEST INFRASTRUCTURE — NOT THE PRODUCT.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the engine in ``g2vlm_amd/`` never
does (it fails loudly when its HIP library is missing).

What it restates (reference paths under /root/reference, which this file never reads):
  * stage drivers      modeling/g2vlm/g2vlm.py:561-1410  (prepare_*, forward_cache_update_*,
                        reconstruct, generate_text, recon, chat_with_recon)
  * MoT LLM            modeling/g2vlm/qwen2vl.py:237-251, 419-664, 751-910, 1267-1337
  * DINOv2 encoder     modeling/g2vlm/dinov2_model.py:26-356,
                        modeling/dinov2_with_registers/modeling_dinov2_with_registers.py:42-171
  * Qwen2-VL pieces    modeling/qwen2vl/modeling_qwen2_vl.py:103-313, 366-404, 457-521, 987-1072
  * Pi3 decoders/heads modeling/pi3/models/layers/{pos_embed,attention,block,transformer_head,
                        camera_head}.py
  * index helpers      data/data_utils.py:78-201

dtype flow is SURVEY.md App. C: the GPU reference runs under ``autocast(bf16)`` with fp32
master weights, so every Linear/conv rounds input, weight and bias to bf16 and returns bf16;
norms, residual streams, RoPE (LLM) and the pointmap/camera heads are fp32; RoPE2D tables
and math are bf16.  The casts are written out explicitly here so the oracle does not depend
on autocast state.  ``precise=True`` switches every bf16 cast off (fp32 everywhere, fp64
attention) to measure the bf16 noise floor.

Pinned by ``tests/golden/*`` generated from the shim-imported reference
(``oracle/gen_golden.py``).  Unpinned: flash-attn 2.7.4's own numerics (third-party, absent
from /root/reference; attention here is an fp32-softmax restatement), rows outside the DINO
windows (defined as 0), real-checkpoint outputs and real tokenizer ids.
"""
import math

import torch
import torch.nn.functional as F

_RESNET_MEAN = [0.485, 0.456, 0.406]
_RESNET_STD = [0.229, 0.224, 0.225]
MROPE_SECTION = [16, 24, 24]


class NaiveCache:
    """modeling/g2vlm/qwen2vl.py:237-251"""

    def __init__(self, num_layers):
        self.key_cache = {k: None for k in range(num_layers)}
        self.value_cache = {k: None for k in range(num_layers)}

    @property
    def num_layers(self):
        return len(self.key_cache)

    @property
    def seq_lens(self):
        return 0 if self.key_cache[0] is None else self.key_cache[0].shape[0]


# ----------------------------------------------------------------------------- primitives
def varlen_attention(q, k, v, cu_q, cu_k, causal, scale=None, precise=False):
    """flash_attn_varlen_func semantics (call sites qwen2vl.py:643-652, dinov2_model.py:49-58,
    modeling_qwen2_vl.py:400): per-window softmax(QK^T*scale)V in fp32, GQA by head repeat,
    bottom-right aligned causal mask; rows in no window are 0 (convention, App. D-H1)."""
    Hq, Hk, D = q.shape[1], k.shape[1], q.shape[2]
    scale = scale if scale is not None else 1.0 / math.sqrt(D)
    out = torch.zeros_like(q)
    rep = Hq // Hk
    ct = torch.float64 if precise else torch.float32
    for i in range(len(cu_q) - 1):
        qs, qe, ks, ke = int(cu_q[i]), int(cu_q[i + 1]), int(cu_k[i]), int(cu_k[i + 1])
        if qe <= qs:
            continue
        qi = q[qs:qe].to(ct).transpose(0, 1)
        ki = k[ks:ke].to(ct).transpose(0, 1).repeat_interleave(rep, dim=0)
        vi = v[ks:ke].to(ct).transpose(0, 1).repeat_interleave(rep, dim=0)
        s = torch.matmul(qi, ki.transpose(1, 2)) * scale
        if causal:
            lq, lk = qe - qs, ke - ks
            row = torch.arange(lq).view(-1, 1)
            col = torch.arange(lk).view(1, -1)
            s = s.masked_fill(col > row + (lk - lq), float("-inf"))
        o = torch.matmul(torch.softmax(s, dim=-1), vi)
        out[qs:qe] = o.transpose(0, 1).to(q.dtype)
    return out


def rotate_half(x):
    """modeling_qwen2_vl.py:170-174"""
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def mrope_tables(position_ids, theta, head_dim=128):
    """Qwen2VLRotaryEmbedding.forward (modeling_qwen2_vl.py:142-166) followed by the section
    select of apply_multimodal_rotary_pos_emb (:223-225).  position_ids [3, L] int64.
    Returns cos, sin fp32 [L, head_dim]."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]          # [3, L, hd/2]
    emb = torch.cat((freqs, freqs), dim=-1)                                      # [3, L, hd]
    cos, sin = emb.cos(), emb.sin()
    sec = MROPE_SECTION * 2
    cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)
    sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1)
    return cos, sin


def rope2d_tables(D, seq_len, dtype, base=100.0):
    """RoPE2D.get_cos_sin (pos_embed.py:120-129): angles are rounded to ``dtype`` (bf16 under
    autocast) BEFORE cos/sin — parity hazard H3."""
    inv_freq = 1.0 / (base ** (torch.arange(0, D, 2).float() / D))
    t = torch.arange(seq_len, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq).to(dtype)
    freqs = torch.cat((freqs, freqs), dim=-1)
    return freqs.cos(), freqs.sin()


def rope2d(tokens, positions):
    """RoPE2D.forward (pos_embed.py:141-159); tokens [B,H,N,d], positions [B,N,2] (y,x).
    All arithmetic in tokens.dtype (three bf16 roundings per element)."""
    D = tokens.size(3) // 2
    cos, sin = rope2d_tables(D, int(positions.max()) + 1, tokens.dtype)

    def rope1d(t, pos1d):
        c = F.embedding(pos1d, cos)[:, None, :, :]
        s = F.embedding(pos1d, sin)[:, None, :, :]
        return (t * c) + (rotate_half(t) * s)

    y, x = tokens.chunk(2, dim=-1)
    return torch.cat((rope1d(y, positions[:, :, 0]), rope1d(x, positions[:, :, 1])), dim=-1)


def get_rope_index_image_3d(t, h, w, base, merge=1):
    """data/data_utils.py:78-137 (merge=1, DINO) and :142-201 (merge=2, ViT).
    Returns position ids [3, t*h*w] int64 and delta = max - min."""
    h, w = h // merge, w // merge
    ti = torch.arange(t).view(-1, 1).expand(-1, h * w).flatten()
    hi = torch.arange(h).view(1, -1, 1).expand(t, -1, w).flatten()
    wi = torch.arange(w).view(1, 1, -1).expand(t, h, -1).flatten()
    pos = torch.stack([ti, hi, wi], dim=0) + base
    return pos, int(pos.max() - pos.min())


def smart_resize(height, width, factor=28, min_pixels=56 * 56, max_pixels=14 * 14 * 4 * 1280):
    """modeling/qwen2vl/image_processing_qwen2_vl.py:56-84"""
    if height < factor or width < factor:
        raise ValueError(f"height:{height} or width:{width} must be larger than factor:{factor}")
    if max(height, width) / min(height, width) > 200:
        raise ValueError("absolute aspect ratio must be smaller than 200")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = math.floor(height / beta / factor) * factor
        w_bar = math.floor(width / beta / factor) * factor
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def vit_patchify(frames, patch=14, temporal=2, merge=2):
    """The reshape/transpose of Qwen2VLImageProcessor._preprocess
    (image_processing_qwen2_vl.py:245-273).  frames: [T,3,H,W] fp32 normalised.
    Returns pixel_values [t*gh*gw, 3*temporal*patch*patch] in merge-block order and (t,gh,gw)."""
    if frames.shape[0] % temporal != 0:
        frames = torch.cat([frames, frames[-1:].repeat(temporal - 1, 1, 1, 1)], 0)
    T, C, Hh, Ww = frames.shape
    gt, gh, gw = T // temporal, Hh // patch, Ww // patch
    p = frames.reshape(gt, temporal, C, gh // merge, merge, patch, gw // merge, merge, patch)
    p = p.permute(0, 3, 6, 4, 7, 2, 1, 5, 8)
    return p.reshape(gt * gh * gw, C * temporal * patch * patch).contiguous(), (gt, gh, gw)


# ----------------------------------------------------------------------------- the model
class OracleG2VLM:
    """Functional restatement over a flat state dict (fp32 tensors keyed as SURVEY §8b)."""

    def __init__(self, state_dict, dims, precise=False):
        self.sd = state_dict
        self.dims = dims
        self.precise = precise
        self.hidden_size = dims["llm"]["hidden"]
        self.num_layers = dims["llm"]["layers"]
        self.has_conf = "conf_head.proj.weight" in state_dict
        # use_dinov3 (g2vlm.py:134, 169-172, 1172-1174): dims["dino"]["v3"] holds the DINOv3ViTConfig fields; heads and grids use 16
        self.patch = dims["dino"].get("patch", 14)
        self.cd = torch.float32 if precise else torch.bfloat16     # "autocast" compute dtype
        # error-growth probes (tests/test_full_depth_gpu.py): when `taps` is a dict, the residual stream after the MoT
        # layers in `tap_layers` (1-based), the DINO tokens and the three decoder outputs are stored in it
        self.taps, self.tap_layers = None, ()

    # ---- helpers
    def lin(self, x, name, bias=True):
        """nn.Linear under autocast: x, W, b -> bf16; fp32 accumulate; bf16 out."""
        w = self.sd[name + ".weight"].to(self.cd)
        b = self.sd[name + ".bias"].to(self.cd) if (bias and name + ".bias" in self.sd) else None
        return F.linear(x.to(self.cd), w, b)

    def lin32(self, x, name):
        """nn.Linear inside an autocast(enabled=False) island."""
        return F.linear(x.float(), self.sd[name + ".weight"], self.sd.get(name + ".bias"))

    def rms(self, x, name):
        """Qwen2RMSNorm.forward (modeling_qwen2_vl.py:496-501)."""
        dt = x.dtype
        h = x.to(torch.float32)
        var = h.pow(2).mean(-1, keepdim=True)
        h = h * torch.rsqrt(var + self.dims["llm"]["eps"])
        return self.sd[name + ".weight"] * h.to(dt)

    def ln(self, x, name, eps=1e-6):
        """nn.LayerNorm under autocast.  fp32 input (DINO / MoT / decoder residual streams): fp32 math, fp32 out on CPU and
        CUDA alike.  bf16 input (the Qwen2-VL ViT's residual stream): autocast(cuda) computes in fp32 and the next Linear
        rounds to bf16; autocast(cpu) - what the shim-imported reference runs, and what the golden vectors pin - calls the
        mixed-dtype kernel (bf16 in, fp32 weights, bf16 out).  Both are bf16(LN(x)) up to fp32 summation order; the CPU
        form is used so the oracle stays bit-identical to the reference run it is checked against."""
        if x.dtype == torch.bfloat16 and not self.precise:
            return F.layer_norm(x, (x.shape[-1],), self.sd[name + ".weight"], self.sd[name + ".bias"], eps)
        return F.layer_norm(x.float(), (x.shape[-1],), self.sd[name + ".weight"], self.sd[name + ".bias"], eps)

    def sdpa(self, q, k, v):
        """torch SDPA call sites pi3/models/layers/attention.py:255-264, 370-375 ([B,H,N,d])."""
        if self.precise:
            s = torch.matmul(q.double(), k.double().transpose(-1, -2)) / math.sqrt(q.shape[-1])
            return torch.matmul(torch.softmax(s, -1), v.double()).to(q.dtype)
        # same torch kernel the reference dispatches to on CPU (bf16 in, fp32 softmax/accumulate)
        with torch.nn.attention.sdpa_kernel(torch.nn.attention.SDPBackend.FLASH_ATTENTION):
            return F.scaled_dot_product_attention(q, k, v)

    # ---- MoT LLM (modeling/g2vlm/qwen2vl.py)
    def _attn(self, i, x, cos, sin, query_lens, packed_query_indexes, cache, key_values_lens,
              packed_key_value_indexes, is_causal, mode, geo_idx, text_idx):
        """PackedAttentionMoT.forward_inference (qwen2vl.py:555-664).  x: normed layer input."""
        L = self.dims["llm"]
        nh, nkv, hd = L["heads"], L["kv_heads"], 128
        a = f"language_model.model.layers.{i}.self_attn."
        if mode == "und":
            q = self.lin(x, a + "q_proj").view(-1, nh, hd).transpose(0, 1)
            k = self.lin(x, a + "k_proj").view(-1, nkv, hd).transpose(0, 1)
            v = self.lin(x, a + "v_proj").view(-1, nkv, hd)
            q = self.rms(q, a + "q_norm")              # bf16 in -> normalised rounded to bf16, * fp32 w
            k = self.rms(k, a + "k_norm")
        else:
            x = x.to(self.cd)                                                    # :579
            n = x.shape[0]
            q = x.new_zeros((n, nh * hd)); k = x.new_zeros((n, nkv * hd)); v = x.new_zeros((n, nkv * hd))
            xt, xg = x[text_idx], x[geo_idx]
            q[text_idx] = self.lin(xt, a + "q_proj"); q[geo_idx] = self.lin(xg, a + "q_proj_moe_geo")
            k[text_idx] = self.lin(xt, a + "k_proj"); k[geo_idx] = self.lin(xg, a + "k_proj_moe_geo")
            v[text_idx] = self.lin(xt, a + "v_proj"); v[geo_idx] = self.lin(xg, a + "v_proj_moe_geo")
            q = q.view(-1, nh, hd).to(torch.float32)                             # :600
            k = k.view(-1, nkv, hd).to(torch.float32)
            v = v.view(-1, nkv, hd)
            q[text_idx] = self.rms(q[text_idx], a + "q_norm")
            q[geo_idx] = self.rms(q[geo_idx], a + "q_norm_moe_geo")
            k[text_idx] = self.rms(k[text_idx], a + "k_norm")
            k[geo_idx] = self.rms(k[geo_idx], a + "k_norm_moe_geo")
            q, k = q.transpose(0, 1), k.transpose(0, 1)
        # mRoPE in fp32 (:611-615); cos/sin [L,hd] broadcast over heads
        q = (q * cos) + (rotate_half(q) * sin)
        k = (k * cos) + (rotate_half(k) * sin)
        q = q.to(self.cd).transpose(0, 1)
        k = k.to(self.cd).transpose(0, 1)
        v = v.to(self.cd)
        if cache is not None and cache.key_cache[i] is not None:
            pk, pv = cache.key_cache[i], cache.value_cache[i]
            tot = int(sum(query_lens)) + int(sum(key_values_lens))
            mk = pk.new_zeros((tot, nkv, hd)); mv = pk.new_zeros((tot, nkv, hd))
            mk[packed_query_indexes] = k; mk[packed_key_value_indexes] = pk
            mv[packed_query_indexes] = v; mv[packed_key_value_indexes] = pv
            kv_lens = key_values_lens + query_lens
        else:
            mk, mv, kv_lens = k, v, query_lens
        cu_q = F.pad(torch.cumsum(query_lens, 0), (1, 0))
        cu_k = F.pad(torch.cumsum(kv_lens, 0), (1, 0))
        o = varlen_attention(q, mk, mv, cu_q, cu_k, is_causal, precise=self.precise)
        o = o.reshape(-1, nh * hd)
        if mode == "und":
            o = self.lin(o, a + "o_proj", bias=False)
        else:
            o[text_idx] = self.lin(o[text_idx], a + "o_proj", bias=False)
            o[geo_idx] = self.lin(o[geo_idx], a + "o_proj_moe_geo", bias=False)
        if cache is not None:
            cache.key_cache[i], cache.value_cache[i] = mk, mv
        return o

    def _mlp(self, x, name):
        """Qwen2MLP.forward (modeling_qwen2_vl.py:519-521); every op result rounds to bf16."""
        g = self.lin(x, name + ".gate_proj", bias=False)
        u = self.lin(x, name + ".up_proj", bias=False)
        return self.lin(F.silu(g) * u, name + ".down_proj", bias=False)

    def _layer(self, i, x, **kw):
        """Qwen2VLMoTDecoderLayer.forward_inference (qwen2vl.py:842-910)."""
        p = f"language_model.model.layers.{i}."
        mode, gi, ti = kw["mode"], kw["geo_idx"], kw["text_idx"]
        res = x
        if mode == "und":
            h = self.rms(x, p + "input_layernorm")
        else:
            h = torch.zeros_like(x)
            h[ti] = self.rms(x[ti], p + "input_layernorm")
            h[gi] = self.rms(x[gi], p + "input_layernorm_moe_geo")
        o = self._attn(i, h, **kw)
        if mode == "geo":
            o[gi] = (o[gi] * self.sd[p + "ls1.gamma"]).to(self.cd)               # :885
        x = res + o
        res = x
        if mode == "und":
            y = self._mlp(self.rms(x, p + "post_attention_layernorm"), p + "mlp")
        else:
            t = self.rms(x[ti], p + "post_attention_layernorm").to(self.cd)
            g = self.rms(x[gi], p + "post_attention_layernorm_moe_geo").to(self.cd)
            y = torch.zeros_like(x).to(self.cd)
            y[ti] = self._mlp(t, p + "mlp")
            y[gi] = self._mlp(g, p + "mlp_moe_geo")
            y[gi] = (y[gi] * self.sd[p + "ls2.gamma"]).to(self.cd)               # :907
        return res + y

    def llm_forward_inference(self, packed_query_sequence, query_lens, packed_query_position_ids,
                              packed_query_indexes, past_key_values, key_values_lens,
                              packed_key_value_indexes, is_causal, mode="und",
                              packed_geo_token_indexes=None, packed_text_indexes=None, num_layers=None):
        """Qwen2VLModel.forward_inference (qwen2vl.py:1267-1337).  Returns final-normed hidden fp32."""
        cos, sin = mrope_tables(packed_query_position_ids, self.dims["llm"]["theta"])
        x = packed_query_sequence
        for i in range(self.num_layers if num_layers is None else num_layers):
            x = self._layer(i, x, cos=cos, sin=sin, query_lens=query_lens,
                            packed_query_indexes=packed_query_indexes, cache=past_key_values,
                            key_values_lens=key_values_lens, packed_key_value_indexes=packed_key_value_indexes,
                            is_causal=is_causal, mode=mode, geo_idx=packed_geo_token_indexes,
                            text_idx=packed_text_indexes)
            if self.taps is not None and mode == "geo" and (i + 1) in self.tap_layers:
                self.taps[f"mot{i + 1}"] = x.clone()
        p = "language_model.model."
        if mode == "und":
            return self.rms(x, p + "norm")
        out = torch.zeros_like(x)
        out[packed_text_indexes] = self.rms(x[packed_text_indexes], p + "norm")
        out[packed_geo_token_indexes] = self.rms(x[packed_geo_token_indexes], p + "norm_moe_geo")
        return out

    def embed_tokens(self, ids):
        return F.embedding(ids, self.sd["language_model.model.embed_tokens.weight"])

    # ---- DINOv2 (dinov2_model.py, modeling_dinov2_with_registers.py)
    def dino_pos_embed(self, height, width):
        """interpolate_pos_encoding (modeling_dinov2_with_registers.py:93-145)."""
        pe = self.sd["dino_model.embeddings.position_embeddings"]
        n_pos = pe.shape[1] - 1
        gh, gw = height // 14, width // 14
        if gh * gw == n_pos and height == width:
            return pe
        cls_pe, patch_pe = pe[:, 0], pe[:, 1:]
        dim = pe.shape[-1]
        s = int(n_pos ** 0.5)
        patch_pe = patch_pe.reshape(1, s, s, dim).permute(0, 3, 1, 2)
        patch_pe = F.interpolate(patch_pe.float(), size=(gh, gw), mode="bicubic", align_corners=False, antialias=True)
        patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, -1, dim)
        return torch.cat((cls_pe.unsqueeze(0), patch_pe), dim=1)

    def dino_embeddings(self, pixel_values):
        """Dinov2WithRegistersEmbeddings.forward (:147-171): conv14/14 as bf16 GEMM, cls, pos, regs."""
        e = "dino_model.embeddings."
        n, _, hh, ww = pixel_values.shape
        w = self.sd[e + "patch_embeddings.projection.weight"].to(self.cd)
        b = self.sd[e + "patch_embeddings.projection.bias"].to(self.cd)
        emb = F.conv2d(pixel_values.to(self.cd), w, b, stride=14).flatten(2).transpose(1, 2)
        emb = torch.cat((self.sd[e + "cls_token"].expand(n, -1, -1), emb.float()), dim=1)
        emb = emb + self.dino_pos_embed(hh, ww)
        return torch.cat((emb[:, :1], self.sd[e + "register_tokens"].expand(n, -1, -1), emb[:, 1:]), dim=1)

    def dino_layer(self, i, x, cu):
        """Dinov2WithRegistersLayer.forward (dinov2_model.py:218-249)."""
        p = f"dino_model.encoder.layer.{i}."
        nh = self.dims["dino"]["heads"]
        h = self.ln(x, p + "norm1")
        q = self.lin(h, p + "attention.attention.query").view(x.shape[0], nh, -1)
        k = self.lin(h, p + "attention.attention.key").view(x.shape[0], nh, -1)
        v = self.lin(h, p + "attention.attention.value").view(x.shape[0], nh, -1)
        ctx = varlen_attention(q, k, v, cu, cu, False, precise=self.precise).reshape(x.shape[0], -1)
        a = self.lin(ctx, p + "attention.output.dense")
        x = a * self.sd[p + "layer_scale1.lambda1"] + x
        m = self.lin(F.gelu(self.lin(self.ln(x, p + "norm2"), p + "mlp.fc1")), p + "mlp.fc2")
        return m * self.sd[p + "layer_scale2.lambda1"] + x

    def dino_forward(self, pixel_values, cu_seqlens, num_layers=None):
        """Dinov2WithRegistersModel.forward (dinov2_model.py:301-356) -> [N,P,C] fp32.
        use_dinov3: the call the reference's training forward makes for that variant (g2vlm.py:380-386), DINOv3ViTModel with
        the same cu_seqlens; its inference method passes `packed_pixel_values=` (g2vlm.py:997-1001) and cannot reach it."""
        if self.dims["dino"].get("v3"):
            from oracle import dinov3_oracle
            pre = "dino_model."
            sub = {k[len(pre):]: v for k, v in self.sd.items() if k.startswith(pre)}
            return dinov3_oracle.forward(sub, self.dims["dino"]["v3"], pixel_values, cu_seqlens, num_layers, precise=self.precise)
        emb = self.dino_embeddings(pixel_values)
        n, s, d = emb.shape
        x = emb.reshape(n * s, d)
        for i in range(self.dims["dino"]["layers"] if num_layers is None else num_layers):
            x = self.dino_layer(i, x, cu_seqlens)
        x = self.ln(x, "dino_model.layernorm").reshape(n, s, d)
        return x[:, 5:]

    # ---- Qwen2-VL ViT (modeling_qwen2_vl.py:987-1072)
    def vit_rot_pos(self, t, h, w, merge=2):
        hd = self.dims["vit"]["embed"] // self.dims["vit"]["heads"]
        hp = torch.arange(h).unsqueeze(1).expand(-1, w).reshape(h // merge, merge, w // merge, merge)
        hp = hp.permute(0, 2, 1, 3).flatten()
        wp = torch.arange(w).unsqueeze(0).expand(h, -1).reshape(h // merge, merge, w // merge, merge)
        wp = wp.permute(0, 2, 1, 3).flatten()
        pos_ids = torch.stack([hp, wp], dim=-1).repeat(t, 1)
        dim = hd // 2
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
        freqs = torch.outer(torch.arange(max(h, w), dtype=torch.float), inv_freq)
        rot = freqs[pos_ids].flatten(1)
        emb = torch.cat((rot, rot), dim=-1)
        return emb.cos(), emb.sin()

    def vit_forward(self, pixel_values, grid_thw, num_layers=None):
        V = self.dims["vit"]
        nh = V["heads"]
        w = self.sd["vit_model.patch_embed.proj.weight"].to(self.cd)
        # PatchEmbed.forward (modeling_qwen2_vl.py:280-286): Conv3d with stride = kernel, no bias.  Kept as a conv (not the
        # equivalent [T,1176] x [1176,C] GEMM): on CPU the two sum in different orders and differ by a bf16 ulp at K = 1176
        x = F.conv3d(pixel_values.view(-1, 3, w.shape[2], w.shape[3], w.shape[4]).to(self.cd), w, stride=tuple(w.shape[2:])).view(-1, w.shape[0])
        t, h, wd = [int(v) for v in grid_thw]
        cos, sin = self.vit_rot_pos(t, h, wd)
        cu = torch.tensor([0] + [h * wd * (i + 1) for i in range(t)])
        n = x.shape[0]
        for i in range(V["depth"] if num_layers is None else num_layers):
            p = f"vit_model.blocks.{i}."
            qkv = self.lin(self.ln(x, p + "norm1"), p + "attn.qkv").reshape(n, 3, nh, -1).permute(1, 0, 2, 3)
            q, k, v = qkv[0], qkv[1], qkv[2]
            c, s = cos.unsqueeze(-2), sin.unsqueeze(-2)
            qf, kf = q.float(), k.float()
            q = ((qf * c) + (rotate_half(qf) * s)).to(self.cd)
            k = ((kf * c) + (rotate_half(kf) * s)).to(self.cd)
            o = varlen_attention(q, k, v, cu, cu, False, precise=self.precise).reshape(n, -1)
            x = x + self.lin(o, p + "attn.proj")
            hdn = self.lin(self.ln(x, p + "norm2"), p + "mlp.fc1")
            hdn = hdn * torch.sigmoid(1.702 * hdn)                               # quick_gelu, bf16 ops
            x = x + self.lin(hdn, p + "mlp.fc2")
        m = "vit_model.merger."
        y = self.ln(x, m + "ln_q").view(-1, 4 * V["embed"])
        return self.lin(F.gelu(self.lin(y, m + "mlp.0")), m + "mlp.2")

    # ---- Pi3 decoders / heads
    def _self_attn_rope(self, x, p, pos):
        b, n, c = x.shape
        nh = self.dims["dec"]["heads"]
        qkv = self.lin(x, p + "qkv").reshape(b, n, 3, nh, c // nh).transpose(1, 3)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        q, k = rope2d(q, pos), rope2d(k, pos)
        o = self.sdpa(q, k, v).transpose(1, 2).reshape(b, n, c)
        return self.lin(o, p + "proj")

    def _cross_attn_rope(self, x, y, p, qpos, kpos):
        b, n, c = x.shape
        m = y.shape[1]
        nh = self.dims["dec"]["heads"]
        q = self.lin(x, p + "q_proj").reshape(b, n, nh, c // nh).permute(0, 2, 1, 3)
        k = self.lin(y, p + "k_proj").reshape(b, m, nh, c // nh).permute(0, 2, 1, 3)
        v = self.lin(y, p + "v_proj").reshape(b, m, nh, c // nh).permute(0, 2, 1, 3)
        q, k = rope2d(q, qpos), rope2d(k, kpos)
        o = self.sdpa(q, k, v).transpose(1, 2).reshape(b, n, c)
        return self.lin(o, p + "proj")

    def _mlp_gelu(self, x, p):
        return self.lin(F.gelu(self.lin(x, p + "fc1")), p + "fc2")

    def decoder(self, name, hidden, pos, context=None, depth=None):
        """Pi3TransformerDecoder / Pi3ContextTransformerDecoder (transformer_head.py:9-56, 84-130)."""
        x = hidden
        for i in range(self.dims["dec"]["depth"] if depth is None else depth):
            p = f"{name}.blocks.{i}."
            x = x + self._self_attn_rope(self.ln(x, p + "norm1"), p + "attn.", pos)
            if context is not None:
                y_ = self.ln(context, p + "norm_y")
                x = x + self._cross_attn_rope(self.ln(x, p + "norm2"), y_, p + "cross_attn.", pos, pos)
                x = x + self._mlp_gelu(self.ln(x, p + "norm3"), p + "mlp.")
            else:
                x = x + self._mlp_gelu(self.ln(x, p + "norm2"), p + "mlp.")
        return self.lin(x, name + ".linear_out")

    def pts_head(self, name, tokens, H, W):
        """Pi3LinearPts3d.forward (transformer_head.py:69-81), fp32 island."""
        b = tokens.shape[0]
        feat = self.lin32(tokens, name + ".proj")
        feat = feat.transpose(-1, -2).reshape(b, -1, H // self.patch, W // self.patch)
        return F.pixel_shuffle(feat, self.patch).permute(0, 2, 3, 1)

    def camera_head(self, feat):
        """Pi3CameraHead.forward (camera_head.py:48-93), fp32 island.  feat [N,P,512]."""
        for i in range(2):
            r = f"camera_head.res_conv.{i}."
            x = F.relu(self.lin32(feat, r + "res_conv1"))
            x = F.relu(self.lin32(x, r + "res_conv2"))
            x = F.relu(self.lin32(x, r + "res_conv3"))
            feat = feat + x
        feat = feat.mean(dim=1)                                                  # AdaptiveAvgPool2d(1)
        feat = F.relu(self.lin32(feat, "camera_head.more_mlps.0"))
        feat = F.relu(self.lin32(feat, "camera_head.more_mlps.2"))
        out_t = self.lin32(feat, "camera_head.fc_t")
        out_r = self.lin32(feat, "camera_head.fc_rot").reshape(-1, 3, 3)
        mt = torch.transpose(F.normalize(out_r, p=2, dim=-1), -1, -2)
        u, s, vh = torch.linalg.svd(mt)
        v = vh.transpose(-2, -1)
        det = torch.det(torch.matmul(v, u.transpose(-2, -1)))
        r = torch.matmul(torch.cat([v[:, :, :-1], v[:, :, -1:] * det.view(-1, 1, 1)], dim=2), u.transpose(-2, -1))
        pose = torch.zeros((feat.shape[0], 4, 4))
        pose[:, :3, :3] = r
        pose[:, :3, 3] = out_t
        pose[:, 3, 3] = 1.0
        return pose

    # ---- stage drivers (modeling/g2vlm/g2vlm.py)
    def prepare_prompts(self, curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids, bos=False, eos=False):
        """prepare_prompts_addbos (:561-594, bos=True), prepare_prompts_pure_text (:632-664),
        prepare_prompts (:666-699, bos and eos)."""
        ids_all, pos_all, lens, idx, kv_idx = [], [], [], [], []
        curr = 0
        newlens, new_rope = [], []
        for prompt, kvlen, pos in zip(prompts, curr_kvlens, curr_rope):
            kv_idx.extend(range(curr, curr + kvlen)); curr += kvlen
            ids = tokenizer.encode(prompt)
            if bos:
                ids = [new_token_ids["bos_token_id"]] + ids
            if eos:
                ids = ids + [new_token_ids["eos_token_id"]]
            lens.append(len(ids)); ids_all.extend(ids)
            pos_all.extend(range(pos, pos + len(ids)))
            idx.extend(range(curr, curr + len(ids)))
            newlens.append(kvlen + len(ids)); new_rope.append(pos + len(ids))
            curr += len(ids)
        gi = {
            "text_token_lens": torch.tensor(lens, dtype=torch.int),
            "packed_text_ids": torch.tensor(ids_all, dtype=torch.long),
            "packed_text_position_ids": torch.tensor(pos_all, dtype=torch.long).expand(3, -1),
            "packed_text_indexes": torch.tensor(idx, dtype=torch.long),
            "packed_key_value_indexes": torch.tensor(kv_idx, dtype=torch.long),
            "key_values_lens": torch.tensor(curr_kvlens, dtype=torch.int),
        }
        return gi, newlens, new_rope

    def forward_cache_update_text(self, cache, packed_text_ids, packed_text_position_ids, text_token_lens,
                                  packed_text_indexes, packed_key_value_indexes, key_values_lens):
        """g2vlm.py:701-733"""
        x = self.embed_tokens(packed_text_ids)
        self.llm_forward_inference(x, text_token_lens, packed_text_position_ids, packed_text_indexes, cache,
                                   key_values_lens, packed_key_value_indexes, True, "und")
        return cache

    def prepare_images(self, curr_kvlens, curr_rope, grids, new_token_ids, merge=1):
        """Index bookkeeping shared by prepare_dino_images_pi3 (:868-966; merge=1, all views in ONE
        sample) and prepare_vit_images (:735-810; merge=2, one image per call).  grids: list of
        (t,h,w) patch grids."""
        text_ids, text_idx, tok_idx, seqlens, pos_ids, idx, kv_idx, tok_lens = [], [], [], [], [], [], [], []
        _curr = curr = 0
        kvlen, pos = curr_kvlens[0], curr_rope[0]
        kv_idx.extend(range(curr, curr + kvlen)); curr += kvlen
        for (t, h, w) in grids:
            text_ids.append(new_token_ids["start_of_image"]); text_idx.append(_curr); idx.append(curr)
            curr += 1; _curr += 1
            pos_ids.append(torch.full((3, 1), pos, dtype=torch.long)); pos += 1
            n_tok = t * (h // merge) * (w // merge)
            tok_lens.append(n_tok)
            tok_idx.extend(range(_curr, _curr + n_tok)); idx.extend(range(curr, curr + n_tok))
            curr += n_tok; _curr += n_tok
            p, delta = get_rope_index_image_3d(t, h, w, pos, merge)
            pos_ids.append(p); pos += delta + 1
            text_ids.append(new_token_ids["end_of_image"]); text_idx.append(_curr); idx.append(curr)
            curr += 1; _curr += 1
            pos_ids.append(torch.full((3, 1), pos, dtype=torch.long)); pos += 1
            seqlens.append(n_tok + 2)
            kvlen += n_tok + 2
        gi = {
            "packed_text_ids": torch.tensor(text_ids, dtype=torch.long),
            "packed_text_indexes": torch.tensor(text_idx, dtype=torch.long),
            "token_seqlens": torch.tensor(tok_lens, dtype=torch.int),
            "packed_token_indexes": torch.tensor(tok_idx, dtype=torch.long),
            "packed_position_ids": torch.cat(pos_ids, dim=1),
            "packed_seqlens": torch.tensor([sum(seqlens)], dtype=torch.int),
            "packed_indexes": torch.tensor(idx, dtype=torch.long),
            "packed_key_value_indexes": torch.tensor(kv_idx, dtype=torch.long),
            "key_values_lens": torch.tensor(curr_kvlens, dtype=torch.int),
        }
        return gi, [kvlen], [pos]

    def prepare_dino_images(self, curr_kvlens, curr_rope, images01, new_token_ids):
        """prepare_dino_images_pi3 minus the PIL loading: images01 is [N,3,H,W] in [0,1]."""
        n, _, hh, ww = images01.shape
        gi, newlens, new_rope = self.prepare_images(curr_kvlens, curr_rope, [(1, hh // self.patch, ww // self.patch)] * n, new_token_ids)
        mean = torch.tensor(_RESNET_MEAN).view(1, 3, 1, 1); std = torch.tensor(_RESNET_STD).view(1, 3, 1, 1)
        gi["packed_dino_images"] = (images01 - mean) / std
        gi["original_images"] = images01.clone()
        gi["dino_token_seqlens"] = gi.pop("token_seqlens")
        gi["packed_dino_token_indexes"] = gi.pop("packed_token_indexes")
        return gi, newlens, new_rope

    def forward_cache_update_dino(self, cache, gi, num_layers=None, dino_layers=None):
        """g2vlm.py:968-1039.  Returns (cache, last_hidden [Lq,H] fp32)."""
        H = self.hidden_size
        text_emb = self.embed_tokens(gi["packed_text_ids"])
        seq = text_emb.new_zeros((int(gi["packed_seqlens"].sum()), H))
        seq[gi["packed_text_indexes"]] = text_emb
        cu = F.pad(torch.cumsum(gi["dino_token_seqlens"], 0), (1, 0))         # windows of length P (H1)
        tok = self.dino_forward(gi["packed_dino_images"], cu, dino_layers)
        if self.taps is not None:
            self.taps["dino_tokens"] = tok.clone()
        tok = self.lin(tok.reshape(-1, tok.shape[-1]), "dino2llm")
        seq[gi["packed_dino_token_indexes"]] = tok.to(seq.dtype)
        last = self.llm_forward_inference(seq, gi["packed_seqlens"], gi["packed_position_ids"], gi["packed_indexes"],
                                          cache, gi["key_values_lens"], gi["packed_key_value_indexes"], False, "geo",
                                          gi["packed_dino_token_indexes"], gi["packed_text_indexes"], num_layers)
        return cache, last

    def reconstruct(self, last_hidden, gi):
        """g2vlm.py:1143-1238"""
        imgs = gi["packed_dino_images"]
        n, _, H, W = imgs.shape
        ph, pw = H // self.patch, W // self.patch
        hidden = last_hidden[gi["packed_dino_token_indexes"]].reshape(n, ph * pw, -1)
        pos = torch.cartesian_prod(torch.arange(ph), torch.arange(pw)).view(1, ph * pw, 2).expand(n, -1, 2).clone()
        point_hidden = self.decoder("point_decoder", hidden, pos)
        camera_hidden = self.decoder("camera_decoder", hidden, pos)
        context = hidden[0:1].repeat(n, 1, 1)
        global_hidden = self.decoder("global_points_decoder", hidden, pos, context=context)
        if self.taps is not None:
            self.taps.update(point_hidden=point_hidden.clone(), camera_hidden=camera_hidden.clone(), global_hidden=global_hidden.clone())
        points, local_points, camera_poses, global_points = self.heads(point_hidden, camera_hidden, global_hidden, H, W)
        conf = None
        if self.has_conf:                                  # g2vlm.py:1192-1193, 1208-1210 (train_conf_pi3 checkpoints)
            conf_hidden = self.decoder("conf_decoder", hidden, pos)
            conf = self.pts_head("conf_head", conf_hidden.float(), H, W).reshape(1, n, H, W, -1)
        return dict(points=points, local_points=local_points, conf=conf, camera_poses=camera_poses,
                    global_points=global_points, images=gi["original_images"].unsqueeze(0))

    def heads(self, point_hidden, camera_hidden, global_hidden, H, W):
        """The fp32 islands of G2VLM.reconstruct (g2vlm.py:1200-1226, autocast off): Pi3LinearPts3d + exp / xy*z
        (transformer_head.py:58-81), Pi3CameraHead (camera_head.py:32-93), unprojection through the poses.  The two point
        heads are per patch (Linear + pixel_shuffle), so `point_hidden` / `global_hidden` [n, p, 1024] may hold any p = (H/ps)
        (W/ps) patches per view - a sub-grid of the image - while `camera_hidden` [n, P, 512] carries ALL patches of the
        view (the camera head averages over them)."""
        n = point_hidden.shape[0]
        ret = self.pts_head("point_head", point_hidden.float(), H, W).reshape(1, n, H, W, -1)
        xy, z = ret.split([2, 1], dim=-1)
        z = torch.exp(z)
        local_points = torch.cat([xy * z, z], dim=-1)
        camera_poses = self.camera_head(camera_hidden.float()).reshape(1, n, 4, 4)
        global_points = self.pts_head("global_point_head", global_hidden.float(), H, W).reshape(1, n, H, W, -1)
        homo = torch.cat([local_points, torch.ones_like(local_points[..., :1])], dim=-1)
        points = torch.einsum("bnij, bnhwj -> bnhwi", camera_poses, homo)[..., :3]
        return points, local_points, camera_poses, global_points

    def recon(self, tokenizer, new_token_ids, images01, prompt="Reconstruct the 3D scene."):
        """g2vlm.py:1240-1303 with images already loaded as [N,3,H,W] in [0,1]."""
        cache = NaiveCache(self.num_layers)
        gi, newlens, new_rope = self.prepare_prompts([0], [0], ["Reconstruct the 3D scene."], tokenizer,
                                                     new_token_ids, bos=True)
        self.forward_cache_update_text(cache, **gi)
        gi, newlens, new_rope = self.prepare_dino_images(newlens, new_rope, images01, new_token_ids)
        cache, last = self.forward_cache_update_dino(cache, gi)
        return self.reconstruct(last, gi)

    def prepare_vit_image(self, curr_kvlens, curr_rope, pixel_values, grid_thw, new_token_ids):
        gi, newlens, new_rope = self.prepare_images(curr_kvlens, curr_rope, [tuple(int(v) for v in grid_thw)],
                                                    new_token_ids, merge=2)
        gi["packed_vit_images"] = pixel_values
        gi["packed_image_grid_thw"] = torch.tensor([list(grid_thw)])
        gi["vit_token_seqlens"] = gi.pop("token_seqlens")
        gi["packed_vit_token_indexes"] = gi.pop("packed_token_indexes")
        return gi, newlens, new_rope

    def forward_cache_update_vit(self, cache, gi, vit_layers=None):
        """g2vlm.py:812-866"""
        text_emb = self.embed_tokens(gi["packed_text_ids"])
        seq = text_emb.new_zeros((int(gi["packed_seqlens"].sum()), self.hidden_size))
        seq[gi["packed_text_indexes"]] = text_emb
        emb = self.vit_forward(gi["packed_vit_images"], gi["packed_image_grid_thw"][0], vit_layers)
        seq[gi["packed_vit_token_indexes"]] = emb.to(seq.dtype)
        self.llm_forward_inference(seq, gi["packed_seqlens"], gi["packed_position_ids"], gi["packed_indexes"], cache,
                                   gi["key_values_lens"], gi["packed_key_value_indexes"], False, "und")
        return cache

    def generate_text(self, cache, kvlen, rope_pos, start_token, max_length, end_token_id=None, return_logits=False):
        """generate_text (g2vlm.py:1070-1141), batch 1, greedy.  bf16 logits, argmax = first max."""
        seq, logits_all = [], []
        tok = torch.tensor([start_token], dtype=torch.long)
        step = 0
        while step < max_length:
            seq.append(int(tok[0]))
            x = self.embed_tokens(tok)
            pos = torch.full((3, 1), rope_pos, dtype=torch.long)
            h = self.llm_forward_inference(x, torch.tensor([1], dtype=torch.int), pos, torch.tensor([kvlen]), cache,
                                           torch.tensor([kvlen], dtype=torch.int), torch.arange(kvlen), True, "und")
            logits = self.lin(h, "language_model.lm_head", bias=False)
            if return_logits:
                logits_all.append(logits[0].float())
            tok = torch.argmax(logits, dim=-1)
            kvlen += 1; rope_pos += 1; step += 1
            if end_token_id is not None and int(tok[0]) == end_token_id:
                break
        return (seq, logits_all) if return_logits else seq

    def chat_prefill(self, tokenizer, new_token_ids, images01, vit_inputs, prompt):
        """The cache-building half of chat_with_recon (g2vlm.py:1305-1398).  Returns (cache, kv length, rope position,
        start token) = what generate_text continues from."""
        cache = NaiveCache(self.num_layers)
        sys_p = "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n"
        gi, newlens, new_rope = self.prepare_prompts([0], [0], [sys_p], tokenizer, new_token_ids)
        self.forward_cache_update_text(cache, **gi)
        gi, newlens, new_rope = self.prepare_dino_images(newlens, new_rope, images01, new_token_ids)
        self.forward_cache_update_dino(cache, gi)
        for pv, thw in vit_inputs:
            gi, newlens, new_rope = self.prepare_vit_image(newlens, new_rope, pv, thw, new_token_ids)
            self.forward_cache_update_vit(cache, gi)
        gi, newlens, new_rope = self.prepare_prompts(newlens, new_rope, [prompt + "<|im_end|>\n<|im_start|>assistant"],
                                                     tokenizer, new_token_ids)
        self.forward_cache_update_text(cache, **gi)
        template = "<|im_start|>user\\your text<|im_end|>\n<|im_start|>assistant\n"
        start = tokenizer.encode(template, add_special_tokens=False)[-1]
        return cache, newlens[0], new_rope[0], start

    def chat_with_recon(self, tokenizer, new_token_ids, images01, vit_inputs, prompt, max_length, return_logits=False):
        """chat_with_recon (g2vlm.py:1305-1410); images01 [N,3,H,W]; vit_inputs list of
        (pixel_values, grid_thw).  Returns generated ids (start token first, as the reference)."""
        cache, kvlen, rope_pos, start = self.chat_prefill(tokenizer, new_token_ids, images01, vit_inputs, prompt)
        return self.generate_text(cache, kvlen, rope_pos, start, max_length, new_token_ids["eos_token_id"], return_logits)
