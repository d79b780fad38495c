"""G2VLM: the reference's inference surface (reference modeling/g2vlm/g2vlm.py:119-1410) on the
MI355X engine.  Same method names, keyword names and return conventions, so
`inference_recon.py` / `inference_chat.py` / the notebook drive this class unchanged:

    recon(tokenizer, new_token_ids, dino_image_transform, images) -> dict
    chat_with_recon(tokenizer, new_token_ids, image_transform, dino_image_transform, images, prompt, max_length) -> str
    prepare_prompts*, prepare_dino_images_pi3, prepare_vit_images, prepare_start_tokens,
    forward_cache_update_{text,dino,vit}, reconstruct, generate_text

Training (`forward`) is out of scope.  There is no CPU path: every stage
method calls into libg2vlm_hip.so and raises if it is missing.
"""
import os

import torch

from ... import hip, host
from ...engine import Engine
from ...weights import Weights
from .qwen2vl import NaiveCache


class G2VLMConfig:
    """reference g2vlm.py:79-116"""

    def __init__(self, visual_und=True, visual_recon=True, joint_train_recon=False, pretrain_train_recon=False,
                 use_dinov3=False, ce_loss_dino=False, train_conf_pi3=False, llm_config=None, vit_config=None,
                 dino_config=None, latent_patch_size=2, max_latent_size=32, vit_max_num_patch_per_side=70,
                 dino_max_num_patch_per_side=37, interpolate_pos=False, use_registers=False, **kwargs):
        self.visual_und, self.visual_recon, self.train_conf_pi3 = visual_und, visual_recon, train_conf_pi3
        self.llm_config, self.vit_config, self.dino_config = llm_config, vit_config, dino_config
        self.vit_max_num_patch_per_side, self.dino_max_num_patch_per_side = vit_max_num_patch_per_side, dino_max_num_patch_per_side
        self.use_dinov3, self.use_registers, self.interpolate_pos = use_dinov3, use_registers, interpolate_pos
        if use_registers:
            raise NotImplementedError("LLM-side register tokens (g2vlm.py:146-155) are a training-time option the inference "
                                      "path never reads (patch_start_idx is unused in reconstruct); SURVEY §8(f)")


V3_FIELDS = ("patch_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "hidden_act",
             "layer_norm_eps", "rope_theta", "query_bias", "key_bias", "value_bias", "proj_bias", "mlp_bias", "layerscale_value",
             "use_gated_mlp", "num_register_tokens")


def dims_from_configs(llm, vit, dino, use_dinov3=False):
    """Engine dims dict from the three config objects; asserts the hard-coded reference assumptions.
    use_dinov3 (reference g2vlm.py:134, 169-172): `dino` is a DINOv3ViTConfig; the heads and grids then use patch 16."""
    hd = llm.hidden_size // llm.num_attention_heads
    assert hd == 128, "mrope sections [16,24,24] are hard-coded for head_dim 128 (modeling_qwen2_vl.py:561-566)"
    if use_dinov3:
        assert dino.patch_size == 16, ("use_dinov3 hard-codes patch 16 in the heads and the position grid (reference g2vlm.py:170, "
                                       f"1172-1174); got dino_config.patch_size={dino.patch_size}")
    else:
        assert dino.patch_size == 14 and dino.num_register_tokens == 4 and not dino.use_swiglu_ffn, (
            "dino_config.json must describe DINOv2-L/14 with 4 registers and a GELU MLP (patch_size 14: the heads hard-code it, "
            f"reference g2vlm.py:172); got patch_size={dino.patch_size}, registers={dino.num_register_tokens}")
    assert llm.hidden_size % 16 == 0
    d = {
        "llm": dict(hidden=llm.hidden_size, layers=llm.num_hidden_layers, heads=llm.num_attention_heads,
                    kv_heads=llm.num_key_value_heads, ffn=llm.intermediate_size, vocab=llm.vocab_size,
                    eps=llm.rms_norm_eps, theta=float(llm.rope_theta)),
        "dino": dict(hidden=dino.hidden_size, layers=dino.num_hidden_layers, heads=dino.num_attention_heads),
        "dec": dict(depth=5, heads=16),
    }
    if use_dinov3:
        d["dino"].update(patch=16, v3={k: getattr(dino, k) for k in V3_FIELDS})
    if vit is not None:
        d["vit"] = dict(embed=vit.embed_dim, depth=vit.depth, heads=vit.num_heads, mlp_ratio=vit.mlp_ratio, out=vit.hidden_size)
    else:
        d["vit"] = dict(embed=0, depth=0, heads=1, mlp_ratio=4, out=0)
    return d


def _cpu(t):
    return t.cpu() if torch.is_tensor(t) and t.is_cuda else t


class G2VLM:
    config_class = G2VLMConfig

    def __init__(self, language_model, vit_model, dino_model, config):
        self.language_model, self.vit_model, self.dino_model, self.config = language_model, vit_model, dino_model, config
        self.dims = dims_from_configs(config.llm_config, config.vit_config if config.visual_und else None, config.dino_config,
                                      use_dinov3=config.use_dinov3)
        self.use_dinov3 = config.use_dinov3
        self.dino_patch_size = self.dims["dino"].get("patch", 14)          # reference g2vlm.py:140
        self.hidden_size = self.dims["llm"]["hidden"]
        self.use_moe = "Mo" in getattr(config.llm_config, "layer_module", "Qwen2VLMoTDecoderLayer")
        self.use_decode_graph = True         # capture the per-token step in a hipGraph (generate_text)
        self.sample_seed = 0                 # Philox key of the next do_sample call (incremented per call)
        self.batch_vit_prefill = True        # chat prefill: consecutive equal-grid images as one ViT + und pass (forward_cache_update_vit_multi)
        self._sd = None
        self.weights = None
        self._idx_cache = {}
        self.engine = None
        self.device = None

    # ---- nn.Module-ish surface used by the loader / scripts
    def load_state_dict(self, state_dict, strict=False):
        from collections import namedtuple
        self._sd = state_dict                              # any mapping key -> fp32 tensor (dict, safetensors, synthetic)
        return namedtuple("IncompatibleKeys", "missing_keys unexpected_keys")([], [])

    def to(self, device):
        if self._sd is None:
            raise RuntimeError("load_state_dict() first")
        hip.lib()                                         # fail loudly before touching the device if the .so is absent
        self.device = torch.device(device)
        self.weights = Weights(self._sd, self.dims, self.device)
        if self.config.train_conf_pi3 and not self.weights.has_conf:
            raise KeyError("config.train_conf_pi3 is set but the state dict has no conf_decoder / conf_head tensors "
                           "(reference g2vlm.py:209-219 builds them for such checkpoints)")
        self.engine = Engine(self.weights, self.dims)
        self._sd = None
        return self

    def cuda(self):
        return self.to("cuda")

    def eval(self):
        return self

    def parameters(self):
        return iter(self.weights.t.values())

    def _dev_i32(self, t):
        """int32 device copy of a small host index tensor, memoised by content: the same bookkeeping (rows, positions,
        marker ids) recurs for every scene of a given shape, and a pageable H2D copy is a stream-wide sync point."""
        t = t.to(torch.int32).contiguous()
        if t.is_cuda:
            return t
        key = (tuple(t.shape), hash(t.numpy().tobytes()))
        hit = self._idx_cache.get(key)
        if hit is not None and torch.equal(hit[0], t):
            return hit[1]
        d = hip.h2d(t, self.device, resident=True)
        if len(self._idx_cache) >= 256:
            self._idx_cache.clear()
        self._idx_cache[key] = (t.clone(), d)
        return d

    # ---- text
    def prepare_prompts_addbos(self, curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids):
        return host.prepare_text(curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids, bos=True)

    def prepare_prompts_addeos(self, curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids):
        return host.prepare_text(curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids, eos_bos_assistant=True)

    def prepare_prompts_pure_text(self, curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids):
        return host.prepare_text(curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids)

    def prepare_prompts(self, curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids):
        return host.prepare_text(curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids, bos=True, eos=True)

    @torch.no_grad()
    def forward_cache_update_text(self, past_key_values, packed_text_ids, packed_text_position_ids, text_token_lens,
                                  packed_text_indexes, packed_key_value_indexes, key_values_lens):
        """reference g2vlm.py:701-733: und expert, causal, appends len(ids) KV rows."""
        kv_len = int(_cpu(key_values_lens).sum())
        L = packed_text_ids.numel()
        idx = _cpu(packed_text_indexes)
        assert past_key_values.length == kv_len and torch.equal(idx, torch.arange(kv_len, kv_len + L)), \
            "append-only cache: batch 1, new rows directly after the past rows (reference asserts B==1 too)"
        x = torch.empty((L, self.hidden_size), dtype=torch.float32, device=self.device)
        self.engine.embed(self._dev_i32(packed_text_ids), x)
        self.engine.llm_forward(x, 0, self._dev_i32(_cpu(packed_text_position_ids)), self._dev_i32(idx), past_key_values,
                                kv_len, causal=True, und_rounding=1)
        return past_key_values

    # ---- DINO / geo expert
    def prepare_dino_images_pi3(self, curr_kvlens, curr_rope, images, transforms, new_token_ids):
        """reference g2vlm.py:868-966.  `images`: list of paths / PIL images, or an [N,3,H,W] tensor in [0,1]."""
        # Images go to the device as early and as small as possible: paths / PIL images as the loader's uint8 frames (a quarter
        # of the fp32 bytes), tensors as they are; ToTensor's k/255, the ImageNet normalisation (g2vlm.py:950) and the
        # original_images copy then run in one kernel whose outputs are bit-identical to the host ops.  The dict keeps the
        # reference's keys and values; the two image tensors just live on the model's device already.
        ps = self.dino_patch_size
        if torch.is_tensor(images) or ps != 14:
            # use_dinov3: the reference's own prepare hard-codes load_and_resize14 and a //14 grid next to patchify(., 16)
            # (g2vlm.py:881, 899, 906) and cannot serve that variant; its loader for it is load_and_resize16
            # (transforms_vggt.py:464), whose last step is a real resize - done on the host, fp32 frames uploaded
            imgs = host.load_and_resize14(images if torch.is_tensor(images) else list(images), 518, patch=ps)
            assert imgs.dim() == 4 and imgs.shape[1] == 3
            n, _, hh, ww = imgs.shape
            frames = hip.h2d(imgs, self.device, torch.float32)
        else:
            frames = host.load_images_u8(list(images), 518, device=self.device)    # LANCZOS on the device for plain RGB frames
            n, hh, ww, _ = frames.shape
            if hh % 14 or ww % 14:                          # cannot happen for width 518 (the loader rounds the height to /14)
                frames = hip.h2d(host.load_and_resize14(list(images), 518), self.device, torch.float32)
                n, _, hh, ww = frames.shape
            else:
                frames = hip.h2d(frames, self.device)
        gi, newlen, new_rope = host.prepare_image_tokens(curr_kvlens[0], curr_rope[0], [(1, hh // ps, ww // ps)] * n, new_token_ids)
        gi["packed_dino_images"], gi["original_images"] = hip.dino_preprocess(frames, host.RESNET_MEAN, host.RESNET_STD)
        gi["dino_token_seqlens"] = gi.pop("token_seqlens")
        gi["packed_dino_token_indexes"] = gi.pop("packed_token_indexes")
        return gi, [newlen], [new_rope]

    @torch.no_grad()
    def forward_cache_update_dino(self, past_key_values, packed_text_ids, packed_text_indexes, packed_dino_token_indexes,
                                  dino_token_seqlens, packed_position_ids, packed_seqlens, packed_indexes,
                                  packed_key_value_indexes, key_values_lens, packed_dino_images, original_images,
                                  num_layers=None, dino_layers=None, before_llm=None):
        """reference g2vlm.py:968-1039.  Returns (cache, last_hidden fp32 [Lq,H] in packed order).
        before_llm: called after the DINO encoder has been enqueued and before the first kernel that reads the cache
        (prefill_text_and_dino joins the text prefix's stream there)."""
        hp, eng = hip, self.engine
        H = self.hidden_size
        imgs = hip.h2d(packed_dino_images, self.device, torch.float32).contiguous()
        N, _, Hh, Ww = imgs.shape
        assert N >= 1
        ps = self.dino_patch_size
        P = (Hh // ps) * (Ww // ps)
        lens = _cpu(dino_token_seqlens)
        assert bool((lens == P).all()), "all views share one grid (load_images resizes every view to the first one's size)"
        kv_len = int(_cpu(key_values_lens).sum())
        assert past_key_values.length == kv_len
        geo_idx, text_idx = _cpu(packed_dino_token_indexes).long(), _cpu(packed_text_indexes).long()
        Lq = int(_cpu(packed_seqlens).sum())
        perm = torch.cat([geo_idx, text_idx])
        assert perm.numel() == Lq
        x = torch.empty((Lq, H), dtype=torch.float32, device=self.device)
        # geo rows: DINO tokens -> dino2llm (bf16 Linear, widened to the fp32 stream)
        if self.use_dinov3:
            # the reference's inference method passes `packed_pixel_values=` (g2vlm.py:997-1001), which the DINOv3 module does
            # not take; this is the call its training forward makes for that variant (g2vlm.py:380-386): same cu_seqlens
            # (cumulative PATCH counts, hazard H1), patch tokens [N, P, C] back
            cu = torch.nn.functional.pad(torch.cumsum(lens.long(), 0), (1, 0)).tolist()
            tok = self.weights.dinov3(imgs, cu, int(lens.max()), num_layers=dino_layers)       # fp32 [N, P, C]
            if eng.taps is not None:
                eng.taps["dino_tokens"] = tok.clone()
            hp.linear(hp.cast_bf16(tok.reshape(N * P, -1)), self.weights["dino2llm.w"], self.weights["dino2llm.b"], hp.EPI_RES_F32,
                      out=x[:N * P])
        else:
            tok = eng.dino_forward(imgs, int(lens[0]), dino_layers)                             # bf16 [N*(P+5), C]
            if eng.taps is not None:
                eng.taps["dino_tokens"] = tok.view(N, P + 5, -1)[:, 5:].clone()
            tok32 = hp.linear(tok, self.weights["dino2llm.w"], self.weights["dino2llm.b"], hp.EPI_RES_F32)
            patch_rows = (torch.arange(N).view(-1, 1) * (P + 5) + 5 + torch.arange(P).view(1, -1)).reshape(-1)
            hp.gather_rows(tok32, self._dev_i32(patch_rows), x[:N * P])
        # und rows: <|vision_start|>/<|vision_end|> embeddings
        eng.embed(self._dev_i32(packed_text_ids), x[N * P:])
        pos = self._dev_i32(_cpu(packed_position_ids)[:, perm])
        kv_rows = self._dev_i32(_cpu(packed_indexes)[perm])
        if before_llm is not None:
            before_llm()
        last = eng.llm_forward(x, N * P, pos, kv_rows, past_key_values, kv_len, causal=False, und_rounding=0, num_layers=num_layers)
        last_hidden = torch.empty_like(last)
        hp.scatter_rows(last, self._dev_i32(perm), last_hidden)
        return past_key_values, last_hidden

    @torch.no_grad()
    def prefill_text_and_dino(self, past_key_values, text_inputs, dino_inputs, num_layers=None, dino_layers=None):
        """forward_cache_update_text(**text_inputs) followed by forward_cache_update_dino(**dino_inputs) - the first two stages
        of recon (reference g2vlm.py:1262-1290) - with the text prefix on a side stream.  The prefix (a dozen rows through the
        28 und-expert layers: ~250 launches of a few microseconds each, weight-streaming GEMVs) and the DINO encoder (24
        layers of MFMA-bound kernels) do not depend on each other; the first kernel that needs both is the MoT prefill, which
        reads the prefix's K/V rows.  Same kernels, same bits; the cache is sized here, on the caller's stream, so that no
        buffer changes hands between the streams.  G2V_TEXT_OVERLAP=0: one after the other (A/B)."""
        need = int(_cpu(dino_inputs["key_values_lens"]).sum()) + int(_cpu(dino_inputs["packed_seqlens"]).sum())
        past_key_values.reserve(need)
        if os.environ.get("G2V_TEXT_OVERLAP", "1") == "0" or torch.cuda.is_current_stream_capturing():
            past_key_values = self.forward_cache_update_text(past_key_values, **text_inputs)
            return self.forward_cache_update_dino(past_key_values, **dino_inputs, num_layers=num_layers, dino_layers=dino_layers)
        cur = torch.cuda.current_stream()
        side = self.engine.side_stream(cur)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            past_key_values = self.forward_cache_update_text(past_key_values, **text_inputs)
        return self.forward_cache_update_dino(past_key_values, **dino_inputs, num_layers=num_layers, dino_layers=dino_layers,
                                              before_llm=lambda: cur.wait_stream(side))

    @torch.no_grad()
    def reconstruct(self, past_key_values, packed_key_value_indexes, key_values_lens, selected_hidden_states,
                    packed_dino_token_indexes, packed_dino_images, original_images, **kwargs):
        """reference g2vlm.py:1143-1238"""
        hp, eng = hip, self.engine
        N, _, Hh, Ww = packed_dino_images.shape
        gh, gw = Hh // self.dino_patch_size, Ww // self.dino_patch_size        # reference g2vlm.py:1172-1177
        P = gh * gw
        hidden = torch.empty((N * P, self.hidden_size), dtype=torch.float32, device=self.device)
        hp.gather_rows(selected_hidden_states, self._dev_i32(_cpu(packed_dino_token_indexes)), hidden)
        point_hidden, camera_hidden, global_hidden, points, local, poses, glob = eng.decoders_and_heads(hidden, hidden[:P], N, gh, gw, Hh, Ww)
        if eng.taps is not None:
            eng.taps.update(point_hidden=point_hidden.clone(), camera_hidden=camera_hidden.clone(), global_hidden=global_hidden.clone())
        conf = None
        if self.weights.has_conf:                          # reference g2vlm.py:1192-1193, 1208-1210
            conf = eng.conf_head(eng.decoder("conf_decoder", hidden, N, gh, gw), N, Hh, Ww).unsqueeze(0)
        oi = hip.h2d(original_images, self.device)
        if oi.dim() == 4:
            oi = oi.unsqueeze(0)
        return dict(points=points.unsqueeze(0), local_points=local.unsqueeze(0), conf=conf, camera_poses=poses.unsqueeze(0),
                    global_points=glob.unsqueeze(0), images=oi)

    @torch.no_grad()
    def recon(self, tokenizer, new_token_ids, dino_image_transform, images, prompt="Reconstruct the 3D scene."):
        """reference g2vlm.py:1240-1303"""
        past = NaiveCache(self.dims["llm"]["layers"], self.dims["llm"]["kv_heads"], self.device)
        gi_text, newlens, new_rope = self.prepare_prompts_addbos([0], [0], ["Reconstruct the 3D scene."], tokenizer, new_token_ids)
        gi, newlens, new_rope = self.prepare_dino_images_pi3(newlens, new_rope, images, dino_image_transform, new_token_ids)
        past, last_hidden = self.prefill_text_and_dino(past, gi_text, gi)
        return self.reconstruct(past_key_values=past, selected_hidden_states=last_hidden, **gi)

    # ---- Qwen2-VL ViT / und expert
    def prepare_vit_images(self, curr_kvlens, curr_rope, images, transforms, new_token_ids):
        """reference g2vlm.py:735-810 (one image per call in chat_with_recon)."""
        assert len(images) == 1, "chat_with_recon feeds one image per call (g2vlm.py:1362-1370)"
        pixel_values, grid_thw = transforms([images[0]])
        t, gh, gw = (int(v) for v in grid_thw[0])
        gi, newlen, new_rope = host.prepare_image_tokens(curr_kvlens[0], curr_rope[0], [(t, gh, gw)], new_token_ids, merge=2)
        gi["packed_image_grid_thw"] = grid_thw[:1].clone()
        gi["packed_vit_images"] = pixel_values.unsqueeze(0)
        gi["vit_token_seqlens"] = gi.pop("token_seqlens")
        gi["packed_vit_token_indexes"] = gi.pop("packed_token_indexes")
        return gi, [newlen], [new_rope]

    @torch.no_grad()
    def forward_cache_update_vit(self, past_key_values, packed_text_ids, packed_text_indexes, packed_vit_images,
                                 packed_image_grid_thw, packed_vit_token_indexes, vit_token_seqlens, packed_position_ids,
                                 packed_seqlens, packed_indexes, packed_key_value_indexes, key_values_lens,
                                 packed_vit_tokens=None, packed_vit_position_ids=None, vit_layers=None):
        """reference g2vlm.py:812-866: ViT tokens + markers through the und expert, non-causal."""
        hp, eng = hip, self.engine
        H = self.hidden_size
        kv_len = int(_cpu(key_values_lens).sum())
        assert past_key_values.length == kv_len
        Lq = int(_cpu(packed_seqlens).sum())
        t, gh, gw = (int(v) for v in _cpu(packed_image_grid_thw)[0])
        kp = self.weights["vit.patch.w"].shape[1]
        if packed_vit_images.is_cuda and packed_vit_images.dtype == torch.bfloat16 and packed_vit_images.shape[-1] == kp:
            pv = packed_vit_images.reshape(-1, kp)              # host.QwenVL2ImageTransform(device=...): already the GEMM operand
        else:
            pv = _cpu(packed_vit_images).reshape(-1, packed_vit_images.shape[-1]).float()
            pv = hip.h2d(torch.nn.functional.pad(pv, (0, kp - pv.shape[1])), self.device)     # host zero-pad of K
        D = self.dims["vit"]["embed"] // self.dims["vit"]["heads"]
        cos, sin = host.vit_rot_pos(t, gh, gw, D)
        emb = eng.vit_forward(pv, (t, gh, gw), hip.h2d(cos, self.device), hip.h2d(sin, self.device), vit_layers)
        x = torch.empty((Lq, H), dtype=torch.float32, device=self.device)
        te = torch.empty((packed_text_ids.numel(), H), dtype=torch.float32, device=self.device)
        eng.embed(self._dev_i32(packed_text_ids), te)
        hp.scatter_rows(te, self._dev_i32(_cpu(packed_text_indexes)), x)
        hp.scatter_rows(hp.cast_f32(emb), self._dev_i32(_cpu(packed_vit_token_indexes)), x)
        eng.llm_forward(x, 0, self._dev_i32(_cpu(packed_position_ids)), self._dev_i32(_cpu(packed_indexes)), past_key_values,
                        kv_len, causal=False, und_rounding=1)
        return past_key_values

    @torch.no_grad()
    def forward_cache_update_vit_multi(self, past_key_values, inputs, vit_layers=None):
        """Several consecutive forward_cache_update_vit stages as ONE pass.  `inputs`: the generation-input dicts of
        consecutive prepare_vit_images calls (image j's cache rows directly after image j-1's), all images of one patch grid.
        The reference runs the stages one by one (g2vlm.py:1362-1370): image j's tokens attend to the cache, the earlier
        images and - non-causally - to themselves.  The same dependency structure is one prefill of all images' rows with one
        attention window per image, keys [0, end of image j): every token sees exactly the keys it sees in the reference's
        order, so the cache and every later logit agree with the stage-by-stage path up to GEMM summation order.  Why: a
        731-row prefill leaves most of the chip idle (its N = 1536 Linears are 18-72 tiles on 256 CUs) and pays the
        ~600 launches of a ViT + und pass per image; C5 has 8 images per scene."""
        hp, eng = hip, self.engine
        H, n = self.hidden_size, len(inputs)
        g0 = inputs[0]
        kv_len = int(_cpu(g0["key_values_lens"]).sum())
        assert past_key_values.length == kv_len
        t, gh, gw = (int(v) for v in _cpu(g0["packed_image_grid_thw"])[0])
        S = int(_cpu(g0["packed_seqlens"]).sum())
        kp = self.weights["vit.patch.w"].shape[1]
        pvs = []
        for j, gi in enumerate(inputs):
            assert tuple(int(v) for v in _cpu(gi["packed_image_grid_thw"])[0]) == (t, gh, gw), "one patch grid per batched pass"
            assert int(_cpu(gi["key_values_lens"]).sum()) == kv_len + j * S and int(_cpu(gi["packed_seqlens"]).sum()) == S
            pv = gi["packed_vit_images"]
            if pv.is_cuda and pv.dtype == torch.bfloat16 and pv.shape[-1] == kp:
                pvs.append(pv.reshape(-1, kp))
            else:
                pv = _cpu(pv).reshape(-1, pv.shape[-1]).float()
                pvs.append(hp.cast_bf16(hip.h2d(torch.nn.functional.pad(pv, (0, kp - pv.shape[1])), self.device)))
        D = self.dims["vit"]["embed"] // self.dims["vit"]["heads"]
        cos, sin = host.vit_rot_pos(t, gh, gw, D)
        emb = eng.vit_forward(torch.cat(pvs, 0), (t, gh, gw), hip.h2d(cos.repeat(n, 1), self.device), hip.h2d(sin.repeat(n, 1), self.device),
                              vit_layers, n_images=n)
        cat = lambda k, off=0: torch.cat([_cpu(gi[k]) + j * off for j, gi in enumerate(inputs)], -1)   # noqa: E731
        x = torch.empty((n * S, H), dtype=torch.float32, device=self.device)
        ids = cat("packed_text_ids")
        te = torch.empty((ids.numel(), H), dtype=torch.float32, device=self.device)
        eng.embed(self._dev_i32(ids), te)
        hp.scatter_rows(te, self._dev_i32(cat("packed_text_indexes", S)), x)
        hp.scatter_rows(hp.cast_f32(emb), self._dev_i32(cat("packed_vit_token_indexes", S)), x)
        wins = tuple((j * S, S, 0, kv_len + (j + 1) * S, False) for j in range(n))
        eng.llm_forward(x, 0, self._dev_i32(cat("packed_position_ids")), self._dev_i32(cat("packed_indexes")), past_key_values,
                        kv_len, causal=False, und_rounding=1, windows=wins)
        return past_key_values

    # ---- decode
    def prepare_start_tokens(self, curr_kvlens, curr_rope, tokenizer, new_token_ids):
        """reference g2vlm.py:1042-1068 (the template string, backslash included, is the reference's)."""
        template = "<|im_start|>user\\your text<|im_end|>\n<|im_start|>assistant\n"
        ids = tokenizer.encode(template, add_special_tokens=False)
        start = ids[-1] if ids else (tokenizer.eos_token_id or 151643)
        kv = sum(curr_kvlens)
        gi = {
            "packed_start_tokens": torch.tensor([start] * len(curr_kvlens), dtype=torch.long),
            "packed_query_position_ids": torch.tensor(list(curr_rope), dtype=torch.long).expand(3, -1),
            "key_values_lens": torch.tensor(list(curr_kvlens), dtype=torch.int),
            "packed_key_value_indexes": torch.arange(kv),
        }
        return gi

    @torch.no_grad()
    def generate_text(self, past_key_values, packed_key_value_indexes, key_values_lens, packed_start_tokens,
                      packed_query_position_ids, max_length, do_sample=False, temperature=1.0, end_token_id=None):
        """reference g2vlm.py:1070-1141, batch 1.  Greedy: argmax over the bf16 logits picks the first maximal index.
        do_sample: the next token is drawn from softmax(logits / temperature) on the device (Gumbel-max over a Philox
        stream seeded by `self.sample_seed`, csrc/misc.hip); torch's own multinomial stream is not reproduced, the
        distribution is (tests/test_kernels_gpu.py::test_sample_*)."""
        eng = self.engine
        assert packed_start_tokens.numel() == 1 and past_key_values.length == int(_cpu(key_values_lens).sum())
        pos = int(_cpu(packed_query_position_ids)[0, 0])
        st = eng.decode_begin(past_key_values, int(_cpu(packed_start_tokens)[0]), pos, max_length, use_graph=self.use_decode_graph,
                              sample=self._sample_arg(do_sample, temperature))
        ids = self._greedy_loop(lambda: eng.decode_step(st), st["tok"], 1, max_length, end_token_id)
        eng.decode_end(st)
        return torch.tensor(ids[0], dtype=torch.long).view(-1, 1)

    def _sample_arg(self, do_sample, temperature):
        """(seed, temperature) for the engine's sampler, or None for greedy.  Every sampled call takes the next seed of
        this model's stream, so two calls give different draws and a re-seeded model (`model.sample_seed = s`) repeats."""
        if not do_sample:
            return None
        if not float(temperature) > 0:
            raise ValueError("temperature must be > 0 when do_sample=True")
        self.sample_seed += 1
        return (self.sample_seed, float(temperature))

    def _greedy_loop(self, step_fn, tok, B, max_length, end_token_id, chunk=8):
        """The reference's greedy loop (g2vlm.py:1088-1135) without a host sync per token: `chunk` steps are launched back
        to back, their ids collected on the device, and the host looks for EOS once per chunk (a blocking read per step
        costs a wake-up of the host thread per token while the GPU idles).  Steps launched past a scene's EOS only append
        unused cache rows.  Returns, per scene, [start, t1, ...] cut before its first EOS and at max_length ids."""
        hist = torch.empty((max_length + 1, B), dtype=torch.int32, device=tok.device)
        hist[0].copy_(tok)
        n, stop = 0, [None] * B
        while n < max_length:
            m = min(chunk if end_token_id is not None else max_length, max_length - n)
            for i in range(m):
                hist[n + 1 + i].copy_(step_fn())
            lo, n = n, n + m
            if end_token_id is not None:
                new = hist[lo + 1:n + 1].cpu()
                for j in range(B):
                    if stop[j] is None:
                        hit = (new[:, j] == int(end_token_id)).nonzero()
                        if hit.numel():
                            stop[j] = lo + 1 + int(hit[0])
                if all(v is not None for v in stop):
                    break
        h = hist[:n + 1].cpu()
        return [h[:min(max_length, stop[j] if stop[j] is not None else max_length), j].tolist() for j in range(B)]

    def _chat_prefill(self, tokenizer, new_token_ids, image_transform, dino_image_transform, images, prompt):
        """The cache-building half of chat_with_recon (reference g2vlm.py:1305-1398): system prompt, geometry views,
        ViT views, question.  Returns the filled cache and the start-token inputs of generate_text."""
        past = NaiveCache(self.dims["llm"]["layers"], self.dims["llm"]["kv_heads"], self.device)
        sys_p = "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n"
        gi_text, newlens, new_rope = self.prepare_prompts_pure_text([0], [0], [sys_p], tokenizer, new_token_ids)
        gi, newlens, new_rope = self.prepare_dino_images_pi3(newlens, new_rope, list(images) if not torch.is_tensor(images) else images,
                                                             dino_image_transform, new_token_ids)
        past, _ = self.prefill_text_and_dino(past, gi_text, gi)      # the system prompt under the DINO encoder
        return self._chat_suffix(past, newlens, new_rope, tokenizer, new_token_ids, image_transform, images, prompt)

    def _chat_suffix(self, past, newlens, new_rope, tokenizer, new_token_ids, image_transform, images, prompt):
        """The stages of chat_with_recon after the geometry views (reference g2vlm.py:1360-1398): one ViT stage per image, then
        the question.  Split out because the view-sharded prefill (g2vlm_amd/sharded.py::chat_view_sharded) builds the
        geometry rows of `past` on several ranks and continues here on one."""
        gis = []
        for image in (images if not torch.is_tensor(images) else [None] * images.shape[0]):
            gi, newlens, new_rope = self.prepare_vit_images(newlens, new_rope, [image], image_transform, new_token_ids)
            if not self.batch_vit_prefill:
                past = self.forward_cache_update_vit(past, **gi)
                continue
            if gis and not torch.equal(_cpu(gis[0]["packed_image_grid_thw"]), _cpu(gi["packed_image_grid_thw"])):
                past = self._flush_vit(past, gis)              # another patch grid: close the run of equal-grid images
            gis.append(gi)
        past = self._flush_vit(past, gis)
        gi, newlens, new_rope = self.prepare_prompts_pure_text(newlens, new_rope, [prompt + "<|im_end|>\n<|im_start|>assistant"],
                                                               tokenizer, new_token_ids)
        past = self.forward_cache_update_text(past, **gi)
        return past, self.prepare_start_tokens(newlens, new_rope, tokenizer, new_token_ids)

    def _flush_vit(self, past, gis):
        if len(gis) == 1:
            past = self.forward_cache_update_vit(past, **gis[0])
        elif gis:
            past = self.forward_cache_update_vit_multi(past, gis)
        gis.clear()
        return past

    @torch.no_grad()
    def chat_with_recon(self, tokenizer, new_token_ids, image_transform, dino_image_transform, images, prompt, max_length,
                        do_sample=False, temperature=1.0):
        """reference g2vlm.py:1305-1410"""
        if do_sample and not float(temperature) > 0:
            raise ValueError("temperature must be > 0 when do_sample=True")       # before the prefill, not after it
        past, gi = self._chat_prefill(tokenizer, new_token_ids, image_transform, dino_image_transform, images, prompt)
        ids = self.generate_text(past_key_values=past, max_length=max_length, do_sample=do_sample, temperature=temperature,
                                 end_token_id=new_token_ids["eos_token_id"], **gi)
        return tokenizer.decode(ids[1:, 0])

    # ---- batched decode: several scenes answered together (the reference is batch 1, g2vlm.py:1006, 1137)
    @torch.no_grad()
    def generate_text_batch(self, pasts, start_inputs, max_length, end_token_id=None, do_sample=False, temperature=1.0):
        """generate_text for B scenes at once: pasts[j] / start_inputs[j] are what scene j's generate_text would be given.
        Greedy.  Every scene yields exactly the ids its own batch-1 generate_text yields (a scene that hits end_token_id
        stops contributing; the step keeps running for the others).  Returns a list of B LongTensors [n_j, 1]."""
        eng = self.engine
        B = len(pasts)
        starts = [int(_cpu(gi["packed_start_tokens"])[0]) for gi in start_inputs]
        poss = [int(_cpu(gi["packed_query_position_ids"])[0, 0]) for gi in start_inputs]
        for past, gi in zip(pasts, start_inputs):
            assert past.length == int(_cpu(gi["key_values_lens"]).sum())
        st = eng.decode_begin_batch(pasts, starts, poss, max_length, use_graph=self.use_decode_graph,
                                    sample=self._sample_arg(do_sample, temperature))
        out = self._greedy_loop(lambda: eng.decode_step_batch(st), st["tok"], B, max_length, end_token_id)
        return [torch.tensor(o, dtype=torch.long).view(-1, 1) for o in out]

    @torch.no_grad()
    def generate_text_stream(self, prefills, max_batch, max_length, max_kv_len, end_token_id=None, chunk=8, do_sample=False,
                             temperature=1.0):
        """Continuous batching: `prefills` is an iterable of zero-argument callables, each returning a
        freshly prefilled (cache, start_inputs) pair as generate_text takes them.  At most `max_batch` scenes decode
        together; every `chunk` steps the ids are read back, a scene that reached end_token_id or max_length leaves its
        slot, and the next prefill (run right there, between two replays of the captured step) takes it.  Every scene
        yields exactly the ids of its own batch-1 generate_text.  Returns the LongTensors [n_j, 1] in input order."""
        eng = self.engine
        it = iter(prefills)
        # a scene may run up to chunk - 1 steps past its last id before the host notices: room for those rows too
        st = eng.decode_open_slots(max_batch, max_kv_len + max_length + chunk + 1, use_graph=self.use_decode_graph,
                                   sample=self._sample_arg(do_sample, temperature))
        B = st["B"]
        slot_scene, outs, results = [None] * B, [None] * B, {}
        n_started = 0

        def refill():
            nonlocal n_started
            for j in range(B):
                if slot_scene[j] is not None:
                    continue
                mk = next(it, None)
                if mk is None:
                    eng.decode_idle_slot(st, j)
                    continue
                past, gi = mk()
                start = int(_cpu(gi["packed_start_tokens"])[0])
                eng.decode_set_slot(st, j, past, start, int(_cpu(gi["packed_query_position_ids"])[0, 0]), max_length + chunk)
                slot_scene[j], outs[j] = n_started, [start]
                n_started += 1

        refill()
        hist = torch.empty((chunk, B), dtype=torch.int32, device=self.device)
        while any(s_ is not None for s_ in slot_scene):
            for i in range(chunk):
                hist[i].copy_(eng.decode_step_batch(st))
            new = hist.cpu()
            for j in range(B):
                if slot_scene[j] is None:
                    continue
                for i in range(chunk):
                    t = int(new[i, j])
                    if (end_token_id is not None and t == int(end_token_id)) or len(outs[j]) >= max_length:
                        results[slot_scene[j]] = outs[j]
                        slot_scene[j] = None
                        break
                    outs[j].append(t)
                else:
                    if len(outs[j]) >= max_length:
                        results[slot_scene[j]] = outs[j]
                        slot_scene[j] = None
            refill()
        return [torch.tensor(results[i], dtype=torch.long).view(-1, 1) for i in range(n_started)]

    @torch.no_grad()
    def chat_with_recon_batch(self, tokenizer, new_token_ids, image_transform, dino_image_transform, scenes, max_length,
                              do_sample=False, temperature=1.0):
        """chat_with_recon over a list of (images, prompt) scenes: prefills run scene by scene (each is already a
        full-GPU job), the greedy decode runs for all scenes together so the und-expert weights stream once per step."""
        pasts, starts = [], []
        for images, prompt in scenes:
            past, gi = self._chat_prefill(tokenizer, new_token_ids, image_transform, dino_image_transform, images, prompt)
            pasts.append(past); starts.append(gi)
        ids = self.generate_text_batch(pasts, starts, max_length, end_token_id=new_token_ids["eos_token_id"], do_sample=do_sample,
                                       temperature=temperature)
        return [tokenizer.decode(i[1:, 0]) for i in ids]
