"""Model blocks of the G2VLM hot path, expressed as sequences of libg2vlm_hip.so kernel calls.

Host side only orchestrates: which rows go to which expert, which buffers alias, which windows
the attention covers.  Every FLOP and every byte moved on the device is in csrc/*.hip.

Row layout of the MoT prefill.  The reference keeps the packed sequence in arrival order and
routes with index gathers/scatters (qwen2vl.py:584-606, 861-865, 894-907: ~10 full-tensor
passes per layer).  Here the geo (patch) rows are kept FIRST and the und (marker / text) rows
LAST for the whole 28-layer stack, so each expert sees one contiguous row range: norms take a
`split` row, GEMMs run as a 2-group launch, nothing is gathered.  Attention is order-agnostic for
the non-causal geo prefill; K/V rows are still written to the cache at their reference positions
(kv_rows), so the cache content is identical to the reference's.
"""
import math
import os

import torch

from . import hip


class KVCache:
    """Pre-allocated contiguous replacement of NaiveCache (qwen2vl.py:237-251): same logical content
    (K post-RoPE, bf16, [len, Hkv, 128] per layer) without the per-call realloc + 4 scatters."""

    def __init__(self, num_layers, n_kv_heads=2, device="cuda", capacity=0):
        self.num_layers_, self.hkv, self.device = num_layers, n_kv_heads, device
        self.k = [None] * num_layers
        self.v = [None] * num_layers
        self.capacity = 0
        self.length = 0
        if capacity:
            self.reserve(capacity)

    def reserve(self, n):
        if n <= self.capacity:
            return
        cap = max(n, int(self.capacity * 1.5))
        for i in range(self.num_layers_):
            nk = torch.empty((cap, self.hkv, 128), dtype=torch.bfloat16, device=self.device)
            nv = torch.empty_like(nk)
            if self.k[i] is not None and self.length:
                nk[:self.length].copy_(self.k[i][:self.length]); nv[:self.length].copy_(self.v[i][:self.length])
            self.k[i], self.v[i] = nk, nv
        self.capacity = cap

    # NaiveCache-compatible views
    @property
    def num_layers(self):
        return self.num_layers_

    @property
    def seq_lens(self):
        return self.length

    @property
    def key_cache(self):
        return {i: (self.k[i][:self.length] if self.length else None) for i in range(self.num_layers_)}

    @property
    def value_cache(self):
        return {i: (self.v[i][:self.length] if self.length else None) for i in range(self.num_layers_)}


def rope2d_tables(D, seq_len, base=100.0):
    """RoPE2D.get_cos_sin (reference pos_embed.py:120-129) under autocast: the angle is rounded to
    bf16 BEFORE cos/sin (hazard H3).  One-off host table, uploaded by the caller."""
    inv_freq = 1.0 / (base ** (torch.arange(0, D, 2).float() / D))
    t = torch.arange(seq_len, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq).to(torch.bfloat16)
    freqs = torch.cat((freqs, freqs), dim=-1)
    return freqs.cos(), freqs.sin()


class Engine:
    def __init__(self, weights, dims):
        self.w = weights
        self.dims = dims
        self.dev = weights.device
        self._tiles = {}
        self._rope2d = {}
        self._dino_pos = {}
        self._zeros = {}
        self.attn_events = None      # bench.py: list collecting (start, stop) HIP events around every MoT prefill attention launch
        self.gemm_events = None      # bench.py: the same around every gate/up GEMM launch of the MoT prefill (the largest Linear)
        self.patch = dims["dino"].get("patch", 14)   # geometry encoder's patch size: 14 (DINOv2) or 16 (use_dinov3, g2vlm.py:170)
        self._decode_cached = {}     # capacity bucket -> captured batch-1 decode state (decode_begin)
        self._side_streams = {}
        # parity probes (tests/test_full_depth_gpu.py): with `taps` a dict, the fp32 residual stream after the MoT layers listed
        # in `tap_layers` (1-based; split row order), the DINO tokens and the decoder outputs are cloned into it
        self.taps, self.tap_layers = None, ()
        self._decode_gen = 2         # batch-1 decode kernels: 2 = persistent grids (csrc/decode_layer.hip), 1 = csrc/decode.hip

    @property
    def decode_gen(self):
        return self._decode_gen

    @decode_gen.setter
    def decode_gen(self, gen):
        """A/B switch of the decode kernels.  A captured step belongs to the generation it was captured with: drop it."""
        if gen not in (1, 2):
            raise ValueError("decode_gen: 1 (csrc/decode.hip) or 2 (csrc/decode_layer.hip, default)")
        if gen != self._decode_gen:
            self._decode_cached.clear()
        self._decode_gen = gen

    # ------------------------------------------------------------------ small caches
    def plan(self, windows, Hq):
        key = (tuple(windows), Hq)
        if key not in self._tiles:
            # long KV ranges (MoT global attention, ViT): 8-wave / 256-row workgroups share each K/V tile between twice
            # the queries (0.93 vs 0.82 PF at C3); short per-view windows keep the 4-wave form
            # ... when there are enough query rows to make 256-row items worth it: a 731-row ViT prefill or a long text
            # prompt against a 15 k-row cache runs 3-6 % faster on 128-row tiles (tools/attn_small_q.py)
            long_kv = sum(w[3] for w in windows if w[0] == windows[0][0]) >= 2048 and max(w[1] for w in windows) >= 2048
            # per-view windows (DINO, the Pi3 decoders): 256-row items are 5-8 % faster there too (tools/attn_windows.py: 118 ->
            # 109 us, 150 -> 142 us at 8 x 1369) as long as the last, partly filled tile does not eat the gain and there are
            # enough items to fill the chip twice over
            tall = False
            if not long_kv:
                pad = lambda t: sum((w[1] + t - 1) // t * t for w in windows) / max(1, sum(w[1] for w in windows))  # noqa: E731
                items = sum((w[1] + 255) // 256 for w in windows) * Hq
                tall = items >= 512 and pad(256) <= 1.12 * pad(128)
            self._tiles[key] = hip.make_attn_plan(windows, Hq, self.dev, tile_rows=256 if (long_kv or tall) else 128)
        return self._tiles[key]

    def rope2d_tab(self, D, gh, gw):
        key = (D, gh, gw)
        if key not in self._rope2d:
            cos, sin = rope2d_tables(D // 2, max(gh, gw))
            pos = torch.cartesian_prod(torch.arange(gh), torch.arange(gw)).to(torch.int32)
            self._rope2d[key] = (cos.to(self.dev), sin.to(self.dev), pos.to(self.dev))
        return self._rope2d[key]

    def dino_pos(self, H, W):
        """interpolate_pos_encoding (modeling_dinov2_with_registers.py:93-145); host, cached per shape."""
        key = (H, W)
        if key not in self._dino_pos:
            pe = self.w.dino_pos_cpu
            n_pos = pe.shape[1] - 1
            gh, gw = H // 14, W // 14
            if not (gh * gw == n_pos and H == W):
                dim = pe.shape[-1]
                s = int(n_pos ** 0.5)
                patch = pe[:, 1:].reshape(1, s, s, dim).permute(0, 3, 1, 2)
                patch = torch.nn.functional.interpolate(patch.float(), size=(gh, gw), mode="bicubic", align_corners=False,
                                                        antialias=True)
                pe = torch.cat((pe[:, :1], patch.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)
            self._dino_pos[key] = pe[0].contiguous().to(self.dev)
        return self._dino_pos[key]

    # ------------------------------------------------------------------ MoT LLM
    def llm_forward(self, x, split, pos_i32, kv_rows, cache, kv_len, causal, und_rounding, num_layers=None,
                    final_norm_dtype=torch.float32, kv_total=None, kv_exchange=None, local_kv=None, windows=None):
        """Qwen2VLModel.forward_inference (reference qwen2vl.py:1267-1337) on the split row layout.

        x fp32 [L,H] (updated in place): rows [0,split) use the geo expert, rows [split,L) the und
        expert.  K/V rows are written to cache rows kv_rows; attention covers cache rows
        [0, kv_len + L).  Returns the routed final norm of x.

        View-sharded prefill (g2vlm_amd/sharded.py): x holds only this rank's rows, `kv_total` is the GLOBAL number of
        cache rows after this call, `local_kv` = (first row, rows) of this rank's own K/V block in the cache and
        `kv_exchange` the per-layer K/V exchange: .start(layer) launches the all-gather of the ranks' blocks right after the
        cache write, the attention then runs over the LOCAL block while the remote blocks travel (phase 0 of the plan),
        .wait(layer) joins, and the second launch attends to the prefix and the remote blocks and merges.

        `windows`: attention windows (q0, q_len, k0, k_len, causal) replacing the single [0, L) x [0, kv_len + L) one - several
        stages of the reference's stage-by-stage prefill run as ONE pass (G2VLM.forward_cache_update_vit_multi).
        """
        w, hp = self.w, hip
        Lc = self.dims["llm"]
        H, Hq, Hkv, eps = Lc["hidden"], Lc["heads"], Lc["kv_heads"], Lc["eps"]
        L = x.shape[0]
        tot = kv_len + L if kv_total is None else kv_total
        cache.reserve(tot)
        cos, sin = hp.mrope_table(pos_i32, w["inv_freq"])
        if kv_exchange is not None:
            assert not causal and local_kv is not None
            r0, nr = local_kv
            wins = [(0, L, r0, nr, False, 0)]
            if r0 > 0:
                wins.append((0, L, 0, r0, False, 1))
            if r0 + nr < tot:
                wins.append((0, L, r0 + nr, tot - r0 - nr, False, 1))
            plan = self.plan(tuple(wins), Hq)
        elif windows is not None:
            plan = self.plan(tuple(windows), Hq)
        else:
            plan = self.plan(((0, L, 0, tot, bool(causal)),), Hq)
        nq, nqkv = Hq * 128, (Hq + 2 * Hkv) * 128
        h = torch.empty((L, H), dtype=torch.bfloat16, device=self.dev)
        qkv = torch.empty((L, nqkv), dtype=torch.bfloat16, device=self.dev)
        qb = torch.empty((L, nq), dtype=torch.bfloat16, device=self.dev)
        ao = torch.empty((L, nq), dtype=torch.bfloat16, device=self.dev)
        act = torch.empty((L, Lc["ffn"]), dtype=torch.bfloat16, device=self.dev)
        ng, nu = split, L - split

        def groups(A, C, wname, bname=None, res=None, gamma=None, lda=None, ldc=None):
            gs = []
            for tag, r0, m, gam in (("geo", 0, ng, gamma), ("und", split, nu, None)):
                if m == 0:
                    continue
                gs.append(dict(A=A[r0:], W=w[wname.format(tag)], bias=w[bname.format(tag)] if bname else None,
                               C=C[r0:], res=res[r0:] if res is not None else None, gamma=gam, M=m))
            return gs

        for i in range(Lc["layers"] if num_layers is None else num_layers):
            p = f"L{i}."
            hp.rmsnorm(x, w[p + "geo.ln1"], w[p + "und.ln1"], split, eps, out=h)
            hp.gemm_bf16(groups(h, qkv, p + "{}.qkv.w", p + "{}.qkv.b"), nqkv, H, hp.EPI_BF16, out_ld=nqkv)
            hp.qknorm_mrope_cache(qkv, Hq, Hkv, w[p + "geo.qn"], w[p + "und.qn"], w[p + "geo.kn"], w[p + "und.kn"], split, eps,
                                  und_rounding, cos, sin, qb, cache.k[i], cache.v[i], kv_rows)
            if self.attn_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            kk, vv_ = cache.k[i][:tot].view(tot, Hkv * 128), cache.v[i][:tot].view(tot, Hkv * 128)
            if kv_exchange is not None:
                kv_exchange.start(i)
                hp.flash_attn(qb, kk, vv_, ao, plan, Hq, Hkv, 128, phase=0)
                kv_exchange.wait(i)
                for ph in range(1, len(plan.phases)):
                    hp.flash_attn(qb, kk, vv_, ao, plan, Hq, Hkv, 128, phase=ph)
            else:
                hp.flash_attn(qb, kk, vv_, ao, plan, Hq, Hkv, 128)
            if self.attn_events is not None:
                ev[1].record()
                self.attn_events.append((ev, L, tot))
            hp.gemm_bf16(groups(ao, x, p + "{}.o.w", None, res=x, gamma=w[p + "ls1"]), H, nq, hp.EPI_RES_F32, out_ld=H, ldres=H,
                         flags=hp.GAMMA_ROUND_BF16)
            hp.rmsnorm(x, w[p + "geo.ln2"], w[p + "und.ln2"], split, eps, out=h)
            if self.gemm_events is not None:
                gev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                gev[0].record()
            hp.gemm_bf16(groups(h, act, p + "{}.gu.w"), 2 * Lc["ffn"], H, hp.EPI_SWIGLU, out_ld=Lc["ffn"])
            if self.gemm_events is not None:
                gev[1].record()
                self.gemm_events.append((gev, L))
            hp.gemm_bf16(groups(act, x, p + "{}.down.w", None, res=x, gamma=w[p + "ls2"]), H, Lc["ffn"], hp.EPI_RES_F32, out_ld=H,
                         ldres=H, lda=Lc["ffn"], flags=hp.GAMMA_ROUND_BF16)
            if self.taps is not None and split > 0 and (i + 1) in self.tap_layers:
                self.taps[f"mot{i + 1}"] = x.clone()
        cache.length = max(cache.length, tot)
        return hp.rmsnorm(x, w["norm.geo"], w["norm.und"], split, eps, out_dtype=final_norm_dtype)

    def embed(self, ids_i32, out):
        return hip.gather_rows(self.w["embed"], ids_i32, out)

    # ------------------------------------------------------------------ DINOv2 encoder
    def dino_embed(self, images_norm):
        """Dinov2WithRegistersEmbeddings.forward (reference modeling_dinov2_with_registers.py:147-171): im2col GEMM,
        CLS, position table, 4 registers.  fp32 [N*(P+5), C]."""
        w, hp = self.w, hip
        N, _, H, W = images_norm.shape
        P = (H // 14) * (W // 14)
        cols = hp.im2col14(images_norm, w["dino.patch.w"].shape[1])
        emb = hp.linear(cols, w["dino.patch.w"], w["dino.patch.b"])
        return hp.dino_assemble(emb, w["dino.cls"], w["dino.regs"], self.dino_pos(H, W), N, P)

    def dino_layers(self, x, n_windows, window_len, num_layers=None):
        """Dinov2WithRegistersEncoder + final LayerNorm (reference dinov2_model.py:251-275, 351) on fp32 token rows x
        (updated in place).  Attention windows are [i*window_len, (i+1)*window_len) for i < n_windows; rows outside every
        window get a zero attention output (hazard H1).  Returns final-LN tokens bf16 [rows, C]."""
        w, hp = self.w, hip
        Dn = self.dims["dino"]
        C, nh = Dn["hidden"], Dn["heads"]
        T = x.shape[0]
        plan = self.plan(tuple((i * window_len, window_len, i * window_len, window_len, False) for i in range(n_windows)), nh)
        h = torch.empty((T, C), dtype=torch.bfloat16, device=self.dev)
        qkv = torch.empty((T, 3 * C), dtype=torch.bfloat16, device=self.dev)
        ao = torch.zeros((T, C), dtype=torch.bfloat16, device=self.dev)        # rows outside every window stay 0
        mid = torch.empty((T, 4 * C), dtype=torch.bfloat16, device=self.dev)
        for i in range(Dn["layers"] if num_layers is None else num_layers):
            p = f"D{i}."
            hp.layernorm(x, w[p + "norm1.w"], w[p + "norm1.b"], 1e-6, out=h)
            hp.linear(h, w[p + "qkv.w"], w[p + "qkv.b"], out=qkv)
            if n_windows > 0:
                hp.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], ao, plan, nh, nh, C // nh)
            hp.linear(ao, w[p + "dense.w"], w[p + "dense.b"], hp.EPI_RES_F32, out=x, res=x, gamma=w[p + "ls1"])
            hp.layernorm(x, w[p + "norm2.w"], w[p + "norm2.b"], 1e-6, out=h)
            hp.linear(h, w[p + "fc1.w"], w[p + "fc1.b"], hp.EPI_GELU, out=mid)
            hp.linear(mid, w[p + "fc2.w"], w[p + "fc2.b"], hp.EPI_RES_F32, out=x, res=x, gamma=w[p + "ls2"])
        return hp.layernorm(x, w["dino.ln.w"], w["dino.ln.b"], 1e-6, out=h)

    def dino_forward(self, images_norm, window_len, num_layers=None):
        """Dinov2WithRegistersModel.forward (reference dinov2_model.py:301-356).  images_norm fp32 [N,3,H,W] on device.
        Windows are [i*window_len, (i+1)*window_len) of the flat [N*(P+5)] token axis exactly as the reference builds
        them (hazard H1: window_len = P, so the last 5N rows get a zero attention output)."""
        return self.dino_layers(self.dino_embed(images_norm), images_norm.shape[0], window_len, num_layers)

    # ------------------------------------------------------------------ Pi3 decoders
    def decoder(self, name, hidden, N, gh, gw, context=None, depth=None):
        """Pi3TransformerDecoder / Pi3ContextTransformerDecoder (reference transformer_head.py:9-56,
        84-130; blocks block.py:259-405).  hidden fp32 [N*P, C]; context fp32 [P, C] = view 0
        (identical for every view, g2vlm.py:1196, so its K/V are computed once).  Returns bf16."""
        w, hp = self.w, hip
        C = self.dims["llm"]["hidden"]
        nh = self.dims["dec"]["heads"]
        D = C // nh
        P = gh * gw
        M = N * P
        cos, sin, pos = self.rope2d_tab(D, gh, gw)
        x = hidden.clone()
        h = torch.empty((M, C), dtype=torch.bfloat16, device=self.dev)
        qkv = torch.empty((M, 3 * C), dtype=torch.bfloat16, device=self.dev)
        ao = torch.empty((M, C), dtype=torch.bfloat16, device=self.dev)
        mid = torch.empty((M, 4 * C), dtype=torch.bfloat16, device=self.dev)
        self_plan = self.plan(tuple((v * P, P, v * P, P, False) for v in range(N)), nh)
        if context is not None:
            cross_plan = self.plan(tuple((v * P, P, 0, P, False) for v in range(N)), nh)
            yn = torch.empty((P, C), dtype=torch.bfloat16, device=self.dev)
            ckv = torch.empty((P, 2 * C), dtype=torch.bfloat16, device=self.dev)
            cq = torch.empty((M, C), dtype=torch.bfloat16, device=self.dev)
        for i in range(self.dims["dec"]["depth"] if depth is None else depth):
            p = f"{name}.{i}."
            hp.layernorm(x, w[p + "norm1.w"], w[p + "norm1.b"], 1e-6, out=h)
            hp.linear(h, w[p + "attn.qkv.w"], w[p + "attn.qkv.b"], out=qkv)
            hp.rope2d(qkv, 0, 2 * nh, D, cos, sin, pos, P)
            hp.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], ao, self_plan, nh, nh, D)
            hp.linear(ao, w[p + "attn.proj.w"], w[p + "attn.proj.b"], hp.EPI_RES_F32, out=x, res=x)
            if context is not None:
                hp.layernorm(context, w[p + "norm_y.w"], w[p + "norm_y.b"], 1e-6, out=yn)
                hp.linear(yn, w[p + "ckv.w"], w[p + "ckv.b"], out=ckv)
                hp.rope2d(ckv, 0, nh, D, cos, sin, pos, P)
                hp.layernorm(x, w[p + "norm2.w"], w[p + "norm2.b"], 1e-6, out=h)
                hp.linear(h, w[p + "cq.w"], w[p + "cq.b"], out=cq)
                hp.rope2d(cq, 0, nh, D, cos, sin, pos, P)
                hp.flash_attn(cq, ckv[:, :C], ckv[:, C:], ao, cross_plan, nh, nh, D)
                hp.linear(ao, w[p + "cproj.w"], w[p + "cproj.b"], hp.EPI_RES_F32, out=x, res=x)
                n_mlp = "norm3"
            else:
                n_mlp = "norm2"
            hp.layernorm(x, w[p + n_mlp + ".w"], w[p + n_mlp + ".b"], 1e-6, out=h)
            hp.linear(h, w[p + "mlp.fc1.w"], w[p + "mlp.fc1.b"], hp.EPI_GELU, out=mid)
            hp.linear(mid, w[p + "mlp.fc2.w"], w[p + "mlp.fc2.b"], hp.EPI_RES_F32, out=x, res=x)
        return hp.linear(hp.cast_bf16(x), w[name + ".out.w"], w[name + ".out.b"])

    def decoders_and_heads(self, hidden, context, N, gh, gw, H, W):
        """The three Pi3 decoders and the fp32 heads of G2VLM.reconstruct (reference g2vlm.py:1186-1226: point, camera, global
        decoder, then the heads).  Returns (point_hidden, camera_hidden, global_hidden, points, local_points, poses,
        global_points).  The global decoder and its head share nothing with the other two until the results are returned, so
        they run on a side stream; the caller's stream runs camera decoder -> camera head -> point decoder -> point head (the
        point head needs the poses).  Two sequences side by side fill each other's low-power stretches (layer norms, RoPE, the
        fp32 heads, single-round Linears' epilogues); the order of independent kernels changes nothing in their results.
        G2V_HEADS_OVERLAP: 0 = everything in sequence, 1 = only the camera / point heads on the side stream (A/B).
        Buffers cross the two streams without record_stream: every use of the side stream starts with side.wait_stream(caller's)
        and ends with the caller's stream waiting for it, so a block freed after the join is only ever reused behind that join
        in either stream's order - and a deferred free would make the caching allocator hipMalloc in steady state
        (tools/step_outliers.py: 13 device mallocs in 40 steps with record_stream, a device-wide stall each)."""
        P = gh * gw
        mode = "0" if torch.cuda.is_current_stream_capturing() else os.environ.get("G2V_HEADS_OVERLAP", "2")
        if mode == "0":
            point_hidden = self.decoder("point_decoder", hidden, N, gh, gw)
            camera_hidden = self.decoder("camera_decoder", hidden, N, gh, gw)
            global_hidden = self.decoder("global_points_decoder", hidden, N, gh, gw, context=context)
            points, local, poses, glob = self.heads(point_hidden, camera_hidden, global_hidden, N, H, W)
            return point_hidden, camera_hidden, global_hidden, points, local, poses, glob
        cur = torch.cuda.current_stream()
        side = self.side_stream(cur)
        if mode == "1":
            camera_hidden = self.decoder("camera_decoder", hidden, N, gh, gw)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                poses = self.camera_poses(camera_hidden, N, P)
            point_hidden = self.decoder("point_decoder", hidden, N, gh, gw)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                points, local = self.point_maps_local(point_hidden, poses, N, H, W)
            global_hidden = self.decoder("global_points_decoder", hidden, N, gh, gw, context=context)
            cur.wait_stream(side)
            glob = self.point_maps_global(global_hidden, N, H, W)
            return point_hidden, camera_hidden, global_hidden, points, local, poses, glob
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            global_hidden = self.decoder("global_points_decoder", hidden, N, gh, gw, context=context)
            glob = self.point_maps_global(global_hidden, N, H, W)
        camera_hidden = self.decoder("camera_decoder", hidden, N, gh, gw)
        poses = self.camera_poses(camera_hidden, N, P)
        point_hidden = self.decoder("point_decoder", hidden, N, gh, gw)
        points, local = self.point_maps_local(point_hidden, poses, N, H, W)
        cur.wait_stream(side)
        return point_hidden, camera_hidden, global_hidden, points, local, poses, glob

    def side_stream(self, cur):
        """One side stream per caller's stream (kept: stream creation is not free), for work that is independent of what
        the caller's stream is doing (G2VLM.prefill_text_and_dino)."""
        key = cur.cuda_stream
        if key not in self._side_streams:
            self._side_streams[key] = torch.cuda.Stream(device=self.dev)
        return self._side_streams[key]

    def conf_head(self, conf_hidden, N, H, W):
        """Confidence map of a train_conf_pi3 checkpoint (reference g2vlm.py:1208-1210): fp32 Linear 1024 -> patch^2 +
        pixel_shuffle(patch) -> [N, H, W, 1]."""
        cf = hip.gemm_f32(hip.cast_f32(conf_hidden), self.w["conf_head.w"], self.w["conf_head.b"])
        return hip.pixel_shuffle(cf, N, H, W, 1, self.patch)

    def camera_poses(self, camera_hidden, N, P):
        """Pi3CameraHead (reference camera_head.py:32-93; fp32, autocast off, g2vlm.py:1213-1215): 2 x [3 Linear + ReLU + skip]
        on the view's P tokens, mean over P, 2 x (Linear + ReLU), fc_t / fc_rot, row-normalise, SVD, det fix.
        camera_hidden bf16 [N*P, 512] -> fp32 [N, 4, 4]."""
        w, hp = self.w, hip
        feat = hp.cast_f32(camera_hidden)
        for i in range(2):
            t = hp.gemm_f32(feat, w[f"cam.res{i}.1.w"], w[f"cam.res{i}.1.b"], relu=True)
            t = hp.gemm_f32(t, w[f"cam.res{i}.2.w"], w[f"cam.res{i}.2.b"], relu=True)
            feat = hp.gemm_f32(t, w[f"cam.res{i}.3.w"], w[f"cam.res{i}.3.b"], relu=True, res=feat)
        return hp.camera_tail(feat, N, P, w["cam.mlp0.w"], w["cam.mlp0.b"], w["cam.mlp1.w"], w["cam.mlp1.b"],
                              w["cam.fc_t.w"], w["cam.fc_t.b"], w["cam.fc_rot.w"], w["cam.fc_rot.b"])

    def point_maps(self, point_hidden, global_hidden, poses, N, H, W):
        """Pi3LinearPts3d x 2 + the post-math (reference transformer_head.py:58-81, g2vlm.py:1200-1205, 1219-1226): fp32 Linear
        1024 -> 3 patch^2, pixel_shuffle, z = exp(z), (x z, y z, z), world points = pose . [local, 1].  Per patch: the hidden
        rows may be any (H / patch) x (W / patch) grid of patches per view.  Returns (points, local_points, global_points)."""
        points, local = self.point_maps_local(point_hidden, poses, N, H, W)
        return points, local, self.point_maps_global(global_hidden, N, H, W)

    def point_maps_local(self, point_hidden, poses, N, H, W):
        """The point head's half of point_maps: (world points, local points)."""
        w, hp = self.w, hip
        pf = hp.gemm_f32(hp.cast_f32(point_hidden), w["point_head.w"], w["point_head.b"])
        local, points = hp.pts_epilogue(pf, N, H, W, 1, poses, patch=self.patch)
        return points, local

    def point_maps_global(self, global_hidden, N, H, W):
        """The global point head's half of point_maps."""
        w, hp = self.w, hip
        gf = hp.gemm_f32(hp.cast_f32(global_hidden), w["global_point_head.w"], w["global_point_head.b"])
        glob, _ = hp.pts_epilogue(gf, N, H, W, 0, patch=self.patch)
        return glob

    def heads(self, point_hidden, camera_hidden, global_hidden, N, H, W):
        """fp32 islands of G2VLM.reconstruct (reference g2vlm.py:1200-1226)."""
        poses = self.camera_poses(camera_hidden, N, (H // self.patch) * (W // self.patch))
        points, local, glob = self.point_maps(point_hidden, global_hidden, poses, N, H, W)
        return points, local, poses, glob

    # ------------------------------------------------------------------ Qwen2-VL ViT
    def vit_forward(self, pixel_values, grid_thw, cos, sin, num_layers=None, n_images=1):
        """Qwen2VisionTransformerPretrainedModel.forward (reference modeling_qwen2_vl.py:1048-1072).
        pixel_values fp32 [T, Kpad] on device (K = 1176 zero-padded on the host to the patch GEMM's
        K); cos/sin fp32 [T, head_dim] on device.  Returns bf16 [T/4, out].
        n_images > 1: that many images of the same grid stacked along T (cos / sin tiled by the caller) - the reference
        runs them one call each (g2vlm.py:1362-1370); the tokens do not interact (one attention window per frame), so one
        pass over the stack gives each image its own result."""
        w, hp = self.w, hip
        V = self.dims["vit"]
        C, nh = V["embed"], V["heads"]
        D = C // nh
        T = pixel_values.shape[0]
        t, gh, gw = grid_thw
        t = t * n_images
        assert pixel_values.shape[1] == w["vit.patch.w"].shape[1] and T == t * gh * gw
        x = hp.linear(pixel_values if pixel_values.dtype == torch.bfloat16 else hp.cast_bf16(pixel_values), w["vit.patch.w"], None)
        plan = self.plan(tuple((i * gh * gw, gh * gw, i * gh * gw, gh * gw, False) for i in range(t)), nh)
        h = torch.empty((T, C), dtype=torch.bfloat16, device=self.dev)
        qkv = torch.empty((T, 3 * C), dtype=torch.bfloat16, device=self.dev)
        ao = torch.empty((T, C), dtype=torch.bfloat16, device=self.dev)
        mid = torch.empty((T, int(C * V["mlp_ratio"])), dtype=torch.bfloat16, device=self.dev)
        for i in range(V["depth"] if num_layers is None else num_layers):
            p = f"V{i}."
            hp.layernorm(x, w[p + "norm1.w"], w[p + "norm1.b"], 1e-6, out=h)
            hp.linear(h, w[p + "attn.qkv.w"], w[p + "attn.qkv.b"], out=qkv)
            hp.rope_vision(qkv, 2 * nh, D, cos, sin)
            hp.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], ao, plan, nh, nh, D)
            hp.linear(ao, w[p + "attn.proj.w"], w[p + "attn.proj.b"], hp.EPI_RES_BF16, out=x, res=x)
            hp.layernorm(x, w[p + "norm2.w"], w[p + "norm2.b"], 1e-6, out=h)
            hp.linear(h, w[p + "mlp.fc1.w"], w[p + "mlp.fc1.b"], hp.EPI_QUICKGELU, out=mid)
            hp.linear(mid, w[p + "mlp.fc2.w"], w[p + "mlp.fc2.b"], hp.EPI_RES_BF16, out=x, res=x)
        hp.layernorm(x, w["vit.ln_q.w"], w["vit.ln_q.b"], 1e-6, out=h)
        m = hp.linear(h.view(T // 4, 4 * C), w["vit.m0.w"], w["vit.m0.b"], hp.EPI_GELU)
        return hp.linear(m, w["vit.m2.w"], w["vit.m2.b"])

    # ------------------------------------------------------------------ batch-1 decode step
    def _decode_body(self, cache, st):
        """One token through 28 und-expert layers + final norm + lm_head + argmax (reference generate_text loop body,
        g2vlm.py:1088-1125).  Allocation-free and host-state-free: position, cache row and KV length live in `st`
        on the device and are advanced by the last kernel, so the whole step can be captured in a hipGraph."""
        w, hp = self.w, hip
        Lc = self.dims["llm"]
        H, Hq, Hkv, eps, Fd = Lc["hidden"], Lc["heads"], Lc["kv_heads"], Lc["eps"], Lc["ffn"]
        x = st["x"]
        xr = x.view(-1)
        hp.gather_rows(w["embed"], st["tok"], x)
        hp.mrope_table_into(st["pos"], w["inv_freq"], st["cos"], st["sin"])
        if self.decode_gen == 2:
            # persistent-grid kernels (csrc/decode_layer.hip): 256 workgroups with an equal share of the bytes per launch
            for i in range(Lc["layers"]):
                p = f"L{i}.und."
                hp.gemv_pg(xr, w[p + "qkv.w"], norm_w=w[p + "ln1"], eps=eps, bias=w[p + "qkv.b"], out=st["qkv"].view(-1))
                hp.decode_attn_pg(st["qkv"], w[p + "qn"], w[p + "kn"], eps, 1, st["cos"], st["sin"], cache.k[i], cache.v[i], st["ao"],
                                  st["len"], cache.capacity, st["attn_cap"], Hq, Hkv, 128 ** -0.5, st["ws2"])
                hp.gemv_pg(st["ao"].view(-1), w[p + "o.w"], res=xr)
                hp.gemv_pg(xr, w[p + "gu.w"], norm_w=w[p + "ln2"], eps=eps, out=st["act"], act=True)
                hp.gemv_pg(st["act"], w[p + "down.w"], res=xr)
            hp.gemv_pg(xr, w["lm_head"], norm_w=w["norm.und"], eps=eps, out=st["logits"])
        else:
            for i in range(Lc["layers"]):
                p = f"L{i}.und."
                hp.gemv_rmsnorm_bf16(xr, w[p + "ln1"], eps, w[p + "qkv.w"], w[p + "qkv.b"], st["qkv"].view(-1))
                hp.decode_attn_fused(st["qkv"], w[p + "qn"], w[p + "kn"], eps, 1, st["cos"], st["sin"], cache.k[i], cache.v[i], st["ao"],
                                     st["len"], cache.capacity, cache.capacity, Hq, Hkv, 128 ** -0.5, st["ws"])
                hp.gemv_bf16(st["ao"].view(-1), w[p + "o.w"], None, None, res=xr)
                hp.gemv_rmsnorm_swiglu_bf16(xr, w[p + "ln2"], eps, w[p + "gu.w"], st["act"])
                hp.gemv_bf16(st["act"], w[p + "down.w"], None, None, res=xr)
            hp.gemv_rmsnorm_bf16(xr, w["norm.und"], eps, w["lm_head"], None, st["logits"])
        if st.get("rng") is not None:                      # do_sample (reference g2vlm.py:1119-1122)
            hp.sample_rows_bf16(st["logits"], st["tok"], st["amax"], st["rng"])
        else:
            hp.argmax_bf16(st["logits"], st["tok"], st["amax"])
        hp.decode_advance(st["pos"], st["row"], st["len"])

    def _decode_state(self, cache, capacity):
        """Device-side state of the batch-1 decode step over `cache` (a KVCache whose tensors must not move any more)."""
        Lc = self.dims["llm"]
        H, Hq, Hkv, Fd = Lc["hidden"], Lc["heads"], Lc["kv_heads"], Lc["ffn"]
        d, bf = self.dev, torch.bfloat16
        i32 = lambda shape: torch.zeros(shape, dtype=torch.int32, device=d)
        return dict(pos=i32((3, 1)), row=i32((1,)), len=torch.ones((1,), dtype=torch.int32, device=d), tok=i32((1,)),
                    x=torch.empty((1, H), dtype=torch.float32, device=d), cos=torch.empty((1, 128), dtype=torch.float32, device=d),
                    sin=torch.empty((1, 128), dtype=torch.float32, device=d), h=torch.empty((1, H), dtype=bf, device=d),
                    qkv=torch.empty((1, (Hq + 2 * Hkv) * 128), dtype=bf, device=d), q=torch.empty((1, Hq * 128), dtype=bf, device=d),
                    ao=torch.empty((1, Hq * 128), dtype=bf, device=d), gu=torch.empty(2 * Fd, dtype=bf, device=d),
                    act=torch.empty(Fd, dtype=bf, device=d), logits=torch.empty(Lc["vocab"], dtype=bf, device=d),
                    ws=torch.empty(hip.decode_attn_workspace(capacity, Hq) // 4, dtype=torch.float32, device=d),
                    ws2=torch.empty(hip.decode_attn_pg_workspace(Hq, Hkv, 1) // 4, dtype=torch.float32, device=d),
                    amax=torch.zeros(129, dtype=torch.int32, device=d), graph=None, cache=cache, user_cache=None, base_len=0, steps=0)

    def decode_begin(self, cache, start_token, pos, max_new_tokens, use_graph=True, sample=None):
        """Point the device-side decode state at the first step after `cache` (a prefilled KVCache).

        use_graph: the step is replayed from a hipGraph.  Capturing it (a warm-up step, ~200 launches recorded, the graph
        instantiated) costs ~10 ms, 5 % of a 128-token answer, so the captured step is kept: it runs over an engine-owned
        KV block sized in 4096-row buckets, the caller's prefill rows are copied into it (630 MB at 11 k rows: 0.3 ms) and
        `decode_end` copies the appended rows back, which keeps NaiveCache's append semantics (qwen2vl.py:626-634) for the
        caller's cache.  Eager mode decodes in the caller's cache directly.

        sample = (seed, temperature): the next token is drawn from softmax(logits / temperature) (the reference's
        do_sample branch, g2vlm.py:1119-1122) instead of argmax; the sampler state lives on the device like the rest."""
        d = self.dev
        kv_len = cache.length
        need = kv_len + max_new_tokens + 1
        cap = (need + 4095) // 4096 * 4096                    # the attention splits its keys by this bucket: graph and eager alike
        if not use_graph:
            cache.reserve(cap)
            st = self._decode_state(cache, cache.capacity)
            st["attn_cap"] = cap
            if sample is not None:
                st["rng"] = hip.make_rng(sample[0], sample[1], d)
        else:
            key = (cap, sample is not None, self._decode_gen)
            st = self._decode_cached.get(key)
            if st is None:
                self._decode_cached.clear()                   # one bucket resident (0.35-0.6 GB each)
                own = KVCache(len(cache.k), self.dims["llm"]["kv_heads"], d, capacity=cap)
                st = self._decode_state(own, own.capacity)
                st["attn_cap"] = cap
                if sample is not None:
                    st["rng"] = hip.make_rng(sample[0], sample[1], d)
                s = torch.cuda.Stream(device=d)               # warm up once on a side stream: lazy module loads must not
                s.wait_stream(torch.cuda.current_stream())    # happen during capture
                with torch.cuda.stream(s):
                    self._decode_body(own, st)
                torch.cuda.current_stream().wait_stream(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._decode_body(own, st)
                st["graph"] = g
                self._decode_cached[key] = st
            own = st["cache"]
            for i in range(len(cache.k)):
                own.k[i][:kv_len].copy_(cache.k[i][:kv_len]); own.v[i][:kv_len].copy_(cache.v[i][:kv_len])
            own.length = kv_len
            st["user_cache"] = cache
        st["pos"].fill_(pos); st["row"].fill_(kv_len); st["len"].fill_(kv_len + 1); st["tok"].fill_(int(start_token))
        if sample is not None:
            st["rng"].copy_(hip.make_rng(sample[0], sample[1], d))      # step 0 of this call's stream (the capture warm-up drew once)
        st["base_len"], st["steps"] = kv_len, 0
        return st

    def decode_step(self, st):
        """Run one token.  Returns the device tensor holding the NEXT token id (int32 [1], overwritten every step)."""
        if st["graph"] is not None:
            st["graph"].replay()
        else:
            self._decode_body(st["cache"], st)
        st["steps"] += 1
        st["cache"].length = st["base_len"] + st["steps"]
        return st["tok"]

    def decode_end(self, st):
        """Give the caller's cache the rows the decode appended (graph mode decodes in an engine-owned block)."""
        user = st.get("user_cache")
        if user is None:
            return
        lo, hi = st["base_len"], st["base_len"] + st["steps"]
        user.reserve(hi)
        own = st["cache"]
        for i in range(len(user.k)):
            user.k[i][lo:hi].copy_(own.k[i][lo:hi]); user.v[i][lo:hi].copy_(own.v[i][lo:hi])
        user.length = hi
        st["user_cache"] = None

    # ------------------------------------------------------------------ batched decode
    def _decode_batch_body(self, st):
        """One token for each of B scenes (same weights, one KV cache each): the reference loop body (g2vlm.py:1088-1125)
        with its batch = 1 limit lifted.  Weights are streamed once per step for all scenes: every Linear is an M = B
        GEMM (skinny MFMA kernel), norms / RoPE / cache write are the row-batched prefill kernels addressing the packed
        cache [B * cap] by row, attention is the split-KV kernel with one grid slice per scene."""
        w, hp = self.w, hip
        Lc = self.dims["llm"]
        H, Hq, Hkv, eps, Fd = Lc["hidden"], Lc["heads"], Lc["kv_heads"], Lc["eps"], Lc["ffn"]
        B, cap = st["B"], st["cap"]
        x, h = st["x"], st["h"]
        nq, nqkv = Hq * 128, (Hq + 2 * Hkv) * 128
        hp.gather_rows(w["embed"], st["tok"], x)
        hp.mrope_table_into(st["pos"], w["inv_freq"], st["cos"], st["sin"])
        pgb = B <= 8 and st["attn_pg"]      # the persistent-grid GEMVs with B rows per weight pass (csrc/decode_batch.hip)
        for i in range(Lc["layers"]):
            p = f"L{i}.und."
            if pgb:
                hp.gemv_pg_batch(x, w[p + "qkv.w"], norm_w=w[p + "ln1"], eps=eps, bias=w[p + "qkv.b"], out=st["qkv"])
                hp.decode_attn_pg(st["qkv"], w[p + "qn"], w[p + "kn"], eps, 1, st["cos"], st["sin"], st["k"][i], st["v"][i], st["ao"],
                                  st["len"], cap, cap, Hq, Hkv, 128 ** -0.5, st["ws2"])
                hp.gemv_pg_batch(st["ao"], w[p + "o.w"], res=x)
                hp.gemv_pg_batch(x, w[p + "gu.w"], norm_w=w[p + "ln2"], eps=eps, out=st["act"], act=True)
                hp.gemv_pg_batch(st["act"], w[p + "down.w"], res=x)
                continue
            hp.rmsnorm(x, w[p + "ln1"], w[p + "ln1"], 0, eps, out=h)
            hp.linear(h, w[p + "qkv.w"], w[p + "qkv.b"], hp.EPI_BF16, out=st["qkv"], ws=st["gws"])
            if st["attn_pg"]:
                hp.decode_attn_pg(st["qkv"], w[p + "qn"], w[p + "kn"], eps, 1, st["cos"], st["sin"], st["k"][i], st["v"][i], st["ao"],
                                  st["len"], cap, cap, Hq, Hkv, 128 ** -0.5, st["ws2"])
            elif st["fused_attn"]:
                hp.decode_attn_fused(st["qkv"], w[p + "qn"], w[p + "kn"], eps, 1, st["cos"], st["sin"], st["k"][i], st["v"][i], st["ao"],
                                     st["len"], cap, cap, Hq, Hkv, 128 ** -0.5, st["ws"])
            else:
                hp.qknorm_mrope_cache(st["qkv"], Hq, Hkv, w[p + "qn"], w[p + "qn"], w[p + "kn"], w[p + "kn"], 0, eps, 1, st["cos"],
                                      st["sin"], st["q"], st["k"][i], st["v"][i], st["row"])
                hp.decode_attn_batch(st["q"], st["k"][i], st["v"][i], st["ao"], st["len"], cap, cap, Hq, Hkv, 128 ** -0.5, st["ws"])
            hp.linear(st["ao"], w[p + "o.w"], None, hp.EPI_RES_F32, out=x, res=x, ws=st["gws"])
            hp.rmsnorm(x, w[p + "ln2"], w[p + "ln2"], 0, eps, out=h)
            hp.linear(h, w[p + "gu.w"], None, hp.EPI_SWIGLU, out=st["act"], ws=st["gws"])
            hp.linear(st["act"], w[p + "down.w"], None, hp.EPI_RES_F32, out=x, res=x, ws=st["gws"])
        if pgb:
            hp.gemv_pg_batch(x, w["lm_head"], norm_w=w["norm.und"], eps=eps, out=st["logits"])
        else:
            hp.rmsnorm(x, w["norm.und"], w["norm.und"], 0, eps, out=h)
            hp.linear(h, w["lm_head"], None, hp.EPI_BF16, out=st["logits"], ws=st["gws"])
        if st.get("rng") is not None:
            hp.sample_rows_bf16(st["logits"], st["tok"], st["amax"], st["rng"])
        else:
            hp.argmax_rows_bf16(st["logits"], st["tok"], st["amax"])
        hp.decode_advance_batch(st["pos"], st["row"], st["len"])

    def decode_open_slots(self, n_slots, cap_rows, use_graph=True, sample=None):
        """Device-side state of a batched decode with `n_slots` scene slots of `cap_rows` cache rows each, all idle
        (an idle slot attends to one zero key; its row of every GEMM is independent of the others and its ids are
        ignored).  Scenes enter and leave through decode_set_slot while the captured step keeps replaying: the graph
        only holds pointers into this state."""
        Lc = self.dims["llm"]
        H, Hq, Hkv, Fd, NL = Lc["hidden"], Lc["heads"], Lc["kv_heads"], Lc["ffn"], Lc["layers"]
        d, bf = self.dev, torch.bfloat16
        B = int(n_slots)
        if not (1 <= B <= 64):
            raise ValueError("batched decode: 1..64 scene slots")
        cap = (int(cap_rows) + 63) // 64 * 64
        i32 = lambda vals: torch.tensor(vals, dtype=torch.int32, device=d)
        # the fused norm + RoPE + append form of the attention kernel holds 206 VGPRs (2 workgroups per CU): one launch less per
        # layer while the grid fits the chip at once (15.0 vs 12.9 + 4.6 us at B = 1), slower once it does not (34.6 vs 26.9 +
        # 4.7 us at B = 8, 688 workgroups)
        fused_attn = B * ((cap // 64 + 3) // 4) * Hkv <= 512
        st = dict(B=B, cap=cap, steps=0, graph=None, fused_attn=fused_attn, attn_pg=self.decode_gen == 2,
                  ws2=torch.empty(hip.decode_attn_pg_workspace(Hq, Hkv, B) // 4, dtype=torch.float32, device=d),
                  k=[torch.zeros((B, cap, Hkv, 128), dtype=bf, device=d) for _ in range(NL)],
                  v=[torch.zeros((B, cap, Hkv, 128), dtype=bf, device=d) for _ in range(NL)],
                  pos=i32([[0] * B] * 3), row=i32([j * cap for j in range(B)]), len=i32([1] * B), tok=i32([0] * B),
                  x=torch.empty((B, H), dtype=torch.float32, device=d), h=torch.empty((B, H), dtype=bf, device=d),
                  cos=torch.empty((B, 128), dtype=torch.float32, device=d), sin=torch.empty((B, 128), dtype=torch.float32, device=d),
                  qkv=torch.empty((B, (Hq + 2 * Hkv) * 128), dtype=bf, device=d), q=torch.empty((B, Hq * 128), dtype=bf, device=d),
                  ao=torch.empty((B, Hq * 128), dtype=bf, device=d), act=torch.empty((B, Fd), dtype=bf, device=d),
                  logits=torch.empty((B, Lc["vocab"]), dtype=bf, device=d),
                  ws=torch.empty(B * hip.decode_attn_workspace(cap, Hq) // 4, dtype=torch.float32, device=d),
                  amax=torch.zeros(129 * B, dtype=torch.int32, device=d),
                  gws=torch.zeros(hip.GEMM_WS_WORDS, dtype=torch.int32, device=d))
        if sample is not None:                               # (seed, temperature): draw instead of argmax, every slot its own stream
            st["rng"] = hip.make_rng(sample[0], sample[1], d)
        if use_graph:
            init = {n: st[n].clone() for n in ("pos", "row", "len", "tok") + (("rng",) if sample is not None else ())}
            s = torch.cuda.Stream(device=d)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._decode_batch_body(st)                  # lazy module loads must not happen during capture
            torch.cuda.current_stream().wait_stream(s)
            for n, t in init.items():
                st[n].copy_(t)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._decode_batch_body(st)
            st["graph"] = g
        return st

    def decode_set_slot(self, st, j, cache, start_token, position, max_new_tokens):
        """Put a prefilled scene into slot j: copy its cache rows into the slot's block and point the slot's device-side
        state at its first decode step.  Runs between replays of the captured step (same stream)."""
        n, cap = cache.length, st["cap"]
        if n + max_new_tokens + 1 > cap:
            raise ValueError(f"scene needs {n + max_new_tokens + 1} cache rows, the slots hold {cap}")
        for i in range(len(st["k"])):
            st["k"][i][j, :n].copy_(cache.k[i][:n]); st["v"][i][j, :n].copy_(cache.v[i][:n])
        st["pos"][:, j] = int(position)
        st["row"][j] = j * cap + n
        st["len"][j] = n + 1
        st["tok"][j] = int(start_token)

    def decode_idle_slot(self, st, j):
        """Park slot j (its scene left): the captured step keeps advancing every slot, so an idle one is rewound to its
        first row before it can run past its block."""
        st["pos"][:, j] = 0
        st["row"][j] = j * st["cap"]
        st["len"][j] = 1
        st["tok"][j] = 0

    def decode_begin_batch(self, caches, start_tokens, positions, max_new_tokens, use_graph=True, sample=None):
        """Pack B prefilled caches into one [B, cap, Hkv, 128] block per layer and set up the device-side decode state.
        caches: list of KVCache (one per scene, after their prefills); start_tokens / positions: one int per scene."""
        B = len(caches)
        if not (1 <= B <= 64) or len(start_tokens) != B or len(positions) != B:
            raise ValueError("decode_begin_batch: 1..64 scenes, one start token and one position each")
        st = self.decode_open_slots(B, max(c.length for c in caches) + max_new_tokens + 1, use_graph, sample)
        for j, c in enumerate(caches):
            self.decode_set_slot(st, j, c, start_tokens[j], positions[j], max_new_tokens)
        return st

    def decode_step_batch(self, st):
        """One token per scene.  Returns the device tensor of NEXT token ids (int32 [B], overwritten every step)."""
        if st["graph"] is not None:
            st["graph"].replay()
        else:
            self._decode_batch_body(st)
        st["steps"] += 1
        return st["tok"]
