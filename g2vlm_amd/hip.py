"""ctypes binding of libg2vlm_hip.so (include/g2vlm_hip.h) + thin torch-tensor front ends.

PyTorch is plumbing here: device memory (tensors), streams, and nothing else.  Every compute
call below lands in a hand-written gfx950 kernel.  There is NO fallback: if the library is
missing or a call fails, this module raises.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("G2V_LIB_PATH") or os.path.join(_HERE, "lib", "libg2vlm_hip.so")   # override: experiment builds only

EPI_BF16, EPI_GELU, EPI_QUICKGELU, EPI_SWIGLU, EPI_RES_F32, EPI_RES_BF16 = range(6)
GAMMA_ROUND_BF16 = 1
FORCE_SMALL_TILE = 2
SUPERTILE = 4
FORCE_BIG_TILE = 8
FORCE_8P = 16
P8_H192, P8_H128, P8_H256, P8_TWO_BARRIER, P8_H288, P8_H224, P8_H160 = 128, 256, 512, 1024, 4096, 8192, 16384   # A/B only
P8_PIPELINED = 32768                                        # A/B only: round 1's main loop instead of the staggered two-barrier one
P8_FOUR_WAVES = 65536       # force gemm_4w.hip (four waves, one per SIMD, 128-column wave tiles); without a flag the launcher picks by shape class
P8_EIGHT_WAVES = 131072     # force gemm_8p.hip's staggered eight-wave loop (round 2's default)
F32, BF16 = 0, 1
NO_CAUSAL = 2 ** 30
_DEBUG_GEMM_FLAGS = int(os.environ.get("G2V_GEMM_FLAGS", "0"))     # A/B experiments only (tools/, tests -k ...)


class GemmGroup(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("C", C.c_void_p),
                ("res", C.c_void_p), ("gamma", C.c_void_p), ("M", C.c_int32), ("_pad", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [("g", GemmGroup * 2), ("ngroups", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldc", C.c_int32), ("ldres", C.c_int32), ("epilogue", C.c_int32),
                ("flags", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
_SIGS = {
    "g2v_version": ([], C.c_int),
    "g2v_arch": ([], C.c_char_p),
    "g2v_gemm_bf16": ([C.POINTER(GemmDesc), _P], C.c_int),
    "g2v_gemm_f32": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P], C.c_int),
    "g2v_layernorm": ([_P, _I, _I, _P, _P, _F, _P, _I, _I, _I, _I, _P], C.c_int),
    "g2v_rmsnorm": ([_P, _I, _P, _P, _I, _F, _P, _I, _I, _I, _I, _P], C.c_int),
    "g2v_mrope_table": ([_P, _I, _P, _P, _P, _P], C.c_int),
    "g2v_qknorm_mrope_cache": ([_P, _I, _I, _I, _P, _P, _P, _P, _I, _F, _I, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "g2v_flash_attn": ([_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _F, _P, _P, _I, _P, _I, _I, _P, _P], C.c_int),
    "g2v_flash_attn_workspace": ([_I], C.c_int64),
    "g2v_debug_attn_form": ([_I], C.c_int),
    "g2v_rope2d": ([_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P], C.c_int),
    "g2v_rope_vision": ([_P, _I, _I, _I, _I, _P, _P, _P], C.c_int),
    "g2v_im2col14": ([_P, _I, _I, _I, _P, _I, _P], C.c_int),
    "g2v_dino_preprocess": ([_P, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, _P], C.c_int),
    "g2v_dino_assemble": ([_P, _P, _P, _P, _P, _I, _I, _I, _P], C.c_int),
    "g2v_im2col_patch": ([_P, _I, _I, _I, _I, _P, _I, _P], C.c_int),
    "g2v_qwen_patchify_u8": ([_P, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _I, _P], C.c_int),
    "g2v_vit_assemble": ([_P, _P, _P, _P, _I, _I, _I, _I, _P], C.c_int),
    "g2v_gather_rows_f32": ([_P, _I, _P, _P, _I, _I, _I, _P], C.c_int),
    "g2v_scatter_rows_f32": ([_P, _I, _P, _P, _I, _I, _I, _P], C.c_int),
    "g2v_cast_f32_bf16": ([_P, _P, _L, _P], C.c_int),
    "g2v_cast_bf16_f32": ([_P, _P, _L, _P], C.c_int),
    "g2v_pts_epilogue": ([_P, _I, _I, _I, _I, _P, _P, _P, _P], C.c_int),
    "g2v_pixel_shuffle14": ([_P, _I, _I, _I, _I, _P, _P], C.c_int),
    "g2v_lanczos_resize_u8": ([_P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _I, _P, _P, _I, _P], C.c_int),
    "g2v_pts_epilogue_ps": ([_P, _I, _I, _I, _I, _I, _P, _P, _P, _P], C.c_int),
    "g2v_pixel_shuffle": ([_P, _I, _I, _I, _I, _I, _P, _P], C.c_int),
    "g2v_camera_tail": ([_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "g2v_argmax_bf16": ([_P, _I, _P, _P, _P], C.c_int),
    "g2v_gemv_bf16": ([_P, _P, _P, _P, _P, _I, _I, _P], C.c_int),
    "g2v_decode_attn_workspace": ([_I, _I], C.c_int64),
    "g2v_gemv_rmsnorm_bf16": ([_P, _P, _F, _P, _P, _P, _I, _I, _P], C.c_int),
    "g2v_gemv_swiglu_bf16": ([_P, _P, _P, _I, _I, _P], C.c_int),
    "g2v_gemv_rmsnorm_swiglu_bf16": ([_P, _P, _F, _P, _P, _I, _I, _P], C.c_int),
    "g2v_decode_attn": ([_P, _P, _P, _P, _I, _I, _I, _F, _P, _P], C.c_int),
    "g2v_swiglu_bf16": ([_P, _P, _I, _P], C.c_int),
    "g2v_decode_attn_dyn": ([_P, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P], C.c_int),
    "g2v_decode_advance": ([_P, _P, _P, _P], C.c_int),
    "g2v_decode_attn_batch": ([_P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _F, _P, _P], C.c_int),
    "g2v_decode_advance_batch": ([_P, _P, _P, _I, _P], C.c_int),
    "g2v_decode_attn_fused": ([_P, _P, _P, _F, _I, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _F, _P, _P], C.c_int),
    "g2v_argmax_rows_bf16": ([_P, _I, _I, _L, _P, _P, _P], C.c_int),
    "g2v_gemv_pg": ([_P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _P], C.c_int),
    "g2v_gemv_pg_batch": ([_P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P], C.c_int),
    "g2v_decode_attn_pg_workspace": ([_I, _I, _I], C.c_int64),
    "g2v_decode_attn_pg": ([_P, _P, _P, _F, _I, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _F, _P, _P], C.c_int),
    "g2v_sample_rows_bf16": ([_P, _I, _I, _L, _P, _P, _P, _P], C.c_int),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built — no CPU fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -m g2vlm_amd.build` (hipcc, gfx950). "
                               "g2vlm_amd has no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.argtypes, fn.restype = args, res
        # the kernels are written for one architecture (wave64 MFMA shapes, LDS-DMA, and cross-workgroup hand-offs that lean on
        # its sc1 write-through / L2-bypass behaviour, csrc/gemm_skinny.hip): refuse any other device instead of misbehaving
        if torch.cuda.is_available():
            want = _lib.g2v_arch().decode()
            have = getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "gcnArchName", "")
            if have and not have.split(":")[0] == want:
                _lib = None
                raise RuntimeError(f"libg2vlm_hip.so is built for {want}; this device is {have}")
        if os.environ.get("G2V_ATTN_FORM"):                   # A/B experiments only (tools/bench_ab_env.sh): 0 = the 8 x 32 attention form
            _lib.g2v_debug_attn_form(int(os.environ["G2V_ATTN_FORM"]))
    return _lib


class HipError(RuntimeError):
    pass


def _ck(rc, what):
    if rc != 0:
        raise HipError(f"{what} failed with code {rc}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the raw getter is ~20x cheaper than building a
    torch.cuda.Stream object per launch: 8 us x ~800 launches per scene otherwise)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "libg2vlm_hip takes device pointers only"
    return C.c_void_p(t.data_ptr())


def _rowmajor(t):
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D view expected"
    return t.stride(0)


# ------------------------------------------------------------------------------------------ GEMM
def h2d(t, device, dtype=None, resident=False):
    """Host -> device copy that never blocks the host: the tensor is staged through pinned memory (torch's caching host
    allocator keeps the staging buffers alive until the copy has run) and copied with non_blocking=True.  A pageable
    `.to(device)` is a stream-wide sync point whose wake-up costs milliseconds while the GPU is busy: 174 of them cost a
    C5 step 0.45 s of GPU idle time."""
    if t.is_cuda:
        return t.to(dtype) if dtype is not None else t
    if dtype is not None:
        t = t.to(dtype)
    if torch.device(device).type != "cuda":
        return t.to(device)
    out = t.contiguous().pin_memory().to(device, non_blocking=True)
    if resident:
        # a table that is cached and may later be read by kernels on OTHER streams (attention plans, memoised indices): make
        # sure the upload has landed before anyone can see the cache entry.  Paid once per table, never in steady state.
        torch.cuda.current_stream(out.device).synchronize()
    return out


def scratch_owner(device_index):
    """Owner of a piece of multi-launch kernel scratch: (device, raw stream, host thread).

    Scratch that lives across launches (the attention's partial slots between its forward and combine launches, the skinny
    GEMM's K-split tickets, the argmax ticket) is only safe to share between launches that the stream orders ONE CALL AFTER
    THE OTHER.  Two streams break that, and so do two host threads enqueueing on one stream: ctypes releases the GIL inside a
    call, so thread B's forward can land between thread A's forward and A's combine (round 2's red GPU run: 8 rank threads
    of the sharding simulation shared one plan's slots).  Keyed by all three, every (stream, thread) pair owns its scratch."""
    return (device_index, int(torch._C._cuda_getCurrentRawStream(device_index)), threading.get_ident())


_gemm_ws = {}


GEMM_WS_WORDS = 2 << 20                                     # 8 MiB of int32: covers every skinny shape of the path


def _gemm_workspace(device):
    """Scratch of the skinny GEMM's cross-workgroup K split, one per scratch_owner (device, stream, host thread) so that
    launches that can overlap or interleave never share tickets: zeroed once, tickets self-reset."""
    key = scratch_owner(device.index)
    ws = _gemm_ws.get(key)
    if ws is None:
        ws = _gemm_ws[key] = torch.zeros(GEMM_WS_WORDS, dtype=torch.int32, device=device)
    return ws


def gemm_bf16(groups, N, K, epilogue, out_ld, lda=None, ldres=0, flags=0, ws=None):
    """groups: list (<=2) of dicts {A, W, bias, C, res, gamma, M}; tensors are bf16 except res/gamma/C per epilogue."""
    d = GemmDesc()
    flags |= _DEBUG_GEMM_FLAGS
    d.ngroups, d.N, d.K, d.epilogue, d.flags = len(groups), N, K, epilogue, flags
    d.lda = lda if lda is not None else K
    d.ldc, d.ldres = out_ld, ldres
    if max(int(g["M"]) for g in groups) <= 64:
        # skinny path: scratch for its cross-workgroup K split.  `ws` (int32, zeroed once) when the caller owns one - a
        # captured decode step must not allocate - else the per-(device, stream) default
        if ws is None:
            ws = _gemm_workspace(groups[0]["A"].device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    for i, g in enumerate(groups):
        gg = d.g[i]
        gg.A, gg.W, gg.bias, gg.C = _p(g["A"]), _p(g["W"]), _p(g.get("bias")), _p(g["C"])
        gg.res, gg.gamma, gg.M = _p(g.get("res")), _p(g.get("gamma")), int(g["M"])
    _ck(lib().g2v_gemm_bf16(C.byref(d), _stream()), "g2v_gemm_bf16")


def linear(x, w, bias=None, epilogue=EPI_BF16, out=None, res=None, gamma=None, flags=0, ws=None):
    """Single-problem convenience wrapper.  x bf16 [M,K] (row-major view), w bf16 [N,K]."""
    M, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if out is None:
        dt = torch.float32 if epilogue == EPI_RES_F32 else torch.bfloat16
        out = torch.empty((M, n_out), dtype=dt, device=x.device)
    gemm_bf16([dict(A=x, W=w, bias=bias, C=out, res=res, gamma=gamma, M=M)], N, K, epilogue,
              out_ld=_rowmajor(out), lda=_rowmajor(x), ldres=_rowmajor(res) if res is not None else 0, flags=flags, ws=ws)
    return out


def gemm_f32(x, w, bias=None, relu=False, res=None, out=None):
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _ck(lib().g2v_gemm_f32(_p(x), _p(w), _p(bias), _p(out), _p(res), M, N, K, _rowmajor(x), _rowmajor(out),
                           _rowmajor(res) if res is not None else 0, int(relu), _stream()), "g2v_gemm_f32")
    return out


# ----------------------------------------------------------------------------------------- norms
def _dt(t):
    return BF16 if t.dtype == torch.bfloat16 else F32


def layernorm(x, w, b, eps, out_dtype=torch.bfloat16, out=None):
    M, Cc = x.shape
    if out is None:
        out = torch.empty((M, Cc), dtype=out_dtype, device=x.device)
    _ck(lib().g2v_layernorm(_p(x), _dt(x), _rowmajor(x), _p(w), _p(b), eps, _p(out), _dt(out), _rowmajor(out), M, Cc,
                            _stream()), "g2v_layernorm")
    return out


def rmsnorm(x, w_lo, w_hi, split, eps, out_dtype=torch.bfloat16, out=None):
    M, Cc = x.shape
    if out is None:
        out = torch.empty((M, Cc), dtype=out_dtype, device=x.device)
    _ck(lib().g2v_rmsnorm(_p(x), _rowmajor(x), _p(w_lo), _p(w_hi), int(split), eps, _p(out), _dt(out), _rowmajor(out),
                          M, Cc, _stream()), "g2v_rmsnorm")
    return out


def mrope_table(pos_i32, inv_freq):
    L = pos_i32.shape[1]
    cos = torch.empty((L, 128), dtype=torch.float32, device=pos_i32.device)
    sin = torch.empty_like(cos)
    _ck(lib().g2v_mrope_table(_p(pos_i32), L, _p(inv_freq), _p(cos), _p(sin), _stream()), "g2v_mrope_table")
    return cos, sin


def qknorm_mrope_cache(qkv, Hq, Hkv, qw_lo, qw_hi, kw_lo, kw_hi, split, eps, und_rounding, cos, sin, q_out, k_cache,
                       v_cache, kv_rows):
    L = qkv.shape[0]
    _ck(lib().g2v_qknorm_mrope_cache(_p(qkv), L, Hq, Hkv, _p(qw_lo), _p(qw_hi), _p(kw_lo), _p(kw_hi), int(split), eps,
                                     int(und_rounding), _p(cos), _p(sin), _p(q_out), _p(k_cache), _p(v_cache),
                                     _p(kv_rows), _stream()), "g2v_qknorm_mrope_cache")


# ------------------------------------------------------------------------------------- attention
class AttnSeg(C.Structure):
    _fields_ = [("desc", C.c_int32), ("head", C.c_int32), ("kt0", C.c_int32), ("kt1", C.c_int32), ("slot", C.c_int32),
                ("_pad", C.c_int32 * 3)]


class AttnPlan:
    """Device-side schedule of one attention shape (see attn.hip): tile descriptors, per-phase segment tables, the merge
    table and the partial-result workspace."""

    def __init__(self, tiles, n_tiles, phases, comb, n_comb, n_slots, tile_rows, device):
        self.tiles, self.n_tiles, self.phases = tiles, n_tiles, phases          # phases: list of (segs, seg_ptr, n_blocks)
        self.comb, self.n_comb, self.n_slots, self.tile_rows = comb, n_comb, n_slots, tile_rows
        self.n_blocks = max((p[2] for p in phases), default=0)
        self.n_split = n_comb                                                    # (name kept for the tests' introspection)
        self.ws_words = max(4, int(lib().g2v_flash_attn_workspace(n_slots)) // 4)
        self.device = device
        self._by_owner = {}

    def ws(self, owner):
        """Partial-result scratch of the launches `owner` (hip.scratch_owner: device, stream, host thread) issues: the slots
        are written by the forward launch(es) and read by the combine, so launches of one plan that overlap on two streams
        (scenes issued on two streams) or interleave from two host threads on one stream must not share them."""
        w = self._by_owner.get(owner)
        if w is None:
            w = self._by_owner[owner] = torch.empty(self.ws_words, dtype=torch.float32, device=self.device)
        return w


def _schedule_items(items, n_blocks, align_short_items, n_desc_tiles):
    """items: list of (desc, head, nkt) in launch order.  Returns per-block lists of (item index, kt0, kt1).

    Whole items first: with equal items and few left over (n_items = q n_blocks + r, r <= n_blocks / 8) every workgroup takes
    q whole items and the r left-over items are cut into s pieces each, one piece per workgroup of an evenly spread
    subset, s just large enough that a piece is <= 2 % of a workgroup's work: few partial slots, balance within 2 %.
    Otherwise the (item, KV tile) units are cut into n_blocks equal ranges (stream-K; for short items the cuts are
    snapped to item boundaries when that costs less balance than the partials cost traffic)."""
    import bisect
    n_items = len(items)
    per_block = [[] for _ in range(n_blocks)]
    sizes = [it[2] for it in items]
    U = sum(sizes)
    if n_items == 0 or U == 0:
        return per_block
    equal = min(sizes) == max(sizes)
    q, r = divmod(n_items, n_blocks)
    if equal and q >= 1 and 8 * r <= n_blocks:
        K0 = sizes[0]
        for bq in range(n_blocks):
            for i in range(bq * q, (bq + 1) * q):
                per_block[bq].append((i, 0, K0))
        if r:
            s_ = 1
            while s_ * q < 50 and 2 * s_ * r <= n_blocks and 2 * s_ <= K0:
                s_ *= 2
            for li in range(r):
                i = q * n_blocks + li
                for pc in range(s_):
                    kt0, kt1 = pc * K0 // s_, (pc + 1) * K0 // s_
                    if kt1 > kt0:
                        per_block[((li * s_ + pc) * n_blocks) // (r * s_)].append((i, kt0, kt1))
        return per_block
    prefix = [0]
    for n in sizes:
        prefix.append(prefix[-1] + n)
    bounds = [bq * U // n_blocks for bq in range(n_blocks + 1)]
    if align_short_items and U // max(1, n_items) < 48 and n_items >= 2 * n_blocks:
        for bq in range(1, n_blocks):
            jx = bisect.bisect_left(prefix, bounds[bq])
            lo_, hi_ = prefix[max(0, jx - 1)], prefix[min(jx, len(prefix) - 1)]
            bounds[bq] = lo_ if bounds[bq] - lo_ <= hi_ - bounds[bq] else hi_
        for bq in range(1, n_blocks + 1):
            bounds[bq] = max(bounds[bq], bounds[bq - 1])
    for bq in range(n_blocks):
        u, u_end = bounds[bq], bounds[bq + 1]
        while u < u_end:
            i = bisect.bisect_right(prefix, u) - 1
            kt0 = u - prefix[i]
            kt1 = min(sizes[i], kt0 + (u_end - u))
            per_block[bq].append((i, kt0, kt1))
            u += kt1 - kt0
    return per_block


def make_attn_plan(windows, Hq, device, max_blocks=None, tile_rows=128, align_short_items=True):
    """windows: list of (q_start, q_len, k_start, k_len, causal[, phase]).  One descriptor per tile_rows-row query tile of
    every window.  Windows with the same query rows and disjoint KV ranges (the view-sharded prefill: local K/V block in
    phase 0, the other ranks' blocks in phase 1) are merged by the combine pass that follows the LAST phase; every phase is
    its own launch (flash_attn(..., phase=i)), so a collective can run between them."""
    rows, desc_nkt, desc_phase, out_tiles = [], [], [], {}
    for win in windows:
        qs, ql, ks, kl, causal = win[:5]
        ph = win[5] if len(win) > 5 else 0
        shift = (kl - ql) if causal else NO_CAUSAL
        for t0 in range(0, ql, tile_rows):
            qr = min(tile_rows, ql - t0)
            k_need = min(kl, t0 + qr - 1 + shift + 1)
            if k_need <= 0:
                continue
            out_tiles.setdefault((qs + t0, qr), []).append(len(rows))
            rows.append([qs + t0, qr, ks, kl, shift, qs, 0, 0])
            desc_nkt.append((k_need + 63) // 64)
            desc_phase.append(ph)
    n_tiles = len(rows)
    n_phases = max(desc_phase, default=0) + 1
    if max_blocks is None:
        max_blocks = 512 if tile_rows == 128 else 256          # resident workgroups: 2 (4-wave) or 1 (8-wave) per CU
    # ---- schedule every phase
    pieces = {}                                                # (desc, head) -> list of [phase, block, position, kt0, kt1]
    phase_blocks = []
    for ph in range(n_phases):
        descs = [d for d in range(n_tiles) if desc_phase[d] == ph]
        items = [(d, h, desc_nkt[d]) for h in range(Hq) for d in descs]
        U = sum(it[2] for it in items)
        # more workgroups than items is fine (a 731-row ViT prefill has 36 items but 8 000 KV tiles: 36 CUs would do all the
        # work); keep at least ~8 KV tiles per workgroup so the per-segment prologue / partial write stays amortised
        n_blocks = max(1, min(max_blocks, max(len(items), U // 8))) if items else 0
        per_block = _schedule_items(items, n_blocks, align_short_items, len(descs)) if items else []
        for bq, lst in enumerate(per_block):
            for pos, (i, kt0, kt1) in enumerate(lst):
                pieces.setdefault((items[i][0], items[i][1]), []).append([ph, bq, pos, kt0, kt1])
        phase_blocks.append((n_blocks, per_block, items))
    # ---- slots: an output tile (query rows, head) whose work is ONE segment over ONE descriptor is written directly
    comb, slot_of, n_slots = [], {}, 0
    for (q0, qr), descs in out_tiles.items():
        for h in range(Hq):
            pcs = [(d, pc) for d in descs for pc in pieces.get((d, h), [])]
            if len(pcs) == 1 and pcs[0][1][3] == 0 and pcs[0][1][4] == desc_nkt[pcs[0][0]]:
                slot_of[(pcs[0][0], h, 0)] = -1
                continue
            comb += [descs[0], h, n_slots, len(pcs)]
            for d, pc in pcs:
                slot_of[(d, h, pc[3])] = n_slots
                n_slots += 1
    phases = []
    for ph, (n_blocks, per_block, items) in enumerate(phase_blocks):
        segs = (AttnSeg * max(1, sum(len(x) for x in per_block)))()
        ptr, k = [0], 0
        for lst in per_block:
            for (i, kt0, kt1) in lst:
                d, h = items[i][0], items[i][1]
                sg = segs[k]
                sg.desc, sg.head, sg.kt0, sg.kt1, sg.slot = d, h, kt0, kt1, slot_of[(d, h, kt0)]
                k += 1
            ptr.append(k)
        seg_t = torch.frombuffer(bytearray(bytes(segs)), dtype=torch.int32).clone()
        phases.append((h2d(seg_t, device, resident=True), h2d(torch.tensor(ptr, dtype=torch.int32), device, resident=True), n_blocks))
    tiles = h2d(torch.tensor(rows, dtype=torch.int32).reshape(-1, 8), device, resident=True)
    comb_t = h2d(torch.tensor(comb if comb else [0, 0, 0, 0], dtype=torch.int32), device, resident=True)
    return AttnPlan(tiles, n_tiles, phases, comb_t, len(comb) // 4, n_slots, tile_rows, device)


def flash_attn(q, k, v, out, plan, Hq, Hkv, D, scale=None, phase=None):
    """q [Lq, >=Hq*D] / k,v [Lk, >=Hkv*D] bf16 row-major views (strides in elements); out [Lq, Hq*D] bf16.
    phase=None: every phase of the plan, then the merge; phase=i: that launch only (the merge follows the last one)."""
    scale = scale if scale is not None else D ** -0.5
    ws = plan.ws(scratch_owner(q.device.index))
    last = len(plan.phases) - 1
    for i in (range(len(plan.phases)) if phase is None else (phase,)):
        segs, seg_ptr, n_blocks = plan.phases[i]
        n_comb = plan.n_comb if i == last else 0
        _ck(lib().g2v_flash_attn(_p(q), _rowmajor(q), _p(k), _rowmajor(k), _p(v), _rowmajor(v), _p(out), _rowmajor(out),
                                 _p(plan.tiles), plan.n_tiles, Hq, Hkv, D, scale, _p(segs), _p(seg_ptr), n_blocks, _p(plan.comb), n_comb,
                                 plan.tile_rows, _p(ws), _stream()), "g2v_flash_attn")
    return out


def rope2d(x, col0, n_heads, D, cos, sin, pos, P):
    _ck(lib().g2v_rope2d(_p(x), _rowmajor(x), x.shape[0], col0, n_heads, D, _p(cos), _p(sin), _p(pos), P, _stream()),
        "g2v_rope2d")


def rope_vision(x, n_heads, D, cos, sin):
    _ck(lib().g2v_rope_vision(_p(x), _rowmajor(x), x.shape[0], n_heads, D, _p(cos), _p(sin), _stream()), "g2v_rope_vision")


# ------------------------------------------------------------------------------------- data movers
def im2col14(img, Kpad=640):
    N, _, H, W = img.shape
    out = torch.empty((N * (H // 14) * (W // 14), Kpad), dtype=torch.bfloat16, device=img.device)
    _ck(lib().g2v_im2col14(_p(img), N, H, W, _p(out), Kpad, _stream()), "g2v_im2col14")
    return out


def dino_preprocess(frames, mean, std, want_orig=True):
    """frames: uint8 [N,H,W,3] or fp32 [N,3,H,W] in [0,1], on the device -> (normalised fp32 [N,3,H,W], original fp32
    [N,3,H,W] or None); see g2v_dino_preprocess."""
    u8 = frames.dtype == torch.uint8
    if u8:
        N, H, W, _ = frames.shape
    else:
        assert frames.dtype == torch.float32
        N, _, H, W = frames.shape
    frames = frames.contiguous()
    norm = torch.empty((N, 3, H, W), dtype=torch.float32, device=frames.device)
    orig = torch.empty_like(norm) if want_orig else None      # a copy, as the reference's .clone() (g2vlm.py:952)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    _ck(lib().g2v_dino_preprocess(_p(frames), int(u8), N, H, W, m3, s3, _p(norm), _p(orig), _stream()),
        "g2v_dino_preprocess")
    return norm, orig


def dino_assemble(patch, cls, regs, pos, N, P):
    Cc = patch.shape[1]
    x = torch.empty((N * (P + 5), Cc), dtype=torch.float32, device=patch.device)
    _ck(lib().g2v_dino_assemble(_p(patch), _p(cls), _p(regs), _p(pos), _p(x), N, P, Cc, _stream()), "g2v_dino_assemble")
    return x


def im2col_patch(img, patch, Kpad):
    N, _, H, W = img.shape
    out = torch.empty((N * (H // patch) * (W // patch), Kpad), dtype=torch.bfloat16, device=img.device)
    _ck(lib().g2v_im2col_patch(_p(img), N, H, W, patch, _p(out), Kpad, _stream()), "g2v_im2col_patch")
    return out


def vit_assemble(patch, cls, regs, N, P, R):
    Cc = patch.shape[1]
    x = torch.empty((N * (P + 1 + R), Cc), dtype=torch.float32, device=patch.device)
    _ck(lib().g2v_vit_assemble(_p(patch), _p(cls), _p(regs), _p(x), N, P, R, Cc, _stream()), "g2v_vit_assemble")
    return x


def qwen_patchify_u8(frames_u8, mean, std, Kpad):
    """frames_u8 uint8 [F,H,W,3] on the device -> bf16 [T, Kpad] patch matrix (see g2v_qwen_patchify_u8)."""
    Fn, H, W, _ = frames_u8.shape
    out = torch.empty((((Fn + 1) // 2) * (H // 14) * (W // 14), Kpad), dtype=torch.bfloat16, device=frames_u8.device)
    m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    _ck(lib().g2v_qwen_patchify_u8(_p(frames_u8), Fn, H, W, m3, s3, _p(out), Kpad, _stream()), "g2v_qwen_patchify_u8")
    return out


def gather_rows(src, idx_i32, out):
    _ck(lib().g2v_gather_rows_f32(_p(src), _rowmajor(src), _p(idx_i32), _p(out), _rowmajor(out), idx_i32.numel(),
                                  src.shape[1], _stream()), "g2v_gather_rows_f32")
    return out


def scatter_rows(src, idx_i32, out):
    _ck(lib().g2v_scatter_rows_f32(_p(src), _rowmajor(src), _p(idx_i32), _p(out), _rowmajor(out), idx_i32.numel(),
                                   src.shape[1], _stream()), "g2v_scatter_rows_f32")
    return out


def cast_bf16(x):
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _ck(lib().g2v_cast_f32_bf16(_p(x), _p(out), x.numel(), _stream()), "g2v_cast_f32_bf16")
    return out


def cast_f32(x):
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _ck(lib().g2v_cast_bf16_f32(_p(x), _p(out), x.numel(), _stream()), "g2v_cast_bf16_f32")
    return out


_LANCZOS_DEV = {}


def lanczos_resize_u8(frames, out_h, out_w):
    """PIL's LANCZOS resize of uint8 RGB frames [N, H, W, 3] on the device, bit-exact (g2v_lanczos_resize_u8); the tap
    tables come from host.lanczos_tables and stay resident per (device, in size, out size)."""
    from . import host
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3 and frames.is_contiguous()
    N, H, W, _ = frames.shape
    if (H, W) == (out_h, out_w):
        return frames.clone()                                # Image.resize returns a copy when the size is unchanged
    tabs = []
    for n_in, n_out in ((W, out_w), (H, out_h)):
        key = (frames.device, n_in, n_out)
        if n_in != n_out and key not in _LANCZOS_DEV:
            b, k = host.lanczos_tables(n_in, n_out)
            _LANCZOS_DEV[key] = (h2d(b, frames.device, resident=True), h2d(k, frames.device, resident=True), k.shape[1])
        tabs.append(_LANCZOS_DEV.get(key, (None, None, 0)) if n_in != n_out else (None, None, 0))
    out = torch.empty((N, out_h, out_w, 3), dtype=torch.uint8, device=frames.device)
    tmp = torch.empty((N, H, out_w, 3), dtype=torch.uint8, device=frames.device) if (W != out_w and H != out_h) else None
    (bh, kh, nh), (bv, kv, nv) = tabs
    _ck(lib().g2v_lanczos_resize_u8(_p(frames), N, H, W, _p(out), out_h, out_w, _p(tmp), _p(bh), _p(kh), nh, _p(bv), _p(kv), nv, _stream()),
        "g2v_lanczos_resize_u8")
    return out


def pts_epilogue(feat, N, H, W, mode, pose=None, patch=14):
    assert feat.shape == (N * (H // patch) * (W // patch), 3 * patch * patch) and feat.dtype == torch.float32
    out = torch.empty((N, H, W, 3), dtype=torch.float32, device=feat.device)
    out2 = torch.empty_like(out) if mode == 1 else None
    _ck(lib().g2v_pts_epilogue_ps(_p(feat), N, H, W, patch, mode, _p(pose), _p(out), _p(out2), _stream()), "g2v_pts_epilogue_ps")
    return out, out2


def pixel_shuffle(feat, N, H, W, C_, patch=14):
    assert feat.shape == (N * (H // patch) * (W // patch), C_ * patch * patch) and feat.dtype == torch.float32
    out = torch.empty((N, H, W, C_), dtype=torch.float32, device=feat.device)
    _ck(lib().g2v_pixel_shuffle(_p(feat), N, H, W, C_, patch, _p(out), _stream()), "g2v_pixel_shuffle")
    return out


def pixel_shuffle14(feat, N, H, W, C_):
    return pixel_shuffle(feat, N, H, W, C_, 14)


def camera_tail(feat, N, P, w0, b0, w1, b1, wt, bt, wr, br):
    pose = torch.empty((N, 4, 4), dtype=torch.float32, device=feat.device)
    _ck(lib().g2v_camera_tail(_p(feat), N, P, _p(w0), _p(b0), _p(w1), _p(b1), _p(wt), _p(bt), _p(wr), _p(br), _p(pose),
                              _stream()), "g2v_camera_tail")
    return pose


_argmax_scratch = {}


def argmax_bf16(x, out, scratch=None):
    if scratch is None:
        # one ticket word per scratch_owner: two streams (or two host threads) reducing at once must not share it
        key = scratch_owner(x.device.index)
        scratch = _argmax_scratch.get(key)
        if scratch is None:
            scratch = _argmax_scratch[key] = torch.zeros(129, dtype=torch.int32, device=x.device)
    _ck(lib().g2v_argmax_bf16(_p(x), x.numel(), _p(out), _p(scratch), _stream()), "g2v_argmax_bf16")
    return out


# ------------------------------------------------------------------------------------------ decode
def gemv_bf16(x, w, bias=None, out=None, res=None):
    """y = bf16(w[N,K] . x[K] + bias); res (f32, in place) += y if given, else out (bf16) = y."""
    N, K = w.shape
    _ck(lib().g2v_gemv_bf16(_p(x), _p(w), _p(bias), _p(out), _p(res), N, K, _stream()), "g2v_gemv_bf16")
    return res if res is not None else out


def gemv_rmsnorm_bf16(x_f32, norm_w, eps, w, bias, out):
    """out = bf16(W . bf16(rmsnorm(x_f32)) + bias): input norm fused into the weight-streaming GEMV."""
    N, K = w.shape
    _ck(lib().g2v_gemv_rmsnorm_bf16(_p(x_f32), _p(norm_w), eps, _p(w), _p(bias), _p(out), N, K, _stream()), "g2v_gemv_rmsnorm_bf16")
    return out


def gemv_rmsnorm_swiglu_bf16(x_f32, norm_w, eps, w_gu, act_out):
    """act_out bf16[F] = SwiGLU(W_gu . bf16(rmsnorm(x_f32))): norm, gate/up GEMV and activation in one launch."""
    N2, K = w_gu.shape
    _ck(lib().g2v_gemv_rmsnorm_swiglu_bf16(_p(x_f32), _p(norm_w), eps, _p(w_gu), _p(act_out), N2, K, _stream()),
        "g2v_gemv_rmsnorm_swiglu_bf16")
    return act_out


def gemv_swiglu_bf16(gu, w, res):
    """res (f32, in place) += bf16(W . swiglu(gu)): activation fused into the down-projection GEMV."""
    N, K = w.shape
    _ck(lib().g2v_gemv_swiglu_bf16(_p(gu), _p(w), _p(res), N, K, _stream()), "g2v_gemv_swiglu_bf16")
    return res


def gemv_pg(x, w, norm_w=None, eps=0.0, bias=None, out=None, res=None, act=False):
    """Persistent-grid GEMV of the decode step (g2v_gemv_pg): y = w[N,K] . x with the fused RMSNorm / SwiGLU / residual forms."""
    N, K = w.shape
    _ck(lib().g2v_gemv_pg(_p(x), _p(norm_w), float(eps), _p(w), _p(bias), _p(out), _p(res), N, K, int(act), _stream()), "g2v_gemv_pg")
    return res if res is not None else out


def gemv_pg_batch(x, w, norm_w=None, eps=0.0, bias=None, out=None, res=None, act=False):
    """g2v_gemv_pg_batch: Y[B,N] = x[B,K] . w[N,K]^T for B <= 8 decode rows, weights streamed once; fused forms as gemv_pg."""
    N, K = w.shape
    B = x.shape[0]
    assert x.dim() == 2 and x.shape[1] == K and x.is_contiguous() and (x.dtype == torch.float32) == (norm_w is not None)
    tgt = res if res is not None else out
    assert tgt.is_contiguous() and tgt.shape == (B, N // 2 if act else N)
    _ck(lib().g2v_gemv_pg_batch(_p(x), _p(norm_w), float(eps), _p(w), _p(bias), _p(out), _p(res), B, N, K, int(act), _stream()),
        "g2v_gemv_pg_batch")
    return tgt


def decode_attn_pg_workspace(Hq, Hkv, batch=1):
    return int(lib().g2v_decode_attn_pg_workspace(Hq, Hkv, batch))


def decode_attn_pg(qkv, qw, kw, eps, und_rounding, cos, sin, k_cache, v_cache, out, len_dev, scene_rows, max_len, Hq, Hkv, scale,
                   workspace):
    """g2v_decode_attn_pg: as decode_attn_fused, on a grid of 256 equal shares of the max_len cache rows per scene."""
    _ck(lib().g2v_decode_attn_pg(_p(qkv), _p(qw), _p(kw), eps, int(und_rounding), _p(cos), _p(sin), _p(k_cache), _p(v_cache), _p(out),
                                 _p(len_dev), qkv.shape[0], int(scene_rows), int(max_len), Hq, Hkv, scale, _p(workspace), _stream()),
        "g2v_decode_attn_pg")
    return out


def decode_attn_workspace(Lk, Hq):
    return int(lib().g2v_decode_attn_workspace(Lk, Hq))


def decode_attn(q, k_cache, v_cache, out, Lk, Hq, Hkv, scale, workspace):
    _ck(lib().g2v_decode_attn(_p(q), _p(k_cache), _p(v_cache), _p(out), Lk, Hq, Hkv, scale, _p(workspace), _stream()),
        "g2v_decode_attn")
    return out


def decode_attn_dyn(q, k_cache, v_cache, out, len_dev, max_len, Hq, Hkv, scale, workspace):
    _ck(lib().g2v_decode_attn_dyn(_p(q), _p(k_cache), _p(v_cache), _p(out), _p(len_dev), max_len, Hq, Hkv, scale,
                                  _p(workspace), _stream()), "g2v_decode_attn_dyn")
    return out


def decode_advance(pos3, row, length):
    _ck(lib().g2v_decode_advance(_p(pos3), _p(row), _p(length), _stream()), "g2v_decode_advance")


def decode_attn_batch(q, k_cache, v_cache, out, len_dev, scene_rows, max_len, Hq, Hkv, scale, workspace):
    """q / out [B, Hq*128]; caches [B, scene_rows, Hkv, 128]; len_dev int32 [B]."""
    _ck(lib().g2v_decode_attn_batch(_p(q), _p(k_cache), _p(v_cache), _p(out), _p(len_dev), q.shape[0], int(scene_rows), max_len, Hq,
                                    Hkv, scale, _p(workspace), _stream()), "g2v_decode_attn_batch")
    return out


def decode_attn_fused(qkv, qw, kw, eps, und_rounding, cos, sin, k_cache, v_cache, out, len_dev, scene_rows, max_len, Hq, Hkv, scale,
                      workspace):
    """Attention of one decode step with q/k-norm, mRoPE and the cache append folded in.  qkv [B, (Hq+2Hkv)*128] raw."""
    _ck(lib().g2v_decode_attn_fused(_p(qkv), _p(qw), _p(kw), eps, int(und_rounding), _p(cos), _p(sin), _p(k_cache), _p(v_cache),
                                    _p(out), _p(len_dev), qkv.shape[0], int(scene_rows), max_len, Hq, Hkv, scale, _p(workspace),
                                    _stream()), "g2v_decode_attn_fused")
    return out


def decode_advance_batch(pos3, row, length):
    _ck(lib().g2v_decode_advance_batch(_p(pos3), _p(row), _p(length), row.numel(), _stream()), "g2v_decode_advance_batch")


def argmax_rows_bf16(x, out, scratch):
    """x bf16 [rows, n] (row stride >= n); out int32 [rows]; scratch int32 [rows*129] zeroed once."""
    _ck(lib().g2v_argmax_rows_bf16(_p(x), x.shape[0], x.shape[1], _rowmajor(x), _p(out), _p(scratch), _stream()), "g2v_argmax_rows_bf16")
    return out


def make_rng(seed, temperature, device):
    """Device-side sampler state of g2v_sample_rows_bf16: int32[4] = {seed lo, seed hi, step = 0, float bits of 1 / T}."""
    import struct
    if not temperature > 0:
        raise ValueError("temperature must be > 0")
    seed = int(seed) & (2 ** 64 - 1)
    to_i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v     # noqa: E731
    words = [to_i32(seed & 0xffffffff), to_i32(seed >> 32), 0, struct.unpack("<i", struct.pack("<f", 1.0 / float(temperature)))[0]]
    return h2d(torch.tensor(words, dtype=torch.int32), device)


def sample_rows_bf16(x, out, scratch, rng):
    """x bf16 [rows, n]: out[r] ~ softmax(x[r] / T) (reference g2vlm.py:1119-1122); advances rng's step word."""
    if x.dim() == 1:
        x = x.view(1, -1)
    _ck(lib().g2v_sample_rows_bf16(_p(x), x.shape[0], x.shape[1], _rowmajor(x), _p(out), _p(scratch), _p(rng), _stream()),
        "g2v_sample_rows_bf16")
    return out


def mrope_table_into(pos_i32, inv_freq, cos, sin):
    _ck(lib().g2v_mrope_table(_p(pos_i32), pos_i32.shape[1], _p(inv_freq), _p(cos), _p(sin), _stream()), "g2v_mrope_table")


def swiglu_bf16(gu, out):
    """out[i] = bf16(bf16(silu(g[i])) * u[i]); gu holds gate/up interleaved per 16 (decode MLP)."""
    _ck(lib().g2v_swiglu_bf16(_p(gu), _p(out), out.numel(), _stream()), "g2v_swiglu_bf16")
    return out
