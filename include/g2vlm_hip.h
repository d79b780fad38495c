/* libg2vlm_hip.so — C ABI of the MI355X (gfx950) kernels behind the G2VLM inference hot path.
 *
 * The reference (ushariRanasinghe/G2VLM) has NO in-repo native/FFI interface: its native work is
 * done by third-party wheels reached from Python.  Each entry point below names
 * the reference call site(s) it replaces; INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add.  Conventions (last row):
 *   - plain pointers and ints only; every pointer is DEVICE memory owned by the caller unless
 *     marked "host"; kernels allocate nothing and keep no global state;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *     and re-entrant across streams;
 *   - return 0 on success, negative errno-style code otherwise (-22 bad argument, -5 launch
 *     failure); nothing throws.
 *   - bf16 = raw uint16 storage, row-major; "ld*" are row strides in ELEMENTS.
 */
#ifndef G2VLM_HIP_H
#define G2VLM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int g2v_version(void);              /* ABI version, currently 1 */
const char* g2v_arch(void);         /* "gfx950" */

/* ---- GEMM: every nn.Linear / patch conv under autocast(bf16) (SURVEY K-k, K-l, K-m) ---------- */
enum {
  G2V_EPI_BF16 = 0,      /* C(bf16) = bf16(acc + bias)                                              */
  G2V_EPI_GELU = 1,      /* C(bf16) = bf16(gelu_erf(bf16(acc + bias)))     dinov2_model.py:174-177   */
  G2V_EPI_QUICKGELU = 2, /* x*sigmoid(1.702x) with the reference's 3 bf16 roundings (ViT MLP)       */
  G2V_EPI_SWIGLU = 3,    /* W = gate/up interleaved per 16 columns; C(bf16)[M,N/2] = silu(g)*u
                            modeling_qwen2_vl.py:519-521                                            */
  G2V_EPI_RES_F32 = 4,   /* C(f32) = res(f32 or NULL) + [bf16]( bf16(acc+bias) * gamma(f32 or NULL) )
                            dinov2_model.py:233-245, qwen2vl.py:883-909                             */
  G2V_EPI_RES_BF16 = 5   /* C(bf16) = bf16(res(bf16) + bf16(acc+bias))   modeling_qwen2_vl.py:476-482 */
};
#define G2V_GEMM_GAMMA_ROUND_BF16 1 /* flags: round (linear*gamma) to bf16 (MoT ls1/ls2 `.to(bf16)`) */
#define G2V_GEMM_SUPERTILE 4        /* flags: big-tile kernel walks 4x8 supertiles (L2 reuse experiment)         */
#define G2V_GEMM_FORCE_BIG_TILE 8   /* flags: always use the 256-wide DMA-staged kernel when K%64==0 && N%256==0  */
#define G2V_GEMM_FORCE_8P 16        /* flags: always use the 256x256 8-phase kernel (gemm_8p.hip) when K%64==0 && N%256==0  */
/* A/B testing of the 8-phase kernel's variants (tools/bench_kernels.py, tests): a forced tile height, or the two-barrier main loop */
#define G2V_GEMM_8P_H192 128
#define G2V_GEMM_8P_H128 256
#define G2V_GEMM_8P_H256 512
#define G2V_GEMM_8P_TWO_BARRIER 1024
#define G2V_GEMM_8P_H288 4096
#define G2V_GEMM_8P_H224 8192
#define G2V_GEMM_8P_H160 16384
#define G2V_GEMM_8P_PIPELINED 32768 /* A/B: round 1's main loop (one barrier per phase, reads of phase p+1 before the MFMAs of p) */
/* 256x256x64 tile, two bit-identical forms; without either flag the launcher picks by shape class (gemm_8p.hip, G2V_GEMM_4W_MASK) */
#define G2V_GEMM_8P_FOUR_WAVES 65536   /* force gemm_4w.hip: four waves, one per SIMD, 128-column wave tiles, accumulators in AGPRs */
#define G2V_GEMM_8P_NO_ROW_SKIP 262144 /* A/B: wave rows without a valid row run their MFMAs anyway (gemm_8p.hip) */
#define G2V_GEMM_8P_EIGHT_WAVES 131072 /* force gemm_8p.hip: eight waves, the two wave rows staggered by one barrier (round 2) */
#define G2V_GEMM_FORCE_SMALL_TILE 2 /* flags: always use the 128x128 register-staged kernel (A/B testing)  */

typedef struct {
  const void* A;     /* bf16 [M, lda]                         */
  const void* W;     /* bf16 [N, K]  (nn.Linear.weight)       */
  const void* bias;  /* bf16 [N] or NULL                      */
  void* C;           /* output, row stride ldc                */
  const void* res;   /* residual (epilogue 4: f32, 5: bf16), row stride ldres, may alias C */
  const void* gamma; /* f32 [N] layer-scale or NULL           */
  int32_t M;
  int32_t _pad;
} g2v_gemm_group;

typedef struct {
  g2v_gemm_group g[2]; /* two row-partitioned problems sharing N,K,epilogue = und / geo experts */
  int32_t ngroups, N, K, lda, ldc, ldres, epilogue, flags;
  /* optional device scratch for the skinny (M <= 64) path's cross-workgroup K split: `workspace_bytes` bytes, ZEROED ONCE
   * by the caller (arrival tickets live at its head and reset themselves), not shared by launches that may overlap on
   * different streams.  NULL: the split stays inside a workgroup (slower for small N).  A few MiB cover every shape.   */
  void* workspace;
  int64_t workspace_bytes;
} g2v_gemm_desc;       /* host struct */

int g2v_gemm_bf16(const g2v_gemm_desc* desc, void* stream);

/* fp32 islands (autocast disabled): Pi3LinearPts3d.proj, Pi3CameraHead linears
 * (transformer_head.py:75, camera_head.py:26-29).  C = [relu](A x W^T + bias) [+ res]            */
int g2v_gemm_f32(const void* A, const void* W, const void* bias, void* C, const void* res,
                 int M, int N, int K, int lda, int ldc, int ldres, int relu, void* stream);

/* ---- norms ------------------------------------------------------------------------------------ */
#define G2V_F32 0
#define G2V_BF16 1
/* nn.LayerNorm in fp32 (autocast fp32 op), dinov2_model.py:225,240,351; block.py:316-320.
 * x [M,C] (f32 or bf16) -> out [M,C] (f32 or bf16)                                               */
int g2v_layernorm(const void* x, int x_dtype, int ldx, const void* w, const void* b, float eps,
                  void* out, int out_dtype, int ldo, int M, int C, void* stream);
/* Qwen2RMSNorm (modeling_qwen2_vl.py:496-501) with expert routing by row range: rows < split use
 * w_lo, rows >= split use w_hi (qwen2vl.py:861-865, 897-898, 1325-1329).  x f32 [M,C].            */
int g2v_rmsnorm(const void* x, int ldx, const void* w_lo, const void* w_hi, int split, float eps,
                void* out, int out_dtype, int ldo, int M, int C, void* stream);

/* ---- LLM rotary + qk-norm + KV-cache write ---------------------------------------------------- */
/* Qwen2VLRotaryEmbedding.forward + mrope section select (modeling_qwen2_vl.py:142-166, 223-225).
 * pos int32 [3,L] -> cos,sin f32 [L,128]                                                         */
int g2v_mrope_table(const void* pos, int L, const void* inv_freq /* f32[64] */, void* cos, void* sin, void* stream);
/* qwen2vl.py:572-576 / 596-619 / 626-634 fused: per-head RMSNorm(128) of q,k with routed weights,
 * mRoPE in fp32, cast to bf16; q -> q_out [L,Hq,128]; k,v -> cache rows kv_rows[i] of
 * k_cache/v_cache [*,Hkv,128].  qkv bf16 [L, (Hq+2Hkv)*128].  und_rounding=1 replicates the
 * bf16 round of the normalised value in und mode (bf16 input to Qwen2RMSNorm).                   */
int g2v_qknorm_mrope_cache(const void* qkv, int L, int Hq, int Hkv,
                           const void* qw_lo, const void* qw_hi, const void* kw_lo, const void* kw_hi,
                           int split, float eps, int und_rounding, const void* cos, const void* sin,
                           void* q_out, void* k_cache, void* v_cache, const void* kv_rows, void* stream);

/* ---- attention: flash_attn_varlen_func / SDPA-flash call sites (SURVEY K-a,b,c,d) ------------- */
typedef struct {
  int32_t q0, q_rows;     /* query rows [q0, q0+q_rows) (<= tile_rows) of one window               */
  int32_t k0, k_len;      /* keys [k0, k0+k_len) of that window                                    */
  int32_t causal_shift;   /* key j allowed iff j - k0 <= (q - q_win0) + shift; INT32_MAX/2 = none  */
  int32_t q_win0;         /* first query row of the window                                         */
  int32_t _pad[2];
} g2v_attn_tile;          /* device array, one entry per (window, 128-row query tile)              */

/* Persistent schedule, built by the host (g2vlm_amd/hip.py::make_attn_plan).  An ITEM is (tile descriptor, head); its KV
 * window is walked in 64-key tiles.  Workgroup b runs segments segs[seg_ptr[b] .. seg_ptr[b+1]):                          */
typedef struct {
  int32_t desc;           /* index into `tiles`                                                            */
  int32_t head;           /* query head                                                                    */
  int32_t kt0, kt1;       /* 64-key KV tiles [kt0, kt1) of the descriptor's window                           */
  int32_t slot;           /* < 0: the segment covers the item's whole KV work and writes the normalised bf16 output;
                             >= 0: it leaves unnormalised fp32 partials (m, l, O) in workspace slot `slot`            */
  int32_t _pad[3];
} g2v_attn_seg;
/* `comb` (device int32 [n_comb][4]) = (descriptor, head, first slot, number of slots): the output rows of that descriptor
 * and head are merged from those consecutive slots by a second launch (n_comb = 0: none).  Slots of one output tile may
 * come from different descriptors with the same query rows (disjoint KV windows: local block / remote blocks of the
 * view-sharded prefill) and from EARLIER calls on the same workspace: a call with n_blocks = 0 only merges.
 * tile_rows = query rows per descriptor at most: 128 (4-wave workgroups) or 256 (8-wave workgroups).
 * `workspace`: g2v_flash_attn_workspace(n_slots) bytes of fp32 scratch.                                                  */
int64_t g2v_flash_attn_workspace(int n_slots);
/* A/B switch for tools and tests (process-wide, not for production use): which kernel serves head dim 128 with 256-row items.
 * 1 (default) = 4 waves x 64 query rows, one wave per SIMD (flash_fwd64_kernel); 0 = the 8 waves x 32 rows form.  Both implement
 * the same contract; they differ in fp32 summation order and in the softmax reference (exact row maximum vs a power of two).     */
int g2v_debug_attn_form(int form);
int g2v_flash_attn(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                   void* o, int ldo, const g2v_attn_tile* tiles, int n_tiles,
                   int Hq, int Hkv, int D, float scale,
                   const g2v_attn_seg* segs, const int32_t* seg_ptr, int n_blocks, const int32_t* comb, int n_comb,
                   int tile_rows, void* workspace, void* stream);

/* ---- decoders' RoPE2D (pos_embed.py:112-159), in place on the q and k thirds of a fused qkv ---- */
/* x bf16 rows [M, ld]; for each of n_heads heads at column col0 + h*D: 2-D rope with the bf16 tables
 * cos/sin [max_pos, D/2] (built on the host exactly as the reference does, hazard H3);
 * pos int32 [P,2] (y,x), row r uses pos[r % P].                                                  */
int g2v_rope2d(void* x, int ld, int M, int col0, int n_heads, int D, const void* cos, const void* sin,
               const void* pos, int P, void* stream);

/* Qwen2-VL ViT rope (apply_rotary_pos_emb_vision, modeling_qwen2_vl.py:235-246): fp32 rotate-half
 * rope on n_heads heads of width D starting at x (bf16 rows [M, ld]); cos/sin f32 [M, D]            */
int g2v_rope_vision(void* x, int ld, int M, int n_heads, int D, const void* cos, const void* sin, void* stream);

/* ---- DINO front end (modeling_dinov2_with_registers.py:62-71, 147-171) ------------------------ */
/* ToTensor + Normalize + the original_images copy of prepare_dino_images_pi3 (g2vlm.py:947-953) on the device:
 * in = the loader's uint8 frames [N,H,W,3] (in_is_u8) or an f32 [N,3,H,W] image in [0,1]; norm f32 [N,3,H,W] =
 * (x - mean) / std, orig f32 [N,3,H,W] = x (may be NULL); mean3 / std3 host floats.  Bit-identical to the host ops.  */
int g2v_dino_preprocess(const void* in, int in_is_u8, int N, int H, int W, const float* mean3, const float* std3, void* norm,
                        void* orig, void* stream);

/* PIL's Image.resize(size, LANCZOS) on uint8 RGB frames (what the reference's loader calls per image,
 * data/transforms_vggt.py:437), bit-exact: src u8 [N, Hin, Win, 3] -> dst u8 [N, Hout, Wout, 3].  bounds_* int32 [out, 2]
 * (first tap, taps) and kk_* int32 [out, ksize] are Pillow's 22-bit fixed-point tap tables for the x and the y axis, built on
 * the host (g2vlm_amd/host.py::lanczos_tables); an axis whose size does not change takes no pass (tables may be NULL);
 * tmp: u8 [N, Hin, Wout, 3] scratch when both axes change.                                                              */
int g2v_lanczos_resize_u8(const void* src, int N, int Hin, int Win, void* dst, int Hout, int Wout, void* tmp,
                          const void* bounds_h, const void* kk_h, int ksize_h,
                          const void* bounds_v, const void* kk_v, int ksize_v, void* stream);
/* im2col of 14x14/14 patches: img f32 [N,3,H,W] -> bf16 [N*P, Kpad] (K=588 zero-padded)           */
int g2v_im2col14(const void* img, int N, int H, int W, void* out, int Kpad, void* stream);
/* x f32 [N, 5+P, C] = {cls+pos[0], reg0..3, patch[p] + pos[1+p]}; patch bf16 [N*P, C]             */
int g2v_dino_assemble(const void* patch, const void* cls, const void* regs, const void* pos,
                      void* x, int N, int P, int C, void* stream);

/* DINOv3 front end (modeling/dinov3/dinov3_model.py:47-69): Conv2d patch/patch under autocast = im2col (K = 3*patch^2,
 * zero-padded to Kpad) + g2v_gemm_bf16; then [cls | R register tokens | patch rows] per view, fp32, no position table
 * (positions enter as RoPE on q/k: g2v_rope_vision with an identity row for the prefix tokens).                         */
int g2v_im2col_patch(const void* img, int N, int H, int W, int patch, void* out, int Kpad, void* stream);
int g2v_vit_assemble(const void* patch, const void* cls, const void* regs, void* x, int N, int P, int R, int C,
                     void* stream);

/* Qwen2VLImageProcessor._preprocess after the PIL resize (image_processing_qwen2_vl.py:218-273) on the device: uint8
 * frames [F,H,W,3] (H, W multiples of 28) -> rescale, normalise (host floats mean3 / std3), temporal pairing, patch
 * reorder -> bf16 [ceil(F/2) * H/14 * W/14, Kpad] (columns >= 1176 zero): the A operand of the patch-embed GEMM.
 * Bit-identical to the bf16 cast of the host transform's fp32 matrix.                                            */
int g2v_qwen_patchify_u8(const void* img_u8, int F, int H, int W, const float* mean3, const float* std3, void* out,
                         int Kpad, void* stream);

/* ---- small data movers ------------------------------------------------------------------------ */
int g2v_gather_rows_f32(const void* src, int ld_src, const void* idx, void* dst, int ld_dst,
                        int rows, int C, void* stream);      /* dst[i] = src[idx[i]]  (nn.Embedding) */
int g2v_scatter_rows_f32(const void* src, int ld_src, const void* idx, void* dst, int ld_dst,
                         int rows, int C, void* stream);     /* dst[idx[i]] = src[i]                 */
int g2v_cast_f32_bf16(const void* src, void* dst, int64_t n, void* stream);
int g2v_cast_bf16_f32(const void* src, void* dst, int64_t n, void* stream);

/* ---- pointmap / camera heads (g2vlm.py:1200-1226, transformer_head.py:69-81, camera_head.py) -- */
/* feat f32 [N*P, 3*patch^2] -> out f32 [N,H,W,3] via pixel_shuffle(patch); patch = the geometry encoder's patch size,
 * 14 (DINOv2, g2vlm.py:172) or 16 (use_dinov3, g2vlm.py:170); H, W multiples of it.  mode 0: raw (global points);
 * mode 1: local points (xy*exp(z), exp(z)) into `out` AND world points = pose[:3,:4] . (local,1)
 * into `out2` (pose f32 [N,4,4]).                                                                */
int g2v_pts_epilogue_ps(const void* feat, int N, int H, int W, int patch, int mode, const void* pose,
                        void* out, void* out2, void* stream);
/* the patch-14 form (kept for callers bound before the DINOv3 variant existed) */
int g2v_pts_epilogue(const void* feat, int N, int H, int W, int mode, const void* pose,
                     void* out, void* out2, void* stream);
/* F.pixel_shuffle(., patch) of a Pi3LinearPts3d output with C channels (transformer_head.py:69-81; the confidence head of
 * train_conf_pi3 checkpoints has C = 1, g2vlm.py:1208-1219): feat f32 [N*P, C*patch^2] -> out f32 [N, H, W, C]        */
int g2v_pixel_shuffle(const void* feat, int N, int H, int W, int C, int patch, void* out, void* stream);
int g2v_pixel_shuffle14(const void* feat, int N, int H, int W, int C, void* out, void* stream);   /* patch = 14 */

/* mean over P tokens of f32 [N,P,512] -> 2x(Linear+ReLU) -> fc_t, fc_rot -> SVD-orthogonalise ->
 * pose f32 [N,4,4].  Weights f32 in nn.Linear layout.                                            */
int g2v_camera_tail(const void* feat, int N, int P, const void* w0, const void* b0, const void* w1,
                    const void* b1, const void* wt, const void* bt, const void* wr, const void* br,
                    void* pose, void* stream);

/* torch.argmax over bf16 logits (g2vlm.py:1125), first maximal index -> int32 out[0].
 * scratch: int32[129] device words, zeroed ONCE by the caller (the kernel resets its arrival ticket).   */
int g2v_argmax_bf16(const void* x, int n, void* out, void* scratch, void* stream);


/* ---- batch-1 decode (g2vlm.py:1086-1135) ------------------------------------------------------- */
/* nn.Linear at M=1: y = bf16(W[N,K] . x[K] + bias); res != NULL: res[n] (f32) += y, else out[n] = y   */
int g2v_gemv_bf16(const void* x, const void* W, const void* bias, void* out, void* res, int N, int K, void* stream);
/* the same Linear with its producer fused in: (a) x = bf16(Qwen2RMSNorm(x_f32[K]; norm_w, eps)) -> out bf16[N];
 * (b) x = SwiGLU of the interleaved gate/up vector gu bf16[2K] -> res f32[N] += y (the MLP's down_proj + residual)  */
int g2v_gemv_rmsnorm_bf16(const void* x_f32, const void* norm_w, float eps, const void* W, const void* bias, void* out,
                          int N, int K, void* stream);
int g2v_gemv_swiglu_bf16(const void* gu, const void* W, void* res, int N, int K, void* stream);
/* (c) Qwen2MLP's first half in one launch (modeling_qwen2_vl.py:519-521 at q_len 1): x = bf16(Qwen2RMSNorm(x_f32[K])),
 * W_gu = gate/up interleaved per 16 rows [N2 = 2F, K] -> act_out bf16[F] = bf16(bf16(silu(g)) * u); the down projection
 * is then g2v_gemv_bf16(act_out, W_down, NULL, NULL, res)                                               */
int g2v_gemv_rmsnorm_swiglu_bf16(const void* x_f32, const void* norm_w, float eps, const void* W_gu, void* act_out,
                                 int N2, int K, void* stream);
/* Qwen2MLP activation on the fused gate/up GEMV output (interleaved per 16, as G2V_EPI_SWIGLU's W):
 * gu bf16[2n] -> out bf16[n] = bf16(bf16(silu(g)) * u)                                                */
int g2v_swiglu_bf16(const void* gu, void* out, int n, void* stream);
/* flash_attn_varlen_func with q_len 1 (qwen2vl.py:643-652): q bf16 [Hq,128]; caches bf16 [*,Hkv,128];
 * out bf16 [Hq,128]; workspace f32 of g2v_decode_attn_workspace(Lk,Hq) bytes                          */
int64_t g2v_decode_attn_workspace(int Lk, int Hq);
int g2v_decode_attn(const void* q, const void* k_cache, const void* v_cache, void* out, int Lk, int Hq,
                    int Hkv, float scale, void* workspace, void* stream);
/* hipGraph-replayable forms of the decode step: the KV length lives in device memory (int32 Lk_dev[1]) and the grid
 * is sized for max_len keys; g2v_decode_advance bumps {rope pos[3], cache row, kv length} by one on the device.   */
int g2v_decode_attn_dyn(const void* q, const void* k_cache, const void* v_cache, void* out, const void* Lk_dev,
                        int max_len, int Hq, int Hkv, float scale, void* workspace, void* stream);
int g2v_decode_advance(void* pos3, void* row, void* len, void* stream);

/* ---- batched decode (lifts the reference's batch = 1 limit, g2vlm.py:1006, 1137) -------------------
 * `batch` scenes share the weights; each has its own KV cache and length.  The Linears run as M = batch GEMMs through
 * g2v_gemm_bf16 (skinny kernel), norms / RoPE / cache write through the row-batched prefill entry points with cache rows
 * scene * scene_rows + len; only attention, argmax and the state bump need batch-aware forms:
 *   q / out bf16 [batch, Hq*128]; caches bf16 [batch, scene_rows, Hkv, 128]; Lk_dev int32[batch] (device);
 *   workspace >= batch * g2v_decode_attn_workspace(max_len, Hq) bytes; the grid covers max_len keys per scene.          */
int g2v_decode_attn_batch(const void* q, const void* k_cache, const void* v_cache, void* out, const void* Lk_dev,
                          int batch, int64_t scene_rows, int max_len, int Hq, int Hkv, float scale, void* workspace,
                          void* stream);
/* The decode step's attention with the q/k-norm, mRoPE and KV-cache append folded in (replaces g2v_qknorm_mrope_cache +
 * g2v_decode_attn_* for the one new token per scene; reference qwen2vl.py:596-652 at q_len 1):
 *   qkv bf16 [batch, (Hq + 2 Hkv) * 128] = the step's RAW fused projections; q_norm_w / k_norm_w f32 [128];
 *   cos / sin f32 [batch, 128] (g2v_mrope_table); Lk_dev[b] = cache length INCLUDING the new token - its K (normalised,
 *   rotated) and V are written to row Lk_dev[b] - 1 of scene b's block by this call; the rest as g2v_decode_attn_batch. */
int g2v_decode_attn_fused(const void* qkv, const void* q_norm_w, const void* k_norm_w, float eps, int und_rounding,
                          const void* cos, const void* sin, void* k_cache, void* v_cache, void* out, const void* Lk_dev,
                          int batch, int64_t scene_rows, int max_len, int Hq, int Hkv, float scale, void* workspace,
                          void* stream);
/* pos3 int32 [3, batch], row / len int32 [batch]: all += 1                                                             */
int g2v_decode_advance_batch(void* pos3, void* row, void* len, int batch, void* stream);
/* torch.argmax(logits, dim=-1) for bf16 [rows, ld >= n], first maximal index per row -> int32 out[rows];
 * scratch int32[rows * 129], zeroed once by the caller                                                                */
int g2v_argmax_rows_bf16(const void* x, int rows, int n, int64_t ld, void* out, void* scratch, void* stream);
/* torch.multinomial(softmax(logits / temperature, -1), 1) of generate_text's do_sample branch (g2vlm.py:1119-1122), per
 * row of bf16 [rows, ld >= n] -> int32 out[rows].  Drawn as argmax_i(logit_i / T - log(-log u_i)) (Gumbel-max: the same
 * distribution in one pass), u_i = Philox4x32-10(counter = {i, row, step, 0}, key = seed) top 24 bits.
 * rng: int32[4] on the device = {seed lo, seed hi, step, float bits of 1 / T}; this call advances `step` by one, so a
 * hipGraph-captured decode step draws fresh numbers at each replay.  torch's own CUDA Philox offsets are not
 * reproduced: parity with the reference is distributional.  scratch as g2v_argmax_rows_bf16.                          */
int g2v_sample_rows_bf16(const void* x, int rows, int n, int64_t ld, void* out, void* scratch, void* rng, void* stream);


/* ---- batch-1 decode, persistent-grid kernels (csrc/decode_layer.hip): every launch is 256 workgroups with an equal share
 * of the bytes each, which is what streams HBM at the full rate (one CU pulls ~1/256 of it) -------------------------- */
/* nn.Linear at M = 1 (g2vlm.py:1086-1125).  norm_w != NULL: x is the fp32 residual row and Qwen2RMSNorm(norm_w, eps) is
 * applied on the fly (K <= 1536), else x bf16[K] (K <= 9216).  act: W = gate/up interleaved per 16 rows [N = 2F, K], out
 * bf16[F] = bf16(bf16(silu(g)) * u) (norm_w required).  res != NULL: res[n] f32 += bf16(y[n] + bias[n]), else out[n] = it. */
int g2v_gemv_pg(const void* x, const void* norm_w, float eps, const void* W, const void* bias, void* out, void* res, int N,
                int K, int act, void* stream);
/* The same Linear for B = 1..8 decode rows at once (batched decode; the reference asserts B == 1,
 * g2vlm.py:1006 / 1137): Y[B, N] = X[B, K] . W[N, K]^T with the weights streamed ONCE.  Argument meaning as g2v_gemv_pg;
 * x / out / res are row-major [B, K] / [B, N (N/2 for act)] / [B, N].  K % 8 == 0, K <= 12288 (<= 1536 with norm_w).     */
int g2v_gemv_pg_batch(const void* x, const void* norm_w, float eps, const void* W, const void* bias,
                      void* out, void* res, int B, int N, int K, int act, void* stream);

/* g2v_decode_attn_fused on a persistent grid (same arguments): 256 / Hkv blocks per kv head and scene, each an equal share
 * of the max_len cache rows (the share is fixed by the capacity so that no address depends on the device-side length), one
 * partial per (head, block).  Rows in [Lk_dev[b], max_len) may hold anything.  Hkv <= 128.
 * workspace >= g2v_decode_attn_pg_workspace(Hq, Hkv, batch) bytes.                                                       */
int64_t g2v_decode_attn_pg_workspace(int Hq, int Hkv, int batch);
int g2v_decode_attn_pg(const void* qkv, const void* q_norm_w, const void* k_norm_w, float eps, int und_rounding,
                       const void* cos, const void* sin, void* k_cache, void* v_cache, void* out, const void* Lk_dev,
                       int batch, int64_t scene_rows, int max_len, int Hq, int Hkv, float scale, void* workspace, void* stream);
#ifdef __cplusplus
}
#endif
#endif
