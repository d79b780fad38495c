// Data movers and the fp32 head tails: patch im2col, DINO token assembly, row gather/scatter,
// dtype casts, pixel-shuffle + exp + unprojection epilogue, camera-head tail with 3x3 SVD.
#include "common.h"
#include "g2vlm_hip.h"

namespace {

// out[(n*P + py*gw + px), k] = img[n, c, py*14+ky, px*14+kx], k = c*196 + ky*14 + kx  (Conv2d weight order)
__global__ void im2col14_kernel(const float* img, int N, int H, int W, __bf16* out, int Kpad) {
  int gw = W / 14, gh = H / 14, P = gw * gh;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)N * P * Kpad;
  if (i >= total) return;
  int k = (int)(i % Kpad);
  long rp = i / Kpad;
  float v = 0.f;
  if (k < 588) {
    int p = (int)(rp % P), n = (int)(rp / P);
    int py = p / gw, px = p - py * gw;
    int c = k / 196, r = k - c * 196, ky = r / 14, kx = r - ky * 14;
    v = img[(((size_t)n * 3 + c) * H + py * 14 + ky) * W + px * 14 + kx];
  }
  out[i] = f2bf(v);
}

// DINO input preprocessing on the device (reference g2vlm.py:947-953: ToTensor's k/255, torchvision Normalize, the
// original_images copy): one thread per pixel and channel of the NCHW outputs.  U8 = true reads the loader's uint8 frame
// [N,H,W,3] (a quarter of the bytes over PCIe), else an fp32 [N,3,H,W] image already in [0,1].  IEEE sub / div in the
// reference's order, so both outputs are bit-identical to the host tensors.
template <bool U8>
__global__ void dino_preprocess_kernel(const void* in, float* norm, float* orig, int N, int H, int W, float m0, float m1, float m2,
                                       float s0, float s1, float s2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long hw = (long)H * W;
  if (i >= (long)N * 3 * hw) return;
  const int c = (int)((i / hw) % 3);
  float x;
  if constexpr (U8) {
    const long n = i / (3 * hw), p = i % hw;
    x = __fdiv_rn((float)reinterpret_cast<const unsigned char*>(in)[(n * hw + p) * 3 + c], 255.0f);
  } else {
    x = reinterpret_cast<const float*>(in)[i];
  }
  const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
  if (orig) orig[i] = x;
  norm[i] = __fdiv_rn(__fsub_rn(x, m), sd);
}

__global__ void dino_assemble_kernel(const __bf16* patch, const float* cls, const float* regs, const float* pos, float* x,
                                     int N, int P, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int S = P + 5;
  if (i >= (long)N * S * C) return;
  int c = (int)(i % C);
  long rt = i / C;
  int t = (int)(rt % S), n = (int)(rt / S);
  float v;
  if (t == 0) v = cls[c] + pos[c];
  else if (t < 5) v = regs[(t - 1) * C + c];
  else v = bf2f(patch[((size_t)n * P + (t - 5)) * C + c]) + pos[(size_t)(t - 4) * C + c];
  x[i] = v;
}

// generic patch size (DINOv3: Conv2d 16/16, reference modeling/dinov3/dinov3_model.py:47-49): K = 3 * ps * ps, zero-padded to Kpad
__global__ void im2col_patch_kernel(const float* img, int N, int H, int W, int ps, __bf16* out, int Kpad) {
  int gw = W / ps, gh = H / ps, P = gw * gh, pp = ps * ps;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)N * P * Kpad;
  if (i >= total) return;
  int k = (int)(i % Kpad);
  long rp = i / Kpad;
  float v = 0.f;
  if (k < 3 * pp) {
    int p = (int)(rp % P), n = (int)(rp / P);
    int py = p / gw, px = p - py * gw;
    int c = k / pp, r = k - c * pp, ky = r / ps, kx = r - ky * ps;
    v = img[(((size_t)n * 3 + c) * H + py * ps + ky) * W + px * ps + kx];
  }
  out[i] = f2bf(v);
}

// DINOv3ViTEmbeddings.forward (dinov3_model.py:51-69): [cls | R registers | patches] per view, no position table
__global__ void vit_assemble_kernel(const __bf16* patch, const float* cls, const float* regs, float* x, int N, int P, int R, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int S = P + 1 + R;
  if (i >= (long)N * S * C) return;
  int c = (int)(i % C);
  long rt = i / C;
  int t = (int)(rt % S), n = (int)(rt / S);
  float v;
  if (t == 0) v = cls[c];
  else if (t <= R) v = regs[(t - 1) * C + c];
  else v = bf2f(patch[((size_t)n * P + (t - 1 - R)) * C + c]);
  x[i] = v;
}

// Qwen2VLImageProcessor._preprocess after the PIL resize (image_processing_qwen2_vl.py:218-273), on the device: uint8 frames
// [F, H, W, 3] -> rescale (x * (1/255) in float64, as the HF rescale does, then float32) -> (x - mean) / std in fp32 ->
// temporal pairs (an odd last frame repeated) -> the 9-d patch reorder -> bf16 A operand of the patch GEMM [T, Kpad]:
// row = ((t * gh/2 + bh) * gw/2 + bw) * 4 + mh * 2 + mw,  col = ((c * 2 + tt) * 14 + ky) * 14 + kx, zero beyond 1176.
__global__ void qwen_patchify_u8_kernel(const unsigned char* img, int F, int H, int W, float m0, float m1, float m2, float s0, float s1,
                                        float s2, __bf16* out, int Kpad) {
  const int gh = H / 14, gw = W / 14, gt = (F + 1) / 2;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)gt * gh * gw * Kpad;
  if (i >= total) return;
  int k = (int)(i % Kpad);
  long row = i / Kpad;
  float v = 0.f;
  if (k < 1176) {
    int mw = (int)(row & 1), mh = (int)((row >> 1) & 1);
    long rb = row >> 2;
    int bw = (int)(rb % (gw / 2)); rb /= (gw / 2);
    int bh = (int)(rb % (gh / 2));
    int t = (int)(rb / (gh / 2));
    int kx = k % 14, r = k / 14, ky = r % 14; r /= 14;
    int tt = r & 1, c = r >> 1;
    int f = min(2 * t + tt, F - 1);
    int y = (2 * bh + mh) * 14 + ky, x = (2 * bw + mw) * 14 + kx;
    float a = (float)((double)img[(((size_t)f * H + y) * W + x) * 3 + c] * (1.0 / 255.0));
    float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    v = __fdiv_rn(__fsub_rn(a, mean), sd);
  }
  out[i] = f2bf(v);
}

template <bool SCATTER>
__global__ void move_rows_kernel(const float* src, int ld_src, const int* idx, float* dst, int ld_dst, int rows, int C4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * C4) return;
  int c = (int)(i % C4), r = (int)(i / C4);
  int sr = SCATTER ? r : idx[r], dr = SCATTER ? idx[r] : r;
  reinterpret_cast<f32x4*>(dst + (size_t)dr * ld_dst)[c] = reinterpret_cast<const f32x4*>(src + (size_t)sr * ld_src)[c];
}

__global__ void cast_f32_bf16_kernel(const float* s, __bf16* d, long n4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = reinterpret_cast<const f32x4*>(s)[i];
  u32x2 w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  reinterpret_cast<u32x2*>(d)[i] = w;
}
__global__ void cast_bf16_f32_kernel(const __bf16* s, float* d, long n4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  u32x2 w = reinterpret_cast<const u32x2*>(s)[i];
  reinterpret_cast<f32x4*>(d)[i] = f32x4{bits2f_lo(w[0]), bits2f_hi(w[0]), bits2f_lo(w[1]), bits2f_hi(w[1])};
}

// Pi3LinearPts3d tail (transformer_head.py:69-81): feat [N*P, 3*PS*PS] -> [N,H,W,3]; one thread per pixel.  PS = the
// encoder's patch size: 14 (DINOv2, g2vlm.py:172) or 16 (DINOv3, g2vlm.py:170)
template <int PS>
__global__ void pts_epilogue_kernel(const float* feat, int N, int H, int W, int mode, const float* pose, float* out,
                                    float* out2) {
  constexpr int PP = PS * PS;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * H * W) return;
  int x = (int)(i % W);
  long t = i / W;
  int y = (int)(t % H), n = (int)(t / H);
  int gw = W / PS, P = gw * (H / PS);
  int py = y / PS, iy = y - py * PS, px = x / PS, ix = x - px * PS;
  const float* f = feat + ((size_t)n * P + py * gw + px) * (3 * PP) + iy * PS + ix;
  float a = f[0], b = f[PP], c = f[2 * PP];
  if (mode == 0) {
    out[i * 3 + 0] = a; out[i * 3 + 1] = b; out[i * 3 + 2] = c;
    return;
  }
  float z = expf(c);
  float lx = __fmul_rn(a, z), ly = __fmul_rn(b, z);
  out[i * 3 + 0] = lx; out[i * 3 + 1] = ly; out[i * 3 + 2] = z;
  const float* T = pose + (size_t)n * 16;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float v = T[r * 4 + 0] * lx;
    v = fmaf(T[r * 4 + 1], ly, v);
    v = fmaf(T[r * 4 + 2], z, v);
    v = v + T[r * 4 + 3];
    out2[i * 3 + r] = v;
  }
}

// F.pixel_shuffle(feat [N, C*PS*PS, h, w], PS) -> [N, H, W, C] for a Pi3LinearPts3d of any output_dim (the confidence head
// has C = 1, transformer_head.py:58-81): feat [N*P, C*PS*PS] row-major, one thread per output element
template <int PS>
__global__ void pixel_shuffle_kernel(const float* feat, int N, int H, int W, int C, float* out) {
  constexpr int PP = PS * PS;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * H * W * C) return;
  int c = (int)(i % C);
  long t = i / C;
  int x = (int)(t % W);
  t /= W;
  int y = (int)(t % H), n = (int)(t / H);
  int gw = W / PS, P = gw * (H / PS);
  int py = y / PS, iy = y - py * PS, px = x / PS, ix = x - px * PS;
  out[i] = feat[((size_t)n * P + py * gw + px) * (PP * C) + c * PP + iy * PS + ix];
}

// ---- Pillow's 8-bit LANCZOS resample on the device (reference data/transforms_vggt.py:437 calls PIL's Image.resize) ----------
// Integer restatement of ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc (Pillow src/libImaging/Resample.c):
// out = clip8((2^21 + sum_t pixel[first + t] * k[t]) >> 22) with the 22-bit fixed-point taps of host.lanczos_tables.
// AXIS 0: along x (src [N, H, Win, 3] -> dst [N, H, Wout, 3]); AXIS 1: along y (src [N, Hin, W, 3] -> dst [N, Hout, W, 3]).
// One thread per output pixel (3 channels).
template <int AXIS>
__global__ void lanczos_pass_kernel(const unsigned char* src, unsigned char* dst, int N, int Hs, int Ws, int Hd, int Wd, const int* bounds,
                                    const int* kk, int ksize) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * Hd * Wd) return;
  const int x = (int)(i % Wd);
  const long t = i / Wd;
  const int y = (int)(t % Hd), n = (int)(t / Hd);
  const int o = AXIS == 0 ? x : y;
  const int first = bounds[2 * o], cnt = bounds[2 * o + 1];
  const int* k = kk + (size_t)o * ksize;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  const unsigned char* p = AXIS == 0 ? src + (((size_t)n * Hs + y) * Ws + first) * 3 : src + (((size_t)n * Hs + first) * Ws + x) * 3;
  const size_t step = AXIS == 0 ? 3 : (size_t)Ws * 3;
  for (int j = 0; j < cnt; ++j) {
    const int c = k[j];
    s0 += (int)p[0] * c; s1 += (int)p[1] * c; s2 += (int)p[2] * c;
    p += step;
  }
  unsigned char* q = dst + (size_t)i * 3;
  q[0] = (unsigned char)min(max(s0 >> 22, 0), 255);
  q[1] = (unsigned char)min(max(s1 >> 22, 0), 255);
  q[2] = (unsigned char)min(max(s2 >> 22, 0), 255);
}

// ---- camera tail: one 512-thread block per view --------------------------------------------------
__device__ void matvec512(const float* W, const float* b, const float* vin, float* vout, int n_out, bool relu) {
  int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int o = w; o < n_out; o += 8) {
    const float* r = W + (size_t)o * 512;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = fmaf(r[k * 64 + lane], vin[k * 64 + lane], s);
    s = wave_sum(s);
    if (lane == 0) { s += b[o]; vout[o] = relu ? fmaxf(s, 0.f) : s; }
  }
}

// R = V diag(1,1,det(V U^T)) U^T for svd(normalize_rows(m)^T) = U S V^T (camera_head.py:75-93), i.e. the
// rotation nearest to A = normalize_rows(m).  One-sided Jacobi in double on the 3x3.
__device__ void svd_orthogonalize(const float* m9, float* R9) {
  double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i) {
    float n = sqrtf(m9[i * 3] * m9[i * 3] + m9[i * 3 + 1] * m9[i * 3 + 1] + m9[i * 3 + 2] * m9[i * 3 + 2]);
    n = fmaxf(n, 1e-12f);
    for (int j = 0; j < 3; ++j) A[i][j] = (double)(m9[i * 3 + j] / n);
  }
  // Hestenes: rotate column pairs of A (and V) until columns are orthogonal: A_in = (A) V^T
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < 3; ++i) { al += A[i][p] * A[i][p]; be += A[i][q] * A[i][q]; ga += A[i][p] * A[i][q]; }
        off += ga * ga;
        if (fabs(ga) < 1e-300) continue;
        double zeta = (be - al) / (2.0 * ga);
        double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
        for (int i = 0; i < 3; ++i) {
          double ap = A[i][p], aq = A[i][q];
          A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
          double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
    if (off < 1e-30) break;
  }
  // A(now) = Us * diag(sig); A_in = Us diag(sig) V^T
  double sig[3], Us[3][3];
  for (int j = 0; j < 3; ++j) {
    sig[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    double inv = sig[j] > 1e-150 ? 1.0 / sig[j] : 0.0;
    for (int i = 0; i < 3; ++i) Us[i][j] = A[i][j] * inv;
  }
  int jmin = 0;
  for (int j = 1; j < 3; ++j) if (sig[j] < sig[jmin]) jmin = j;
  // nearest rotation to A_in: Us D V^T, D flips the smallest singular direction when det(Us V^T) < 0
  double M[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i][j] = Us[i][0] * V[j][0] + Us[i][1] * V[j][1] + Us[i][2] * V[j][2];
  double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
               M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
  double d = det < 0 ? -1.0 : 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double v = 0;
      for (int k = 0; k < 3; ++k) v += Us[i][k] * (k == jmin ? d : 1.0) * V[j][k];
      R9[i * 3 + j] = (float)v;
    }
}

__global__ __launch_bounds__(512) void camera_tail_kernel(const float* feat, int P, const float* w0, const float* b0,
                                                          const float* w1, const float* b1, const float* wt, const float* bt,
                                                          const float* wr, const float* br, float* pose) {
  __shared__ float va[512], vb[512], small[12];
  int n = blockIdx.x, c = threadIdx.x;
  const float* f = feat + (size_t)n * P * 512;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = 0;
  for (; p + 3 < P; p += 4) {
    s0 += f[(size_t)p * 512 + c]; s1 += f[(size_t)(p + 1) * 512 + c];
    s2 += f[(size_t)(p + 2) * 512 + c]; s3 += f[(size_t)(p + 3) * 512 + c];
  }
  for (; p < P; ++p) s0 += f[(size_t)p * 512 + c];
  va[c] = ((s0 + s1) + (s2 + s3)) / (float)P;
  __syncthreads();
  matvec512(w0, b0, va, vb, 512, true);
  __syncthreads();
  matvec512(w1, b1, vb, va, 512, true);
  __syncthreads();
  matvec512(wt, bt, va, small, 3, false);
  matvec512(wr, br, va, small + 3, 9, false);
  __syncthreads();
  if (c == 0) {
    float R[9];
    svd_orthogonalize(small + 3, R);
    float* T = pose + (size_t)n * 16;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
      T[i * 4 + 3] = small[i];
    }
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
  }
}

// argmax over bf16 logits (first maximal index): 64 blocks reduce slices to (value, index) pairs, the last block to
// arrive (atomic ticket) reduces the 64 partials.  scratch: int32[1 + 2*64] zeroed once by the caller (ticket is reset).
// NaN ranks above everything (torch.argmax returns the index of a NaN), so the result is ALWAYS an index of the input:
// a decode loop that feeds it to an embedding gather must not be able to leave the table.
__device__ __forceinline__ void argmax_merge(float& best, int& bi, float ov, int oi) {
  if (ov != ov) ov = INFINITY;
  if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
}

// Philox4x32-10 (Salmon et al., SC'11): counter-based, so every (row, index) of every decode step draws its own number
// with no state shared between lanes; the host restatement in tests/test_kernels_gpu.py reproduces it bit for bit.
__device__ __forceinline__ uint32_t philox_first(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}

// blockIdx.y = row of a [rows, ld] logits matrix (batched decode): own output slot and own 129-int scratch slab.
// SAMPLE: torch.multinomial(softmax(logits / temperature)) of the reference's do_sample branch (g2vlm.py:1119-1122) as a
// Gumbel-max draw: argmax_i (logit_i / T - log(-log u_i)), u_i = Philox(seed, step, row, i) - the same distribution, one
// pass, no normaliser.  rng = int32[4] on the device: {seed lo, seed hi, step, float bits of 1 / T}; the block that
// finishes a row's reduction bumps nothing: the step counter is advanced by rng_step_kernel (launched behind the draw by g2v_sample_rows_bf16: one writer per step).
template <bool SAMPLE>
__global__ __launch_bounds__(256) void argmax_bf16_kernel(const __bf16* x, int n, int* out, int* scratch, long ld, const int* rng) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int last;
  const int nb = gridDim.x;
  x += (size_t)blockIdx.y * ld;
  out += blockIdx.y;
  scratch += blockIdx.y * 129;
  float best = -INFINITY;
  int bi = 0x7fffffff;                                     // "no candidate yet": loses every tie; replaced below if still unset
  const int per = (n + nb - 1) / nb, lo = blockIdx.x * per, hi = min(n, lo + per);
  if constexpr (SAMPLE) {
    const uint32_t k0 = (uint32_t)rng[0], k1 = (uint32_t)rng[1], step = (uint32_t)rng[2];
    const float inv_t = __int_as_float(rng[3]);
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
      const uint32_t r = philox_first((uint32_t)i, blockIdx.y, step, 0u, k0, k1);
      const float u = ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);          // 24 bits, strictly inside (0, 1)
      argmax_merge(best, bi, bf2f(x[i]) * inv_t - logf(-logf(u)), i);
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += 256) argmax_merge(best, bi, bf2f(x[i]), i);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) argmax_merge(best, bi, __shfl_xor(best, o, 64), __shfl_xor(bi, o, 64));
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) argmax_merge(best, bi, sv[k], si[k]);
    __hip_atomic_store(reinterpret_cast<float*>(scratch + 1) + 2 * blockIdx.x, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(scratch + 2 + 2 * blockIdx.x, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int t = __hip_atomic_fetch_add(scratch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (t == nb - 1);
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float b2 = -INFINITY; int i2 = 0x7fffffff;
    if ((int)threadIdx.x < nb) {
      b2 = __hip_atomic_load(reinterpret_cast<float*>(scratch + 1) + 2 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      i2 = __hip_atomic_load(scratch + 2 + 2 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(b2, i2, __shfl_xor(b2, o, 64), __shfl_xor(i2, o, 64));
    if (threadIdx.x == 0) { out[0] = (i2 >= 0 && i2 < n) ? i2 : 0; __hip_atomic_store(scratch, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
}

__global__ void rng_step_kernel(int* rng) { rng[2] += 1; }

inline unsigned blocks_for(long n, int b = 256) { return (unsigned)((n + b - 1) / b); }

}  // namespace

extern "C" int g2v_im2col14(const void* img, int N, int H, int W, void* out, int Kpad, void* stream) {
  if (!img || !out || N < 0 || H % 14 || W % 14 || Kpad < 588 || (Kpad & 7)) return G2V_ERR_ARG;
  long total = (long)N * (H / 14) * (W / 14) * Kpad;
  if (total == 0) return G2V_OK;
  hipLaunchKernelGGL(im2col14_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)img, N, H, W,
                     (__bf16*)out, Kpad);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_dino_preprocess(const void* in, int in_is_u8, int N, int H, int W, const float* mean3, const float* std3, void* norm,
                                   void* orig, void* stream) {
  if (!in || !norm || !mean3 || !std3 || N < 0 || H <= 0 || W <= 0) return G2V_ERR_ARG;
  const long total = (long)N * 3 * H * W;
  if (total == 0) return G2V_OK;
  if (in_is_u8)
    hipLaunchKernelGGL(dino_preprocess_kernel<true>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, in, (float*)norm,
                       (float*)orig, N, H, W, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  else
    hipLaunchKernelGGL(dino_preprocess_kernel<false>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, in, (float*)norm,
                       (float*)orig, N, H, W, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_lanczos_resize_u8(const void* src, int N, int Hin, int Win, void* dst, int Hout, int Wout, void* tmp,
                                     const void* bounds_h, const void* kk_h, int ksize_h, const void* bounds_v, const void* kk_v,
                                     int ksize_v, void* stream) {
  const bool need_h = Win != Wout, need_v = Hin != Hout;
  if (!src || !dst || N < 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || (!need_h && !need_v) ||
      (need_h && (!bounds_h || !kk_h || ksize_h <= 0)) || (need_v && (!bounds_v || !kk_v || ksize_v <= 0)) || (need_h && need_v && !tmp))
    return G2V_ERR_ARG;
  if (N == 0) return G2V_OK;
  hipStream_t s = (hipStream_t)stream;
  const unsigned char* in = (const unsigned char*)src;
  if (need_h) {                                              // Pillow's order: horizontal first, on the source rows
    unsigned char* o = need_v ? (unsigned char*)tmp : (unsigned char*)dst;
    hipLaunchKernelGGL(lanczos_pass_kernel<0>, dim3(blocks_for((long)N * Hin * Wout)), dim3(256), 0, s, in, o, N, Hin, Win, Hin, Wout,
                       (const int*)bounds_h, (const int*)kk_h, ksize_h);
    G2V_CHECK_LAUNCH();
    in = o;
  }
  if (need_v) {
    hipLaunchKernelGGL(lanczos_pass_kernel<1>, dim3(blocks_for((long)N * Hout * Wout)), dim3(256), 0, s, in, (unsigned char*)dst, N, Hin,
                       Wout, Hout, Wout, (const int*)bounds_v, (const int*)kk_v, ksize_v);
    G2V_CHECK_LAUNCH();
  }
  return G2V_OK;
}

extern "C" int g2v_dino_assemble(const void* patch, const void* cls, const void* regs, const void* pos, void* x, int N, int P,
                                 int C, void* stream) {
  if (!patch || !cls || !regs || !pos || !x || N < 0 || P <= 0 || C <= 0) return G2V_ERR_ARG;
  long total = (long)N * (P + 5) * C;
  if (total == 0) return G2V_OK;
  hipLaunchKernelGGL(dino_assemble_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)patch,
                     (const float*)cls, (const float*)regs, (const float*)pos, (float*)x, N, P, C);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_im2col_patch(const void* img, int N, int H, int W, int patch, void* out, int Kpad, void* stream) {
  if (!img || !out || N < 0 || patch <= 0 || H % patch || W % patch || Kpad < 3 * patch * patch || (Kpad & 7)) return G2V_ERR_ARG;
  long total = (long)N * (H / patch) * (W / patch) * Kpad;
  if (total == 0) return G2V_OK;
  hipLaunchKernelGGL(im2col_patch_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)img, N, H, W, patch,
                     (__bf16*)out, Kpad);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_vit_assemble(const void* patch, const void* cls, const void* regs, void* x, int N, int P, int R, int C, void* stream) {
  if (!patch || !cls || !x || N < 0 || P <= 0 || R < 0 || C <= 0 || (R > 0 && !regs)) return G2V_ERR_ARG;
  long total = (long)N * (P + 1 + R) * C;
  if (total == 0) return G2V_OK;
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)patch,
                     (const float*)cls, (const float*)regs, (float*)x, N, P, R, C);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_qwen_patchify_u8(const void* img_u8, int F, int H, int W, const float* mean3, const float* std3, void* out, int Kpad,
                                    void* stream) {
  if (!img_u8 || !mean3 || !std3 || !out || F <= 0 || H <= 0 || W <= 0 || H % 28 || W % 28 || Kpad < 1176 || (Kpad & 7)) return G2V_ERR_ARG;
  long total = (long)((F + 1) / 2) * (H / 14) * (W / 14) * Kpad;
  hipLaunchKernelGGL(qwen_patchify_u8_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)img_u8,
                     F, H, W, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], (__bf16*)out, Kpad);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_gather_rows_f32(const void* src, int ld_src, const void* idx, void* dst, int ld_dst, int rows, int C,
                                   void* stream) {
  if (!src || !idx || !dst || rows < 0 || C <= 0 || (C & 3) || (ld_src & 3) || (ld_dst & 3)) return G2V_ERR_ARG;
  if (rows == 0) return G2V_OK;
  hipLaunchKernelGGL((move_rows_kernel<false>), dim3(blocks_for((long)rows * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const float*)src, ld_src, (const int*)idx, (float*)dst, ld_dst, rows, C / 4);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_scatter_rows_f32(const void* src, int ld_src, const void* idx, void* dst, int ld_dst, int rows, int C,
                                    void* stream) {
  if (!src || !idx || !dst || rows < 0 || C <= 0 || (C & 3) || (ld_src & 3) || (ld_dst & 3)) return G2V_ERR_ARG;
  if (rows == 0) return G2V_OK;
  hipLaunchKernelGGL((move_rows_kernel<true>), dim3(blocks_for((long)rows * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const float*)src, ld_src, (const int*)idx, (float*)dst, ld_dst, rows, C / 4);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_cast_f32_bf16(const void* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n < 0 || (n & 3)) return G2V_ERR_ARG;
  if (n == 0) return G2V_OK;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)src,
                     (__bf16*)dst, (long)(n / 4));
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_cast_bf16_f32(const void* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n < 0 || (n & 3)) return G2V_ERR_ARG;
  if (n == 0) return G2V_OK;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src,
                     (float*)dst, (long)(n / 4));
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_pts_epilogue_ps(const void* feat, int N, int H, int W, int patch, int mode, const void* pose, void* out,
                                   void* out2, void* stream) {
  if (!feat || !out || N < 0 || (patch != 14 && patch != 16) || H % patch || W % patch || (mode == 1 && (!pose || !out2)))
    return G2V_ERR_ARG;
  long total = (long)N * H * W;
  if (total == 0) return G2V_OK;
  auto k = patch == 14 ? pts_epilogue_kernel<14> : pts_epilogue_kernel<16>;
  hipLaunchKernelGGL(k, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)feat, N, H, W, mode,
                     (const float*)pose, (float*)out, (float*)out2);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_pts_epilogue(const void* feat, int N, int H, int W, int mode, const void* pose, void* out, void* out2,
                                void* stream) {
  return g2v_pts_epilogue_ps(feat, N, H, W, 14, mode, pose, out, out2, stream);
}

extern "C" int g2v_pixel_shuffle(const void* feat, int N, int H, int W, int C, int patch, void* out, void* stream) {
  if (!feat || !out || N < 0 || C <= 0 || (patch != 14 && patch != 16) || H % patch || W % patch) return G2V_ERR_ARG;
  long total = (long)N * H * W * C;
  if (total == 0) return G2V_OK;
  auto k = patch == 14 ? pixel_shuffle_kernel<14> : pixel_shuffle_kernel<16>;
  hipLaunchKernelGGL(k, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)feat, N, H, W, C, (float*)out);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_pixel_shuffle14(const void* feat, int N, int H, int W, int C, void* out, void* stream) {
  return g2v_pixel_shuffle(feat, N, H, W, C, 14, out, stream);
}

extern "C" int g2v_camera_tail(const void* feat, int N, int P, const void* w0, const void* b0, const void* w1, const void* b1,
                               const void* wt, const void* bt, const void* wr, const void* br, void* pose, void* stream) {
  if (!feat || !w0 || !b0 || !w1 || !b1 || !wt || !bt || !wr || !br || !pose || N < 0 || P <= 0) return G2V_ERR_ARG;
  if (N == 0) return G2V_OK;
  hipLaunchKernelGGL(camera_tail_kernel, dim3(N), dim3(512), 0, (hipStream_t)stream, (const float*)feat, P, (const float*)w0,
                     (const float*)b0, (const float*)w1, (const float*)b1, (const float*)wt, (const float*)bt, (const float*)wr,
                     (const float*)br, (float*)pose);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_argmax_bf16(const void* x, int n, void* out, void* scratch, void* stream) {
  if (!x || !out || !scratch || n <= 0) return G2V_ERR_ARG;
  hipLaunchKernelGGL(argmax_bf16_kernel<false>, dim3(n >= 65536 ? 64 : 1), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, n, (int*)out,
                     (int*)scratch, 0L, (const int*)nullptr);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// row-wise argmax of bf16 [rows, ld] (first maximal index per row); out int32[rows]; scratch int32[rows * 129], zeroed once
extern "C" int g2v_argmax_rows_bf16(const void* x, int rows, int n, int64_t ld, void* out, void* scratch, void* stream) {
  if (!x || !out || !scratch || n <= 0 || rows <= 0 || rows > 65535 || ld < n) return G2V_ERR_ARG;
  hipLaunchKernelGGL(argmax_bf16_kernel<false>, dim3(n >= 65536 ? 64 : 1, rows), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, n,
                     (int*)out, (int*)scratch, (long)ld, (const int*)nullptr);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

// row-wise draw from softmax(x / T) (the reference's do_sample branch, g2vlm.py:1119-1122) by Gumbel-max over a Philox
// stream; rng int32[4] on the device = {seed lo, seed hi, step, float bits of 1/T}.  The step word is advanced by this call
// (a one-thread kernel behind the draw), so a captured decode step draws fresh numbers at every replay.
extern "C" int g2v_sample_rows_bf16(const void* x, int rows, int n, int64_t ld, void* out, void* scratch, void* rng, void* stream) {
  if (!x || !out || !scratch || !rng || n <= 0 || rows <= 0 || rows > 65535 || ld < n) return G2V_ERR_ARG;
  hipLaunchKernelGGL(argmax_bf16_kernel<true>, dim3(n >= 65536 ? 64 : 1, rows), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, n,
                     (int*)out, (int*)scratch, (long)ld, (const int*)rng);
  G2V_CHECK_LAUNCH();
  hipLaunchKernelGGL(rng_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)rng);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

extern "C" int g2v_version(void) { return 1; }
extern "C" const char* g2v_arch(void) { return "gfx950"; }
