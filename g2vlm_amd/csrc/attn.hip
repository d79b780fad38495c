// Flash attention forward for gfx950: varlen windows, GQA, optional bottom-right causal mask.
// Replaces flash_attn_varlen_func (reference modeling/g2vlm/qwen2vl.py:643-652,
// modeling/g2vlm/dinov2_model.py:49-58, modeling/qwen2vl/modeling_qwen2_vl.py:400) and the SDPA
// flash calls of the Pi3 decoders (modeling/pi3/models/layers/attention.py:255-264, 370-375).
//
// Structure (MI355X guide, "Fused attention prefill"): block = 4 waves x 32 query rows; K/V tiles of
// 64 keys double-buffered in LDS (register-staged, issue-early / write-late); S^T = K.Q^T with
// mfma_f32_32x32x16_bf16 so that a lane owns ONE query column and 32 of the tile's 64 keys in
// registers -> the softmax row reduction is in-register plus one lane^32 exchange; the S^T
// accumulator is re-used directly as the B operand of O^T += V^T.P^T (guide §3 "accumulator tile as
// the next MFMA's operand", permuted-k order), with V^T fragments fetched by ds_read_b64_tr_b16 from
// a row-major V tile.  LDS image = 256-byte rows with the dual-use XOR (guide T10 layout (b)),
// conflict-free for both the b128 row reads of K and the transposed reads of V.
//
// Numerics: S and softmax statistics in fp32, P rounded to bf16 for the PV product (as flash-attn
// does), O accumulated in fp32 and normalised once at the end.
#include "common.h"
#include "g2vlm_hip.h"

namespace {

constexpr int KV_TILE = 64;
constexpr int ROWB = 256;                      // LDS bytes per key row (head dim padded to 128)
constexpr int TILE_B = KV_TILE * ROWB;         // 16 KiB

struct FlashArgs {
  const __bf16* q; const __bf16* k; const __bf16* v; __bf16* o;
  const g2v_attn_tile* tiles;
  int ldq, ldk, ldv, ldo, n_tiles, Hq, Hkv;
  float scale_log2;
};

__device__ __forceinline__ int lds_off(int row, int ch) {
  return ROWB * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

template <int D>
__global__ __launch_bounds__(256, 2) void flash_fwd_kernel(FlashArgs a) {
  constexpr int KSTEPS = D / 16;               // k-steps of the QK^T product
  constexpr int DBLK = (D + 31) / 32;          // 32-wide d blocks of O^T
  constexpr int CH = D / 8;                    // 16-byte chunks per row
  constexpr int NLD = (KV_TILE * CH + 255) / 256;
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];   // K0 V0 K1 V1

  // XCD-aware bijective remap: blocks that share an XCD (bid % 8) get a contiguous logical range,
  // i.e. (nearly) one kv head per XCD, so K/V tiles are L2 hits for all but the first reader
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    int xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  }
  const int head = bid / a.n_tiles;
  const g2v_attn_tile T = a.tiles[bid - head * a.n_tiles];
  const int kvh = head / (a.Hq / a.Hkv);

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;

  // ---- Q fragments (B operand: Q^T[k=d][col=query]) straight from global
  const int qi = min(32 * w + r, T.q_rows - 1);            // clamp: padded rows duplicate the last one
  const __bf16* qp = a.q + (size_t)(T.q0 + qi) * a.ldq + head * D + 8 * hh;
  bf16x8 qf[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);

  // ---- causal extent
  const int q_rel = (T.q0 - T.q_win0) + 32 * w + r;       // query index inside its window
  const int kmax_row = q_rel + T.causal_shift;            // last allowed key (window-local), may be huge
  int k_need = T.k_len;
  {
    long last = (long)(T.q0 - T.q_win0) + T.q_rows - 1 + T.causal_shift + 1;
    if (last < k_need) k_need = (int)last;
  }
  const int n_kt = (k_need + KV_TILE - 1) / KV_TILE;

  // ---- K/V staging map
  const __bf16* kbase = a.k + (size_t)T.k0 * a.ldk + kvh * D;
  const __bf16* vbase = a.v + (size_t)T.k0 * a.ldv + kvh * D;
  u32x4 rk[NLD], rv[NLD];
  int s_row[NLD], s_col[NLD], s_off[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int id = min(tid + 256 * i, KV_TILE * CH - 1);
    s_row[i] = id / CH;
    int ch = id - s_row[i] * CH;
    s_col[i] = ch * 8;
    s_off[i] = lds_off(s_row[i], ch);
  }
  auto stage_load = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int kr = min(kt * KV_TILE + s_row[i], T.k_len - 1);
      rk[i] = *reinterpret_cast<const u32x4*>(kbase + (size_t)kr * a.ldk + s_col[i]);
      rv[i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)kr * a.ldv + s_col[i]);
    }
  };
  auto stage_write = [&](int buf) {
    char* sk = smem + buf * 2 * TILE_B;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if (tid + 256 * i < KV_TILE * CH) {
        *reinterpret_cast<u32x4*>(sk + s_off[i]) = rk[i];
        *reinterpret_cast<u32x4*>(sk + TILE_B + s_off[i]) = rv[i];
      }
    }
  };

  // ---- per-lane LDS read offsets
  // K row read (A operand, key rows): row = 32b + r, chunk = 2ks + hh
  int koff[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) koff[b] = ROWB * (32 * b + r);
  const int kx = ((r & 3) << 2) | ((r >> 2) & 3);         // XOR term of the row (same for 32b + r)
  // V transposed read: group g = lane>>4 covers d columns 16*(g&1).., keys +4*(g>>1); lane i = 4q+p
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int t_row = 4 * hh + tq;                          // + 32b + 16s + 8jj
  const int t_ch = 2 * ((lane >> 4) & 1) + (tp >> 1);     // + 4db
  const int t_sub = 8 * (tp & 1);

  f32x16 O[DBLK];
#pragma unroll
  for (int d = 0; d < DBLK; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[d][e] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c = a.scale_log2;

  stage_load(0);
  stage_write(0);
  __syncthreads();

  for (int kt = 0; kt < n_kt; ++kt) {
    const int buf = kt & 1;
    const char* sK = smem + buf * 2 * TILE_B;
    const char* sV = sK + TILE_B;
    if (kt + 1 < n_kt) stage_load(kt + 1);

    // ---- S^T = K . Q^T
    f32x16 S[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) S[b][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + koff[b] + 16 * ((2 * ks + hh) ^ kx));
        S[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], S[b], 0, 0, 0);
      }
    }

    // ---- scale (log2 domain), mask, online softmax
    const int kb = kt * KV_TILE;
    const bool need_mask = (kb + KV_TILE > T.k_len) || ((long)kb + KV_TILE - 1 > (long)(T.q0 - T.q_win0) + T.causal_shift);
    float rmax = -1e30f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float t = S[b][e] * c;
        if (need_mask) {
          int key = kb + 32 * b + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (key >= T.k_len || key > kmax_row) t = -1e30f;
        }
        S[b][e] = t;
        rmax = fmaxf(rmax, t);
      }
    rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
    const float m_new = fmaxf(m_run, rmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float p = __builtin_amdgcn_exp2f(S[b][e] - m_new);
        if (need_mask && S[b][e] <= -1e30f) p = 0.f;
        S[b][e] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int e = 0; e < 16; ++e) O[d][e] *= alpha;

    // ---- O^T += V^T . P^T   (P^T taken from the S accumulator, permuted-k order)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = f2bf(S[b][8 * s + j]);
#pragma unroll
        for (int d = 0; d < DBLK; ++d) {
          int row0 = 32 * b + 16 * s + t_row;
          const char* p0 = sV + lds_off(row0, 4 * d + t_ch) + t_sub;
          const char* p1 = sV + lds_off(row0 + 8, 4 * d + t_ch) + t_sub;
          s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
          s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
          union { struct { s16x4 a, b; } s; bf16x8 v; } u;
          u.s.a = v0; u.s.b = v1;
          O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, pf, O[d], 0, 0, 0);
        }
      }

    if (kt + 1 < n_kt) stage_write(buf ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane = query row, registers = d
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (32 * w + r < T.q_rows) {
    __bf16* op = a.o + (size_t)(T.q0 + 32 * w + r) * a.ldo + head * D;
#pragma unroll
    for (int d = 0; d < DBLK; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int dc = 32 * d + 8 * g + 4 * hh;
        if (dc < D) {
          u32x2 wv = {pack_bf16x2(O[d][4 * g] * inv, O[d][4 * g + 1] * inv),
                      pack_bf16x2(O[d][4 * g + 2] * inv, O[d][4 * g + 3] * inv)};
          *reinterpret_cast<u32x2*>(op + dc) = wv;
        }
      }
  }
}

template <int D>
int launch_flash(const FlashArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(flash_fwd_kernel<D>, dim3(a.n_tiles * a.Hq), dim3(256), 0, s, a);
  G2V_CHECK_LAUNCH();
  return G2V_OK;
}

}  // namespace

extern "C" int g2v_flash_attn(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int ldo,
                              const g2v_attn_tile* tiles, int n_tiles, int Hq, int Hkv, int D, float scale, void* stream) {
  if (!q || !k || !v || !o || !tiles || n_tiles < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv) return G2V_ERR_ARG;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3)) return G2V_ERR_ARG;
  if (n_tiles == 0) return G2V_OK;
  FlashArgs a{(const __bf16*)q, (const __bf16*)k, (const __bf16*)v, (__bf16*)o, tiles, ldq, ldk, ldv, ldo, n_tiles, Hq, Hkv,
              scale * 1.4426950408889634f};
  hipStream_t s = (hipStream_t)stream;
  switch (D) {
    case 16: return launch_flash<16>(a, s);
    case 64: return launch_flash<64>(a, s);
    case 80: return launch_flash<80>(a, s);
    case 96: return launch_flash<96>(a, s);
    case 128: return launch_flash<128>(a, s);
    default: return G2V_ERR_ARG;
  }
}
