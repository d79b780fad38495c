// Split-KV decode attention of the persistent-grid step (decode_layer.hip) as a device function.
#pragma once
#include "common.h"
#include "decode_util.h"

namespace {

// exchange accesses (see decode_attn_pg_body): plain, or agent scope through the buffer path (aux 16 = sc1 on gfx940+).
// `base` must be wave-uniform (it becomes the buffer resource in SGPRs), `off` is the lane's byte offset.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xch_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <bool XCH, typename T>
__device__ __forceinline__ T xch_load(const void* base, int off) {
  if constexpr (!XCH) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off);
  } else if constexpr (sizeof(T) == 16) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xch_rsrc(base), off, 0, 16);
    return *reinterpret_cast<const T*>(&v);
  } else if constexpr (sizeof(T) == 8) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xch_rsrc(base), off, 0, 16);
    return *reinterpret_cast<const T*>(&v);
  } else {
    static_assert(sizeof(T) == 4, "exchange loads are 4, 8 or 16 bytes");
    const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(xch_rsrc(base), off, 0, 16);
    return *reinterpret_cast<const T*>(&v);
  }
}
template <bool XCH>
__device__ __forceinline__ void xch_store(float* base, int off, float v) {
  if constexpr (XCH) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), xch_rsrc(base), off, 0, 16);
  else *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) = v;
}

constexpr int KB = 32, GMAX = 8;

struct AttnArgs {
  const __bf16* qkv; const float* qw; const float* kw; const float* cs; const float* sn; float eps; int und_rounding;
  __bf16* kc; __bf16* vc; float* ws; const int* Lk_dev; int Hq, Hkv; float scale; long scene_rows; int cap, S, SW;
};

// byte offset of 16-byte chunk `ch` of row `row` in the dual-use LDS image (attn.hip lds_off, guide T10 layout (a))
__device__ __forceinline__ int v_img_off(int row, int ch) {
  return 2048 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

// LDS of one workgroup of the attention (4 waves): 58.4 KB
struct AttnLds {
  __attribute__((aligned(16))) __bf16 sq[4][GMAX + 1][128];   // per wave: normalised q heads + the new k
  __attribute__((aligned(16))) char sv[4][KB * 256];           // per wave: the V batch, dual-use image
  float wm[4][GMAX], wl[4][GMAX];
  __attribute__((aligned(16))) float wo[4][GMAX][128];
};

// The body of the kernel for block (bx of NBH, kv head kvh, scene z).  The workgroup has NWB >= 4 waves: waves 0..3 compute,
// all of them take part in the block barrier and in the merge (the merge is element-wise: the same arithmetic whatever NWB).
// XCH: the partials are written with agent-scope (sc1: write-through) stores, for a caller whose OTHER workgroups consume them
// within the same launch; XLD: the step's q / k / v row is read with sc1 loads as well (needed when its address may be stale
// in this XCD's L2).  The per-phase launch of decode_layer.hip needs neither (round 2's one-launch step, removed in round 3
// after it measured 1.76 ms per token against 1.10, was the caller that did).  The cache rows, norm weights and RoPE row
// come from earlier launches either way.
template <bool XCH, bool XLD, int NWB>
__device__ __forceinline__ void decode_attn_pg_body(const AttnArgs& a, AttnLds& lds, const int bx, const int kvh, const int z, const int NBH,
                                                    const int tid G2V_STAMP_ARG) {
  auto& sq = lds.sq; auto& sv = lds.sv; auto& wm = lds.wm; auto& wl = lds.wl; auto& wo = lds.wo;
  G2V_STAMP_RT(10);
  G2V_STAMP(0);
  const int Hq = a.Hq, Hkv = a.Hkv, G = Hq / Hkv;
  const __bf16* q = a.qkv + (size_t)z * (Hq + 2 * Hkv) * 128;
  __bf16* kc = a.kc + (size_t)z * a.scene_rows * Hkv * 128;
  __bf16* vc = a.vc + (size_t)z * a.scene_rows * Hkv * 128;
  const int lane = tid & 63, w = tid >> 6;
  if (NWB == 4 || w < 4) {
  const int r32 = lane & 31, hh = lane >> 5;                 // MFMA 32x32: row / column index, k half
  const int fr = lane & 15, fg = lane >> 4;
  const int row_stride = Hkv * 128;                          // elements
  const int S = a.S, SW = a.SW;                             // keys per block / per wave, by capacity (host: ceil(cap / NBH), ceil(S / 4))
  const int wlo = bx * S + w * SW;
  const int wcap = min(min(wlo + SW, (bx + 1) * S), a.cap);   // end of this wave's range if the cache were full
  const int last_row = a.cap - 1;
  const __bf16* kbase = kc + kvh * 128 + 8 * hh;            // A operand: lane (r32, hh) takes K[key r32][16 ks + 8 hh ..]
  const __bf16* vbase = vc + kvh * 128 + 8 * fr;            // staging: lane (fg, fr) takes V[row 4 i + fg][8 fr ..]

  // ---- every load first, the step's own rows before the cache (vmcnt retires in issue order: the norms below run while
  // K / V are in flight).  Rows are clamped to the cache block (always mapped); what lies past the length is masked below.
  const int j = lane & 15;
  u32x2 x0r[3], x1r[3];
#pragma unroll
  for (int ps = 0; ps < 3; ++ps) {
    const int item = min(4 * ps + (lane >> 4), G);          // G = the new token's k row
    const int src = 2 * ((item < G ? kvh * G + item : Hq + kvh) * 128 + 4 * j);      // byte offset into the step's qkv row
    x0r[ps] = xch_load<XLD, u32x2>(q, src);
    x1r[ps] = xch_load<XLD, u32x2>(q, src + 128);
  }
  const u32x4 vnew = xch_load<XLD, u32x4>(q, 2 * ((Hq + Hkv + kvh) * 128 + 8 * fr));
  const float* cs = a.cs + (size_t)z * 128;
  const float* sn = a.sn + (size_t)z * 128;
  const f32x4 qw0 = *reinterpret_cast<const f32x4*>(a.qw + 4 * j), qw1 = *reinterpret_cast<const f32x4*>(a.qw + 64 + 4 * j);
  const f32x4 kw0 = *reinterpret_cast<const f32x4*>(a.kw + 4 * j), kw1 = *reinterpret_cast<const f32x4*>(a.kw + 64 + 4 * j);
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + 4 * j), c1 = *reinterpret_cast<const f32x4*>(cs + 64 + 4 * j);
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(sn + 4 * j), s1 = *reinterpret_cast<const f32x4*>(sn + 64 + 4 * j);
  bf16x8 kf[8];
  u32x4 vv[8];
  auto load_batch = [&](int k0) {
    const __bf16* kp = kbase + (size_t)min(k0 + r32, last_row) * row_stride;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 16 * ks);
#pragma unroll
    for (int i = 0; i < 8; ++i) vv[i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)min(k0 + 4 * i + fg, last_row) * row_stride);
  };
  load_batch(wlo);
  const int Lk = a.Lk_dev[z];
  G2V_STAMP(1);

  const int whi = min(wcap, Lk);                            // the wave's real range is [wlo, whi)
  const bool has_new = wlo < whi && whi == Lk;              // it ends with the new token's row (wave-uniform)

  // ---- q / k norm + rotation of the step's rows: every wave, unconditionally (the loads above must not end up behind the
  // wait for the length word; only the STORES of the new k row depend on it)
#pragma unroll
  for (int ps = 0; ps < 3; ++ps) {
    if (4 * ps < G + 1) {                                   // uniform over the launch
      const int c = 4 * ps + (lane >> 4);
      const int item = min(c, G);                           // what this group loaded above
      const bool isq = item < G;
      const u32x2 a0 = x0r[ps], a1 = x1r[ps];
      float x0[4] = {bits2f_lo(a0[0]), bits2f_hi(a0[0]), bits2f_lo(a0[1]), bits2f_hi(a0[1])};
      float x1[4] = {bits2f_lo(a1[0]), bits2f_hi(a1[0]), bits2f_lo(a1[1]), bits2f_hi(a1[1])};
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) ss += x0[e] * x0[e] + x1[e] * x1[e];
      ss = row16_sum(ss);
      const float rstd = 1.0f / sqrtf(ss / 128.f + a.eps);
      const f32x4 w0 = isq ? qw0 : kw0, w1 = isq ? qw1 : kw1;
      float o0[4], o1[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float n0 = __fmul_rn(x0[e], rstd), n1 = __fmul_rn(x1[e], rstd);
        if (a.und_rounding) { n0 = bfround(n0); n1 = bfround(n1); }
        n0 = __fmul_rn(w0[e], n0); n1 = __fmul_rn(w1[e], n1);
        o0[e] = __fadd_rn(__fmul_rn(n0, c0[e]), __fmul_rn(-n1, s0[e]));
        o1[e] = __fadd_rn(__fmul_rn(n1, c1[e]), __fmul_rn(n0, s1[e]));
      }
      const u32x2 p0 = {pack_bf16x2(o0[0], o0[1]), pack_bf16x2(o0[2], o0[3])}, p1 = {pack_bf16x2(o1[0], o1[1]), pack_bf16x2(o1[2], o1[3])};
      if (c <= G) {                                          // strip rows 0..G (row G is only read by the wave that owns the new row)
        *reinterpret_cast<u32x2*>(&sq[w][item][4 * j]) = p0;
        *reinterpret_cast<u32x2*>(&sq[w][item][64 + 4 * j]) = p1;
      }
      if (c == G && has_new) {                               // the new token's K row -> cache row Lk - 1 of this scene
        __bf16* krow = kc + (size_t)(Lk - 1) * row_stride + kvh * 128 + 4 * j;
        *reinterpret_cast<u32x2*>(krow) = p0;
        *reinterpret_cast<u32x2*>(krow + 64) = p1;
      }
    }
  }
  G2V_STAMP(2);
  float m_run = -INFINITY, l_run = 0.f;                      // this lane's head (column r32), raw-score units / its half's keys
  f32x16 O[4];                                               // O^T[d = 32 blk + row][head r32]
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[d][e] = 0.f;
  const float c2 = a.scale * 1.4426950408889634f;            // p = 2^((s - m) c2)

  if (wlo < whi) {
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // the strip is written and read by this wave only
    __builtin_amdgcn_wave_barrier();
    bf16x8 qf[8];                                            // B operand: Q^T[d = 16 ks + 8 hh + j][head r32]
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(&sq[w][min(r32, G - 1)][16 * ks + 8 * hh]);
    if (has_new && fg == 0) *reinterpret_cast<u32x4*>(vc + (size_t)(Lk - 1) * row_stride + kvh * 128 + 8 * fr) = vnew;
    char* sV = sv[w];
    // V^T fragment addresses (attn.hip): row 16 s + 8 jj + 4 hh + tq, chunk 4 d + t_ch -> v_lb[jj] + 2048 (2 s + jj) + 512 d
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int t_ch = 2 * ((lane >> 4) & 1) + (tp >> 1);
    int v_lb[2];
    v_lb[0] = 64 * (4 * hh + tq) + 16 * (t_ch ^ hh) + 8 * (tp & 1);
    v_lb[1] = v_lb[0] ^ 32;

    for (int k0 = wlo; k0 < whi; k0 += KB) {
      const int nk = min(KB, whi - k0);
      if (has_new && k0 + nk == whi) {
        // the batch that ends with the new row: the loads above read whatever the cache row held BEFORE this step
        const int new_local = Lk - 1 - k0;
        if (r32 == new_local) {
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(&sq[w][G][16 * ks + 8 * hh]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (4 * i + fg == new_local) vv[i] = vnew;
      }
      // ---- V batch -> LDS image (rows at or past nk as zeros: 0 x NaN must not reach the MFMA)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + fg;
        const u32x4 val = row < nk ? vv[i] : u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(sV + v_img_off(row, fr)) = val;
      }
      // ---- S^T = K . Q^T: register e of lane (r32, hh) is S[key (e & 3) + 8 (e >> 2) + 4 hh][head r32]
      f32x16 Sx;
#pragma unroll
      for (int e = 0; e < 16; ++e) Sx[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) Sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], Sx, 0, 0, 0);
      if (k0 + KB < whi) {                                   // next batch's K / V under this batch's softmax and P.V
        // (kf and vv are free: the MFMAs above have issued, the LDS stores have read vv)
        load_batch(k0 + KB);
      }
      float rmax = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = (e & 3) + 8 * (e >> 2) + 4 * hh;
        Sx[e] = key < nk ? Sx[e] : -INFINITY;
        rmax = fmaxf(rmax, Sx[e]);
      }
      {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
        rmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      const float m_new = fmaxf(m_run, rmax);                // finite: nk >= 1
      if (k0 > wlo) {                                        // wave-uniform: a second batch rescales what the first left
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int e = 0; e < 16; ++e) O[d][e] *= alpha;
      }
      m_run = m_new;
      const float mc = m_new * c2;
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(Sx[e], c2, -mc));      // masked keys: exp2(-inf) = 0
        Sx[e] = pv;
        psum += pv;
      }
      l_run += psum;
      bf16x8 pf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) pf[s2][jj] = f2bf(Sx[8 * s2 + jj]);
      // ---- O^T += V^T . P^T
      __builtin_amdgcn_s_waitcnt(0xC07F);                    // this wave's V stores have landed (wave-private image)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int s2 = i >> 2, d = i & 3;
        union { struct { s16x4 a, b; } s; bf16x8 v; } uu;
        uu.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sV + v_lb[0] + 2048 * (2 * s2) + 512 * d));
        uu.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sV + v_lb[1] + 2048 * (2 * s2 + 1) + 512 * d));
        O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uu.v, pf[s2], O[d], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();                       // the reads are issued before the next batch's stores (same wave, in order)
      if (k0 == wlo) G2V_STAMP(3);
    }
  }
  G2V_STAMP(4);
  // ---- the wave's result to LDS: lanes r32 < G hold head r32; the two halves hold disjoint d rows and partial l
  {
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
    const float l_tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    if (r32 < G) {
      if (hh == 0) { wm[w][r32] = m_run * a.scale; wl[w][r32] = l_tot; }      // natural-log units, as the combine expects
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(&wo[w][r32][32 * d + 8 * g + 4 * hh]) = f32x4{O[d][4 * g], O[d][4 * g + 1], O[d][4 * g + 2], O[d][4 * g + 3]};
    }
  }
  }
  G2V_STAMP(5);
  __syncthreads();
  G2V_STAMP(6);
  // ---- merge the four waves: one partial per (head, block)
  for (int idx = tid; idx < G * 128; idx += 64 * NWB) {
    const int h = idx >> 7, d = idx & 127;
    float M = fmaxf(fmaxf(wm[0][h], wm[1][h]), fmaxf(wm[2][h], wm[3][h]));
    float L = 0.f, Ov = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float f = wm[k][h] == -INFINITY ? 0.f : __expf(wm[k][h] - M);
      L = fmaf(wl[k][h], f, L);
      Ov = fmaf(wo[k][h][d], f, Ov);
    }
    float* o = a.ws + (((size_t)z * Hq + kvh * G) * NBH + bx) * 130;          // wave-uniform base; head h, word d per lane
    const int ob = 4 * (h * NBH * 130);
    if (d == 0) { xch_store<XCH>(o, ob, M); xch_store<XCH>(o, ob + 4, L); }
    xch_store<XCH>(o, ob + 4 * (2 + d), Ov);
  }
  G2V_STAMP(7);
  G2V_STAMP_RT(11);
}

}  // namespace
