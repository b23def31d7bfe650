// Shared pieces of the implicit-GEMM conv kernels (conv_igemm.hip: register-staged + 3-slot LDS-DMA ring;
// conv_igemm_p8.hip: the deep-pipelined 256 x 256 tile): the kernel argument block, the fragment tile helpers and the
// epilogues (channels-last stores, BatchNorm partial sums, fused output transforms).  Everything here is inline device
// code in a named namespace, so both translation units agree on the types they hand each other.
#pragma once
#include "sfk_common.h"
#include <stdlib.h>

namespace sfk_igemm {


struct ConvK {
  const void* x;
  void* y;
  const void* w;
  float* stats;
  int xt, xh, xw, xld, xoff;
  int yt, yh, yw, yld, yoff;
  int M;
  FastDiv drw, drh, drt;
  int gst, gsh, gsw, ost, osh, osw, oot, ooh, oow;
  int cin, cout, wtaps, ntaps, KC, accumulate;
  int mtiles, ntiles;
  int wide_store;   // bf16 output with 16-byte addressable 8-channel groups
  int kshort;       // K-step count up to which the exact-count K loop runs
  int lin_out;      // output pixel index == row index (os = 1, oo = 0, row extents = y extents)
  // fused BatchNorm-backward reduce (sfk_conv_desc.bnb): the stored value becomes dz = result * mask
  const void* bn_y;        // the conv output the BatchNorm normalised (same pixel grid as y)
  const void* bn_mask;     // activation whose sign is the ReLU mask, or NULL
  int bn_yld, bn_yoff, bn_mld, bn_moff, bn_relu;
  const float *bn_mean, *bn_invstd, *bn_scale, *bn_shift;
  float* bn_parts;         // [mtiles][cout][2] = (sum dz, sum dz * x_hat) per row tile; NULL = fusion off
  const uint8_t* obits;    // ReLU bitmap applied to the stored result (sfk_conv_desc.out_relu_bits), or NULL
  // fused output transform (sfk_conv_desc.ep): v = acc*scale + shift (+ old) (+ res*rscale + rshift), ReLU (+ bitmap)
  const float *ep_scale, *ep_shift, *ep_rscale, *ep_rshift;
  const void* ep_res;
  int ep_rld, ep_roff, ep_relu, ep_on;
  uint8_t* ep_bits;
  FastDiv dspt;   // 16-byte channel segments per tap (cin / VEC)
  FastDiv dkct;   // K-steps per tap of the uniform walk (cin / 32)
  FastDiv dk64;   // by ntaps: K-tile -> (channel chunk, tap) of conv_igemm_p8's walk
  uint32_t xbytes, wbytes;   // extents of the two buffer resources
  uint32_t ybytes, ep_rbytes; // ... and of y / the shortcut map (fused epilogue: branch-free loads)
  int ybig;                   // y holds 4 GiB or more: the plain epilogue's += reads it through 64-bit pointers
  uint32_t bn_ybytes, bn_mbytes, obits_bytes;   // ... and of the fused BatchNorm-backward epilogue's operands
  sfk_tap taps[SFK_MAX_TAPS];
};

constexpr int BK = 32;

template <typename T> struct Tile;
template <> struct Tile<bf16_t> {
  static constexpr int VEC = 8, SEGS = 4, ROWB = 64;
  // 16-byte slot s of row r; the XOR makes the four 16-lane groups of ds_read_b128 hit 16 distinct slots
  static __device__ __forceinline__ int off(int r, int s) { return r * ROWB + ((s ^ ((4 - ((r >> 2) & 3)) & 3)) << 4); }
  typedef bf16x8 frag;
  static __device__ __forceinline__ frag load(const char* tile, int r, int g) {
    return *reinterpret_cast<const frag*>(tile + off(r, g));
  }
  static __device__ __forceinline__ void mma(f32x4& acc, const frag& a, const frag& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
};
template <> struct Tile<float> {
  static constexpr int VEC = 4, SEGS = 8, ROWB = 144;  // 128 B of data + 16 B pad
  static __device__ __forceinline__ int off(int r, int s) { return r * ROWB + (s << 4); }
  struct frag { float4 lo, hi; };
  static __device__ __forceinline__ frag load(const char* tile, int r, int g) {
    frag f;
    f.lo = *reinterpret_cast<const float4*>(tile + off(r, 2 * g));
    f.hi = *reinterpret_cast<const float4*>(tile + off(r, 2 * g + 1));
    return f;
  }
  // lane group g holds k = 8g..8g+7; MFMA step s consumes element s of every group (A and B agree on k)
  static __device__ __forceinline__ void mma(f32x4& acc, const frag& a, const frag& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.x, b.lo.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.y, b.lo.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.z, b.lo.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo.w, b.lo.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.x, b.hi.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.y, b.hi.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.z, b.hi.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi.w, b.hi.w, acc, 0, 0, 0);
  }
};

// mbits: ReLU bitmap of the 4 channels (bit e = keep channel e), or -1 (no mask)
__device__ __forceinline__ void store4(float* p, const f32x4& v, bool acc, int mbits = -1) {
  float4 o = make_float4(v[0], v[1], v[2], v[3]);
  if (acc) {
    const float4 old = *reinterpret_cast<const float4*>(p);
    o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
  }
  if (mbits >= 0) {
    o.x = (mbits & 1) ? o.x : 0.f; o.y = (mbits & 2) ? o.y : 0.f;
    o.z = (mbits & 4) ? o.z : 0.f; o.w = (mbits & 8) ? o.w : 0.f;
  }
  *reinterpret_cast<float4*>(p) = o;
}
__device__ __forceinline__ void store4(bf16_t* p, const f32x4& v, bool acc, int = -1) {   // (bitmaps: wide stores only)
  float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
  if (acc) {
    const bf16x4 old = *reinterpret_cast<const bf16x4*>(p);
    a0 += (float)old[0]; a1 += (float)old[1]; a2 += (float)old[2]; a3 += (float)old[3];
  }
  bf16x4 o;
  o[0] = (bf16_t)a0; o[1] = (bf16_t)a1; o[2] = (bf16_t)a2; o[3] = (bf16_t)a3;
  *reinterpret_cast<bf16x4*>(p) = o;
}

// bf16 epilogue with 16-byte stores (guide T21 for the 16x16 fragment): a lane holds 4 consecutive channels of one
// pixel per co fragment, so the natural store is 8 B and a wave-instruction scatters 16 x 32-B pieces -- the store
// tail of the output-heavy layers (conv_c, data gradients of conv_a) was issue-bound on them.  v_permlane16_swap
// between fragments i and i+1 (lanes g^1 are 16 apart) leaves every lane with 8 CONSECUTIVE channels:
//   g even: fragment i, channels 8*(g>>1)..+7        g odd: fragment i+1, channels 8*(g>>1)..+7
// half the store instructions, each writing 64 contiguous bytes per pixel.
__device__ __forceinline__ void swap16(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void store8_pair(bf16_t* pix, int co_base, int cout, f32x4 a, f32x4 b, int g, bool acc,
                                            const bf16x8& old, int mbits = -1) {
  float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int e = 0; e < 4; ++e) swap16(v[e], v[4 + e]);
  const int co = co_base + 16 * (g & 1) + 8 * (g >> 1);
  if (co >= cout) return;
  bf16_t* p = pix + co;
  if (acc) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)old[e];
  }
  if (mbits >= 0) {   // the stored tensor is a gradient w.r.t. a ReLU output: keep it where the activation was positive
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ((mbits >> e) & 1) ? v[e] : 0.f;
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
  *reinterpret_cast<bf16x8*>(p) = o;
}
// the 8 channels a lane will own after the swap, as they are in memory now (accumulate mode).  ALL of a tile's old values
// are fetched before the first store: interleaved, every load waited out a full round trip behind the previous store
// (the compiler cannot prove they do not alias), 8 serial trips per lane -- accumulate-mode data gradients ran at 1.9 TB/s
__device__ __forceinline__ bf16x8 load8_old(const bf16_t* pix, int co_base, int cout, int g) {
  const int co = co_base + 16 * (g & 1) + 8 * (g >> 1);
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
  return co < cout ? *reinterpret_cast<const bf16x8*>(pix + co) : z;
}
__device__ __forceinline__ bf16x8 load8_old(const float*, int, int, int) { return bf16x8{}; }
__device__ __forceinline__ void store8_pair(float*, int, int, f32x4, f32x4, int, bool, const bf16x8&, int = -1) {}   // f32 stores are 16 B already

// 16-lane row sum with DPP shifts (4 VALU ops; __shfl_xor goes through ds_bpermute): the total ends in lane 15 of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
  return v;
}


// Fused BatchNorm-backward reduce (bf16, 16-byte channel groups, an even number of co fragments per wave).  The pass
// that produces dA -- the data gradient into a BatchNorm'ed activation -- already holds the finished values in its
// accumulators, so its epilogue applies the ReLU mask, stores dz instead of dA, and leaves the two per-channel sums the
// BatchNorm backward needs (sum dz, sum dz * x_hat) as per-tile partial rows, exactly as the forward pass leaves its
// statistics.  The stand-alone reduce kernel (read dA, read y, read the mask, write dz) disappears; the epilogue reads
// y (and the mask source) for its own tile only.
// What makes or breaks it is memory-level parallelism.  A lane owns FM x FN/2 sixteen-byte groups of the tile; every global
// load is a BRANCH-FREE buffer load (rows past M re-read row M - 1, absent operands are zero-sized resources that read
// zeros: with `if (m < M) load` hipcc drains vmcnt behind every load) and ALL of a lane's loads -- y_bn, the old values of a
// += pass (ACC), the mask source (MSRC) -- are issued before the first use (with three operands: one fragment pair at a
// time, 48 VGPRs); the per-channel coefficients go through LDS once per tile (x_hat = y * ca + cb, mask = y * cs + ch > 0)
// instead of 32 registers per fragment pair.  The first version kept two pixel rows in flight, i.e. FM / 2 x FN / 2 serial
// round trips plus FN / 2 for the coefficients per tile: +24 .. +78 us on layers whose reduce kernel takes 17 .. 45.
// `red` = (WM x BN x 2 + 4 x BN) floats of LDS (aliases the ring; the caller has drained it).
template <int FM, int FN, int BM, int BN, int WM, int WN, bool ACC, bool MSRC>
__device__ __forceinline__ void epilogue_bn_bwd(const ConvK& k, const f32x4 (&acc)[FN][FM], float* red, int mt, int nt,
                                                int wm, int wn, int lane, int tid) {
  const int l15 = lane & 15, g = lane >> 4;
  bf16_t* __restrict__ yp = static_cast<bf16_t*>(k.y);
  const bool relu_y = k.bn_relu && !MSRC;                       // mask from y * scale + shift > 0
  const int co_w = nt * BN + wn * (BN / WN);
  const __amdgpu_buffer_rsrc_t r_old = sfk_make_rsrc(k.y, (ACC || MSRC) && k.accumulate ? k.ybytes : 0u);
  const __amdgpu_buffer_rsrc_t r_y = sfk_make_rsrc(k.bn_y, k.bn_ybytes);
  const __amdgpu_buffer_rsrc_t r_m = sfk_make_rsrc(k.bn_mask, MSRC ? k.bn_mbytes : 0u);
  // ---- coefficients of this tile's BN channels -> LDS (one thread per channel)
  float* coef = red + WM * BN * 2;                              // [4][BN]: ca | cb | cs | ch
  if (tid < BN) {
    const int co = nt * BN + tid;
    const int cc = co < k.cout ? co : 0;
    const float is = k.bn_invstd[cc], mu = k.bn_mean[cc];
    float cs = 0.f, ch = 1.f;                                   // no mask from y: y * 0 + 1 > 0
    if (relu_y) { cs = k.bn_scale[cc]; ch = k.bn_shift[cc]; }
    coef[tid] = is;
    coef[BN + tid] = -mu * is;
    coef[2 * BN + tid] = cs;
    coef[3 * BN + tid] = ch;
  }
  // the pixel a row maps to (clamped to the last row: nothing is stored for rows past M, their sums are masked out)
  int64_t plin[FM];
  bool rok[FM];
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m0 = mt * BM + wm * (BM / WM) + 16 * j + l15;
    rok[j] = m0 < k.M;
    const int m = rok[j] ? m0 : k.M - 1;
    if (k.lin_out) {
      plin[j] = m;
    } else {
      uint32_t q1, rw_, q2, rh_, n_, rt_;
      k.drw.divmod((uint32_t)m, q1, rw_);
      k.drh.divmod(q1, q2, rh_);
      k.drt.divmod(q2, n_, rt_);
      const int to = (int)rt_ * k.ost + k.oot, ho = (int)rh_ * k.osh + k.ooh, wo = (int)rw_ * k.osw + k.oow;
      plin[j] = (((int64_t)n_ * k.yt + to) * k.yh + ho) * k.yw + wo;
    }
  }
  constexpr int PG = ((MSRC || ACC) ? 1 : FN / 2);       // fragment pairs whose loads are in flight together (VGPR budget: 168)
  bool synced = false;
#pragma unroll
  for (int p0 = 0; p0 < FN; p0 += 2 * PG) {
    bf16x8 yv[PG][FM], oldv[(ACC || MSRC) ? PG : 1][(ACC || MSRC) ? FM : 1], mv[MSRC ? PG : 1][MSRC ? FM : 1];
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const int p = p0 + 2 * q;
      const int co = co_w + 16 * (p + (g & 1)) + 8 * (g >> 1);
      const int cc = co < k.cout ? co : 0;
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        yv[q][j] = __builtin_bit_cast(bf16x8, sfk_buffer_load16(r_y, (uint32_t)((plin[j] * k.bn_yld + k.bn_yoff + cc) * 2)));
        if constexpr (ACC || MSRC)
          oldv[q][j] = __builtin_bit_cast(bf16x8, sfk_buffer_load16(r_old, (uint32_t)((plin[j] * k.yld + k.yoff + cc) * 2)));
        if constexpr (MSRC)
          mv[q][j] = __builtin_bit_cast(bf16x8, sfk_buffer_load16(r_m, (uint32_t)((plin[j] * k.bn_mld + k.bn_moff + cc) * 2)));
      }
    }
    if (!synced) {             // the coefficients are in LDS (and every wave is past the K loop's last fragment reads)
      __syncthreads();
      synced = true;
    }
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const int p = p0 + 2 * q;
      const int col0 = wn * (BN / WN) + 16 * (p + (g & 1)) + 8 * (g >> 1);     // this lane's 8 channels within the tile
      const int co = nt * BN + col0;
      const bool cok = co < k.cout;
      float ca[8], cb[8], cs[8], ch[8];
      {
        const float4 a0 = *reinterpret_cast<const float4*>(coef + col0), a1 = *reinterpret_cast<const float4*>(coef + col0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(coef + BN + col0), b1 = *reinterpret_cast<const float4*>(coef + BN + col0 + 4);
        const float4 c0 = *reinterpret_cast<const float4*>(coef + 2 * BN + col0), c1 = *reinterpret_cast<const float4*>(coef + 2 * BN + col0 + 4);
        const float4 d0 = *reinterpret_cast<const float4*>(coef + 3 * BN + col0), d1 = *reinterpret_cast<const float4*>(coef + 3 * BN + col0 + 4);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) { ca[e] = av[e]; cb[e] = bv[e]; cs[e] = cv[e]; ch[e] = dv[e]; }
      }
      float s1[8], s2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        float v[8] = {acc[p][j][0], acc[p][j][1], acc[p][j][2], acc[p][j][3],
                      acc[p + 1][j][0], acc[p + 1][j][1], acc[p + 1][j][2], acc[p + 1][j][3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) swap16(v[e], v[4 + e]);
        const bool live = rok[j] && cok;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float vv = v[e];
          if constexpr (ACC || MSRC) vv += (float)oldv[q][j][e];       // (no accumulate: zero-sized resource, + 0)
          const float yf = (float)yv[q][j][e];
          bool keep;
          if constexpr (MSRC) keep = (float)mv[q][j][e] > 0.f;
          else keep = yf * cs[e] + ch[e] > 0.f;
          // dz is what the BatchNorm backward sees: the value as it is STORED (bf16), masked
          const bf16_t dzb = (bf16_t)(keep ? vv : 0.f);
          const float dz = live ? (float)dzb : 0.f;
          o[e] = dzb;
          s1[e] += dz;
          s2[e] += dz * (yf * ca[e] + cb[e]);
        }
        if (live) *reinterpret_cast<bf16x8*>(yp + plin[j] * k.yld + k.yoff + co) = o;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = row16_sum(s1[e]), b = row16_sum(s2[e]);
        if (l15 == 15) {
          red[(wm * BN + col0 + e) * 2 + 0] = a;
          red[(wm * BN + col0 + e) * 2 + 1] = b;
        }
      }
    }
  }
  __syncthreads();
  if (tid < BN) {
    const int co = nt * BN + tid;
    if (co < k.cout) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WM; ++w_) {
        a += red[(w_ * BN + tid) * 2 + 0];
        b += red[(w_ * BN + tid) * 2 + 1];
      }
      float* o = k.bn_parts + ((int64_t)mt * k.cout + co) * 2;
      o[0] = a;
      o[1] = b;
    }
  }
}

// The bitmap flavour of the above (bnb.y_bn == NULL: the data gradient that finishes the output gradient of a block with a
// fused tail -- dX = (dX + dY W^T) * bit, partial rows (sum dz, 0)) with ALL of a lane's loads in flight at once: the old
// rows and the bitmap bytes of its FM x FN/2 sixteen-byte stores (40 VGPRs beside the accumulators).  The general routine
// keeps two pixel rows in flight (it also carries y_bn, the mask source and four coefficient vectors), i.e. FM / 2 x FN / 2
// serial round trips per tile -- on the fast pathway's conv_a layers, whose K loop is 1..6 steps, that WAS the kernel:
// 156 us for a pass whose bytes take 59.
template <int FM, int FN, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void epilogue_bits_sum(const ConvK& k, const f32x4 (&acc)[FN][FM], float* red, int mt, int nt,
                                                  int wm, int wn, int lane, int tid) {
  const int l15 = lane & 15, g = lane >> 4;
  bf16_t* __restrict__ yp = static_cast<bf16_t*>(k.y);
  const int co_w = nt * BN + wn * (BN / WN);
  const __amdgpu_buffer_rsrc_t r_old = sfk_make_rsrc(k.y, k.accumulate ? k.ybytes : 0u);
  const __amdgpu_buffer_rsrc_t r_b = sfk_make_rsrc(k.obits, k.obits_bytes);
  int64_t plin[FM];
  bool rok[FM];
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m0 = mt * BM + wm * (BM / WM) + 16 * j + l15;
    rok[j] = m0 < k.M;
    const int m = rok[j] ? m0 : k.M - 1;
    if (k.lin_out) {
      plin[j] = m;
    } else {
      uint32_t q1, rw_, q2, rh_, n_, rt_;
      k.drw.divmod((uint32_t)m, q1, rw_);
      k.drh.divmod(q1, q2, rh_);
      k.drt.divmod(q2, n_, rt_);
      const int to = (int)rt_ * k.ost + k.oot, ho = (int)rh_ * k.osh + k.ooh, wo = (int)rw_ * k.osw + k.oow;
      plin[j] = (((int64_t)n_ * k.yt + to) * k.yh + ho) * k.yw + wo;
    }
  }
  bf16x8 oldv[FN / 2][FM];
  uint32_t mbyte[FN / 2][FM];
#pragma unroll
  for (int p = 0; p < FN; p += 2) {
    const int co = co_w + 16 * (p + (g & 1)) + 8 * (g >> 1);
    const int cc = co < k.cout ? co : 0;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      oldv[p / 2][j] = __builtin_bit_cast(bf16x8, sfk_buffer_load16(r_old, (uint32_t)((plin[j] * k.yld + k.yoff + cc) * 2)));
      mbyte[p / 2][j] = (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(r_b, (int)(plin[j] * (k.cout >> 3) + (cc >> 3)), 0, 0);
    }
  }
#pragma unroll
  for (int p = 0; p < FN; p += 2) {
    const int co = co_w + 16 * (p + (g & 1)) + 8 * (g >> 1);
    const bool cok = co < k.cout;
    float s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = 0.f;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      float v[8] = {acc[p][j][0], acc[p][j][1], acc[p][j][2], acc[p][j][3],
                    acc[p + 1][j][0], acc[p + 1][j][1], acc[p + 1][j][2], acc[p + 1][j][3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) swap16(v[e], v[4 + e]);
      const bool live = rok[j] && cok;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float vv = v[e] + (float)oldv[p / 2][j][e];             // (no accumulate: zero-sized resource, + 0)
        const bf16_t dzb = (bf16_t)(((mbyte[p / 2][j] >> e) & 1u) ? vv : 0.f);   // the value as it is STORED, masked
        o[e] = dzb;
        s1[e] += live ? (float)dzb : 0.f;
      }
      if (live) *reinterpret_cast<bf16x8*>(yp + plin[j] * k.yld + k.yoff + co) = o;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = row16_sum(s1[e]);
      if (l15 == 15) {
        const int col = wn * (BN / WN) + 16 * (p + (g & 1)) + 8 * (g >> 1) + e;
        red[(wm * BN + col) * 2 + 0] = a;
        red[(wm * BN + col) * 2 + 1] = 0.f;
      }
    }
  }
  __syncthreads();
  if (tid < BN) {
    const int co = nt * BN + tid;
    if (co < k.cout) {
      float a = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WM; ++w_) a += red[(w_ * BN + tid) * 2 + 0];
      float* o = k.bn_parts + ((int64_t)mt * k.cout + co) * 2;
      o[0] = a;
      o[1] = 0.f;
    }
  }
}

// Fused output transform (EPI == 3, sfk_conv_epilogue): BatchNorm scale / shift, shortcut, ReLU and its bitmap on the
// accumulators -- the conv output of a bottleneck's conv_c never reaches HBM; also the "+ bias" of the second
// data-gradient pass of that tail.  Rows are the output pixels (lin_out).  All loads of a fragment row group (old values,
// shortcut) are issued before the first store (a load behind a store waits out a round trip: see load8_old).
// shortcut rows of one tile, fetched BEFORE the K loop by the kernels that can afford the registers (a conv_c has 2..4
// K-steps: with the shortcut read only in the epilogue a tile's life is DMA round trip + shortcut round trip + stores,
// strictly one after the other -- 279 us for a layer whose bytes take 190 us)
template <int FM, int FN> struct ResPre { bf16x8 v[FM][(FN + 1) / 2]; };

template <typename T, int FM, int FN, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void prefetch_res(const ConvK& k, ResPre<FM, FN>& pre, int mt, int nt, int wm, int wn, int lane) {
  if constexpr (sizeof(T) == 2 && (FN % 2) == 0) {
    const int l15 = lane & 15, g = lane >> 4;
    const int co_w = nt * BN + wn * (BN / WN);
    const __amdgpu_buffer_rsrc_t r_res = sfk_make_rsrc(k.ep_res, k.ep_res ? k.ep_rbytes : 0u);
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = mt * BM + wm * (BM / WM) + 16 * j + l15;
      const int64_t row = m < k.M ? m : k.M - 1;
#pragma unroll
      for (int p = 0; p < FN; p += 2) {
        const int co = co_w + 16 * (p + (g & 1)) + 8 * (g >> 1);
        const int cc = co < k.cout ? co : 0;
        pre.v[j][p / 2] = __builtin_bit_cast(bf16x8, sfk_buffer_load16(r_res, (uint32_t)((row * k.ep_rld + k.ep_roff + cc) * 2)));
      }
    }
  }
}

template <typename T, int FM, int FN, int BM, int BN, int WM, int WN, bool PRE = false>
__device__ __forceinline__ void epilogue_fused(const ConvK& k, const f32x4 (&acc)[FN][FM], int mt, int nt, int wm, int wn,
                                               int lane, const ResPre<FM, FN>* pre = nullptr) {
  const int l15 = lane & 15, g = lane >> 4;
  T* __restrict__ yp = static_cast<T*>(k.y);
  const T* __restrict__ rp = static_cast<const T*>(k.ep_res);
  const int co_w = nt * BN + wn * (BN / WN);
  int64_t rows[FM];
  bool rok[FM];
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = mt * BM + wm * (BM / WM) + 16 * j + l15;
    rok[j] = m < k.M;
    rows[j] = rok[j] ? m : k.M - 1;          // rows past M re-read the last row (branch-free loads), nothing is stored
  }
  if constexpr (sizeof(T) == 2 && (FN % 2) == 0) {
    // bf16, 16-byte channel groups: fragment pairs, 8 consecutive channels per lane after the permlane swap
#pragma unroll
    for (int p = 0; p < FN; p += 2) {
      const int co = co_w + 16 * (p + (g & 1)) + 8 * (g >> 1);
      const bool cok = co < k.cout;
      const int cc = cok ? co : 0;
      // per-channel coefficients: 16-byte BUFFER loads -- an absent vector is a zero-sized resource that reads zeros, so
      // nothing is loaded under a branch (a conditional load makes hipcc drain vmcnt behind it: eight serial round trips
      // per tile measured +53 us on a 411 MB map; one scalar load per channel, 48 of them, +66 us)
      float sc[8], sh[8], rs[8];
      {
        const uint32_t cb = (uint32_t)cc * 4u, nb = (uint32_t)k.cout * 4u;
        const __amdgpu_buffer_rsrc_t r_sc = sfk_make_rsrc(k.ep_scale, k.ep_scale ? nb : 0u);
        const __amdgpu_buffer_rsrc_t r_sh = sfk_make_rsrc(k.ep_shift, k.ep_shift ? nb : 0u);
        const __amdgpu_buffer_rsrc_t r_rs = sfk_make_rsrc(k.ep_rscale, (rp && k.ep_rscale) ? nb : 0u);
        const __amdgpu_buffer_rsrc_t r_rh = sfk_make_rsrc(k.ep_rshift, (rp && k.ep_rshift) ? nb : 0u);
        const uint4 a0 = sfk_buffer_load16(r_sc, cb), a1 = sfk_buffer_load16(r_sc, cb + 16);
        const uint4 b0 = sfk_buffer_load16(r_sh, cb), b1 = sfk_buffer_load16(r_sh, cb + 16);
        const uint4 c0 = sfk_buffer_load16(r_rh, cb), c1 = sfk_buffer_load16(r_rh, cb + 16);
        const uint4 d0 = sfk_buffer_load16(r_rs, cb), d1 = sfk_buffer_load16(r_rs, cb + 16);
        const uint32_t av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const uint32_t bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const uint32_t cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const uint32_t dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
        const bool has_sc = k.ep_scale != nullptr, has_rs = rp && k.ep_rscale;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sc[e] = has_sc ? __uint_as_float(av[e]) : 1.f;
          sh[e] = __uint_as_float(bv[e]) + __uint_as_float(cv[e]);
          rs[e] = has_rs ? __uint_as_float(dv[e]) : 1.f;
        }
      }
      // old values (+=) and the shortcut: buffer loads as well; absent -> zero-sized resource -> zeros
      const __amdgpu_buffer_rsrc_t r_old = sfk_make_rsrc(k.y, k.accumulate ? k.ybytes : 0u);
      const __amdgpu_buffer_rsrc_t r_res = sfk_make_rsrc(k.ep_res, rp ? k.ep_rbytes : 0u);
      // two pixel rows at a time: the 256x128 tile sits at its 128-VGPR cap, 2 x (old, shortcut) x 16 B is what fits
      constexpr int JB = FM >= 2 ? 2 : 1;
#pragma unroll
      for (int j0 = 0; j0 < FM; j0 += JB) {
        bf16x8 oldv[JB], resv[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          const int j = j0 + jj;
          const uint4 o4 = sfk_buffer_load16(r_old, (uint32_t)((rows[j] * k.yld + k.yoff + cc) * 2));
          oldv[jj] = __builtin_bit_cast(bf16x8, o4);
          if constexpr (PRE) {
            resv[jj] = pre->v[j][p / 2];
          } else {
            const uint4 r4 = sfk_buffer_load16(r_res, (uint32_t)((rows[j] * k.ep_rld + k.ep_roff + cc) * 2));
            resv[jj] = __builtin_bit_cast(bf16x8, r4);
          }
        }
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          const int j = j0 + jj;
          float v[8] = {acc[p][j][0], acc[p][j][1], acc[p][j][2], acc[p][j][3],
                        acc[p + 1][j][0], acc[p + 1][j][1], acc[p + 1][j][2], acc[p + 1][j][3]};
#pragma unroll
          for (int e = 0; e < 4; ++e) swap16(v[e], v[4 + e]);
          if (!rok[j] || !cok) continue;
          uint32_t bits = 0;
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = v[e] * sc[e] + sh[e] + (float)oldv[jj][e] + (float)resv[jj][e] * rs[e];
            if (k.ep_relu) {
              bits |= (f > 0.f ? 1u : 0u) << e;
              f = f > 0.f ? f : 0.f;
            }
            o[e] = (bf16_t)f;
          }
          *reinterpret_cast<bf16x8*>(yp + rows[j] * k.yld + k.yoff + co) = o;
          if (k.ep_bits) k.ep_bits[rows[j] * (k.cout >> 3) + (co >> 3)] = (uint8_t)bits;
        }
      }
    }
  } else {
    // f32 (16-byte groups of 4 channels), and bf16 tiles with one co fragment (scale / shift / += only there)
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int co = co_w + 16 * i + 4 * g;
      const bool cok = co < k.cout;
      const int cc = cok ? co : 0;
      // coefficients: one 16-byte buffer load per vector, nothing under a branch (see the 16-byte path above: a conditional
      // load makes hipcc drain vmcnt behind it -- sixteen serial round trips per tile here, and the += + bias pass of the
      // fast pathway's 8-channel maps, a ONE K-step layer, took 85 us for 154 MB)
      float sc[4], sh[4], rs[4], rh[4];
      {
        const uint32_t cb = (uint32_t)cc * 4u, nb = (uint32_t)k.cout * 4u;
        const bool has_sc = k.ep_scale != nullptr, has_rs = rp && k.ep_rscale;
        const uint4 a0 = sfk_buffer_load16(sfk_make_rsrc(k.ep_scale, has_sc ? nb : 0u), cb);
        const uint4 b0 = sfk_buffer_load16(sfk_make_rsrc(k.ep_shift, k.ep_shift ? nb : 0u), cb);
        const uint4 c0 = sfk_buffer_load16(sfk_make_rsrc(k.ep_rscale, has_rs ? nb : 0u), cb);
        const uint4 d0 = sfk_buffer_load16(sfk_make_rsrc(k.ep_rshift, (rp && k.ep_rshift) ? nb : 0u), cb);
        const uint32_t av[4] = {a0.x, a0.y, a0.z, a0.w}, bv[4] = {b0.x, b0.y, b0.z, b0.w};
        const uint32_t cv[4] = {c0.x, c0.y, c0.z, c0.w}, dv[4] = {d0.x, d0.y, d0.z, d0.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sc[e] = has_sc ? __uint_as_float(av[e]) : 1.f;
          sh[e] = __uint_as_float(bv[e]);
          rs[e] = has_rs ? __uint_as_float(cv[e]) : 1.f;
          rh[e] = __uint_as_float(dv[e]);
        }
      }
      float oldv[FM][4], resv[FM][4];
      if constexpr (sizeof(T) == 2) {
        // one 8-byte buffer load per row (zero-sized resource = zeros when there is nothing to add); element-wise 2-byte
        // loads made the += pass of the narrowest layers run at 1.2 TB/s
        const __amdgpu_buffer_rsrc_t r_old = sfk_make_rsrc(k.y, k.accumulate ? k.ybytes : 0u);
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const auto o2 = __builtin_amdgcn_raw_buffer_load_b64(r_old, (int)((rows[j] * k.yld + k.yoff + cc) * 2), 0, 0);
          const bf16x4 ob = __builtin_bit_cast(bf16x4, o2);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            oldv[j][e] = (float)ob[e];
            resv[j][e] = 0.f;                      // (no shortcut on this path: validate())
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < FM; ++j) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            oldv[j][e] = k.accumulate ? (float)yp[rows[j] * k.yld + k.yoff + cc + e] : 0.f;
            resv[j][e] = rp ? (float)rp[rows[j] * k.ep_rld + k.ep_roff + cc + e] : 0.f;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        if (!rok[j] || !cok) continue;
        uint32_t bits = 0;
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[e] = acc[i][j][e] * sc[e] + sh[e] + oldv[j][e];
          if (rp) f[e] += resv[j][e] * rs[e] + rh[e];
          if (k.ep_relu) {
            bits |= (f[e] > 0.f ? 1u : 0u) << e;
            f[e] = f[e] > 0.f ? f[e] : 0.f;
          }
        }
        T* op = yp + rows[j] * k.yld + k.yoff + co;
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<float4*>(op) = make_float4(f[0], f[1], f[2], f[3]);
          if (k.ep_bits) k.ep_bits[rows[j] * (k.cout >> 2) + (co >> 2)] = (uint8_t)bits;
        } else {
          bf16x4 o;
          o[0] = (bf16_t)f[0]; o[1] = (bf16_t)f[1]; o[2] = (bf16_t)f[2]; o[3] = (bf16_t)f[3];
          *reinterpret_cast<bf16x4*>(op) = o;
        }
      }
    }
  }
}

// The plain epilogue of BOTH conv kernels (one copy: the register-staged and the LDS-DMA kernel differ only in how the
// tiles reach LDS): channels-last stores of the accumulators -- 4 consecutive co per lane per fragment, 8 after the
// permlane swap (`wide`) -- with the optional += of `accumulate`, the optional output ReLU bitmap (EPI == 2), and the
// per-tile BatchNorm partial statistics.
template <typename T, int EPI, int FM, int FN, int BM, int BN, int WM, int WN, bool RING_BUSY>
__device__ __forceinline__ void epilogue_plain(const ConvK& k, const f32x4 (&acc)[FN][FM], float* red, int mt, int nt,
                                               int wm, int wn, int lane, int tid) {
  const int l15 = lane & 15, g = lane >> 4;
  T* __restrict__ yp = static_cast<T*>(k.y);
  const int co_w = nt * BN + wn * (BN / WN);
  const bool wide = sizeof(T) == 2 && (FN % 2) == 0 && k.wide_store;
  int64_t poffs[FM];
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = mt * BM + wm * (BM / WM) + 16 * j + l15;
    if (k.lin_out) {                 // the rows ARE the output pixels (stride-1 passes): no coordinates needed
      poffs[j] = (int64_t)m * k.yld + k.yoff;
    } else {
      uint32_t q1, rw_, q2, rh_, n_, rt_;
      k.drw.divmod((uint32_t)m, q1, rw_);
      k.drh.divmod(q1, q2, rh_);
      k.drt.divmod(q2, n_, rt_);
      const int to = (int)rt_ * k.ost + k.oot, ho = (int)rh_ * k.osh + k.ooh, wo = (int)rw_ * k.osw + k.oow;
      poffs[j] = ((((int64_t)n_ * k.yt + to) * k.yh + ho) * k.yw + wo) * k.yld + k.yoff;
    }
  }
  // old rows of a += pass: BRANCH-FREE buffer loads, all in flight before the first store (rows past M re-read the last
  // row, channels past cout channel 0, a plain pass reads zeros from a zero-sized resource -- nothing of those is stored):
  // with the loads under `if (m < M)` hipcc drains vmcnt behind each row, FM serial round trips per tile
  bf16x8 oldv[FM][(FN + 1) / 2];
  if constexpr (sizeof(T) == 2) {
    if (wide && k.accumulate && k.ybig) {            // a map of 4 GiB or more: 64-bit pointers (rare; the slow way)
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int m = mt * BM + wm * (BM / WM) + 16 * j + l15;
        if (m < k.M) {
#pragma unroll
          for (int i = 0; i < FN; i += 2) oldv[j][i / 2] = load8_old(yp + poffs[j], co_w + 16 * i, k.cout, g);
        }
      }
    } else if (wide) {
      const __amdgpu_buffer_rsrc_t r_old = sfk_make_rsrc(k.y, k.accumulate ? k.ybytes : 0u);
      const int64_t plast = k.lin_out ? ((int64_t)(k.M - 1) * k.yld + k.yoff) : poffs[0];
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int m = mt * BM + wm * (BM / WM) + 16 * j + l15;
        // (gathered output rows past M decode to some in-range pixel of the map already: poffs[j] is a valid address)
        const int64_t po = (k.lin_out && m >= k.M) ? plast : poffs[j];
#pragma unroll
        for (int i = 0; i < FN; i += 2) {
          const int co = co_w + 16 * i + 16 * (g & 1) + 8 * (g >> 1);
          oldv[j][i / 2] = __builtin_bit_cast(bf16x8, sfk_buffer_load16(r_old, (uint32_t)((po + (co < k.cout ? co : 0)) * 2)));
        }
      }
    }
  }
  // ReLU bitmap of the output (lin_out only: the pixel index is the row): byte [pixel][co / VEC], fetched with the old
  // values, before the first store
  static_assert(EPI != 2 || FN <= 8, "one packed bitmap word per fragment row");
  uint32_t mb[EPI == 2 ? FM : 1];            // byte i/2 = the bitmap byte of this lane's store i (packed: registers)
  bool masked = false;
  if constexpr (EPI == 2) {
    masked = wide && k.obits;
    if (masked) {
      // the wave's FN*16 channels of a pixel are FN*2 consecutive bitmap bytes: one aligned 4- or 8-byte load per pixel
      // row (the 4 lanes of a pixel fetch the same word) when the channel count allows, else one byte load per store
      const bool word = (FN == 4 || FN == 2) && (k.cout % (FN * 16)) == 0;
      const int sh = 8 * (2 * (g & 1) + (g >> 1));          // this lane's byte within each 4-byte group
      if (word) {
        // branch-free (rows past M re-read the last row; nothing is stored for them): a load under a branch makes hipcc
        // drain vmcnt after it, one serial round trip per pixel row
        uint2 w[FM];
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const int m = mt * BM + wm * (BM / WM) + 16 * j + l15;
          const uint8_t* bp = k.obits + (int64_t)(m < k.M ? m : k.M - 1) * (k.cout >> 3) + (co_w >> 3);
          if constexpr (FN == 4) w[j] = *reinterpret_cast<const uint2*>(bp);
          else w[j] = make_uint2(*reinterpret_cast<const uint32_t*>(bp), 0u);
        }
#pragma unroll
        for (int j = 0; j < FM; ++j) mb[j] = ((w[j].x >> sh) & 255u) | (((w[j].y >> sh) & 255u) << 8);
      } else {
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const int m = mt * BM + wm * (BM / WM) + 16 * j + l15;
          mb[j] = 0;
          if (m < k.M) {
#pragma unroll
            for (int i = 0; i < FN; i += 2) {
              const int co = co_w + 16 * i + 16 * (g & 1) + 8 * (g >> 1);
              if (co < k.cout) mb[j] |= (uint32_t)k.obits[(int64_t)m * (k.cout >> 3) + (co >> 3)] << (8 * (i / 2));
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = mt * BM + wm * (BM / WM) + 16 * j + l15;
    if (m < k.M) {
      const int64_t poff = poffs[j];
      if (wide) {
#pragma unroll
        for (int i = 0; i < FN; i += 2)
          store8_pair(yp + poff, co_w + 16 * i, k.cout, acc[i][j], acc[(i + 1) % FN][j], g, k.accumulate != 0, oldv[j][i / 2],
                      (EPI == 2 && masked) ? (int)((mb[EPI == 2 ? j : 0] >> (8 * (i / 2))) & 255u) : -1);
      } else {
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          const int co = co_w + 16 * i + 4 * g;
          if (co < k.cout) {
            int mbits = -1;
            if constexpr (EPI == 2 && sizeof(T) == 4) {
              if (k.obits) mbits = k.obits[(int64_t)m * (k.cout >> 2) + (co >> 2)] & 15;
            }
            store4(yp + poff + co, acc[i][j], k.accumulate != 0, mbits);
          }
        }
      }
    }
  }

  // ---- BatchNorm partial statistics of this tile (rows past M accumulated zeros, so they add nothing)
  if (k.stats) {
    // red = [WM][BN][2] floats, aliases the staging ring: the register-staged kernel still has LDS reads of its last
    // K-step in flight (RING_BUSY), the DMA kernel has drained and met at a barrier already
    if constexpr (RING_BUSY) __syncthreads();
#pragma unroll
    for (int i = 0; i < FN; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const float v = acc[i][j][r];
          s1 += v;
          s2 += v * v;
        }
        s1 = row16_sum(s1);
        s2 = row16_sum(s2);
        if (l15 == 15) {
          const int col = wn * (BN / WN) + 16 * i + 4 * g + r;
          red[(wm * BN + col) * 2 + 0] = s1;
          red[(wm * BN + col) * 2 + 1] = s2;
        }
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int co = nt * BN + tid;
      if (co < k.cout) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < WM; ++w_) {
          s1 += red[(w_ * BN + tid) * 2 + 0];
          s2 += red[(w_ * BN + tid) * 2 + 1];
        }
        float* o = k.stats + ((int64_t)mt * k.cout + co) * 2;
        o[0] = s1;
        o[1] = s2;
      }
    }
  }
}

}  // namespace sfk_igemm
