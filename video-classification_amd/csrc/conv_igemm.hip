// sfk_conv_igemm: Conv3d forward / data-gradient as an implicit GEMM on gfx950 MFMA.
//
//   D[co][pixel] = sum_{tap, ci} W[co][tap][ci] * X[gather(pixel, tap)][ci]
//
// MFMA orientation: A = filter tile (rows = co), B = gathered pixel tile (cols = pixel), so a lane ends up
// holding 4 CONSECUTIVE output channels of one pixel -> 8-byte channels-last stores, and the per-channel
// BatchNorm partial sums are a 16-lane butterfly.
// Both operands are K(=ci)-contiguous in HBM (channels-last activations, [co][tap][ci] filters), so tiles are
// staged HBM -> registers (16 B / lane, zero-filled for padding pixels and ragged channels) -> LDS, double
// buffered: the loads of K-step i+1 are in flight while step i runs on the matrix cores.
//   bf16: v_mfma_f32_16x16x32_bf16, LDS rows of 64 B XOR-swizzled so ds_read_b128 is conflict-free
//   f32 : v_mfma_f32_16x16x4_f32 x8 per K-step (bit-exact fp32 fma chain) -- the parity precision
#include "sfk_common.h"
#include <stdlib.h>

namespace {

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

// EPI: 0 plain epilogue, 1 fused BatchNorm-backward reduce (bnb; 5 = its bitmap flavour, 6 = += pass, 7 = mask source), 2 output ReLU bitmap (out_relu_bits), 3 fused output
// transform (sfk_conv_epilogue) -- own
// instantiations: the extra epilogue state must not cost the plain kernel registers (the 256x128 tile sits at 128 VGPRs)
template <typename T, int BM, int BN, int WM, int WN, bool SHORTK, int EPI = 0>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 3 : 1)) void conv_igemm_kernel(const ConvK k) {
  using TL = Tile<T>;
  constexpr int VEC = TL::VEC, SEGS = TL::SEGS, ROWB = TL::ROWB;
  constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
  constexpr int RPI = 256 / SEGS;              // tile rows covered by one pass of the 256 threads
  constexpr int XL = BM / RPI;                 // gathered-pixel loads per thread per K-step
  constexpr int WL = (BN + RPI - 1) / RPI;     // filter loads per thread per K-step
  constexpr int BUF = (BM + BN) * ROWB;
  static_assert(WM * WN == 4 && BM % RPI == 0, "tile shape");
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  // per-tap tables (+ a sentinel entry that gathers nothing, for the ragged K tail)
  __shared__ sfk_tap s_taps[SFK_MAX_TAPS + 1];
  __shared__ int s_xdelta[SFK_MAX_TAPS + 1];   // element offset of the tap inside the input map
  __shared__ int s_woff[SFK_MAX_TAPS + 1];     // widx * cin
  if (threadIdx.x <= SFK_MAX_TAPS) {
    sfk_tap t = k.taps[threadIdx.x < SFK_MAX_TAPS ? threadIdx.x : 0];
    if ((int)threadIdx.x >= k.ntaps) { t.dt = -128; t.dh = 0; t.dw = 0; t.widx = 0; }
    s_taps[threadIdx.x] = t;
    s_xdelta[threadIdx.x] = (((int)t.dt * k.xh + (int)t.dh) * k.xw + (int)t.dw) * k.xld * (int)sizeof(T);   // bytes
    s_woff[threadIdx.x] = (int)t.widx * k.cin * (int)sizeof(T);
  }
  __syncthreads();

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int l15 = lane & 15, g = lane >> 4;

  // XCD-aware tile order: blocks b, b+8, ... share an L2; give them the SAME pixel tile (all its co tiles)
  int mt, nt;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = b & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    nt = logical % k.ntiles;
    mt = logical / k.ntiles;
  }

  // ---- per-thread staging coordinates (fixed over the K loop): pixel coordinates for the bounds test and the
  // element offset of the un-shifted pixel; a tap only ADDS its table delta.
  const int seg = tid % SEGS, row0 = tid / SEGS;
  int xtb[XL], xhb[XL], xwb[XL];
  uint32_t xbase[XL];   // byte offset of the un-shifted pixel inside the input buffer
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int m = mt * BM + row0 + i * RPI;
    uint32_t q1, rw_, q2, rh_, n_, rt_;
    k.drw.divmod((uint32_t)m, q1, rw_);
    k.drh.divmod(q1, q2, rh_);
    k.drt.divmod(q2, n_, rt_);
    xtb[i] = (m < k.M) ? (int)rt_ * k.gst : -(1 << 28);  // rows past M gather nothing
    xhb[i] = (int)rh_ * k.gsh;
    xwb[i] = (int)rw_ * k.gsw;
    xbase[i] = (uint32_t)((((((int64_t)n_ * k.xt + (int)rt_ * k.gst) * k.xh + xhb[i]) * k.xw + xwb[i]) * k.xld + k.xoff) *
                          (int64_t)sizeof(T));
  }
  uint32_t wbase[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    const int r = row0 + i * RPI;
    const int co = nt * BN + r;
    wbase[i] = (r < BN && co < k.cout) ? (uint32_t)(co * k.wtaps * k.cin) * (uint32_t)sizeof(T) : SFK_OOB;
  }
  const __amdgpu_buffer_rsrc_t xrs = sfk_make_rsrc(k.x, k.xbytes);
  const __amdgpu_buffer_rsrc_t wrs = sfk_make_rsrc(k.w, k.wbytes);

  // K is the flattened (tap, channel) axis cut into 16-byte segments; a K-step takes SEGS consecutive segments, so
  // narrow layers (cin < 32) pack several taps into one MFMA step instead of padding each tap to 32 channels.
  // Loads are branch-free buffer loads: a slot that is conv padding / past the tile gets offset SFK_OOB and reads zeros.
  struct Stage { uint4 x[XL]; uint4 w[WL]; };
  auto gload = [&](int step, Stage& st) {
    const uint32_t q = (uint32_t)(step * SEGS + seg);
    uint32_t tap, cseg;
    k.dspt.divmod(q, tap, cseg);
    const bool cok = tap < (uint32_t)k.ntaps;
    const int ti_ = cok ? (int)tap : SFK_MAX_TAPS;
    const sfk_tap tp = s_taps[ti_];
    const uint32_t xd = (uint32_t)(s_xdelta[ti_] + (int)cseg * 16);
    const uint32_t wd = (uint32_t)(s_woff[ti_] + (int)cseg * 16);
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int ti = xtb[i] + tp.dt, hi = xhb[i] + tp.dh, wi = xwb[i] + tp.dw;
      const bool ok = cok && (unsigned)ti < (unsigned)k.xt && (unsigned)hi < (unsigned)k.xh &&
                      (unsigned)wi < (unsigned)k.xw;
      st.x[i] = sfk_buffer_load16(xrs, ok ? xbase[i] + xd : SFK_OOB);
    }
#pragma unroll
    for (int i = 0; i < WL; ++i)
      st.w[i] = sfk_buffer_load16(wrs, (cok && wbase[i] != SFK_OOB) ? wbase[i] + wd : SFK_OOB);
  };
  auto lstore = [&](int buf, const Stage& st) {
    char* xs = smem + buf * BUF;
    char* ws = xs + BM * ROWB;
#pragma unroll
    for (int i = 0; i < XL; ++i) *reinterpret_cast<uint4*>(xs + TL::off(row0 + i * RPI, seg)) = st.x[i];
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int r = row0 + i * RPI;
      if ((BN % RPI == 0) || r < BN) *reinterpret_cast<uint4*>(ws + TL::off(r, seg)) = st.w[i];
    }
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char* xs = smem + buf * BUF;
    const char* ws = xs + BM * ROWB;
    typename TL::frag a[FN], b[FM];
#pragma unroll
    for (int i = 0; i < FN; ++i) a[i] = TL::load(ws, wn * (BN / WN) + 16 * i + l15, g);
#pragma unroll
    for (int j = 0; j < FM; ++j) b[j] = TL::load(xs, wm * (BM / WM) + 16 * j + l15, g);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) TL::mma(acc[i][j], a[i], b[j]);
  };

  // Software pipeline, two K-steps deep in registers: while step `it` runs on the matrix cores out of LDS buffer
  // it&1, the tile of step it+1 waits in one register set and the loads of step it+2 are in flight in the other.
  // K-steps past the end gather nothing (every slot is SFK_OOB -> zeros), so the body runs unconditionally for an even
  // number of steps: no guards around the loads means the compiler counts them exactly (vmcnt(N), never a full drain).
  // Layers with at most k.kshort K-steps run the SHORTK instantiation instead: an exact-count loop (its guards cost the
  // steady loop its exact vmcnt, which does not matter for a handful of steps); the rounded-up look-ahead loop spent up
  // to 3/4 of a short layer's gather instructions on steps that load nothing.  Two instantiations, not one kernel
  // with both loops: together they spill.
  Stage r0, r1;
  if constexpr (SHORTK) {
    // short K (1..5 steps: the fast pathway's 8..64-channel layers): exact step count, nothing loaded past the end
    gload(0, r0);
    lstore(0, r0);
    if (k.KC > 1) gload(1, r1);
    __syncthreads();
    int it = 0;
    for (; it + 3 < k.KC; it += 2) {
      gload(it + 2, r0);
      compute(0);
      lstore(1, r1);
      __syncthreads();
      gload(it + 3, r1);
      compute(1);
      lstore(0, r0);
      __syncthreads();
    }
    const int rem = k.KC - it;            // 1, 2 or 3 steps left; LDS buffer 0 holds step `it`, r1 step it+1
    if (rem == 1) {
      compute(0);
    } else if (rem == 2) {
      compute(0);
      lstore(1, r1);
      __syncthreads();
      compute(1);
    } else {
      gload(it + 2, r0);
      compute(0);
      lstore(1, r1);
      __syncthreads();
      compute(1);
      lstore(0, r0);
      __syncthreads();
      compute(0);
    }
  } else {
    const int nit2 = (k.KC + 1) & ~1;
    gload(0, r0);
    lstore(0, r0);
    gload(1, r1);
    __syncthreads();
    for (int it = 0; it < nit2; it += 2) {
      gload(it + 2, r0);
      compute(0);
      lstore(1, r1);
      __syncthreads();
      gload(it + 3, r1);
      compute(1);
      lstore(0, r0);
      __syncthreads();
    }
  }

  // ---- epilogue: channels-last stores (4 consecutive co per lane per fragment)
  if constexpr ((EPI == 1 || EPI >= 5) && sizeof(T) == 2 && (FN % 2) == 0) {   // fused BatchNorm-backward reduce: own instantiations
    __syncthreads();        // the partial sums go through LDS that aliases the ring
    if constexpr (EPI == 5) epilogue_bits_sum<FM, FN, BM, BN, WM, WN>(k, acc, reinterpret_cast<float*>(smem), mt, nt, wm, wn, lane, tid);
    else epilogue_bn_bwd<FM, FN, BM, BN, WM, WN, EPI == 6, EPI == 7>(k, acc, reinterpret_cast<float*>(smem), mt, nt, wm, wn, lane, tid);
    return;
  }
  if constexpr (EPI == 3) {
    epilogue_fused<T, FM, FN, BM, BN, WM, WN>(k, acc, mt, nt, wm, wn, lane);
    return;
  }
  epilogue_plain<T, EPI, FM, FN, BM, BN, WM, WN, true>(k, acc, reinterpret_cast<float*>(smem), mt, nt, wm, wn, lane, tid);
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 production kernel: the same implicit GEMM with LDS-DMA staging (`buffer_load_dwordx4 ... lds`).
//   * a wave-instruction writes 1 KiB = 16 tile rows x 64 B straight into LDS (dest = wave-uniform base + lane*16):
//     lane l fills physical slot l&3 of row l>>2, and FETCHES the logical K segment (l&3) ^ f(row) -- the XOR
//     swizzle of Tile<bf16>::off lives on the source side, the LDS image stays lane-linear (guide rule 21);
//   * every lane has its own source offset, so the conv gather (taps, strides) costs nothing extra, and a lane whose
//     pixel is padding / past the tile passes SFK_OOB: the buffer range check makes the DMA write ZEROS (verified on
//     MI355X, tools/probe/ldsdma_oob.hip);
//   * no staging VGPRs, no ds_write (the ~79 B/clk ds_write_b128 path was the LDS bottleneck of the register-staged
//     loop); 3-slot LDS ring, DMA of step it+2 in flight while step it runs, one counted s_waitcnt vmcnt(N) + one raw
//     s_barrier per K-step (never a full drain).
typedef __attribute__((address_space(3))) void lds_void_t;

// BMS: pixel rows a tile COMPUTES (= the tile stride over M); BM rows are staged.  BMS < BM (224 of 256, waves as 2 x 4 with
// 7 row fragments each) exists for the tile-count arithmetic: M = 50,176 (res4) is 196 tiles of 256 rows -- 0.77 of one
// generation on 256 CUs -- but 224 tiles of 224 rows; the 32 extra staged rows cost L2 -> LDS traffic an MFMA-bound layer has.
template <int BM, int BN, int WM, int WN, int EPI = 0, int BMS = BM>
__global__ __launch_bounds__(64 * WM * WN, (BN == 256 ? 2 : (WM * WN == 8 ? 4 : 3))) void conv_igemm_dma_kernel(const ConvK k) {
  using T = bf16_t;
  using TL = Tile<bf16_t>;
  constexpr int VEC = 8, SEGS = 4, ROWB = 64;
  constexpr int FM = BMS / WM / 16, FN = BN / WN / 16;
  static_assert(BMS <= BM && BMS % (16 * WM) == 0, "computed rows");
  constexpr int NW = WM * WN;                  // waves per block (4 or 8)
  constexpr int WROWS = BN < 16 * NW ? 16 * NW : BN;   // every wave issues the same number of filter DMAs (rows >= BN are OOB)
  constexpr int XI = BM / (16 * NW), WI = WROWS / (16 * NW);   // DMA wave-instructions per wave per K-step
  constexpr int BUF = (BM + WROWS) * ROWB;
  constexpr int RED = WM * BN * 2 * 4;
  constexpr int SM = 3 * BUF > RED ? 3 * BUF : RED;
  constexpr int TAB = 32;   // ints per table
  static_assert((NW == 4 || NW == 8) && BM % (16 * NW) == 0, "tile shape");
  // ONE LDS object (a second __shared__ array makes hipcc drain vmcnt(0) before every fragment read): ring | tables
  __shared__ __attribute__((aligned(16))) char smem[SM + 3 * TAB * 4];
  sfk_tap* s_taps = reinterpret_cast<sfk_tap*>(smem + SM);
  int* s_xdelta = reinterpret_cast<int*>(smem + SM) + TAB;
  int* s_woff = reinterpret_cast<int*>(smem + SM) + 2 * TAB;
  if (threadIdx.x <= SFK_MAX_TAPS) {
    sfk_tap t = k.taps[threadIdx.x < SFK_MAX_TAPS ? threadIdx.x : 0];
    if ((int)threadIdx.x >= k.ntaps) { t.dt = -128; t.dh = 0; t.dw = 0; t.widx = 0; }
    s_taps[threadIdx.x] = t;
    s_xdelta[threadIdx.x] = (((int)t.dt * k.xh + (int)t.dh) * k.xw + (int)t.dw) * k.xld * 2;
    s_woff[threadIdx.x] = (int)t.widx * k.cin * 2;
  }
  __syncthreads();

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int l15 = lane & 15, g = lane >> 4;
  int mt, nt;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = b & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    nt = logical % k.ntiles;
    mt = logical / k.ntiles;
  }

  // this lane's tile rows: X rows wave*(BM/4) + 16j + (lane>>2), filter rows wave*(WROWS/4) + 16j + (lane>>2);
  // rows 16 apart share (row>>2)&3, so ONE logical segment per lane
  const int lrow = lane >> 2;
  const int seg = (lane & 3) ^ ((4 - (((wave * (BM / NW) + lrow) >> 2) & 3)) & 3);
  static_assert((BM / NW) % 16 == 0 && (WROWS / NW) % 16 == 0, "row blocks");
  int xtb[XI], xhb[XI], xwb[XI];
  uint32_t xbase[XI];
#pragma unroll
  for (int j = 0; j < XI; ++j) {
    const int m = mt * BMS + wave * (BM / NW) + 16 * j + lrow;
    uint32_t q1, rw_, q2, rh_, n_, rt_;
    k.drw.divmod((uint32_t)m, q1, rw_);
    k.drh.divmod(q1, q2, rh_);
    k.drt.divmod(q2, n_, rt_);
    xtb[j] = (m < k.M) ? (int)rt_ * k.gst : -(1 << 28);
    xhb[j] = (int)rh_ * k.gsh;
    xwb[j] = (int)rw_ * k.gsw;
    xbase[j] = (uint32_t)((((((int64_t)n_ * k.xt + (int)rt_ * k.gst) * k.xh + xhb[j]) * k.xw + xwb[j]) * k.xld + k.xoff) * 2);
  }
  // the filter tile uses the same swizzle function on ITS row index
  const int wseg = (lane & 3) ^ ((4 - (((wave * (WROWS / NW) + lrow) >> 2) & 3)) & 3);
  uint32_t wbase[WI];
#pragma unroll
  for (int j = 0; j < WI; ++j) {
    const int r = wave * (WROWS / NW) + 16 * j + lrow;
    const int co = nt * BN + r;
    wbase[j] = (r < BN && co < k.cout) ? (uint32_t)(co * k.wtaps * k.cin) * 2u : SFK_OOB;
  }
  const __amdgpu_buffer_rsrc_t xrs = sfk_make_rsrc(k.x, k.xbytes);
  const __amdgpu_buffer_rsrc_t wrs = sfk_make_rsrc(k.w, k.wbytes);

  // Two ways to walk K.  uniform (cin % 32 == 0): a K-step stays inside ONE tap; the per-lane offsets (gather +
  // bounds test) are recomputed only when the tap changes and the channel-chunk advance (kc * 64 B) is wave-uniform,
  // so it rides the buffer instruction's SGPR soffset -- zero per-step VALU.  packed (narrow / ragged cin): segments
  // of one step may belong to different taps, offsets are rebuilt per step.
  const bool uniform = (k.cin & 31) == 0;
  const int kct = k.cin >> 5;                       // K-steps per tap (uniform mode)
  constexpr uint32_t FAR = 0x80000000u;             // stays out of range after adding any soffset (extents < 2 GiB)
  uint32_t xv[XI], wv[WI];                          // current voffsets (uniform mode)
  // the K position is a FUNCTION of the step number passed in (wave-uniform, SGPRs): as mutable state captured by the
  // lambda it lived in scratch memory, and every tap entry drained vmcnt(0) behind a scratch load

  auto issue = [&](int buf, const uint32_t (&xo)[XI], const uint32_t (&wo)[WI], int soff) __attribute__((always_inline)) {
    char* xs = smem + buf * BUF + wave * (BM / NW) * ROWB;
    char* ws = smem + buf * BUF + BM * ROWB + wave * (WROWS / NW) * ROWB;
#pragma unroll
    for (int j = 0; j < XI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(xs + 16 * j * ROWB), 16, (int)xo[j], soff, 0, 0);
#pragma unroll
    for (int j = 0; j < WI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void_t*)(ws + 16 * j * ROWB), 16, (int)wo[j], soff, 0, 0);
  };
  auto dma = [&](int buf, int step) __attribute__((always_inline)) {
    if (uniform) {
      const int u_tap = __builtin_amdgcn_readfirstlane((int)k.dkct.div((uint32_t)step));
      const int u_kc = step - u_tap * kct;
      if (u_kc == 0) {                               // entering tap u_tap (wave-uniform branch)
        if (u_tap < k.ntaps) {
          const sfk_tap tp = s_taps[u_tap];
          const int xd = s_xdelta[u_tap] + seg * 16;
          const int wd = s_woff[u_tap] + seg * 16;
#pragma unroll
          for (int j = 0; j < XI; ++j) {
            const int ti = xtb[j] + tp.dt, hi = xhb[j] + tp.dh, wi = xwb[j] + tp.dw;
            const bool ok = (unsigned)ti < (unsigned)k.xt && (unsigned)hi < (unsigned)k.xh && (unsigned)wi < (unsigned)k.xw;
            xv[j] = ok ? xbase[j] + (uint32_t)xd : FAR;
          }
#pragma unroll
          for (int j = 0; j < WI; ++j) wv[j] = wbase[j] != SFK_OOB ? wbase[j] + (uint32_t)wd : FAR;
        } else {                                     // past the last tap: the ring's look-ahead gathers nothing
#pragma unroll
          for (int j = 0; j < XI; ++j) xv[j] = FAR;
#pragma unroll
          for (int j = 0; j < WI; ++j) wv[j] = FAR;
        }
      }
      issue(buf, xv, wv, u_kc * 64);
    } else {
      const int p_step = step;
      uint32_t xo[XI], wo[WI];
      uint32_t tap, cseg;
      k.dspt.divmod((uint32_t)(p_step * SEGS + seg), tap, cseg);
      const bool cok = tap < (uint32_t)k.ntaps;
      const int ti_ = cok ? (int)tap : SFK_MAX_TAPS;
      const sfk_tap tp = s_taps[ti_];
      const uint32_t xd = (uint32_t)(s_xdelta[ti_] + (int)cseg * 16);
      const uint32_t wd = (uint32_t)(s_woff[ti_] + (int)cseg * 16);
#pragma unroll
      for (int j = 0; j < XI; ++j) {
        const int ti = xtb[j] + tp.dt, hi = xhb[j] + tp.dh, wi = xwb[j] + tp.dw;
        const bool ok = cok && (unsigned)ti < (unsigned)k.xt && (unsigned)hi < (unsigned)k.xh && (unsigned)wi < (unsigned)k.xw;
        xo[j] = ok ? xbase[j] + xd : FAR;
      }
#pragma unroll
      for (int j = 0; j < WI; ++j) wo[j] = (cok && wbase[j] != SFK_OOB) ? wbase[j] + wd : FAR;
      issue(buf, xo, wo, 0);
    }
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  ResPre<(EPI == 4 ? FM : 1), (EPI == 4 ? FN : 2)> respre;
  if constexpr (EPI == 4) prefetch_res<T, FM, FN, BMS, BN, WM, WN>(k, respre, mt, nt, wm, wn, lane);

  // fragment addresses inside a ring slot are loop-invariant; the slot base is a compile-time constant (ring unrolled)
  int a_off[FN], b_off[FM];
#pragma unroll
  for (int i = 0; i < FN; ++i) a_off[i] = BM * ROWB + TL::off(wn * (BN / WN) + 16 * i + l15, g);
#pragma unroll
  for (int j = 0; j < FM; ++j) b_off[j] = TL::off(wm * (BMS / WM) + 16 * j + l15, g);
  auto compute = [&](int slot_base) {
    TL::frag a[FN], b[FM];
#pragma unroll
    for (int i = 0; i < FN; ++i) a[i] = *reinterpret_cast<const TL::frag*>(smem + slot_base + a_off[i]);
#pragma unroll
    for (int j = 0; j < FM; ++j) b[j] = *reinterpret_cast<const TL::frag*>(smem + slot_base + b_off[j]);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) TL::mma(acc[i][j], a[i], b[j]);
  };
  // lgkmcnt(0) belongs to the wait: the barrier releases the OTHER waves to DMA into the slot this step just read, and
  // a ds_read that was issued but has not executed yet would then return the new bytes (seen as a rare, timing-dependent
  // slab of wrong outputs: the compiler sinks the last fragment reads' lgkmcnt wait below the barrier otherwise)
  auto ring_wait = [&]() {
    static_assert(XI + WI >= 2 && XI + WI <= 5, "DMA count");
    if constexpr (XI + WI == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (XI + WI == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else if constexpr (XI + WI == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // 3-slot ring, unrolled so that slot addresses are immediates.  Invariant at the top of a sub-step that computes
  // slot c: c has landed for every wave (barrier); the DMAs of the next step are in flight.  Issue the step after
  // next into the slot the previous step vacated, run this step, wait until only this wave's newest XI+WI DMAs are
  // outstanding, meet the other waves.
  dma(0, 0);
  dma(1, 1);
  ring_wait();
  for (int it = 0;;) {
    dma(2, it + 2); compute(0 * BUF); ring_wait();
    if (++it >= k.KC) break;
    dma(0, it + 2); compute(1 * BUF); ring_wait();
    if (++it >= k.KC) break;
    dma(1, it + 2); compute(2 * BUF); ring_wait();
    if (++it >= k.KC) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the (all-out-of-range) look-ahead DMAs before LDS is reused
  __syncthreads();

  // ---- epilogue
  if constexpr (EPI == 1 || EPI == 5 || EPI == 6 || EPI == 7) {   // fused BatchNorm-backward reduce; the ring is drained (vmcnt(0) + barrier above)
    if constexpr (EPI == 5) epilogue_bits_sum<FM, FN, BMS, BN, WM, WN>(k, acc, reinterpret_cast<float*>(smem), mt, nt, wm, wn, lane, tid);
    else epilogue_bn_bwd<FM, FN, BMS, BN, WM, WN, EPI == 6, EPI == 7>(k, acc, reinterpret_cast<float*>(smem), mt, nt, wm, wn, lane, tid);
    return;
  }
  if constexpr (EPI == 3) {
    epilogue_fused<T, FM, FN, BMS, BN, WM, WN>(k, acc, mt, nt, wm, wn, lane);
    return;
  }
  if constexpr (EPI == 4) {
    epilogue_fused<T, FM, FN, BMS, BN, WM, WN, true>(k, acc, mt, nt, wm, wn, lane, &respre);
    return;
  }
  epilogue_plain<T, EPI, FM, FN, BMS, BN, WM, WN, false>(k, acc, reinterpret_cast<float*>(smem), mt, nt, wm, wn, lane, tid);
}

struct TileSel { int bm, bn; bool dma; };
// ONE place decides tile and kernel family for a descriptor: sfk_conv_igemm_mtiles (rows of stats / partials) and the
// launch must agree.
inline TileSel pick_tile(const sfk_conv_desc* d) {
  const int cout = d->cout, dtype = d->x.dtype, ktot = d->ntaps * d->cin;
  const int64_t M = (int64_t)d->x.n * d->rt * d->rh * d->rw;
  if (cout > 64 && dtype == SFK_BF16) {
    // the LDS-DMA ring addresses < 2 GiB per operand (soffset rides on top of a 31-bit voffset); larger maps -- batches
    // beyond the benchmark's -- run the register-staged 128x128 kernel, which only needs the 4 GiB of validate()
    const int64_t wbytes = (int64_t)d->cout * d->wtaps * d->cin * 2;
    if (sfk_fmap_bytes(&d->x) >= 0x7FF00000ll || wbytes >= 0x7FF00000ll) return {128, 128, false};
    // wide outputs in bf16: a 256x128 tile (8 waves) needs 25% less L2->LDS traffic per FLOP than 128x128 -- worth it
    // once the grid still fills the chip
    // a shortcut in the fused epilogue is pre-fetched before the K loop: that needs the 4-wave tile's register budget
    // 256 x 256 (8 waves as 4 x 2, 64 x 128 per wave, ONE workgroup per CU): the pixel tile is fetched once for 256
    // output channels and a barrier interval carries 32 MFMAs per wave instead of 16 -- for the MFMA-bound layers whose
    // grid still covers the chip (plain epilogue only: the fused ones sit at their register caps)
    // Measured (res4, M = 50,176): 1024 -> 256 (3,1,1) 775 -> 863 TFLOP/s, 256 -> 256 (1,3,3) data gradient 844 -> 882; with
    // 1024 output channels (784 workgroups, one per CU: 3.06 generations) 810 -> 713 -- so only where ONE tile spans cout.
    if (sfk_tune().igemm_tile256 && cout == 256 && ktot >= 512 && M >= 256 * 128 && !d->ep.scale && !d->ep.shift &&
        !d->bnb.partials && !d->out_relu_bits) {
      // one workgroup per CU: 224 computed rows per tile when that needs fewer row-generations x rows than 256
      const int64_t g256 = ((M + 255) / 256 + 255) / 256 * 256, g224 = ((M + 223) / 224 + 255) / 256 * 224;
      return {(sfk_tune().igemm_tile256 & 2) && g224 < g256 ? 224 : 256, 256, true};
    }
    // (the fused BatchNorm-backward epilogue keeps two rows of y_bn / mask / old values in flight: the 4-wave tile's register budget)
    if (M >= 256 * 128 && ktot > sfk_tune().igemm_small_k && !d->ep.res.ptr && !d->bnb.partials) return {256, 128, true};
    return {128, 128, true};
  }
  if (cout > 64) return {128, 128, false};
  if (cout > 32) return {256, 64, false};
  if (cout > 16) return {256, 32, false};
  return {256, 16, false};
}

// the fused BatchNorm-backward epilogue exists for bf16 tiles with an even number of co fragments and 16-byte stores
inline bool bnb_ok(const sfk_conv_desc* d) {
  return d->x.dtype == SFK_BF16 && d->cout > 16 && (d->cout % 8) == 0 && (d->y.ld % 8) == 0 && (d->y.c_off % 8) == 0 &&
         (((uintptr_t)d->y.ptr) & 15) == 0 && sfk_fmap_bytes(&d->y) < (1ll << 31);   // (y_bn / the mask share y's pixel grid)
}

// out_relu_bits: the pass must write every pixel of y in row order (the bitmap is indexed by the row), and bf16 needs
// the 16-byte store path (one bitmap byte per store)
inline bool lin_out_of(const sfk_conv_desc* d) {
  return d->os[0] == 1 && d->os[1] == 1 && d->os[2] == 1 && d->oo[0] == 0 && d->oo[1] == 0 && d->oo[2] == 0 &&
         d->rt == d->y.t && d->rh == d->y.h && d->rw == d->y.w;
}
inline bool relu_out_ok(const sfk_conv_desc* d) {
  if (!lin_out_of(d)) return false;
  return d->x.dtype == SFK_F32 ? true : bnb_ok(d);
}

inline bool ep_on(const sfk_conv_desc* d) { return d->ep.scale || d->ep.shift; }

// pointwise pass over the pixels of y in order (the streaming kernels of conv_pw.hip)
inline bool pointwise_lin(const sfk_conv_desc* d) {
  return d->ntaps == 1 && d->taps[0].dt == 0 && d->taps[0].dh == 0 && d->taps[0].dw == 0 && d->gs[0] == 1 && d->gs[1] == 1 &&
         d->gs[2] == 1 && lin_out_of(d) && d->x.t == d->y.t && d->x.h == d->y.h && d->x.w == d->y.w;
}
// plain / += / + bias pointwise passes with K = cout = 64 / 128 (the second data-gradient pass of the block tail's backward on
// slow res2 / res3): streaming kernel of conv_pw.hip
inline bool pw_plain_route(const sfk_conv_desc* d) {
  if (d->x.dtype != SFK_BF16 || !sfk_tune().igemm_pw_stream || (sfk_tune().igemm_pw_stream & 8) || d->stats || d->bnb.partials ||
      d->out_relu_bits || d->ep.res.ptr || d->ep.relu || d->ep.scale || !pointwise_lin(d) || (d->y.ld % 8) || (d->y.c_off % 8))
    return false;
  // the streaming kernel reads the old rows of a += pass through a 32-bit buffer resource (validate() bounds y only for
  // fused epilogues): maps of 4 GiB and more stay on the implicit GEMM, whose epilogue uses 64-bit pointers
  if (sfk_fmap_bytes(&d->y) >= (1ll << 32) - 64) return false;
  // (128 -> 320: the data gradient of slow res3's first conv_a into the concatenated 256 + 64-channel input gradient -- an
  // output-heavy pass, 514 MB written for 205 read: 220 us on the 2.5-tile-wide implicit GEMM)
  return (d->cout == 64 && d->cin == 64) || (d->cout == 128 && d->cin == 128) || (d->cout == 320 && d->cin == 128 && !d->accumulate && !d->ep.shift);
}
// rows of the streaming data-gradient kernel (accumulate + bitmap mask + column sums), 0 when the pass is not one of its
inline int pw_dgrad_rows(const sfk_conv_desc* d) {
  if (d->x.dtype != SFK_BF16 || !sfk_tune().igemm_pw_stream || (sfk_tune().igemm_pw_stream & 4) || !d->accumulate || ep_on(d) || d->stats || !d->out_relu_bits ||
      !d->bnb.partials || d->bnb.y_bn.ptr || !pointwise_lin(d) || !bnb_ok(d))
    return 0;
  return sfk_conv_pw_dgrad_rows(d);
}

int validate(const sfk_conv_desc* d) {
  if (!d || d->struct_size != sizeof(sfk_conv_desc)) return SFK_ERR_INVALID;   // ABI handshake: the caller's layout is ours
  if (!d->w) return SFK_ERR_INVALID;
  if (!sfk_fmap_ok(&d->x) || !sfk_fmap_ok(&d->y)) return SFK_ERR_INVALID;
  if (d->x.dtype != d->y.dtype || d->x.n != d->y.n) return SFK_ERR_INVALID;
  if (d->cin != d->x.c || d->cout != d->y.c) return SFK_ERR_INVALID;
  if (d->rt <= 0 || d->rh <= 0 || d->rw <= 0 || d->ntaps <= 0 || d->ntaps > SFK_MAX_TAPS) return SFK_ERR_INVALID;
  if (d->wtaps <= 0) return SFK_ERR_INVALID;
  for (int i = 0; i < d->ntaps; ++i)
    if (d->taps[i].widx >= d->wtaps) return SFK_ERR_INVALID;
  for (int a = 0; a < 3; ++a)
    if (d->gs[a] <= 0 || d->os[a] <= 0 || d->oo[a] < 0) return SFK_ERR_INVALID;
  // every scattered row must land inside y
  if ((d->rt - 1) * d->os[0] + d->oo[0] >= d->y.t || (d->rh - 1) * d->os[1] + d->oo[1] >= d->y.h ||
      (d->rw - 1) * d->os[2] + d->oo[2] >= d->y.w)
    return SFK_ERR_INVALID;
  if ((int64_t)d->x.n * d->rt * d->rh * d->rw >= (1ll << 31)) return SFK_ERR_UNSUPPORTED;
  if (!sfk_fmap_vec_ok(&d->x)) return SFK_ERR_UNSUPPORTED;
  if ((d->y.c % 4) || (d->y.ld % 4) || (d->y.c_off % 4) || (((uintptr_t)d->y.ptr) & 15)) return SFK_ERR_UNSUPPORTED;
  if (((uintptr_t)d->w) & 15) return SFK_ERR_UNSUPPORTED;
  const bool bnb_bits = d->bnb.partials && !d->bnb.y_bn.ptr;   // mask = out_relu_bits, partials = (sum dz, 0)
  if (d->out_relu_bits) {
    if (d->bnb.partials && !bnb_bits) return SFK_ERR_INVALID;
    if (!relu_out_ok(d)) return SFK_ERR_UNSUPPORTED;
  }
  if (ep_on(d)) {
    const sfk_conv_epilogue& e = d->ep;
    if (d->stats || d->bnb.partials || d->out_relu_bits) return SFK_ERR_INVALID;
    if ((e.relu_bits && !e.relu) || (e.res_scale && !e.res.ptr) || (e.res_shift && !e.res.ptr)) return SFK_ERR_INVALID;
    if (e.res.ptr && (!sfk_fmap_ok(&e.res) || e.res.n != d->y.n || e.res.t != d->y.t || e.res.h != d->y.h ||
                      e.res.w != d->y.w || e.res.c != d->y.c || e.res.dtype != d->y.dtype))
      return SFK_ERR_INVALID;
    if (!lin_out_of(d)) return SFK_ERR_UNSUPPORTED;
    if ((((uintptr_t)e.scale) | ((uintptr_t)e.shift) | ((uintptr_t)e.res_scale) | ((uintptr_t)e.res_shift)) & 15)
      return SFK_ERR_UNSUPPORTED;
    if (d->x.dtype == SFK_BF16) {
      // tiles with an even number of co fragments (cout > 16) store 16-byte channel groups; the one-fragment tile of the
      // narrowest layers does scale / shift / += only
      if (d->cout > 16 ? !bnb_ok(d) : (e.res.ptr || e.relu)) return SFK_ERR_UNSUPPORTED;
    }
    if (e.res.ptr && !sfk_fmap_vec_ok(&e.res)) return SFK_ERR_UNSUPPORTED;
    if (sfk_fmap_bytes(&d->y) >= (1ll << 32) - 64 || (e.res.ptr && sfk_fmap_bytes(&e.res) >= (1ll << 32) - 64))
      return SFK_ERR_UNSUPPORTED;
  }
  if (d->bnb.partials) {   // the fused epilogue reads y (+=), y_bn and the mask through 32-bit buffer resources
    if (sfk_fmap_bytes(&d->y) >= (1ll << 32) - 64) return SFK_ERR_UNSUPPORTED;
    if (d->bnb.y_bn.ptr && sfk_fmap_ok(&d->bnb.y_bn) && sfk_fmap_bytes(&d->bnb.y_bn) >= (1ll << 32) - 64) return SFK_ERR_UNSUPPORTED;
    if (d->bnb.mask_src.ptr && sfk_fmap_ok(&d->bnb.mask_src) && sfk_fmap_bytes(&d->bnb.mask_src) >= (1ll << 32) - 64) return SFK_ERR_UNSUPPORTED;
  }
  if (bnb_bits) {
    if (!d->out_relu_bits || d->stats || d->bnb.mask_src.ptr) return SFK_ERR_INVALID;
    if (!bnb_ok(d) || !lin_out_of(d)) return SFK_ERR_UNSUPPORTED;
  } else if (d->bnb.partials) {
    const sfk_bn_bwd_fuse& b = d->bnb;
    if (d->stats || !sfk_fmap_ok(&b.y_bn) || !b.mean || !b.invstd) return SFK_ERR_INVALID;
    if (b.y_bn.n != d->y.n || b.y_bn.t != d->y.t || b.y_bn.h != d->y.h || b.y_bn.w != d->y.w || b.y_bn.c != d->y.c ||
        b.y_bn.dtype != d->y.dtype)
      return SFK_ERR_INVALID;
    if (b.mask_src.ptr && (!sfk_fmap_ok(&b.mask_src) || b.mask_src.n != d->y.n || b.mask_src.t != d->y.t ||
                           b.mask_src.h != d->y.h || b.mask_src.w != d->y.w || b.mask_src.c != d->y.c ||
                           b.mask_src.dtype != d->y.dtype))
      return SFK_ERR_INVALID;
    if (!b.mask_src.ptr && b.relu && (!b.scale || !b.shift)) return SFK_ERR_INVALID;
    if (!bnb_ok(d)) return SFK_ERR_UNSUPPORTED;
    if (!sfk_fmap_vec_ok(&b.y_bn) || (b.mask_src.ptr && !sfk_fmap_vec_ok(&b.mask_src))) return SFK_ERR_UNSUPPORTED;
  }
  // buffer resources address 32 bits
  const int64_t esz = d->x.dtype == SFK_BF16 ? 2 : 4;
  if (sfk_fmap_bytes(&d->x) >= (1ll << 32) - 64 || (int64_t)d->cout * d->wtaps * d->cin * esz >= (1ll << 32) - 64)
    return SFK_ERR_UNSUPPORTED;
  return SFK_OK;
}

int launch_dma(const ConvK& k, int bm, dim3 grid, hipStream_t s, int bn = 128) {
  if (bn == 256) {
    if (bm == 224) hipLaunchKernelGGL((conv_igemm_dma_kernel<256, 256, 2, 4, 0, 224>), grid, dim3(512), 0, s, k);
    else hipLaunchKernelGGL((conv_igemm_dma_kernel<256, 256, 4, 2>), grid, dim3(512), 0, s, k);
    SFK_CHECK_LAUNCH();
    return SFK_OK;
  }
  if (k.bn_parts) {          // (pick_tile: never 256 x 128); 5 = the bitmap flavour (mask = out_relu_bits, sums of dz only)
    // (6 = with the old values of a += pass, 7 = with a mask source)
    if (k.bn_y == nullptr) hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, 2, 2, 5>), grid, dim3(256), 0, s, k);
    else if (k.bn_mask) hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, 2, 2, 7>), grid, dim3(256), 0, s, k);
    else if (k.accumulate) hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, 2, 2, 6>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, 2, 2, 1>), grid, dim3(256), 0, s, k);
    SFK_CHECK_LAUNCH();
    return SFK_OK;
  }
  if (k.ep_on) {
    if (bm == 256) hipLaunchKernelGGL((conv_igemm_dma_kernel<256, 128, 4, 2, 3>), grid, dim3(512), 0, s, k);
    else if (k.ep_res) hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, 2, 2, 4>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, 2, 2, 3>), grid, dim3(256), 0, s, k);
    SFK_CHECK_LAUNCH();
    return SFK_OK;
  }
  if (k.obits) {
    if (bm == 256) hipLaunchKernelGGL((conv_igemm_dma_kernel<256, 128, 4, 2, 2>), grid, dim3(512), 0, s, k);
    else hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, 2, 2, 2>), grid, dim3(256), 0, s, k);
    SFK_CHECK_LAUNCH();
    return SFK_OK;
  }
  if (bm == 256) hipLaunchKernelGGL((conv_igemm_dma_kernel<256, 128, 4, 2>), grid, dim3(512), 0, s, k);
  else hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 128, 2, 2>), grid, dim3(256), 0, s, k);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

template <typename T>
int launch(const sfk_conv_desc* d, hipStream_t s) {
  ConvK k;
  k.x = d->x.ptr; k.y = d->y.ptr; k.w = d->w; k.stats = d->stats;
  k.xt = d->x.t; k.xh = d->x.h; k.xw = d->x.w; k.xld = d->x.ld; k.xoff = d->x.c_off;
  k.yt = d->y.t; k.yh = d->y.h; k.yw = d->y.w; k.yld = d->y.ld; k.yoff = d->y.c_off;
  k.M = d->x.n * d->rt * d->rh * d->rw;
  k.drw.set(d->rw); k.drh.set(d->rh); k.drt.set(d->rt);
  k.gst = d->gs[0]; k.gsh = d->gs[1]; k.gsw = d->gs[2];
  k.ost = d->os[0]; k.osh = d->os[1]; k.osw = d->os[2];
  k.oot = d->oo[0]; k.ooh = d->oo[1]; k.oow = d->oo[2];
  k.cin = d->cin; k.cout = d->cout; k.wtaps = d->wtaps; k.ntaps = d->ntaps;
  const int vec = sfk_vec_of(d->x.dtype), segs = BK / vec;
  k.dspt.set(d->cin / vec);
  k.dkct.set(d->cin / 32 > 0 ? d->cin / 32 : 1);
  k.KC = (d->ntaps * (d->cin / vec) + segs - 1) / segs;
  k.accumulate = d->accumulate;
  k.obits = d->out_relu_bits;
  k.bn_parts = d->bnb.partials;
  k.bn_y = nullptr; k.bn_mask = nullptr;
  if (k.bn_parts) {
    k.bn_y = d->bnb.y_bn.ptr; k.bn_yld = d->bnb.y_bn.ld; k.bn_yoff = d->bnb.y_bn.c_off;
    k.bn_mask = d->bnb.mask_src.ptr; k.bn_mld = d->bnb.mask_src.ld; k.bn_moff = d->bnb.mask_src.c_off;
    k.bn_relu = d->bnb.relu;
    k.bn_mean = d->bnb.mean; k.bn_invstd = d->bnb.invstd; k.bn_scale = d->bnb.scale; k.bn_shift = d->bnb.shift;
  }
  k.ep_on = ep_on(d) ? 1 : 0;
  k.ep_scale = d->ep.scale; k.ep_shift = d->ep.shift; k.ep_rscale = d->ep.res_scale; k.ep_rshift = d->ep.res_shift;
  k.ep_res = d->ep.res.ptr; k.ep_rld = d->ep.res.ld; k.ep_roff = d->ep.res.c_off;
  k.ep_relu = d->ep.relu; k.ep_bits = d->ep.relu_bits;
  k.ybytes = (uint32_t)sfk_fmap_bytes(&d->y);
  k.ybig = sfk_fmap_bytes(&d->y) >= (1ll << 32) - 64 ? 1 : 0;
  k.ep_rbytes = d->ep.res.ptr ? (uint32_t)sfk_fmap_bytes(&d->ep.res) : 0u;
  k.bn_ybytes = (k.bn_parts && d->bnb.y_bn.ptr) ? (uint32_t)sfk_fmap_bytes(&d->bnb.y_bn) : 0u;
  k.bn_mbytes = (k.bn_parts && d->bnb.mask_src.ptr) ? (uint32_t)sfk_fmap_bytes(&d->bnb.mask_src) : 0u;
  k.obits_bytes = d->out_relu_bits ? (uint32_t)(sfk_fmap_pixels(&d->y) * (d->cout / sfk_vec_of(d->x.dtype))) : 0u;
  k.kshort = sfk_tune().igemm_short_k;
  k.lin_out = lin_out_of(d);
  const int wide_ok = sfk_tune().igemm_wide_store;
  k.wide_store = (wide_ok || k.obits) && (d->cout % 8) == 0 && (d->y.ld % 8) == 0 && (d->y.c_off % 8) == 0;
  k.xbytes = (uint32_t)sfk_fmap_bytes(&d->x);
  k.wbytes = (uint32_t)((int64_t)d->cout * d->wtaps * d->cin * (d->x.dtype == SFK_BF16 ? 2 : 4));
  for (int i = 0; i < SFK_MAX_TAPS; ++i) k.taps[i] = d->taps[i < d->ntaps ? i : 0];
  if constexpr (sizeof(T) == 2) {
    // the block tail's conv_c (small filter, shortcut / ReLU in the epilogue): the streaming kernel of conv_pw.hip
    if (k.ep_on && (d->ep.res.ptr || d->ep.relu) && !d->accumulate && sfk_tune().igemm_pw_stream && d->ntaps == 1 &&
        d->taps[0].dt == 0 && d->taps[0].dh == 0 && d->taps[0].dw == 0 && d->gs[0] == 1 && d->gs[1] == 1 && d->gs[2] == 1 &&
        k.lin_out && d->x.t == d->y.t && d->x.h == d->y.h && d->x.w == d->y.w) {
      const int r = sfk_conv_pw_fused(d, s);
      if (r != SFK_ERR_UNSUPPORTED) return r;
    }
    if (pw_dgrad_rows(d) > 0) return sfk_conv_pw_dgrad(d, s);
    if (pw_plain_route(d)) {
      const int r = sfk_conv_pw_fused(d, s);
      if (r != SFK_ERR_UNSUPPORTED) return r;
    }
  }
  const TileSel ts = pick_tile(d);
  k.mtiles = (k.M + ts.bm - 1) / ts.bm;
  k.ntiles = (d->cout + ts.bn - 1) / ts.bn;
  const dim3 grid((unsigned)(k.mtiles * k.ntiles)), block(256);
  // bf16: LDS-DMA ring for the wide tile; narrow outputs keep the register-staged kernel (higher occupancy, tiny K)
  if (ts.dma) return launch_dma(k, ts.bm, grid, s, ts.bn);
  if (k.bn_parts) {          // fused BatchNorm-backward reduce (bf16, cout > 16: tiles of 32..128 output channels)
    if constexpr (sizeof(T) == 2) {
      if (k.bn_y == nullptr) {      // the bitmap flavour
        if (ts.bn == 128) hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 128, 2, 2, false, 5>), grid, block, 0, s, k);
        else if (ts.bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 64, 4, 1, false, 5>), grid, block, 0, s, k);
        else hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 32, 4, 1, false, 5>), grid, block, 0, s, k);
      } else if (k.bn_mask) {       // with a mask source
        if (ts.bn == 128) hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 128, 2, 2, false, 7>), grid, block, 0, s, k);
        else if (ts.bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 64, 4, 1, false, 7>), grid, block, 0, s, k);
        else hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 32, 4, 1, false, 7>), grid, block, 0, s, k);
      } else if (k.accumulate) {    // with the old values of a += pass
        if (ts.bn == 128) hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 128, 2, 2, false, 6>), grid, block, 0, s, k);
        else if (ts.bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 64, 4, 1, false, 6>), grid, block, 0, s, k);
        else hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 32, 4, 1, false, 6>), grid, block, 0, s, k);
      } else if (ts.bn == 128) hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 128, 2, 2, false, 1>), grid, block, 0, s, k);
      else if (ts.bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 64, 4, 1, false, 1>), grid, block, 0, s, k);
      else hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 32, 4, 1, false, 1>), grid, block, 0, s, k);
      SFK_CHECK_LAUNCH();
      return SFK_OK;
    }
    return SFK_ERR_UNSUPPORTED;
  }
  if (k.ep_on) {             // fused output transform (look-ahead K loop for every K)
    if (ts.bn == 128) hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 128, 2, 2, false, 3>), grid, block, 0, s, k);
    else if (ts.bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 64, 4, 1, false, 3>), grid, block, 0, s, k);
    else if (ts.bn == 32) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 32, 4, 1, false, 3>), grid, block, 0, s, k);
    else hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 16, 4, 1, false, 3>), grid, block, 0, s, k);
    SFK_CHECK_LAUNCH();
    return SFK_OK;
  }
  if (k.obits) {             // output ReLU bitmap: the look-ahead K loop also for short K (few launches, one instantiation less)
    if (ts.bn == 128) hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 128, 2, 2, false, 2>), grid, block, 0, s, k);
    else if (ts.bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 64, 4, 1, false, 2>), grid, block, 0, s, k);
    else if (ts.bn == 32) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 32, 4, 1, false, 2>), grid, block, 0, s, k);
    else if (sizeof(T) == 4) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 16, 4, 1, false, 2>), grid, block, 0, s, k);
    else return SFK_ERR_UNSUPPORTED;
    SFK_CHECK_LAUNCH();
    return SFK_OK;
  }
  if (k.KC <= k.kshort && ts.bn <= 32) {   // (the wider tiles spill with the exact-count loop: 124..228 B per lane)
    if (ts.bn == 32) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 32, 4, 1, true>), grid, block, 0, s, k);
    else hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 16, 4, 1, true>), grid, block, 0, s, k);
  } else {
    if (ts.bn == 128) hipLaunchKernelGGL((conv_igemm_kernel<T, 128, 128, 2, 2, false>), grid, block, 0, s, k);
    else if (ts.bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 64, 4, 1, false>), grid, block, 0, s, k);
    else if (ts.bn == 32) hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 32, 4, 1, false>), grid, block, 0, s, k);
    else hipLaunchKernelGGL((conv_igemm_kernel<T, 256, 16, 4, 1, false>), grid, block, 0, s, k);
  }
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

}  // namespace

extern "C" int sfk_conv_igemm_mtiles(const sfk_conv_desc* d) {
  const int st = validate(d);
  if (st != SFK_OK) return st;
  const int rows = pw_dgrad_rows(d);      // the streaming data-gradient kernel leaves one row per wave of a co group
  if (rows > 0) return rows;
  const int64_t M = (int64_t)d->x.n * d->rt * d->rh * d->rw;
  const int bm = pick_tile(d).bm;
  return (int)((M + bm - 1) / bm);
}

extern "C" int sfk_conv_relu_out_supported(const sfk_conv_desc* d) {
  if (!d || d->struct_size != sizeof(sfk_conv_desc)) return 0;
  sfk_conv_desc c = *d;
  c.out_relu_bits = nullptr;
  if (validate(&c) != SFK_OK) return 0;
  return relu_out_ok(d) ? 1 : 0;
}

extern "C" int sfk_conv_igemm_family(const sfk_conv_desc* d) {
  if (validate(d) != SFK_OK) return -1;
  if (d->x.dtype == SFK_BF16) {
    if (ep_on(d) && (d->ep.res.ptr || d->ep.relu) && !d->accumulate && sfk_tune().igemm_pw_stream && d->ntaps == 1 &&
        d->taps[0].dt == 0 && d->taps[0].dh == 0 && d->taps[0].dw == 0 && d->gs[0] == 1 && d->gs[1] == 1 && d->gs[2] == 1 &&
        lin_out_of(d) && d->x.t == d->y.t && d->x.h == d->y.h && d->x.w == d->y.w && d->cin <= 128 &&
        ((d->cout == 32 || d->cout == 64 || d->cout == 128) ? d->cin <= 32 : (d->cout == 256 ? (d->cin > 32 && d->cin <= 64) : (d->cout == 512 && d->cin > 96))))
      return 3;
    if (pw_dgrad_rows(d) > 0) return 3;
    if (pw_plain_route(d)) return 3;
  }
  return pick_tile(d).dma ? 1 : 0;
}

extern "C" int sfk_conv_epilogue_supported(const sfk_conv_desc* d) {
  return (d && ep_on(d) && validate(d) == SFK_OK) ? 1 : 0;
}

extern "C" int sfk_conv_bnb_supported(const sfk_conv_desc* d) {
  if (!d || d->struct_size != sizeof(sfk_conv_desc)) return 0;
  sfk_conv_desc c = *d;
  c.bnb.partials = nullptr;
  if (validate(&c) != SFK_OK) return 0;
  return bnb_ok(d) ? 1 : 0;
}

extern "C" int sfk_conv_igemm(const sfk_conv_desc* d, sfk_stream_t stream) {
  const int st = validate(d);
  if (st != SFK_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return d->x.dtype == SFK_BF16 ? launch<bf16_t>(d, s) : launch<float>(d, s);
}
