// Streaming pointwise convolution with the fused output transform (sfk_conv_epilogue), bf16:
//     Y[pix][co] = ReLU( scale[co] * sum_k X[pix][k] W[co][k] + shift[co] + res[pix][co] * rs[co] + rh[co] )   (+ ReLU bitmap)
// for the conv_c -> norm_c -> + shortcut -> ReLU tail of a bottleneck block with a SMALL filter (cout * K * 2 B <= 128 KB).
//
// Why not the implicit-GEMM kernel: with K = 8..128 a 128-pixel tile is 1..4 K-steps of work; one tile per workgroup makes
// the tile's life  DMA round trip -> (shortcut round trip) -> stores  with nothing to overlap them but the other resident
// workgroup (measured 262 us for 0.95 GB, the rate of the plain output-heavy conv).  Here nothing is tiled over workgroups:
//   * the whole filter sits in LDS (loaded once per workgroup, the swizzled [row][32 k] slabs of conv_igemm's Tile<bf16>);
//   * a WAVE owns 16-pixel rows of the output and walks the map with a stride of all waves; the MFMA B operand (pixels) is
//     read straight from HBM in fragment layout -- lane (pixel l15, k group g) loads its 16 bytes, no LDS staging, no barrier;
//   * software pipeline one tile deep: the loads of tile i+1 (X and the shortcut rows, up to 12 x 16 B per lane) are in
//     flight while tile i runs its MFMAs and its epilogue; 8 waves per CU keep ~80 KB in flight;
//   * epilogue as conv_igemm's: v_permlane16_swap pairs -> 8 consecutive channels per lane -> 16-byte stores.
//
// DG = true is the same machine as a DATA-GRADIENT pass of a pointwise conv_a whose result finishes the gradient of the
// previous block's output (sfk_conv_desc: accumulate + out_relu_bits + bnb with y_bn = NULL):
//     dX[pix][ci] = (dX[pix][ci] + sum_k dY[pix][k] Wt[ci][k]) * bit[pix][ci]      and the column sums of the STORED values
// -- K = 64 / 128 only, so the implicit-GEMM kernel's tile lives for 2..4 K-steps and is all epilogue (335 us for 0.95 GB on
// slow res2); here the old rows and the bitmap bytes are prefetched one tile ahead like the shortcut of the forward, and
// every wave leaves ONE partial row [cout][2] = (sum, 0): sfk_conv_igemm_mtiles reports the number of waves per co group.
#include "sfk_common.h"

namespace {

struct PwK {
  const void* x;
  void* y;
  const void* w;
  const void* res;
  const float *scale, *shift, *rscale, *rshift;
  uint8_t* bits;
  const uint8_t* mbits;        // DG: the ReLU bitmap to multiply by
  float* parts;                // DG: partial rows [waves per co group][cout][2] = (sum, 0)
  int xld, xoff, yld, yoff, rld, roff;
  int M, K, cout, relu;
  uint32_t xbytes, ybytes, rbytes;
};

// 16-byte slot g of row r of a [rows][32 k] bf16 slab (as Tile<bf16>::off in conv_igemm.hip)
__device__ __forceinline__ int slab_off(int r, int s) { return r * 64 + ((s ^ ((4 - ((r >> 2) & 3)) & 3)) << 4); }

__device__ __forceinline__ void swap16f(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// NF: co fragments (16 channels) per wave; KS: 32-wide K steps; CG: co groups (waves of a group share a co range);
// NT: threads per workgroup (512 when the filter leaves room for ONE workgroup per CU: still 2 waves per SIMD);
// XDB: the next tile's X rows have registers of their own (else they are fetched into the current ones once the MFMAs
// have consumed them: KS = 4 would spill otherwise)
// MODE 0: fused output transform (plain / += / + bias passes included: scale = 1, shortcut = the old rows); 1 (DG): data
// gradient, see above.  (A third flavour -- plain store + BatchNorm partial rows for K = 256 / 512 -> 64 / 128, the forward
// conv_a of slow res2 / res3 -- measured SLOWER than the implicit GEMM, 129 vs 119 us: 32 filter fragments per 16-pixel tile
// no longer fit in registers and are re-read from LDS; it was removed.)
template <int NF, int KS, int CG, int NT, bool XDB, int MODE = 0>
__global__ __launch_bounds__(NT, 2) void conv_pw_fused_kernel(const PwK k) {
  constexpr bool DG = MODE == 1;
  constexpr int CW = NF * 16;                 // channels per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;                            // KS slabs of [cout][32]
  float* coef = reinterpret_cast<float*>(smem + KS * k.cout * 64);   // scale | shift(+rshift) | rscale, cout floats each
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  // ---- filter -> LDS: 16-byte segments, zero beyond K
  {
    const __amdgpu_buffer_rsrc_t wrs = sfk_make_rsrc(k.w, (uint32_t)k.cout * (uint32_t)k.K * 2u);
    const int segs = k.cout * KS * 4;
    for (int i = tid; i < segs; i += NT) {
      const int s = i & 3, ks = (i >> 2) % KS, r = i / (4 * KS);
      const int kk = ks * 32 + s * 8;
      const uint4 v = sfk_buffer_load16(wrs, kk < k.K ? (uint32_t)((r * k.K + kk) * 2) : SFK_OOB);
      *reinterpret_cast<uint4*>(wl + ks * k.cout * 64 + slab_off(r, s)) = v;
    }
    if (MODE == 0) for (int i = tid; i < k.cout; i += NT) {
      coef[i] = k.scale ? k.scale[i] : 1.f;
      coef[k.cout + i] = (k.shift ? k.shift[i] : 0.f) + ((k.res && k.rshift) ? k.rshift[i] : 0.f);
      coef[2 * k.cout + i] = (k.res && k.rscale) ? k.rscale[i] : 1.f;
    }
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t xrs = sfk_make_rsrc(k.x, k.xbytes);
  const __amdgpu_buffer_rsrc_t rrs = sfk_make_rsrc(k.res, k.res ? k.rbytes : 0u);
  const bool has_res = k.res != nullptr;
  // waves of one co group walk the 16-pixel tiles with a stride of (all waves) / CG
  constexpr int WPB = NT / 64;
  const int wg = (blockIdx.x * WPB + wave);
  const int cgi = wg % CG, wi = wg / CG, nw = (gridDim.x * WPB) / CG;
  const int co_w = cgi * CW;
  const int ntiles = (k.M + 15) >> 4;
  const bf16_t* __restrict__ dummy = nullptr;
  (void)dummy;

  uint4 xc[KS], xn[XDB ? KS : 1];
  uint4 rc[NF / 2], rn[NF / 2];
  uint32_t mc[DG ? NF / 2 : 1], mn[DG ? NF / 2 : 1];      // DG: bitmap bytes of the 8-channel groups this lane stores
  float csum[DG ? NF / 2 : 1][8];
  if constexpr (DG) {
#pragma unroll
    for (int p = 0; p < NF / 2; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) csum[p][e] = 0.f;
  }
  auto issue_x = [&](int tile, uint4 (&xv)[KS]) __attribute__((always_inline)) {
    int m = tile * 16 + l15;
    if (m >= k.M) m = k.M - 1;                                   // ragged last tile: re-read the last row (not stored)
    const uint32_t xrow = (uint32_t)(((int64_t)m * k.xld + k.xoff) * 2);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kk = ks * 32 + g * 8;
      xv[ks] = sfk_buffer_load16(xrs, kk < k.K ? xrow + (uint32_t)(kk * 2) : SFK_OOB);
    }
  };
  auto issue_r = [&](int tile, uint4 (&rv)[NF / 2], uint32_t (&mv)[DG ? NF / 2 : 1]) __attribute__((always_inline)) {
    if (has_res) {
      int m = tile * 16 + l15;
      if (m >= k.M) m = k.M - 1;
      const uint32_t rrow = (uint32_t)(((int64_t)m * k.rld + k.roff + co_w) * 2);
#pragma unroll
      for (int p = 0; p < NF / 2; ++p)
        rv[p] = sfk_buffer_load16(rrs, rrow + (uint32_t)((32 * p + 16 * (g & 1) + 8 * (g >> 1)) * 2));
      if constexpr (DG) {
#pragma unroll
        for (int p = 0; p < NF / 2; ++p)
          mv[p] = k.mbits[(int64_t)m * (k.cout >> 3) + ((co_w + 32 * p + 16 * (g & 1) + 8 * (g >> 1)) >> 3)];
      }
    }
  };

  int tile = wi;
  if (MODE == 0 && tile >= ntiles) return;
  if (tile < ntiles) {
    issue_x(tile, xc);
    issue_r(tile, rc, mc);
  }
  bf16_t* __restrict__ yp = static_cast<bf16_t*>(k.y);
  for (; tile < ntiles; tile += nw) {
    const int nxt = tile + nw;
    if (nxt < ntiles) {
      if constexpr (XDB) issue_x(nxt, reinterpret_cast<uint4(&)[KS]>(xn));
      issue_r(nxt, rn, mn);
    }
    // ---- MFMAs: A = filter fragment (rows = co) from LDS, B = the pixel fragment in registers
    f32x4 acc[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The filter fragments do not depend on the tile: with few of them the compiler keeps them in registers across tiles
    // (no LDS traffic in the loop); 32 or more (128+ VGPRs) would spill, so there the LDS base is laundered per tile and
    // the fragments are re-read (LDS time stays under the tile's HBM time: 32 KB per 16 x K x 2 B of pixels)
    const char* wlt = wl;
    if constexpr (NF * KS >= 32) asm volatile("" : "+v"(wlt));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 b = __builtin_bit_cast(bf16x8, xc[ks]);
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(wlt + ks * k.cout * 64 + slab_off(co_w + 16 * i + l15, g));
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      }
    }
    if constexpr (!XDB) {
      if (nxt < ntiles) issue_x(nxt, xc);      // the MFMAs above have consumed xc
    }
    // ---- epilogue: fragment pairs -> 8 consecutive channels per lane
    const int m = tile * 16 + l15;
    const bool rok = m < k.M;
    const int64_t yrow = (int64_t)m * k.yld + k.yoff;
#pragma unroll
    for (int p = 0; p < NF / 2; ++p) {
      float v[8] = {acc[2 * p][0], acc[2 * p][1], acc[2 * p][2], acc[2 * p][3],
                    acc[2 * p + 1][0], acc[2 * p + 1][1], acc[2 * p + 1][2], acc[2 * p + 1][3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) swap16f(v[e], v[4 + e]);
      const int co = co_w + 32 * p + 16 * (g & 1) + 8 * (g >> 1);
      if constexpr (DG) {
        const bf16x8 r8 = __builtin_bit_cast(bf16x8, rc[p]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = ((mc[p] >> e) & 1u) ? v[e] + (float)r8[e] : 0.f;
          o[e] = (bf16_t)f;
          if (rok) csum[p][e] += (float)o[e];        // sums of what is STORED (what the consumer reads back)
        }
        if (rok) *reinterpret_cast<bf16x8*>(yp + yrow + co) = o;
        continue;
      }
      const float4 s0 = *reinterpret_cast<const float4*>(coef + co), s1 = *reinterpret_cast<const float4*>(coef + co + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(coef + k.cout + co), h1 = *reinterpret_cast<const float4*>(coef + k.cout + co + 4);
      const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = v[e] * sc[e] + sh[e];
      if (has_res) {
        const float4 q0 = *reinterpret_cast<const float4*>(coef + 2 * k.cout + co), q1 = *reinterpret_cast<const float4*>(coef + 2 * k.cout + co + 4);
        const float rs[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const bf16x8 r8 = __builtin_bit_cast(bf16x8, rc[p]);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += (float)r8[e] * rs[e];
      }
      uint32_t bits = 0;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (k.relu) {
          bits |= (f[e] > 0.f ? 1u : 0u) << e;
          f[e] = f[e] > 0.f ? f[e] : 0.f;
        }
        o[e] = (bf16_t)f[e];
      }
      if (rok) {
        *reinterpret_cast<bf16x8*>(yp + yrow + co) = o;
        if (k.bits) k.bits[(int64_t)m * (k.cout >> 3) + (co >> 3)] = (uint8_t)bits;
      }
    }
    if constexpr (XDB) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xc[ks] = xn[ks];
    }
#pragma unroll
    for (int p = 0; p < NF / 2; ++p) rc[p] = rn[p];
    if constexpr (DG) {
#pragma unroll
      for (int p = 0; p < NF / 2; ++p) mc[p] = mn[p];
    }
  }
  if constexpr (DG) {
    // this wave's partial row: fold the 16 pixel lanes (fixed butterfly order), lanes l15 == 0 write 8 channels each
    float* row = k.parts + (int64_t)wi * k.cout * 2;
#pragma unroll
    for (int p = 0; p < NF / 2; ++p) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = csum[p][e];
#pragma unroll
        for (int sft = 1; sft < 16; sft <<= 1) t += __shfl_xor(t, sft);
        csum[p][e] = t;
      }
      if (l15 == 0) {
        const int co = co_w + 32 * p + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<float2*>(row + (co + e) * 2) = make_float2(csum[p][e], 0.f);
      }
    }
  }
}

// workgroups of a launch: one resident generation (8 waves per CU either way), every co group the same number of waves
inline int pw_blocks(int M, int CG, int NT) {
  const int WPB = NT / 64;
  const int ntiles = (M + 15) / 16;
  int blocks = 256 * (NT == 512 ? 1 : 2);
  while ((blocks * WPB) % CG) --blocks;            // (CG = 5: 255 workgroups, not a second generation of four)
  int need = (ntiles * CG + WPB - 1) / WPB;
  while ((need * WPB) % CG) ++need;
  return blocks < need ? blocks : need;
}

template <int NF, int KS, int CG, int NT, bool XDB, int MODE = 0>
int pw_launch(const PwK& k, hipStream_t s) {
  const size_t lds = (size_t)KS * k.cout * 64 + (size_t)3 * k.cout * 4;
  static bool attr_set = false;             // > 64 KB of dynamic LDS needs the opt-in (idempotent, set once per process)
  if (lds > 64 * 1024 && !attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_fused_kernel<NF, KS, CG, NT, XDB, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return SFK_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_pw_fused_kernel<NF, KS, CG, NT, XDB, MODE>), dim3(pw_blocks(k.M, CG, NT)), dim3(NT), lds, s, k);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Two-source pointwise pass for the narrowest maps (sfk_conv_pw_dual): y = x1 w1^T + x2 w2^T + bias with 8 or 16 output
// channels.  An MFMA tile of 16 output channels is all epilogue here (and half empty at 8); this is a plain streaming
// kernel: a thread owns a pixel -- its x1 / x2 rows are 16-byte loads, its output row ONE 16 / 32-byte store --, the two
// filters sit in LDS as packed bf16 pairs (every lane reads the same address: a broadcast), products through
// v_dot2c_f32_bf16 (two bf16 x bf16 products, exact in fp32, into an fp32 accumulator).  U pixels per thread are in flight.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

template <int C1, int C2, int CO>
__global__ __launch_bounds__(256, 4) void pw_dual_kernel(const bf16_t* __restrict__ x1, int ld1, int off1,
                                                      const bf16_t* __restrict__ x2, int ld2, int off2,
                                                      const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                      const float* __restrict__ bias, bf16_t* __restrict__ y, int ldy, int offy,
                                                      int64_t pixels) {
  constexpr int KP = (C1 + C2) / 2;                     // bf16 pairs per output channel
  __shared__ __attribute__((aligned(16))) uint32_t wl[CO * KP];   // [co][pair]: w1 row then w2 row
  __shared__ float bl[CO];
  for (int i = threadIdx.x; i < CO * KP; i += 256) {
    const int co = i / KP, kp = i % KP;
    const bf16_t* src = kp < C1 / 2 ? w1 + co * C1 + 2 * kp : w2 + co * C2 + 2 * (kp - C1 / 2);
    wl[i] = *reinterpret_cast<const uint32_t*>(src);
  }
  if (threadIdx.x < CO) bl[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  constexpr int U = 1;         // pixels per thread in flight (two made hipcc spill 82 registers on the 8-channel variant)
  for (int64_t p0 = (int64_t)blockIdx.x * 256 * U + threadIdx.x; p0 < pixels; p0 += (int64_t)gridDim.x * 256 * U) {
    uint4 a[U][C1 / 8], b[U][C2 / 8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * 256 < pixels ? p0 + u * 256 : p0;      // past the end: re-read p0 (not stored)
#pragma unroll
      for (int i = 0; i < C1 / 8; ++i) a[u][i] = *reinterpret_cast<const uint4*>(x1 + p * ld1 + off1 + 8 * i);
#pragma unroll
      for (int i = 0; i < C2 / 8; ++i) b[u][i] = *reinterpret_cast<const uint4*>(x2 + p * ld2 + off2 + 8 * i);
    }
    // eight output channels at a time (one 16-byte store each); the laundered offset after every channel keeps the compiler
    // from hoisting ALL filter reads above the first product (it did: 512 VGPRs + 900 B of scratch)
    uint32_t woff = 0;           // (always 0; laundered after every channel so that the next channel's filter reads depend on it)
    for (int c8 = 0; c8 < CO; c8 += 8) {
      float acc[U][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int co = c8 + j;
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u][j] = bl[co];
#pragma unroll
        for (int q = 0; q < KP / 4; ++q) {                 // four pairs (16 bytes of filter) per LDS read
          const uint4 wv = *reinterpret_cast<const uint4*>(&wl[co * KP + 4 * q + woff]);
          const uint32_t wp[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint4 xv = q < C1 / 8 ? a[u][q < C1 / 8 ? q : 0] : b[u][q >= C1 / 8 ? q - C1 / 8 : 0];
            const uint32_t xp[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[u][j] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xp[e]), __builtin_bit_cast(bf16x2_t, wp[e]),
                                                          acc[u][j], false);
          }
        }
        asm volatile("" : "+v"(woff) : "v"(acc[0][j]));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t p = p0 + u * 256;
        if (p < pixels) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)acc[u][e];
          *reinterpret_cast<bf16x8*>(y + p * ldy + offy + c8) = o;
        }
      }
    }
  }
}

inline bool pw_dual_map_ok(const sfk_fmap* f) {
  return sfk_fmap_ok(f) && f->dtype == SFK_BF16 && (f->c % 8) == 0 && (f->ld % 8) == 0 && (f->c_off % 8) == 0 &&
         (((uintptr_t)f->ptr) & 15) == 0;
}

}  // namespace

// Called by sfk_conv_igemm's dispatch (conv_igemm.hip) for descriptors it has validated: bf16, one tap at the pixel itself,
// stride 1, rows = the pixels of y in order, fused output transform with a shortcut or ReLU.  Returns SFK_ERR_UNSUPPORTED
// for shapes this kernel is not built for (the caller then runs the implicit-GEMM kernel).
int sfk_conv_pw_fused(const sfk_conv_desc* d, hipStream_t s) {
  const sfk_conv_epilogue& e = d->ep;
  const int K = d->cin, C = d->cout;
  if (K > 128 || (K % 8) != 0) return SFK_ERR_UNSUPPORTED;
  // every map is addressed through a 32-bit buffer resource (y too: the old rows of a += pass are READ through one); a map
  // of 4 GiB or more would wrap silently -- the implicit-GEMM kernels, which write through 64-bit pointers, take those
  if (sfk_fmap_bytes(&d->y) >= (1ll << 32) - 64 || sfk_fmap_bytes(&d->x) >= (1ll << 32) - 64) return SFK_ERR_UNSUPPORTED;
  const int KS = (K + 31) / 32;
  PwK k;
  k.x = d->x.ptr; k.y = d->y.ptr; k.w = d->w; k.res = e.res.ptr;
  k.scale = e.scale; k.shift = e.shift; k.rscale = e.res_scale; k.rshift = e.res_shift;
  k.bits = e.relu_bits; k.mbits = nullptr; k.parts = nullptr;
  k.xld = d->x.ld; k.xoff = d->x.c_off; k.yld = d->y.ld; k.yoff = d->y.c_off;
  k.rld = e.res.ptr ? e.res.ld : 0; k.roff = e.res.ptr ? e.res.c_off : 0;
  k.M = (int)sfk_fmap_pixels(&d->y); k.K = K; k.cout = C; k.relu = e.relu;
  k.xbytes = (uint32_t)sfk_fmap_bytes(&d->x); k.ybytes = (uint32_t)sfk_fmap_bytes(&d->y);
  k.rbytes = e.res.ptr ? (uint32_t)sfk_fmap_bytes(&e.res) : 0u;
  if (d->accumulate) {          // y += : the old rows come in as the shortcut of the output transform
    if (k.res) return SFK_ERR_UNSUPPORTED;
    k.res = d->y.ptr; k.rld = d->y.ld; k.roff = d->y.c_off; k.rbytes = k.ybytes;
  }
  // plain / += / + bias passes of the block tail's backward with K = C (64.6 -> 45.4 us on slow res2: 6.8 TB/s)
  if (C == 64 && KS == 2) return pw_launch<4, 2, 1, 256, true>(k, s);
  if (C == 128 && KS == 4) return pw_launch<8, 4, 1, 256, true>(k, s);
  if (C == 32 && KS == 1) return pw_launch<2, 1, 1, 256, true>(k, s);
  if (C == 64 && KS == 1) return pw_launch<4, 1, 1, 256, true>(k, s);
  if (C == 128 && KS == 1) return pw_launch<8, 1, 1, 256, true>(k, s);
  // 128 channels per wave (16 fragments spill at 256 VGPRs): wider outputs are split over co groups of waves, each of
  // which reads the (small) X rows again -- from L1 / L2
  if (C == 256 && KS == 2) return sfk_tune().igemm_pw_stream == 2 ? pw_launch<4, 2, 4, 256, true>(k, s) : pw_launch<8, 2, 2, 256, true>(k, s);
  // 134 KB of filter: one 8-wave workgroup per CU; 64 channels per wave (8 fragments at KS = 4 spill), the eight co groups
  // of a workgroup read the same X rows (L1)
  if (C == 512 && KS == 4) return pw_launch<4, 4, 8, 512, true>(k, s);
  // 320 = 5 co groups of 64 channels (84 KB of filter: one 8-wave workgroup per CU)
  if (C == 320 && KS == 4 && !k.res) return pw_launch<4, 4, 5, 512, true>(k, s);
  return SFK_ERR_UNSUPPORTED;
}

// The data-gradient flavour (DG): shapes (dY channels -> dX channels) (64 -> 256) and (128 -> 512), the pointwise conv_a of the
// slow pathway's res2 / res3.  sfk_conv_pw_dgrad_rows: partial rows such a launch leaves (0: not a shape of this kernel).
int sfk_conv_pw_dgrad_rows(const sfk_conv_desc* d) {
  if (sfk_fmap_bytes(&d->y) >= (1ll << 32) - 64 || sfk_fmap_bytes(&d->x) >= (1ll << 32) - 64) return 0;   // 32-bit buffer resources
  const int M = (int)sfk_fmap_pixels(&d->y);
  if (d->cin == 64 && d->cout == 256) return pw_blocks(M, 2, 256) * 4 / 2;
  if (d->cin == 128 && d->cout == 512) return pw_blocks(M, 8, 512) * 8 / 8;
  return 0;
}

int sfk_conv_pw_dgrad(const sfk_conv_desc* d, hipStream_t s) {
  PwK k;
  k.x = d->x.ptr; k.y = d->y.ptr; k.w = d->w; k.res = d->y.ptr;          // in place: old rows + result
  k.scale = k.shift = k.rscale = k.rshift = nullptr;
  k.bits = nullptr; k.mbits = d->out_relu_bits; k.parts = d->bnb.partials;
  k.xld = d->x.ld; k.xoff = d->x.c_off; k.yld = d->y.ld; k.yoff = d->y.c_off; k.rld = d->y.ld; k.roff = d->y.c_off;
  k.M = (int)sfk_fmap_pixels(&d->y); k.K = d->cin; k.cout = d->cout; k.relu = 0;
  k.xbytes = (uint32_t)sfk_fmap_bytes(&d->x); k.ybytes = (uint32_t)sfk_fmap_bytes(&d->y); k.rbytes = k.ybytes;
  if (d->cin == 64 && d->cout == 256) return pw_launch<8, 2, 2, 256, true, 1>(k, s);
  if (d->cin == 128 && d->cout == 512) return pw_launch<4, 4, 8, 512, true, 1>(k, s);
  return SFK_ERR_UNSUPPORTED;
}

extern "C" int sfk_conv_pw_dual_supported(const sfk_fmap* x1, const sfk_fmap* x2, const sfk_fmap* y) {
  if (!x1 || !x2 || !y || !pw_dual_map_ok(x1) || !pw_dual_map_ok(x2) || !pw_dual_map_ok(y)) return 0;
  if (sfk_fmap_pixels(x1) != sfk_fmap_pixels(y) || sfk_fmap_pixels(x2) != sfk_fmap_pixels(y)) return 0;
  return (x1->c == 32 && x2->c == 8 && y->c == 8) || (x1->c == 64 && x2->c == 16 && y->c == 16);
}

extern "C" int sfk_conv_pw_dual(const sfk_fmap* x1, const void* w1, const sfk_fmap* x2, const void* w2, const float* bias,
                                const sfk_fmap* y, sfk_stream_t stream) {
  if (!x1 || !x2 || !y || !w1 || !w2 || !sfk_fmap_ok(x1) || !sfk_fmap_ok(x2) || !sfk_fmap_ok(y)) return SFK_ERR_INVALID;
  if (!sfk_conv_pw_dual_supported(x1, x2, y)) return SFK_ERR_UNSUPPORTED;
  if ((((uintptr_t)w1) | ((uintptr_t)w2)) & 3) return SFK_ERR_INVALID;        // the filters are read as packed bf16 pairs
  const int64_t px = sfk_fmap_pixels(y);
  int64_t blocks = (px + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bf16_t *a = static_cast<const bf16_t*>(x1->ptr), *b = static_cast<const bf16_t*>(x2->ptr);
  const bf16_t *u = static_cast<const bf16_t*>(w1), *v = static_cast<const bf16_t*>(w2);
  bf16_t* o = static_cast<bf16_t*>(y->ptr);
  if (y->c == 8)
    hipLaunchKernelGGL((pw_dual_kernel<32, 8, 8>), dim3((unsigned)blocks), dim3(256), 0, s, a, x1->ld, x1->c_off, b, x2->ld, x2->c_off,
                       u, v, bias, o, y->ld, y->c_off, px);
  else
    hipLaunchKernelGGL((pw_dual_kernel<64, 16, 16>), dim3((unsigned)blocks), dim3(256), 0, s, a, x1->ld, x1->c_off, b, x2->ld, x2->c_off,
                       u, v, bias, o, y->ld, y->c_off, px);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}
