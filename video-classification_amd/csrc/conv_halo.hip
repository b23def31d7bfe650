// conv_halo: the (1,3,3) stride-1 "same" convolution of slow res2 (conv_b: 64 -> 64 over 56 x 56 frames), forward and data
// gradient, bf16.
//
// On the implicit-GEMM kernels this layer ran at 0.19 of the HBM roof (131 us for 206 MB): every tap gathers its pixels again
// through the texture path -- nine times the map, in 64-byte pieces, with ~7 vector instructions of gather arithmetic per MFMA
// (PMC: VALU : MFMA = 8 : 1, MFMA pipe 20 % busy, no bank conflicts, L2 hit rate 91 %).  Here a PERSISTENT workgroup (4 waves,
// one per SIMD, the whole 512-entry register file each) keeps
//   * the FILTER in registers: a wave owns 32 output channels = 2 co fragments x 18 k-steps = 36 MFMA A operands, 144 VGPRs,
//     loaded once per workgroup -- no filter tile in LDS, no per-K-step filter traffic at all;
//   * a BAND of the input in LDS: 4 output rows of one frame plus one halo row above and below, every row with a zero pixel
//     either side (6 rows x 64 pixel slots x 128 B = 48 KiB, staged once by LDS-DMA; padding pixels and rows outside the frame
//     are out-of-range lanes, which the DMA writes as zeros), double-buffered: the next band's DMAs ride on the first k-steps
//     of this one;
//   * all nine taps as shifted reads of that one image: tap (dh, dw) of fragment j is the ds_read_b128 at
//     addr[dw][j][k-step] + (dh + 1) * row pitch.  The row pitch is a multiple of 8 pixels, so the XOR swizzle (a function of
//     the pixel index mod 8) survives a vertical shift and the address registers depend on dw only; they are all precomputed,
//     the main loop has NO address arithmetic, NO DMA wait and NO barrier: 18 k-steps of 7 ds_read_b128 + 14 MFMAs, the reads
//     of step n + 1 in flight under the MFMAs of step n.
// L2 -> LDS traffic is 1.5 x the map instead of 9 x, LDS reads 0.5 per MFMA.  Rows of the band are the output pixels in order, so
// the shared epilogue (conv_igemm_epi.h: 16-byte channels-last stores, BatchNorm partial sums per band) applies as it is.
// (A first version streamed the filter through a 3-slot LDS ring, one [co][64 ci] slab per tap: 106 us -- with one wave per
// SIMD every DMA issue and every barrier of the K loop is paid in full; it also served 128 -> 128 over 28 x 28 frames, where it
// only tied the implicit GEMM (88 vs 86 us) and was dropped.)
//
// Swizzle (source side of the DMA, rule "both sides or neither"): a pixel record is 128 B = 8 sixteen-byte slots, two pixels
// per 256-byte bank row; physical slot = logical ^ f(pixel), f = ((pixel >> 1) & 3) << 1 -- even values only, so the two
// k-groups a 16-lane read group mixes (slots L and L ^ 1) never meet, and the eight pixels per group and k-group differ in f
// for ANY alignment of the 16 consecutive pixels (the taps shift it).
#include "conv_igemm_epi.h"

namespace sfk_igemm {

typedef __attribute__((address_space(3))) void lds_void_h_t;

struct HaloTaps { int widx[9]; };   // filter slice of tap (dh, dw) at [(dh + 1) * 3 + (dw + 1)]

template <int EPI>
__global__ __launch_bounds__(256, 1) void conv_halo_kernel(const ConvK k, const HaloTaps ht, const int nbands, const int bands_per_frame) {
  using T = bf16_t;
  constexpr int C = 64, W = 56;
  constexpr int R = 4;                         // output rows per band
  constexpr int PW = 64;                       // pixel slots per band row (>= W + 2, a multiple of 8)
  constexpr int PB = C * 2;                    // bytes per pixel record
  constexpr int ROWBYTES = PW * PB;            // 8 KiB
  constexpr int BAND = (R + 2) * ROWBYTES;     // 48 KiB
  constexpr int NBI = BAND / 1024 / 4;         // band DMA instructions per wave (12), 8 pixels each
  constexpr int NK = 18;                       // k-steps: tap (canonical (dh, dw) order) x 2
  constexpr int BM = R * W;                    // output pixels per band (224): waves 2 (pixels) x 2 (co), 7 x 2 fragments each
  constexpr int RED0 = 2 * BAND;
  constexpr uint32_t FAR = 0x80000000u;
  __shared__ __attribute__((aligned(16))) char smem[RED0 + 2048];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int l15 = lane & 15, g4 = lane >> 4;
  // XCD-aware band order (workgroups b, b + 8, ... share an L2): neighbouring bands share their halo rows
  const int b0 = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  if (b0 >= nbands) return;
  const __amdgpu_buffer_rsrc_t xrs = sfk_make_rsrc(k.x, k.xbytes);
  auto fsw = [](int q) { return ((q >> 1) & 3) << 1; };

  // ---- band DMA lanes: instruction i = wave * NBI + ii covers band pixels 8 i + lane / 8 (one band row: wave-uniform)
  uint32_t bsrc[NBI];                          // source offset relative to the band's first output pixel, biased by (W + 1) pixels; FAR = pad
#pragma unroll
  for (int ii = 0; ii < NBI; ++ii) {
    const int qd = (wave * NBI + ii) * 8 + (lane >> 3);
    const int ri = qd / PW, wi = qd % PW;
    const int ls = (lane & 7) ^ fsw(qd);
    bsrc[ii] = (wi >= 1 && wi <= W) ? (uint32_t)((ri * W + wi) * k.xld * 2 + k.xoff * 2 + ls * 16) : FAR;
  }
  int nb_base = 0, nb_r0 = 0;
  auto band_begin = [&](const int b_) __attribute__((always_inline)) {
    const int b = b_ < nbands ? b_ : 0;        // past the end: re-stage band 0, nobody reads it
    const int fr = b / bands_per_frame;
    nb_r0 = (b - fr * bands_per_frame) * R;
    // the un-biasing term (may be negative: it only ever meets a lane whose row is inside the frame, and then the sum is a valid
    // non-negative offset)
    nb_base = ((fr * k.xh + nb_r0) * W - W - 1) * k.xld * 2;
  };
  auto issue_band = [&](const int buf, const int first, const int count) __attribute__((always_inline)) {
    char* dst = smem + buf * BAND + wave * NBI * 1024;
#pragma unroll
    for (int ii = first; ii < first + count; ++ii) {
      const int ri = (wave * NBI + ii) * 8 / PW;                   // wave-uniform
      const bool rok = (unsigned)(nb_r0 - 1 + ri) < (unsigned)k.xh;
      const uint32_t vo = (rok && bsrc[ii] != FAR) ? (uint32_t)((int)bsrc[ii] + nb_base) : FAR;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_h_t*)(dst + ii * 1024), 16, (int)vo, 0, 0, 0);
    }
  };
  band_begin(b0);
  issue_band(0, 0, NBI);

  // ---- the filter, once: MFMA A operand of (co fragment i, k-step n = 2 tap + s) = w[wn*32 + 16 i + l15][widx(tap)][32 s + 8 g4 ..]
  bf16x8 wreg[2][NK];
  {
    const bf16_t* wp = static_cast<const bf16_t*>(k.w);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int co = wn * 32 + 16 * i + l15;
#pragma unroll
      for (int n = 0; n < NK; ++n)
        wreg[i][n] = *reinterpret_cast<const bf16x8*>(wp + ((int64_t)co * k.wtaps + ht.widx[n >> 1]) * k.cin + 32 * (n & 1) + 8 * g4);
    }
  }
  // ---- fragment read addresses, dh = -1: [band buffer][dw + 1][pixel fragment][k-step within the tap] (84 registers: with
  // one set and the buffer as an immediate the largest offset, 48 KiB + 2 rows, is one past the 16-bit field)
  uint32_t xa[2][3][7][2];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int p = wm * 112 + 16 * j + l15;
    const int q0 = (p / W) * PW + (p % W) + 1;          // band pixel of output pixel p at tap (dh, dw) = (-1, 0)
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const int q = q0 + dw - 1;
#pragma unroll
      for (int sk = 0; sk < 2; ++sk)
#pragma unroll
        for (int bf = 0; bf < 2; ++bf) xa[bf][dw][j][sk] = (uint32_t)(bf * BAND + q * PB + (((4 * sk + g4) ^ fsw(q)) << 4));
    }
  }

  f32x4 acc[2][7];
  bf16x8 xfr[2][7];                       // k-step n multiplies out of set n & 1 while the reads of n + 1 land in the other
  // the next band's DMA pieces ride on the first k-steps, 2 per wave and step
  constexpr int BPK = 2, BKS = NBI / BPK;
  static_assert(NBI % BPK == 0 && BKS <= NK, "band pieces");

  // one band out of buffer `buf` (a literal: the loop below is unrolled over the two buffers)
  auto band_body = [&](const int buf, const int b, const bool first) __attribute__((always_inline)) {
    auto rd_x = [&](const int set, const int n) __attribute__((always_inline)) {
      const int tap = n >> 1, dh = tap / 3 - 1, dw = tap % 3;
#pragma unroll
      for (int j = 0; j < 7; ++j) xfr[set][j] = *reinterpret_cast<const bf16x8*>(smem + xa[buf][dw][j][n & 1] + (dh + 1) * ROWBYTES);
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    band_begin(b + (int)gridDim.x);
    // this band has landed (every wave waited for its own pieces before the previous epilogue, or right here for the first
    // band), and every wave is through with the other buffer: the next band's pieces may overwrite it
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    rd_x(0, 0);
#pragma unroll
    for (int n = 0; n < NK; ++n) {
      if (n < BKS) issue_band(buf ^ 1, n * BPK, BPK);
      if (n + 1 < NK) rd_x((n + 1) & 1, n + 1);
      // MFMAs as asm with the filter operand and the accumulator in the accumulator file ("a"): the 144 filter registers cannot
      // live in the 256 arch VGPRs beside everything else, and left to itself hipcc parks them in AGPRs and copies each
      // fragment back (4 v_accvgpr_read per use) instead of letting the MFMA read it there
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "a"(wreg[i][n]), "v"(xfr[n & 1][j]));
    }
    // the next band's pieces have had 12+ k-steps: waited for here, BEFORE the epilogue's stores enter the vmcnt queue;
    // the nops cover the last MFMA's result latency for the compiler-generated readers of the accumulators (asm MFMAs carry none)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 7" ::: "memory");
    // ---- epilogue of this band.  BatchNorm partial sums FIRST, with raw barriers: the shared epilogue's __syncthreads() also
    // waits for vmcnt(0), i.e. for the band's output stores just issued -- with one persistent workgroup per CU nobody covers
    // that round trip, and it was 7,200 of a band's 13,300 cycles.  The stores go last and drain under the next band's K loop.
    if (k.stats) {
      float* red = reinterpret_cast<float*>(smem + RED0);                 // [2 pixel waves][64 co][2]
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int j = 0; j < 7; ++j) {
            const float v = acc[i][j][r];
            s1 += v;
            s2 += v * v;
          }
          s1 = row16_sum(s1);
          s2 = row16_sum(s2);
          if (l15 == 15) {
            const int col = wn * 32 + 16 * i + 4 * g4 + r;
            red[(wm * 64 + col) * 2 + 0] = s1;
            red[(wm * 64 + col) * 2 + 1] = s2;
          }
        }
      __builtin_amdgcn_s_waitcnt(0xC07F);                                 // lgkmcnt(0): the LDS writes have executed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (tid < 64) {
        float* o = k.stats + ((int64_t)b * k.cout + tid) * 2;
        o[0] = red[tid * 2 + 0] + red[(64 + tid) * 2 + 0];
        o[1] = red[tid * 2 + 1] + red[(64 + tid) * 2 + 1];
      }
      // (red is written again only behind the next band's barrier)
    }
    // channels-last stores: rows = the band's BM output pixels in order (tile index = band index)
    ConvK ks = k;
    ks.stats = nullptr;
    epilogue_plain<T, EPI, 7, 2, BM, C, 2, 2, true>(ks, acc, nullptr, b, 0, wm, wn, lane, tid);
  };
  for (int b = b0;;) {
    band_body(0, b, b == b0);
    b += gridDim.x;
    if (b >= nbands) break;
    band_body(1, b, false);
    b += gridDim.x;
    if (b >= nbands) break;
  }
}

// eligibility: bf16, (1,3,3) "same" stride-1 taps over frames of 56 x 56 x 64, cout = cin = 64, plain epilogue
__attribute__((visibility("hidden"))) bool halo_ok(const sfk_conv_desc* d) {
  if (!sfk_tune().igemm_halo || d->x.dtype != SFK_BF16) return false;
  if (d->cin != 64 || d->cout != 64 || d->x.w != 56) return false;
  if (d->x.h % 4 != 0 || d->x.t != d->y.t || d->x.h != d->y.h || d->x.w != d->y.w || d->rt != d->y.t || d->rh != d->y.h || d->rw != d->y.w)
    return false;
  for (int a = 0; a < 3; ++a)
    if (d->gs[a] != 1 || d->os[a] != 1 || d->oo[a] != 0) return false;
  if (d->ntaps != 9) return false;
  int seen = 0;
  for (int i = 0; i < 9; ++i) {
    if (d->taps[i].dt != 0 || d->taps[i].dh < -1 || d->taps[i].dh > 1 || d->taps[i].dw < -1 || d->taps[i].dw > 1) return false;
    seen |= 1 << ((d->taps[i].dh + 1) * 3 + d->taps[i].dw + 1);
  }
  if (seen != 0x1FF) return false;                    // every (dh, dw) of the 3 x 3 window exactly once
  if (d->ep.scale || d->ep.shift || d->bnb.partials || d->out_relu_bits || d->accumulate) return false;
  if ((d->y.ld % 8) || (d->y.c_off % 8) || !sfk_tune().igemm_wide_store) return false;
  return sfk_fmap_bytes(&d->x) < 0x7FF00000ll;
}

__attribute__((visibility("hidden"))) int halo_mtiles(const sfk_conv_desc* d) { return d->x.n * d->x.t * (d->x.h / 4); }

__attribute__((visibility("hidden"))) int launch_halo(const ConvK& k, const sfk_conv_desc* d, hipStream_t s) {
  const int bpf = d->x.h / 4, nbands = d->x.n * d->x.t * bpf;
  const int grid = nbands < 256 ? nbands : 256;
  HaloTaps ht;
  for (int i = 0; i < 9; ++i) ht.widx[(d->taps[i].dh + 1) * 3 + d->taps[i].dw + 1] = d->taps[i].widx;
  hipLaunchKernelGGL((conv_halo_kernel<0>), dim3((unsigned)grid), dim3(256), 0, s, k, ht, nbands, bpf);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

}  // namespace sfk_igemm
