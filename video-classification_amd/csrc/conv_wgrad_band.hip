// conv_wgrad_band: the filter gradient of the (1,3,3) stride-1 "same" convs whose whole dW fits the accumulators of ONE
// workgroup -- slow res2 conv_b (64 -> 64 over 56 x 56 frames): dW = 64 x 576 fp32 = 144 KB -- or of two: slow res3 conv_b
// (128 -> 128 over 28 x 28), one workgroup per input-channel half (128 x 576 each; BandCfg<28, 128, 128>, described at the end).
//
//   dW[co][(dh, dw), ci] = sum over pixels  dY[pixel][co] * X[pixel + (dh, dw)][ci]
//
// The implicit-GEMM filter-gradient kernels walk the pixel axis in K-steps and GATHER X per tap: every X row enters LDS nine
// times and the 576 columns are three column tiles that each re-read dY (154 us, 0.17 of the HBM roof, a third of the LDS
// cycles in bank conflicts).  Here a persistent workgroup owns a BAND of 4 output rows at a time (conv_halo.hip is the forward
// of the same idea): the band's dY (224 pixels) and the 6 input rows around it (one zero pixel either side of a row: padding =
// out-of-range DMA lanes) are staged ONCE, double-buffered over bands, and all nine taps are transposed reads of that one X
// image at an immediate offset.  No gather arithmetic, no wait and no barrier inside a band; one barrier per band.
//
// LDS image of a band (one of two buffers): PLANES of [pixel][16 channels] (32 bytes per pixel): 4 X planes (the workgroup's
// 64 input channels) of (R + 2) * (W + 2) pixel slots, then the dY planes of R * W pixels.  A transposed read
// (ds_read_b64_tr_b16: each 16-lane group turns 4 pixel rows x 16 channels into 4 consecutive K values of one channel per
// lane) then touches 8 CONSECUTIVE 32-byte slots per 32-lane half -- conflict-free at any alignment, so a tap is a pure
// address offset ((dh + 1) * (W + 2) + (dw + 1)) * 32 with no swizzle to undo.  The K index of the MFMA is a permutation of
// the pixel: lane group g4 holds pixels 32 s + 4 g4 + {0..3} (first read) and 32 s + 16 + 4 g4 + {0..3} (second read) of
// K-step s, the same for both operands; 4-pixel groups never straddle an image row (W % 4 == 0).
//
// Waves: 8 = 2 x 4.  wc = wave & 3 owns input-channel block wc (16 channels) of all nine taps, i.e. 9 column fragments, times
// 4 output-channel fragments: 36 accumulator fragments = 144 registers.  The two waves of a SIMD (kg = wave >> 2) hold the
// SAME outputs and alternate K-steps (global K-step counter parity), so that one multiplies (36 MFMAs at raised priority)
// while the other issues its 26 transposed reads; their sums meet through LDS once, at the end.  The workgroup's partial dW
// goes to the caller's workspace, band_reduce_kernel adds the workgroups' partials in a fixed order (deterministic).
// The next band's LDS-DMA instructions are issued a few per K-step behind the fragment reads; the MFMA priority is 2 for kg = 0
// and 1 for kg = 1 (equal priorities interleave the two waves' MFMA blocks, both finish together and then both sit in their
// reads).  Band order is XCD-aware (neighbouring bands share halo rows: HBM reads = 1.0 x the operands, no LDS bank conflicts).
//
// 128 channels (BandCfg<28, 128, 128>): a band of 4 x 28 = 112 pixels = 3.5 K-steps (the last one half-filled: zero fragments),
// a unit = (band, input-channel half); kg picks the wave's four output-channel fragments (8 dY planes), every wave runs every
// K-step and its address registers move to the other buffer after each band; no exchange at the end.
#include "conv_wgrad_common.h"

#ifndef SFK_BAND_EXP
#define SFK_BAND_EXP 0      // timing experiments (tools/gpu_wband.sh; results are wrong): 1 no MFMAs, 2 no per-band DMAs, 3 no fragment reads
#endif

namespace sfk_wgrad {

typedef __attribute__((address_space(3))) void lds_void_wb_t;

struct BandK {
  int nunits;        // bands (x input-channel halves)
  int bpf;           // bands per frame
  int tpos[9];       // position of tap (dh + 1) * 3 + (dw + 1) in WgradK::taps
};

template <int W_, int CI_, int CO_>
struct BandCfg {
  static constexpr int W = W_, R = 4, PW = W + 2, NPX = R * W;
  static constexpr int NS = (NPX + 31) / 32;                      // K-steps of 32 pixels
  static constexpr bool HALF_LAST = (NPX % 32) != 0;              // ... the last one holds 16
  static constexpr int XPX = ((R + 2) * PW + 31) / 32 * 32;       // pixel slots of an X plane (352)
  static constexpr int DPX = NS * 32;                             // ... of a dY plane (224)
  static constexpr int NXP = 4, NDP = CO_ / 16;
  static constexpr int XCH = XPX / 32, DCH = DPX / 32;            // 1 KiB DMA chunks per plane
  static constexpr int XPLANE = XPX * 32, DPLANE = DPX * 32;
  static constexpr int XBYTES = NXP * XPLANE;
  static constexpr int BUF = XBYTES + NDP * DPLANE;               // 73,728
  static constexpr bool KSPLIT = CO_ == 64;
  static constexpr int NHALF = CI_ / 64;
  static constexpr int NXI = NXP * XCH, NDI = NDP * DCH;          // DMA wave-instructions per band: 44 + 28
  static constexpr int NXJ = (NXI + 7) / 8, NDJ = (NDI + 7) / 8;
  static constexpr int NFRAG = (CO_ / 16) * 36;                   // accumulator fragments of a workgroup's partial
  static_assert(W % 4 == 0 && NPX % 16 == 0, "4-pixel groups inside a row; whole half K-steps");
  static_assert(!KSPLIT || 2 * BUF >= 4 * 36 * 1024, "the end-of-kernel exchange reuses the band buffers");
};

template <int OFF>
__device__ __forceinline__ void tr_read(bf16x4& v, const uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read immediate");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

__device__ __forceinline__ bf16x8 frag8(const bf16x4& lo, const bf16x4& hi) {
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

#include "conv_wgrad_band_acc.inc"

// the 36 MFMAs of a K-step: fragment (co block c, tap t) = accumulator F = 4 t + c in fixed AGPRs for t < 8 (hipcc gives a
// 256-register wave at most 128 AGPRs); the ninth tap's four fragments are ordinary VGPR variables
template <int T>
__device__ __forceinline__ void band_tap_mfma(f32x4 (&acc8)[4], const bf16x8 (&fa)[4], const bf16x8& fb) {
  if constexpr (T < 8) {
    band_acc_mfma<4 * T + 0>(fa[0], fb);
    band_acc_mfma<4 * T + 1>(fa[1], fb);
    band_acc_mfma<4 * T + 2>(fa[2], fb);
    band_acc_mfma<4 * T + 3>(fa[3], fb);
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc8[c]) : "v"(fa[c]), "v"(fb));
  }
}

// one K-step S of the band in the buffer the address registers point into: 8 + 18 transposed reads, 36 MFMAs
// `dma()` issues this K-step's share of the NEXT band's LDS-DMAs behind the fragment reads (under their latency); `hi`: the
// wave's MFMA priority (2 for kg = 0, 1 for kg = 1: with equal priorities the two waves of a SIMD interleave their MFMA blocks,
// finish together and then both sit in their reads with the matrix pipe idle)
template <class Cfg, int S, class Dma>
__device__ __forceinline__ void band_kstep(f32x4 (&acc8)[4], const uint32_t abase, const uint32_t xb0, const uint32_t xb1, const bool hi,
                                           Dma&& dma) {
  constexpr bool HALF = Cfg::HALF_LAST && S == Cfg::NS - 1;
  bf16x4 af[4][2], bf[9][2];
#if SFK_BAND_EXP == 3
#pragma unroll
  for (int c = 0; c < 4; ++c) af[c][0] = af[c][1] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t) bf[t][0] = bf[t][1] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
  asm volatile("" : "+v"(af[0][0]), "+v"(bf[0][0]) : "v"(abase), "v"(xb0), "v"(xb1));
#else
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    switch (c) {      // (the immediates must be literals)
      case 0: tr_read<0 * Cfg::DPLANE + S * 1024>(af[0][0], abase); if (!HALF) tr_read<0 * Cfg::DPLANE + S * 1024 + 512>(af[0][1], abase); break;
      case 1: tr_read<1 * Cfg::DPLANE + S * 1024>(af[1][0], abase); if (!HALF) tr_read<1 * Cfg::DPLANE + S * 1024 + 512>(af[1][1], abase); break;
      case 2: tr_read<2 * Cfg::DPLANE + S * 1024>(af[2][0], abase); if (!HALF) tr_read<2 * Cfg::DPLANE + S * 1024 + 512>(af[2][1], abase); break;
      default: tr_read<3 * Cfg::DPLANE + S * 1024>(af[3][0], abase); if (!HALF) tr_read<3 * Cfg::DPLANE + S * 1024 + 512>(af[3][1], abase); break;
    }
  }
#define SFK_BAND_TAP(T9)                                                                  \
  tr_read<((T9 / 3) * Cfg::PW + (T9 % 3)) * 32>(bf[T9][0], xb0);                          \
  if (!HALF) tr_read<((T9 / 3) * Cfg::PW + (T9 % 3)) * 32>(bf[T9][1], xb1);
  SFK_BAND_TAP(0) SFK_BAND_TAP(1) SFK_BAND_TAP(2) SFK_BAND_TAP(3) SFK_BAND_TAP(4)
  SFK_BAND_TAP(5) SFK_BAND_TAP(6) SFK_BAND_TAP(7) SFK_BAND_TAP(8)
#undef SFK_BAND_TAP
#endif
  dma();
  if (HALF) {
#pragma unroll
    for (int c = 0; c < 4; ++c) af[c][1] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) bf[t][1] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the asm reads have returned (hipcc does not track them)
  __builtin_amdgcn_sched_barrier(0);
  if (hi) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
  // asm MFMAs on fixed AGPRs (conv_wgrad_band_acc.inc): from the builtin, and from "+a" operands, hipcc kept the 144 loop-carried
  // accumulators in VGPRs, copied them per K-step and spilled ~200 registers
  bf16x8 fa[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) fa[c] = frag8(af[c][0], af[c][1]);
#if SFK_BAND_EXP == 1
#pragma unroll
  for (int c = 0; c < 4; ++c) asm volatile("" ::"v"(fa[c]));
#pragma unroll
  for (int t = 0; t < 9; ++t) asm volatile("" ::"v"(bf[t][0]), "v"(bf[t][1]));
#else
  // the ninth tap FIRST: its accumulators are VGPR variables, and hipcc may move them right behind the K-step without the
  // result-latency padding an asm MFMA does not get (seen: three of the four fragments wrong); 32 MFMAs later they are written
  band_tap_mfma<8>(acc8, fa, frag8(bf[8][0], bf[8][1]));
  band_tap_mfma<0>(acc8, fa, frag8(bf[0][0], bf[0][1]));
  band_tap_mfma<1>(acc8, fa, frag8(bf[1][0], bf[1][1]));
  band_tap_mfma<2>(acc8, fa, frag8(bf[2][0], bf[2][1]));
  band_tap_mfma<3>(acc8, fa, frag8(bf[3][0], bf[3][1]));
  band_tap_mfma<4>(acc8, fa, frag8(bf[4][0], bf[4][1]));
  band_tap_mfma<5>(acc8, fa, frag8(bf[5][0], bf[5][1]));
  band_tap_mfma<6>(acc8, fa, frag8(bf[6][0], bf[6][1]));
  band_tap_mfma<7>(acc8, fa, frag8(bf[7][0], bf[7][1]));
#endif
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
}

template <class Cfg>
__global__ __launch_bounds__(512, 2) void conv_wgrad_band_kernel(const WgradK k, const BandK bk) {
  constexpr uint32_t FAR = 0x80000000u;
  constexpr int W = Cfg::W, R = Cfg::R, PW = Cfg::PW;
  static_assert(Cfg::KSPLIT ? Cfg::NS == 7 : Cfg::NS == 4, "the K-step lists below are written for the 56- and 28-wide bands");
  __shared__ __attribute__((aligned(16))) char smem[2 * Cfg::BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, wc = wave & 3;
  const int grid = gridDim.x, bid = blockIdx.x;
  // XCD-aware order: neighbouring bands (shared halo rows, the same frame) on one XCD's L2
  const int lb = (grid & 7) == 0 ? (bid & 7) * (grid >> 3) + (bid >> 3) : bid;
  const int nit = (bk.nunits - lb + grid - 1) / grid;          // units lb, lb + grid, ...  (grid <= nunits)
  const __amdgpu_buffer_rsrc_t xrs = sfk_make_rsrc(k.x, k.xbytes);
  const __amdgpu_buffer_rsrc_t drs = sfk_make_rsrc(k.dy, k.dbytes);

  // ---- DMA lanes.  Wave-instruction id = wave + 8 jj covers chunk id % CH (32 pixel slots) of plane id / CH; lane l moves the
  // 16 bytes of channel half l & 1 of slot l >> 1.  Offsets relative to the band's first output pixel, fixed for the kernel.
  int relx[Cfg::NXJ];
  uint32_t reld[Cfg::NDJ];
  uint32_t xnever = 0, xtop = 0, xbot = 0;      // bit jj: a padding slot / a slot of the row above the band / of the row below
  uint32_t dbad = 0;
#pragma unroll
  for (int jj = 0; jj < Cfg::NXJ; ++jj) {
    const int id = wave + 8 * jj;
    const int cb = id / Cfg::XCH, n = id % Cfg::XCH;
    const int xi = 32 * n + (lane >> 1), hf = lane & 1;
    const int ir = xi / PW, ic = xi % PW - 1;
    const bool ok = ic >= 0 && ic < W && ir < R + 2;
    relx[jj] = (((ir - 1) * W + ic) * k.xld + k.xoff + cb * 16 + hf * 8) * 2;
    xnever |= (uint32_t)!ok << jj;
    xtop |= (uint32_t)(ok && ir == 0) << jj;
    xbot |= (uint32_t)(ok && ir == R + 1) << jj;
  }
#pragma unroll
  for (int jj = 0; jj < Cfg::NDJ; ++jj) {
    const int id = wave + 8 * jj;
    const int cob = id / Cfg::DCH, n = id % Cfg::DCH;
    const int p = 32 * n + (lane >> 1), hf = lane & 1;
    reld[jj] = (uint32_t)((p * k.dld + k.doff + cob * 16 + hf * 8) * 2);
    dbad |= (uint32_t)(p >= Cfg::NPX) << jj;
  }
  // the band being fetched: scalars set once per band (band_next), its DMA slots (X instructions, then dY) issued in chunks
  int nx_xbase = 0, nx_dbase = 0, nx_buf = 0;
  uint32_t nx_bad = 0;            // bit jj: X slot jj of this lane reads padding in this band (branch-free: FAR is OR-ed in)
  bool nx_on = false;
  auto band_next = [&](const int u, const int buf, const bool on) __attribute__((always_inline)) {
    const int band = u / Cfg::NHALF, half = u % Cfg::NHALF;
    const int f = band / bk.bpf, hb = band - f * bk.bpf;
    const int pix0 = (f * k.xh + hb * R) * W;
    nx_bad = xnever | (hb > 0 ? 0u : xtop) | (hb < bk.bpf - 1 ? 0u : xbot);
    nx_xbase = pix0 * k.xld * 2 + half * 128; nx_dbase = pix0 * k.dld * 2;
    nx_buf = buf; nx_on = on && SFK_BAND_EXP != 2;
  };
  constexpr int NSLOT = Cfg::NXJ + Cfg::NDJ;
  auto issue_slots = [&](const int lo, const int hi_) __attribute__((always_inline)) {
    if (!nx_on) return;
    char* const base = smem + nx_buf * Cfg::BUF;
#pragma unroll
    for (int jj = 0; jj < Cfg::NXJ; ++jj) {
      const int id = wave + 8 * jj;
      if (jj >= lo && jj < hi_ && id < Cfg::NXI) {
        const uint32_t vo = (uint32_t)(nx_xbase + relx[jj]) | (((nx_bad >> jj) & 1u) << 31);      // (valid offsets < 0x7FF00000)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_wb_t*)(base + (id / Cfg::XCH) * Cfg::XPLANE + (id % Cfg::XCH) * 1024), 16,
                                                 (int)vo, 0, 0, 0);
      }
    }
#pragma unroll
    for (int jj = 0; jj < Cfg::NDJ; ++jj) {
      const int id = wave + 8 * jj;
      if (Cfg::NXJ + jj >= lo && Cfg::NXJ + jj < hi_ && id < Cfg::NDI) {
        const uint32_t vo = ((uint32_t)nx_dbase + reld[jj]) | (((dbad >> jj) & 1u) << 31);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void_wb_t*)(base + Cfg::XBYTES + (id / Cfg::DCH) * Cfg::DPLANE + (id % Cfg::DCH) * 1024),
                                                 16, (int)vo, 0, 0, 0);
      }
    }
  };

  // ---- transposed-read addresses: lane (g4, q, p4) reads pixel 4 g4 + q of a 16-pixel half K-step, channels 4 p4 .. 4 p4 + 3
  // KSPLIT (64 output channels): a wave runs the EVEN K-steps of the bands in buffer kg and the ODD ones of the bands in buffer
  // kg ^ 1 (global K-step parity = kg, 7 K-steps per band): one address register per (K-step, read) with the buffer folded in.
  // Otherwise (128 output channels: kg picks the wave's four co blocks) every wave runs every K-step and the registers move
  // to the other buffer after each band.
  const int g4 = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_wb_t*)smem;
  const uint32_t a_lane = lds0 + (uint32_t)(Cfg::XBYTES + (Cfg::KSPLIT ? 0 : 4 * kg * Cfg::DPLANE) + (4 * g4 + q) * 32 + p4 * 8);
  uint32_t a_even = a_lane + (uint32_t)(Cfg::KSPLIT ? kg * Cfg::BUF : 0);
  const uint32_t a_odd = a_lane + (uint32_t)((kg ^ 1) * Cfg::BUF);
  uint32_t xb[Cfg::NS][2];
#pragma unroll
  for (int s = 0; s < Cfg::NS; ++s)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int p = 32 * s + 16 * r + 4 * g4 + q;
      const int pr = p / W, pc = p - pr * W;
      xb[s][r] = lds0 + (uint32_t)((Cfg::KSPLIT ? ((s & 1) ^ kg) * Cfg::BUF : 0) + wc * Cfg::XPLANE + (pr * PW + pc) * 32 + p4 * 8);
    }

  f32x4 acc8[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc8[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  band_acc_zero();

  // the band in buffer `par` has landed (every wave waits for its own DMAs, then all meet); everybody is past the last reads of
  // buffer par ^ 1 (band it - 1): the next band may overwrite it
  auto band_top = [&](const int it) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    band_next(lb + (it + 1) * grid, (it + 1) & 1, it + 1 < nit);
  };
  const bool hi = kg == 0;
#define SFK_BAND_DMA(C, N) [&]() __attribute__((always_inline)) { issue_slots((C) * NSLOT / (N), ((C) + 1) * NSLOT / (N)); }
  band_next(lb, 0, true);
  nx_on = true;                                  // (the first band is fetched in every experiment build)
  issue_slots(0, NSLOT);
  if constexpr (Cfg::KSPLIT) {
    for (int it = 0; it < nit; ++it) {
      band_top(it);
      if (((it & 1) ^ kg) == 0) {
        band_kstep<Cfg, 0>(acc8, a_even, xb[0][0], xb[0][1], hi, SFK_BAND_DMA(0, 4));
        band_kstep<Cfg, 2>(acc8, a_even, xb[2][0], xb[2][1], hi, SFK_BAND_DMA(1, 4));
        band_kstep<Cfg, 4>(acc8, a_even, xb[4][0], xb[4][1], hi, SFK_BAND_DMA(2, 4));
        band_kstep<Cfg, 6>(acc8, a_even, xb[6][0], xb[6][1], hi, SFK_BAND_DMA(3, 4));
      } else {
        band_kstep<Cfg, 1>(acc8, a_odd, xb[1][0], xb[1][1], hi, SFK_BAND_DMA(0, 3));
        band_kstep<Cfg, 3>(acc8, a_odd, xb[3][0], xb[3][1], hi, SFK_BAND_DMA(1, 3));
        band_kstep<Cfg, 5>(acc8, a_odd, xb[5][0], xb[5][1], hi, SFK_BAND_DMA(2, 3));
      }
    }
  } else {
    for (int it = 0; it < nit; ++it) {
      band_top(it);
      band_kstep<Cfg, 0>(acc8, a_even, xb[0][0], xb[0][1], hi, SFK_BAND_DMA(0, 4));
      band_kstep<Cfg, 1>(acc8, a_even, xb[1][0], xb[1][1], hi, SFK_BAND_DMA(1, 4));
      band_kstep<Cfg, 2>(acc8, a_even, xb[2][0], xb[2][1], hi, SFK_BAND_DMA(2, 4));
      band_kstep<Cfg, 3>(acc8, a_even, xb[3][0], xb[3][1], hi, SFK_BAND_DMA(3, 4));
      const uint32_t d = (it & 1) ? (uint32_t)(-Cfg::BUF) : (uint32_t)Cfg::BUF;       // to the other buffer
      a_even += d;
#pragma unroll
      for (int s = 0; s < Cfg::NS; ++s) { xb[s][0] += d; xb[s][1] += d; }
    }
  }
#undef SFK_BAND_DMA

  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");          // asm MFMAs carry no result-latency padding for the readers below
  // partial of this workgroup: fragment F = (co block * 9 + tap position) * 4 + ci block, 64 float4 each (band_reduce_kernel);
  // with two input-channel halves workgroup lb = (split lb >> 1, half lb & 1): the same image order
  float4* const wp = k.ws + (int64_t)lb * (Cfg::NFRAG * 64) + lane;
  if constexpr (Cfg::KSPLIT) {
    // ---- the two K halves meet: kg = 1 leaves its fragments in LDS (the band buffers are dead), kg = 0 adds and stores
    // (fragment by fragment: with all 144 values in VGPRs at once hipcc spills into AGPRs it believes free -- ours)
    __syncthreads();
    float4* const ex = reinterpret_cast<float4*>(smem);
    if (kg == 1) {
#define SFK_BAND_PUT(T, C) { const f32x4 v = band_acc_read<4 * T + C>(); ex[(wc * 36 + C * 9 + T) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]); }
#define SFK_BAND_PUT4(T) SFK_BAND_PUT(T, 0) SFK_BAND_PUT(T, 1) SFK_BAND_PUT(T, 2) SFK_BAND_PUT(T, 3)
      SFK_BAND_PUT4(0) SFK_BAND_PUT4(1) SFK_BAND_PUT4(2) SFK_BAND_PUT4(3) SFK_BAND_PUT4(4) SFK_BAND_PUT4(5) SFK_BAND_PUT4(6) SFK_BAND_PUT4(7)
#undef SFK_BAND_PUT4
#undef SFK_BAND_PUT
#pragma unroll
      for (int c = 0; c < 4; ++c) ex[(wc * 36 + c * 9 + 8) * 64 + lane] = make_float4(acc8[c][0], acc8[c][1], acc8[c][2], acc8[c][3]);
    }
    __syncthreads();
    if (kg == 0) {
#define SFK_BAND_OUT(T, C) { const f32x4 v = band_acc_read<4 * T + C>(); const float4 o = ex[(wc * 36 + C * 9 + T) * 64 + lane]; \
        wp[((C * 9 + bk.tpos[T]) * 4 + wc) * 64] = make_float4(v[0] + o.x, v[1] + o.y, v[2] + o.z, v[3] + o.w); }
#define SFK_BAND_OUT4(T) SFK_BAND_OUT(T, 0) SFK_BAND_OUT(T, 1) SFK_BAND_OUT(T, 2) SFK_BAND_OUT(T, 3)
      SFK_BAND_OUT4(0) SFK_BAND_OUT4(1) SFK_BAND_OUT4(2) SFK_BAND_OUT4(3) SFK_BAND_OUT4(4) SFK_BAND_OUT4(5) SFK_BAND_OUT4(6) SFK_BAND_OUT4(7)
#undef SFK_BAND_OUT4
#undef SFK_BAND_OUT
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 o = ex[(wc * 36 + c * 9 + 8) * 64 + lane];
        wp[((c * 9 + bk.tpos[8]) * 4 + wc) * 64] = make_float4(acc8[c][0] + o.x, acc8[c][1] + o.y, acc8[c][2] + o.z, acc8[c][3] + o.w);
      }
    }
  } else {
    // every wave owns its 36 fragments: co blocks 4 kg .. 4 kg + 3
#define SFK_BAND_OUT(T, C) { const f32x4 v = band_acc_read<4 * T + C>(); \
      wp[(((4 * kg + C) * 9 + bk.tpos[T]) * 4 + wc) * 64] = make_float4(v[0], v[1], v[2], v[3]); }
#define SFK_BAND_OUT4(T) SFK_BAND_OUT(T, 0) SFK_BAND_OUT(T, 1) SFK_BAND_OUT(T, 2) SFK_BAND_OUT(T, 3)
    SFK_BAND_OUT4(0) SFK_BAND_OUT4(1) SFK_BAND_OUT4(2) SFK_BAND_OUT4(3) SFK_BAND_OUT4(4) SFK_BAND_OUT4(5) SFK_BAND_OUT4(6) SFK_BAND_OUT4(7)
#undef SFK_BAND_OUT4
#undef SFK_BAND_OUT
#pragma unroll
    for (int c = 0; c < 4; ++c)
      wp[(((4 * kg + c) * 9 + bk.tpos[8]) * 4 + wc) * 64] = make_float4(acc8[c][0], acc8[c][1], acc8[c][2], acc8[c][3]);
  }
}

// dw[co][widx][ci] += sum over the workgroups' partials, in workgroup order per split group and then group order: the summation
// tree is fixed by (splits, ZG).  A block owns 256 / ZG consecutive float4 of the partial image.
// Partial image of a split: [input-channel half][fragment][lane]; nfrag fragments per half.
template <int ZG>
__global__ __launch_bounds__(256) void band_reduce_kernel(const WgradK k, const int nfrag, const int nhalf, const int splits) {
  constexpr int E = 256 / ZG;
  __shared__ float4 red[ZG][E];
  const int le = threadIdx.x % E, zg = threadIdx.x / E;
  const int idx = blockIdx.x * E + le, total = nhalf * nfrag * 64;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < total) {
    const float4* p = k.ws + idx;
    int z = zg;
    for (; z + 3 * ZG < splits; z += 4 * ZG) {          // four loads in flight, added in order
      const float4 v0 = p[(int64_t)z * total], v1 = p[(int64_t)(z + ZG) * total], v2 = p[(int64_t)(z + 2 * ZG) * total],
                   v3 = p[(int64_t)(z + 3 * ZG) * total];
      sum.x += v0.x; sum.y += v0.y; sum.z += v0.z; sum.w += v0.w;
      sum.x += v1.x; sum.y += v1.y; sum.z += v1.z; sum.w += v1.w;
      sum.x += v2.x; sum.y += v2.y; sum.z += v2.z; sum.w += v2.w;
      sum.x += v3.x; sum.y += v3.y; sum.z += v3.z; sum.w += v3.w;
    }
    for (; z < splits; z += ZG) {
      const float4 v = p[(int64_t)z * total];
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
  }
  if constexpr (ZG > 1) {
    red[zg][le] = sum;
    __syncthreads();
    if (zg != 0) return;
#pragma unroll
    for (int z = 1; z < ZG; ++z) {
      const float4 v = red[z][le];
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
  }
  if (idx >= total) return;
  const int lane = idx & 63, half = (idx >> 6) / nfrag, F = (idx >> 6) - half * nfrag;
  const int cib = F & 3, tp = (F >> 2) % 9, cbo = (F >> 2) / 9;
  const int co0 = cbo * 16 + 4 * (lane >> 4), ci = half * 64 + cib * 16 + (lane & 15);
  const int widx = k.taps[tp].widx;
  float* dp = k.dw + ((int64_t)co0 * k.wtaps + widx) * k.cin + ci;
  const int64_t rs = (int64_t)k.wtaps * k.cin;
  const float v4[4] = {sum.x, sum.y, sum.z, sum.w};
  float old[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) old[r] = dp[r * rs];
#pragma unroll
  for (int r = 0; r < 4; ++r) dp[r * rs] = old[r] + v4[r];
}

// eligibility: bf16, (1,3,3) "same" stride-1 taps; 64 -> 64 over frames of 56 x 56 (slow res2 conv_b) or 128 -> 128 over 28 x 28
// (slow res3 conv_b: two workgroups per band, 64 input channels each, all 128 output channels)
__attribute__((visibility("hidden"))) bool wgrad_band_ok(const sfk_wgrad_desc* d) {
  const int on = sfk_tune().wgrad_band;
  if (!on || d->x.dtype != SFK_BF16 || d->dy.dtype != SFK_BF16 || d->dg_w) return false;
  const bool c64 = (on & 1) && d->cin == 64 && d->cout == 64 && d->x.w == 56;
  const bool c128 = (on & 2) && d->cin == 128 && d->cout == 128 && d->x.w == 28;
  if (!c64 && !c128) return false;
  if (d->x.h % 4 != 0 || d->x.n != d->dy.n || d->x.t != d->dy.t || d->x.h != d->dy.h || d->x.w != d->dy.w) return false;
  if (d->gs[0] != 1 || d->gs[1] != 1 || d->gs[2] != 1 || d->ntaps != 9) return false;
  int seen = 0;
  for (int i = 0; i < 9; ++i) {
    if (d->taps[i].dt != 0 || d->taps[i].dh < -1 || d->taps[i].dh > 1 || d->taps[i].dw < -1 || d->taps[i].dw > 1) return false;
    seen |= 1 << ((d->taps[i].dh + 1) * 3 + d->taps[i].dw + 1);
  }
  if (seen != 0x1FF) return false;
  if ((d->x.ld % 8) || (d->x.c_off % 8) || (d->dy.ld % 8) || (d->dy.c_off % 8)) return false;
  if ((((uintptr_t)d->x.ptr) | ((uintptr_t)d->dy.ptr)) & 15) return false;
  if (d->x.n * d->x.t * (d->x.h / 4) < 64) return false;               // a handful of bands per workgroup at least
  return sfk_fmap_bytes(&d->x) < 0x7FF00000ll && sfk_fmap_bytes(&d->dy) < 0x7FF00000ll;
}

template <class Cfg>
static int launch_band_cfg(WgradK& k, const sfk_wgrad_desc* d, hipStream_t s, int64_t* dry) {
  BandK bk;
  bk.bpf = d->x.h / 4;
  bk.nunits = d->x.n * d->x.t * bk.bpf * Cfg::NHALF;
  for (int i = 0; i < 9; ++i) bk.tpos[(d->taps[i].dh + 1) * 3 + d->taps[i].dw + 1] = i;
  const int grid = (bk.nunits < 256 ? bk.nunits : 256) / Cfg::NHALF * Cfg::NHALF;     // (a workgroup keeps its channel half)
  const int64_t need = (int64_t)grid * Cfg::NFRAG * 64 * 16;
  if (dry) { *dry = need; return SFK_OK; }
  if (!k.ws || need > d->workspace_bytes) return SFK_ERR_UNSUPPORTED;      // (the caller falls back to the implicit-GEMM kernels)
  hipLaunchKernelGGL((conv_wgrad_band_kernel<Cfg>), dim3((unsigned)grid), dim3(512), 0, s, k, bk);
  SFK_CHECK_LAUNCH();
  const int total = Cfg::NHALF * Cfg::NFRAG * 64, splits = grid / Cfg::NHALF;
  if (splits >= 32)
    hipLaunchKernelGGL((band_reduce_kernel<16>), dim3((unsigned)((total + 15) / 16)), dim3(256), 0, s, k, Cfg::NFRAG, Cfg::NHALF, splits);
  else
    hipLaunchKernelGGL((band_reduce_kernel<1>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k, Cfg::NFRAG, Cfg::NHALF, splits);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

__attribute__((visibility("hidden"))) int launch_wgrad_band(WgradK& k, const sfk_wgrad_desc* d, hipStream_t s, int64_t* dry) {
  return d->cin == 64 ? launch_band_cfg<BandCfg<56, 64, 64>>(k, d, s, dry) : launch_band_cfg<BandCfg<28, 128, 128>>(k, d, s, dry);
}

}  // namespace sfk_wgrad
