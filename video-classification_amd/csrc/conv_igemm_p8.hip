// conv_igemm_p8: the deep-pipelined implicit-GEMM conv kernel for the MFMA-bound layers (bf16, cin % 64 == 0).
//
// Same GEMM and the same operand orientation as conv_igemm.hip (A = filter rows = co, B = gathered pixels), another main
// loop: ONE workgroup per CU, 8 waves, a 256 (pixels) x 256 (co) tile, K-tiles of 64 channels (128-byte LDS rows), all of
// LDS in one array (2 K-tiles x 64 KB), LDS-DMA staging that stays in flight across raw s_barriers behind a COUNTED vmcnt,
// and the two wave groups of a SIMD running half a phase apart:
//
//   wave (g, wc), g = wave >> 2 (pixel half: rows 128g .. 128g+127 of the tile), wc = wave & 3 (co quarter: 64 channels);
//   waves 0-3 and 4-7 land on the four SIMDs once each, so every SIMD hosts one wave of each group.
//   A K-tile is four PHASES (quadrants of the wave's 128 x 64 output, 16 MFMAs each):
//       q0 = (co half 0, px half 0)   q1 = (co half 1, px half 0)   q2 = (co half 1, px half 1)   q3 = (co half 0, px half 1)
//   and a phase is   L: ds_read the fragments the quadrant needs that are not in registers yet, issue 2 LDS-DMAs,
//                       s_waitcnt vmcnt(10), s_barrier
//                    M: 16 MFMAs at raised priority (each behind the counted lgkmcnt of its fragments), s_barrier.
//   Group 1 executes one extra s_barrier before its first phase, so while group 0 is in M(p) group 1 is in L(p) and vice
//   versa: on every SIMD one wave feeds the matrix pipe while its partner reads fragments and issues DMAs.
//   Fragment reads per K-tile and wave: 12 / 4 / 8 / 0 ds_read_b128 (the co-half-0 filter fragments stay in registers from
//   q0 to q3) -- 24 for 64 MFMAs.
//
// LDS image (per K-tile parity b): X_g = rows [128g, +128) of the pixel tile at b*64K + g*16K, filter rows at b*64K + 32K;
// rows are 128 bytes = 8 sixteen-byte slots, physical slot = logical ^ ((row >> 1) & 7): conflict-free for ds_read_b128
// (its four 16-lane groups then hit 16 distinct slots of the 256-byte bank row).  LDS-DMA writes lane-linear (1 KiB = 8
// rows per wave-instruction), so the swizzle sits on the SOURCE address: lane l of the instruction that fills rows 8j' ..
// 8j'+7 fetches logical segment (l & 7) ^ (4 (j' & 1) | (l >> 4)).
//
// DMA schedule.  The units of K-tile t (2 DMA wave-instructions per wave each):
//       X-x0(t): rows 0..63 of X_g (group g's waves stage their own half: nobody else reads it)
//       X-x1(t): rows 64..127 of X_g
//       W-c0(t): for every co quarter its first 32 filter rows          W-c1(t): ... its second 32 rows   (all 8 waves)
//   are read in phases 4t (x0, c0), 4t+1 (c1), 4t+2 (x1).  A slot is restaged as soon as that is race-free:
//       phase 4t+0 issues X-x1(t+1)     phase 4t+1 issues X-x0(t+2)     phase 4t+2 issues W-c0(t+2)     phase 4t+3 issues W-c1(t+2)
//   WAR: a wave's reads of phase p retire at the start of its M(p) (lgkmcnt(0)), i.e. before the barrier that ends M(p).
//     A wave of the same group passes that barrier before its L(p+1); a group-0 wave meets a group-1 reader's barrier only
//     before its L(p+2).  X_g is private to group g (restaged 1 resp. 2 phases after its read), the shared filter units are
//     restaged 2 phases after their read.
//   RAW: every unit is issued >= 6 phases before it is read, and every phase waits until all but the newest 10 DMAs of the
//     wave (5 phases' worth) have landed, then meets the other waves at the barrier that ends L(p); the reads of phase p+1
//     come after that barrier for both groups (group 1's L(p) ends one barrier later, still before anybody's L(p+1) reads).
//   Never a full drain inside the loop; about 80 KB are on the wire per CU.
//
// K walk: cin % 64 == 0, so a K-tile lies inside one tap; K-tile tau = (channel chunk tau / ntaps, tap tau % ntaps).  The
// padding test of every (row, tap) is done once (a bit mask per row); a unit's per-lane source offset is base + tap delta
// with the mask bit blended in (3 VALU per row), the tap comes from the kernel arguments by a scalar load issued one phase
// ahead, and the channel advance (kc * 128 B) and the filter's tap offset ride the SGPR soffset.
#include "conv_igemm_epi.h"

namespace sfk_igemm {

typedef __attribute__((address_space(3))) void lds_void_p8_t;

template <int BMS, int EPI>
__global__ __launch_bounds__(512, 2) void conv_igemm_p8_kernel(const ConvK k) {
  using T = bf16_t;
  constexpr int FM = BMS / 32;            // pixel fragments per wave: 8 (256 rows) or 7 (224 rows computed, 256 staged)
  constexpr int FN = 4;                   // co fragments per wave
  constexpr int FX1 = FM - 4;             // fragments of the second pixel half
  constexpr int PAR = 65536, XH = 16384, WB = 32768, ROWB = 128;
  static_assert(BMS == 256 || BMS == 224, "computed rows");
  // ONE LDS object, and nothing in it that is read at a run-time address: behind a dynamically addressed ds_read hipcc
  // drains vmcnt(0) (it cannot tell the read from the LDS-DMA destinations) -- the per-tap table lives in the kernel
  // argument block and is read with scalar loads instead
  __shared__ __attribute__((aligned(16))) char smem[2 * PAR];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, g4 = lane >> 4;
  int mt, nt;
  {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = b & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    nt = logical % k.ntiles;
    mt = logical / k.ntiles;
  }

  // ---- DMA source coordinates of this lane: row (lane >> 3) of each 8-row instruction, logical segment seg0 ^ 4j
  const int lrow = lane >> 3;
  const int seg0 = (lane & 7) ^ (lane >> 4);
  // Padding test of every (row, tap) ONCE, as a bit mask per row (bit tap = the gathered pixel lies outside the map): vector
  // instructions issued from L compete with the partner wave's MFMA stream, which runs at raised priority -- ~22 VALU per
  // unit for the test cost the phase ~300 cycles.  Per unit it is now add + bfe + bfi per row.
  uint32_t xbase[2][2], xinv[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = mt * BMS + g * (BMS / 2) + 64 * u + 16 * wc + 8 * j + lrow;
      uint32_t q1, rw_, q2, rh_, n_, rt_;
      k.drw.divmod((uint32_t)m, q1, rw_);
      k.drh.divmod(q1, q2, rh_);
      k.drt.divmod(q2, n_, rt_);
      const int tb = (m < k.M) ? (int)rt_ * k.gst : -(1 << 28), hb = (int)rh_ * k.gsh, wb = (int)rw_ * k.gsw;
      xbase[u][j] = (uint32_t)((((((int64_t)n_ * k.xt + (int)rt_ * k.gst) * k.xh + hb) * k.xw + wb) * k.xld + k.xoff) * 2) +
                    (uint32_t)((seg0 ^ (4 * j)) * 16);
      uint32_t inv = 0;
      for (int tap = 0; tap < k.ntaps; ++tap) {
        const sfk_tap tp = k.taps[tap];
        const int ti = tb + tp.dt, hi = hb + tp.dh, wi = wb + tp.dw;
        const bool ok = (unsigned)ti < (unsigned)k.xt && (unsigned)hi < (unsigned)k.xh && (unsigned)wi < (unsigned)k.xw;
        inv |= (ok ? 0u : 1u) << tap;
      }
      xinv[u][j] = inv;
    }
  constexpr uint32_t FAR = 0x80000000u;
  uint32_t wv[2][2];                      // filter voffsets: loop-invariant (the tap offset is wave-uniform -> soffset)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = nt * 256 + 64 * wc + 32 * h + 16 * g + 8 * j + lrow;
      wv[h][j] = co < k.cout ? (uint32_t)(co * k.wtaps * k.cin) * 2u + (uint32_t)((seg0 ^ (4 * j)) * 16) : FAR;
    }
  const __amdgpu_buffer_rsrc_t xrs = sfk_make_rsrc(k.x, k.xbytes);
  const __amdgpu_buffer_rsrc_t wrs = sfk_make_rsrc(k.w, k.wbytes);
  // K order: channel chunk outer, tap inner -- K-tile tau = (kc, tap) with kc = tau / ntaps.  The taps of one 64-channel chunk
  // then follow each other, so the pixel rows that neighbouring tiles (same XCD) read through different taps are fetched
  // within a few K-tiles of each other and hit L2: with the taps outermost a (3,1,1) conv re-read its whole input from the
  // Infinity Cache / HBM once per tap (TCC hit rate 50 %, and the chip clocked 13 % lower for it).
  const int KT = k.ntaps * (k.cin >> 6);
  // tap and chunk of the K-tile whose units are being issued, all wave-uniform (SGPRs): the tap comes from the kernel
  // argument block by a scalar load issued one phase ahead, inside M, where no LDS read is outstanding (a scalar load
  // returns out of order with LDS reads, so waiting for it later would mean lgkmcnt(0) behind the fragment reads)
  // K-tiles past the end (the look-ahead of the last two) re-stage K-tile 0: harmless, nobody reads those slots, and the
  // issue paths need no "live" test.
  int te_tap = 0, te_xd = 0, te_wo = 0, te_kc = 0;     // the K-tile whose units are being issued
  int nx_tap = 0, nx_kc = 0;                           // the next one: scalar load issued, not unpacked yet
  uint32_t nx_raw = 0;
  // first half: (chunk, tap) of K-tile tau and the scalar load of its tap.  Runs in L(q3), the shortest L segment: scalar
  // instructions there cost the partner's MFMA stream nothing, and the load has landed long before M(q3)'s lgkmcnt(0)
  auto fetch_issue = [&](const int tau_) __attribute__((always_inline)) {
    const int tau = tau_ < KT ? tau_ : 0;
    const int kc = __builtin_amdgcn_readfirstlane((int)k.dk64.div((uint32_t)tau));   // (provably uniform: no waterfall loops)
    nx_tap = tau - kc * k.ntaps;
    nx_raw = *reinterpret_cast<const uint32_t*>(&k.taps[nx_tap & (SFK_MAX_TAPS - 1)]);
    nx_kc = kc;
  };
  // second half, at the top of L(q1) (scalar ALU only): the units issued from here on belong to that K-tile
  auto fetch_finish = [&]() __attribute__((always_inline)) {
    const int dt = (int)(int8_t)(nx_raw & 255), dh = (int)(int8_t)((nx_raw >> 8) & 255), dw = (int)(int8_t)((nx_raw >> 16) & 255);
    te_xd = ((dt * k.xh + dh) * k.xw + dw) * k.xld * 2;
    te_wo = (int)(nx_raw >> 24) * k.cin * 2 + nx_kc * 128;
    te_tap = nx_tap;
    te_kc = nx_kc;
  };
  auto fetch_entry = [&](const int tau) __attribute__((always_inline)) { fetch_issue(tau); fetch_finish(); };
  // X unit u (rows 64u .. 64u+63 of X_g) of the current entry's K-tile -> parity b
  auto issue_x = [&](const int b, const int u) __attribute__((always_inline)) {
    char* dst = smem + b * PAR + g * XH + (64 * u + 16 * wc) * ROWB;
    const int so = __builtin_amdgcn_readfirstlane(te_kc * 128);     // (a loop-carried scalar can end up in a VGPR: no waterfall loop)
    const int tap = __builtin_amdgcn_readfirstlane(te_tap);
    const uint32_t xd = (uint32_t)__builtin_amdgcn_readfirstlane(te_xd);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe(xinv[u][j], tap, 1);        // 0 or ~0
      const uint32_t xo = (m & FAR) | (~m & (xbase[u][j] + xd));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_p8_t*)(dst + 8 * j * ROWB), 16, (int)xo, so, 0, 0);
    }
  };
  // filter unit h of the current entry's K-tile -> parity b
  auto issue_w = [&](const int b, const int h) __attribute__((always_inline)) {
    char* dst = smem + b * PAR + WB + (64 * wc + 32 * h + 16 * g) * ROWB;
    const int so = __builtin_amdgcn_readfirstlane(te_wo);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void_p8_t*)(dst + 8 * j * ROWB), 16, (int)wv[h][j], so, 0, 0);
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses: row l15 of a 16-row fragment, logical slot 4s + g4
  const int fx = (l15 >> 1) & 7;
  int ra[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) ra[s] = l15 * ROWB + ((((4 * s + g4) ^ fx)) << 4);
  // one address register per (operand, parity, k sub-step), fragments at immediate offsets: left to itself hipcc keeps
  // the parity-0 addresses and re-adds 64 KiB with eight VALU instructions in every odd K-tile's L segments
  uint32_t xa[2][2], wa[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      xa[b][s] = (uint32_t)(b * PAR + g * XH + ra[s]);
      wa[b][s] = (uint32_t)(b * PAR + WB + wc * 64 * ROWB + ra[s]);
      asm volatile("" : "+v"(xa[b][s]), "+v"(wa[b][s]));
    }
  bf16x8 wf[2][2][2];                     // [co half][fragment][k sub-step]
  bf16x8 xf[4][2];                        // [pixel fragment of the current half][k sub-step]
  auto read_w = [&](const int b, const int h) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        wf[h][i][s] = *reinterpret_cast<const bf16x8*>(smem + wa[b][s] + (2 * h + i) * 16 * ROWB);
  };
  auto read_x = [&](const int b, const int xh) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (xh == 1 && j >= FX1) continue;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        xf[j][s] = *reinterpret_cast<const bf16x8*>(smem + xa[b][s] + (4 * xh + j) * 16 * ROWB);
    }
  };
  auto mfma = [&](const int h, const int xh) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (xh == 1 && j >= FX1) continue;
          acc[2 * h + i][4 * xh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[h][i][s], xf[j][s], acc[2 * h + i][4 * xh + j], 0, 0, 0);
        }
  };
  auto end_l = [&](const int q) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // (no lgkmcnt(0) here: hipcc waits per fragment, lgkmcnt(N) in front of the MFMA that needs it, so the first MFMAs issue
    // while the last fragments are still on their way -- +1..4 %.  Every fragment read in L(p) is consumed by an MFMA of
    // M(p), so all reads have retired before the barrier that ends M(p): the WAR argument above holds.)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
  };
  auto end_m = [&](const int q) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // the four phases of K-tile t (parity b = t & 1, a literal after unrolling)
  auto ktile = [&](const int b, const int t) __attribute__((always_inline)) {
    read_w(b, 0); read_x(b, 0);  issue_x(b ^ 1, 1);                    end_l(0); mfma(0, 0); end_m(0);   // X-x1(t+1)
    fetch_finish(); read_w(b, 1); issue_x(b, 0);                       end_l(1); mfma(1, 0); end_m(1);   // X-x0(t+2)
    read_x(b, 1);                issue_w(b, 0);                        end_l(2); mfma(1, 1); end_m(2);   // W-c0(t+2)
                                 issue_w(b, 1); fetch_issue(t + 3);    end_l(3); mfma(0, 1); end_m(3);   // W-c1(t+2)
  };

  // prologue: K-tile 0 and all of K-tile 1 but its X-x1 (phase 0 issues that one)
  fetch_entry(0);
  issue_x(0, 0); issue_w(0, 0); issue_w(0, 1); issue_x(0, 1);
  fetch_entry(1);
  issue_x(1, 0); issue_w(1, 0); issue_w(1, 1);
  fetch_issue(2);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
  for (int t = 0;;) {
    ktile(0, t);
    if (++t >= KT) break;
    ktile(1, t);
    if (++t >= KT) break;
  }
  if (g == 0) __builtin_amdgcn_s_barrier();      // ... and group 0 meets its last one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the (out-of-range, zero-writing) look-ahead DMAs before LDS is reused
  __syncthreads();

  epilogue_plain<T, EPI, FM, FN, BMS, 256, 2, 4, false>(k, acc, reinterpret_cast<float*>(smem), mt, nt, g, wc, lane, tid);
}

__attribute__((visibility("hidden"))) int launch_p8(const ConvK& k, int bms, dim3 grid, hipStream_t s) {
  if (k.obits) {
    if (bms == 224) hipLaunchKernelGGL((conv_igemm_p8_kernel<224, 2>), grid, dim3(512), 0, s, k);
    else hipLaunchKernelGGL((conv_igemm_p8_kernel<256, 2>), grid, dim3(512), 0, s, k);
  } else {
    if (bms == 224) hipLaunchKernelGGL((conv_igemm_p8_kernel<224, 0>), grid, dim3(512), 0, s, k);
    else hipLaunchKernelGGL((conv_igemm_p8_kernel<256, 0>), grid, dim3(512), 0, s, k);
  }
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

}  // namespace sfk_igemm
