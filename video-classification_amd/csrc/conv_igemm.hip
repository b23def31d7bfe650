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

#include "conv_igemm_epi.h"

using namespace sfk_igemm;

namespace sfk_igemm {
// conv_igemm_p8.hip
__attribute__((visibility("hidden"))) int launch_p8(const ConvK& k, int bms, dim3 grid, hipStream_t s);
// conv_halo.hip
__attribute__((visibility("hidden"))) bool halo_ok(const sfk_conv_desc* d);
__attribute__((visibility("hidden"))) int halo_mtiles(const sfk_conv_desc* d);
__attribute__((visibility("hidden"))) int launch_halo(const ConvK& k, const sfk_conv_desc* d, hipStream_t s);
}

namespace {

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

struct TileSel { int bm, bn; bool dma; bool p8 = false; };
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
    // the deep-pipelined 256 x 256 tile (conv_igemm_p8.hip: one workgroup per CU, 64-channel K-tiles, DMAs in flight across
    // the barriers, the two wave groups of a SIMD half a phase apart) for the MFMA-bound layers: >= 8 K-tiles, a co tile that
    // is all real (cout % 256 == 0), >= 16 K-tiles, plain / += / bitmap epilogues with 16-byte stores
    // (measured in the step's per-layer report: with 12 K-tiles -- res4 conv_a's data gradient, K = 3 x 256 -- or a co tile that is
    // half padding -- 640 = 2.5 tiles -- the one-tile-per-CU launch loses to the 256 x 128 ring, whose second resident workgroup
    // overlaps one tile's epilogue with the other's main loop: 135 vs 117 us and 254 vs 211 us)
    if (sfk_tune().igemm_p8 && (d->cin % 64) == 0 && ktot >= 1024 && M >= 256 * 64 && (cout % 256) == 0 &&
        !d->ep.scale && !d->ep.shift && !d->bnb.partials && (d->cout % 8) == 0 && (d->y.ld % 8) == 0 && (d->y.c_off % 8) == 0 &&
        (sfk_tune().igemm_wide_store || d->out_relu_bits)) {
      const int64_t nt = (cout + 255) / 256;
      const int64_t g256 = ((M + 255) / 256 * nt + 255) / 256 * 256, g224 = ((M + 223) / 224 * nt + 255) / 256 * 224;
      TileSel t{(sfk_tune().igemm_p8 & 2) && g224 < g256 ? 224 : 256, 256, true};
      t.p8 = true;
      return t;
    }
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
  k.dk64.set(d->ntaps);
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
  if constexpr (sizeof(T) == 2) {
    if (halo_ok(d)) {                  // (1,3,3) stride-1 convs of slow res2 / res3: the LDS-band kernel (conv_halo.hip)
      k.mtiles = halo_mtiles(d);
      k.ntiles = 1;
      return launch_halo(k, d, s);
    }
  }
  const TileSel ts = pick_tile(d);
  k.mtiles = (k.M + ts.bm - 1) / ts.bm;
  k.ntiles = (d->cout + ts.bn - 1) / ts.bn;
  const dim3 grid((unsigned)(k.mtiles * k.ntiles)), block(256);
  // bf16: LDS-DMA ring for the wide tile; narrow outputs keep the register-staged kernel (higher occupancy, tiny K)
  if (ts.p8) return launch_p8(k, ts.bm, dim3((unsigned)(k.mtiles * k.ntiles)), s);
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
  if (halo_ok(d)) return halo_mtiles(d);
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
  if (halo_ok(d)) return 5;
  const TileSel ts = pick_tile(d);
  return ts.p8 ? 4 : (ts.dma ? 1 : 0);
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
