// sfk_conv_wgrad: Conv3d filter gradient on gfx950 MFMA.
//
//   dW[co][tap][ci] += sum_{pixel} dY[pixel][co] * X[gather(pixel, tap)][ci]
//
// The reduction runs over PIXELS, which is the strided dimension of both channels-last operands.  Tiles are
// staged as they lie in HBM ([pixel][channel] rows, 16 B per lane) and the MFMA fragments (8 consecutive pixels
// of one channel per lane) come out of LDS already transposed:
//   bf16: ds_read_b64_tr_b16 (two per fragment)          f32: one ds_read_b32 per MFMA (one k per lane)
// The (tap, cin) axis is flattened into the GEMM's column space and gathered per 16-byte segment (as the forward
// kernel packs its K axis), so the taps of narrow layers share one tile and dY is read once per column tile.
// Grid = (co-tile x column-tile, 1, pixel-split); partial tiles are combined with fp32 atomics into dW.

#include "conv_wgrad_common.h"

using namespace sfk_wgrad;

namespace sfk_wgrad {
// conv_wgrad_p8.hip: the deep-pipelined 256 x 256 tile of the MFMA-bound layers (dry != NULL: workspace bytes only)
__attribute__((visibility("hidden"))) int launch_wgrad_p8(WgradK& k, const sfk_wgrad_desc* d, hipStream_t s, int64_t* dry);
__attribute__((visibility("hidden"))) bool wgrad_p8_ok(const sfk_wgrad_desc* d, int M);
// conv_wgrad_band.hip: the LDS-band kernel of the (1,3,3) 64 -> 64 layer over 56 x 56 frames; workspace only
__attribute__((visibility("hidden"))) int launch_wgrad_band(WgradK& k, const sfk_wgrad_desc* d, hipStream_t s, int64_t* dry);
__attribute__((visibility("hidden"))) bool wgrad_band_ok(const sfk_wgrad_desc* d);
}

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <typename T, int TC> struct WT;
template <int TC> struct WT<bf16_t, TC> {
  // rows of TC channels, no padding; the 32-byte blocks (16 channels = one transposed fragment) of row r are stored at
  // block index  b ^ swz(r),  chosen so that the 8 rows a ds_read_b64_tr_b16 half-wave touches (r = 8g+q, g in {0,1},
  // q in 0..3) land on 8 disjoint 32-byte bank windows (the padded layout conflicted 2-way: 35 % of the LDS cycles)
  static constexpr int VEC = 8, SEGS = TC / 8, ROWB = TC * 2, NB = TC / 16;
  typedef bf16x8 frag;
  static __device__ __forceinline__ int swz(int r) {
    if (NB >= 8) return (r & 3) | (((r >> 3) & 1) << 2);
    if (NB == 4) return ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
    return NB == 2 ? ((r >> 3) & 1) : 0;
  }
  // byte offset of 16-byte segment `seg` of row r
  static __device__ __forceinline__ int off(int r, int seg) { return r * ROWB + ((((seg >> 1) ^ swz(r)) << 1) | (seg & 1)) * 16; }
  // fragment for channels c0..c0+15: lane (g = lane>>4, q = (lane&15)>>2, p = lane&3) addresses row 8g+q (+4),
  // columns c0+4p..c0+4p+3; the transpose read hands lane i of the group column c0+i of those 4 rows.
  static __device__ __forceinline__ frag load(const char* tile, int c0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int r = 8 * g + q;                       // rows r and r+4 share swz (bit 2 is not part of it)
    const char* a = tile + r * ROWB + (((c0 >> 4) ^ swz(r)) << 5) + p * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a + 4 * ROWB));
    frag f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
  static __device__ __forceinline__ void mma(f32x4& acc, const frag& a, const frag& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
};
struct F32Frag { float v[8]; };
template <int TC> struct WT<float, TC> {
  static constexpr int VEC = 4, SEGS = TC / 4, ROWB = TC * 4 + 16;
  typedef F32Frag frag;
  static __device__ __forceinline__ int off(int r, int seg) { return r * ROWB + seg * 16; }
  // MFMA step s takes pixel 4s+g of channel c0 + (lane&15)
  static __device__ __forceinline__ frag load(const char* tile, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    frag f;
#pragma unroll
    for (int s = 0; s < 8; ++s) f.v[s] = *reinterpret_cast<const float*>(tile + (4 * s + g) * ROWB + (c0 + i) * 4);
    return f;
  }
  static __device__ __forceinline__ void mma(f32x4& acc, const frag& a, const frag& b) {
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[s], b.v[s], acc, 0, 0, 0);
  }
};

// Block tile: TCO output channels x TCI columns of the flattened (tap, cin) axis; 4 waves as 2 x 2.  A stage is
// KS K-steps (KS*32 pixels) between two barriers.  Column segments (16 B) are gathered independently, so a tile may
// straddle taps and narrow layers put ALL their taps into one tile (9 x 8 channels = 72 columns).
// GRAM (TCO == TCI, x and dy the SAME map, pointwise): one staged tile serves both operands -- half the loads and LDS stores
// of the Gram matrix  G = a^T a  of the fused block tail (sfk_bn_tail_fwd).
template <typename T, int TCO, int TCI, int KS, bool GRAM = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradK k) {
  static_assert(!GRAM || TCO == TCI, "Gram matrix: square tile");
  using WO = WT<T, TCO>;
  using WI = WT<T, TCI>;
  constexpr int VEC = WO::VEC;
  constexpr int SEGO = WO::SEGS, SEGI = WI::SEGS, ROWO = WO::ROWB, ROWI = WI::ROWB;
  constexpr int R = MK * KS;                       // pixels per stage
  constexpr int FO = TCO / 32, FI = TCI / 32;      // 16-wide fragments per wave (co side, column side)
  // A thread OWNS one pixel row of the stage and walks its 16-byte segments (TPR threads per row): the pixel's
  // (n, t, h, w) decomposition -- three divisions -- is done once per thread and stage instead of once per segment
  // (the gather arithmetic outweighed the MFMAs 10..17 : 1 in the narrow layers).
  constexpr int TPR = 256 / R;
  constexpr int NLO = SEGO / TPR, NLI = SEGI / TPR;
  static_assert(TPR >= 1 && SEGO % TPR == 0 && SEGI % TPR == 0 && NLO >= 1 && NLI >= 1, "staging shape");
  constexpr int TILEO = R * ROWO, TILEI = R * ROWI, BUF = TILEO + TILEI;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  __shared__ sfk_tap s_taps[SFK_MAX_TAPS + 1];
  if (threadIdx.x <= SFK_MAX_TAPS) {
    sfk_tap t = k.taps[threadIdx.x < SFK_MAX_TAPS ? threadIdx.x : 0];
    if ((int)threadIdx.x >= k.ntaps) { t.dt = -128; t.dh = 0; t.dw = 0; t.widx = 0; }
    s_taps[threadIdx.x] = t;
  }
  __syncthreads();

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave & 1, wci = wave >> 1;
  int tile_id, split_id;
  wg_block(k.ntiles, tile_id, split_id);
  const int cot = tile_id / k.citiles, cit = tile_id % k.citiles;
  const int stage0 = split_id * k.chunks_per_split;
  const int stage1 = min(stage0 + k.chunks_per_split, k.nchunks);
  if (stage0 >= stage1) return;

  const __amdgpu_buffer_rsrc_t xrs = sfk_make_rsrc(k.x, k.xbytes);
  const __amdgpu_buffer_rsrc_t drs = sfk_make_rsrc(k.dy, k.dbytes);
  const int row = tid / TPR, q = tid % TPR;        // this thread's pixel row of the stage, first segment

  // column segment -> (tap, channel) of this thread's X slots (fixed for the whole kernel)
  int xi_soff[NLI], xi_cb[NLI];
  sfk_tap xi_tap[NLI];
  bool xi_ok[NLI];
#pragma unroll
  for (int i = 0; i < NLI; ++i) {
    const int seg = q + i * TPR;
    uint32_t tap, cseg;
    k.dspt.divmod((uint32_t)(cit * SEGI + seg), tap, cseg);
    xi_ok[i] = tap < (uint32_t)k.ntaps;
    xi_tap[i] = s_taps[xi_ok[i] ? tap : SFK_MAX_TAPS];
    xi_soff[i] = WI::off(row, seg);
    xi_cb[i] = ((int)cseg * VEC + k.xoff) * (int)sizeof(T);
  }
  // dY slots: channel segments of the same row
  uint32_t di_off[NLO];
  int di_soff[NLO];
#pragma unroll
  for (int i = 0; i < NLO; ++i) {
    const int seg = q + i * TPR;
    const int co = cot * TCO + seg * VEC;
    di_off[i] = co < k.cout ? (uint32_t)((k.doff + co) * (int)sizeof(T)) : SFK_OOB;
    di_soff[i] = WO::off(row, seg);
  }
  uint4 dr[NLO], xr[NLI];
  const bool linear = k.ntaps == 1 && k.taps[0].dt == 0 && k.taps[0].dh == 0 && k.taps[0].dw == 0 && k.gst == 1 &&
                      k.gsh == 1 && k.gsw == 1 && k.xt == (int)k.drt.d && k.xh == (int)k.drh.d && k.xw == (int)k.drw.d;
  // branch-free buffer loads: padding taps, rows past M and ragged channels read zeros through SFK_OOB
  auto gload = [&](int stage) {
    const int m = stage * R + row;
    const bool mok = stage < k.nchunks && m < k.M;       // past the end: every slot OOB (zeros), keeps the body branch-free
    const uint32_t drow = (uint32_t)m * (uint32_t)(k.dld * (int)sizeof(T));
    if constexpr (!GRAM) {
#pragma unroll
      for (int i = 0; i < NLO; ++i) dr[i] = sfk_buffer_load16(drs, (mok && di_off[i] != SFK_OOB) ? drow + di_off[i] : SFK_OOB);
    }
    if (linear) {   // pointwise stride-1 conv: the gathered pixel IS the row
      const uint32_t xrow = (uint32_t)m * (uint32_t)(k.xld * (int)sizeof(T));
#pragma unroll
      for (int i = 0; i < NLI; ++i) xr[i] = sfk_buffer_load16(xrs, (mok && xi_ok[i]) ? xrow + (uint32_t)xi_cb[i] : SFK_OOB);
      return;
    }
    uint32_t q1, rw_, q2, rh_, n_, rt_;
    k.drw.divmod((uint32_t)m, q1, rw_);
    k.drh.divmod(q1, q2, rh_);
    k.drt.divmod(q2, n_, rt_);
    const int tb = (int)rt_ * k.gst, hb = (int)rh_ * k.gsh, wb = (int)rw_ * k.gsw;
    const uint32_t nb = (uint32_t)n_ * k.xt;
#pragma unroll
    for (int i = 0; i < NLI; ++i) {
      const int ti = tb + xi_tap[i].dt, hi = hb + xi_tap[i].dh, wi = wb + xi_tap[i].dw;
      const bool ok = mok && xi_ok[i] && (unsigned)ti < (unsigned)k.xt && (unsigned)hi < (unsigned)k.xh &&
                      (unsigned)wi < (unsigned)k.xw;
      const uint32_t pix = ((nb + ti) * k.xh + hi) * k.xw + wi;
      xr[i] = sfk_buffer_load16(xrs, ok ? pix * (uint32_t)(k.xld * (int)sizeof(T)) + (uint32_t)xi_cb[i] : SFK_OOB);
    }
  };
  auto lstore = [&](int buf) {
    char* ds = smem + buf * BUF;
    char* xs = ds + TILEO;
    if constexpr (!GRAM) {
#pragma unroll
      for (int i = 0; i < NLO; ++i) *reinterpret_cast<uint4*>(ds + di_soff[i]) = dr[i];
    }
#pragma unroll
    for (int i = 0; i < NLI; ++i) *reinterpret_cast<uint4*>(xs + xi_soff[i]) = xr[i];
  };

  f32x4 acc[FO][FI];
#pragma unroll
  for (int i = 0; i < FO; ++i)
#pragma unroll
    for (int j = 0; j < FI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  gload(stage0);
  lstore(0);
  __syncthreads();
  for (int st = stage0; st < stage1; ++st) {
    const int buf = (st - stage0) & 1;
    gload(st + 1 < stage1 ? st + 1 : k.nchunks);   // past the end: every slot OOB (zeros), keeps the body branch-free
    const char* ds = smem + buf * BUF;
    const char* xs = ds + TILEO;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      typename WO::frag a[FO];
      typename WI::frag b[FI];
#pragma unroll
      for (int i = 0; i < FO; ++i) a[i] = WO::load((GRAM ? xs : ds) + ks * MK * ROWO, wco * (TCO / 2) + 16 * i, lane);
#pragma unroll
      for (int j = 0; j < FI; ++j) b[j] = WI::load(xs + ks * MK * ROWI, wci * (TCI / 2) + 16 * j, lane);
#pragma unroll
      for (int i = 0; i < FO; ++i)
#pragma unroll
        for (int j = 0; j < FI; ++j) WO::mma(acc[i][j], a[i], b[j]);
    }
    lstore(buf ^ 1);
    __syncthreads();
  }

  if (k.ws) {   // partial tile as it lies in the accumulators: 1 KiB per wave-instruction, summed by wgrad_reduce_kernel
    float4* wp = k.ws + ((((int64_t)split_id * k.ntiles + tile_id) * 4 + wave) * (FO * FI)) * 64 + lane;
#pragma unroll
    for (int i = 0; i < FO; ++i)
#pragma unroll
      for (int j = 0; j < FI; ++j)
        wp[(i * FI + j) * 64] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    return;
  }
  // D[row = co][col]: lane holds co = 4*(lane>>4) + r and column lane & 15 -> 16 lanes add 16 consecutive floats
  const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int j = 0; j < FI; ++j) {
    const int col = cit * TCI + wci * (TCI / 2) + 16 * j + l15;
    uint32_t tap, ci;
    k.dcin.divmod((uint32_t)col, tap, ci);
    if (tap >= (uint32_t)k.ntaps) continue;
    const int widx = s_taps[tap].widx;
#pragma unroll
    for (int i = 0; i < FO; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * TCO + wco * (TCO / 2) + 16 * i + 4 * g + r;
        if (co < k.cout) atomicAdd(k.dw + ((int64_t)co * k.wtaps + widx) * k.cin + ci, acc[i][j][r]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 production kernel for wide layers: the same GEMM with LDS-DMA staging (see conv_igemm.hip for the scheme).
//   * tiles [32 pixels][TC channels] are written by `buffer_load_dwordx4 ... lds`; a wave-instruction covers 1 KiB =
//     4 rows of 256 B (TC = 128) or 2 rows of 512 B (TC = 256); lane l fills physical 16-B slot l % (L/16) of its row
//     and FETCHES logical slot  phys ^ f(row),  f(row) = ((row&3) | ((row>>3)&1)<<2) << 1 : the 8 rows a
//     ds_read_b64_tr_b16 half-wave touches then land on 8 disjoint 32-B bank windows (unswizzled: 8-way conflict);
//   * dY, and X of pointwise stride-1 convs, advance by a wave-uniform byte count per stage -> SGPR soffset, zero VALU;
//     gathered X (taps / strides) rebuilds its per-lane offsets per stage (each lane's column = one fixed tap);
//   * 3-slot ring, counted vmcnt, one raw barrier per 32-pixel stage; fp32 atomics epilogue as the generic kernel.
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ int wg_swz(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }

__device__ __forceinline__ void wg_swap16f(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// DG (TCO = 256, 8 waves, taps x cin = 64): the four waves of the second column half have no filter-gradient columns; they
// compute the data gradient  dg_y[pixel][ci] = sum_co dY[pixel][co] dg_w[ci][co]  of the stage's 32 pixels from the dY rows
// the ring already holds -- dY is read from HBM once for both gradients.  Wave d of those four: pixels 16 (d & 1) .. +15,
// channels 32 (d >> 1) .. +31; its 16 filter fragments (2 x 8 K-steps of 32 co) live in registers for the whole kernel.
// TCI = 256 (256 x 256: 8 waves as 4 x 2 of 64 co x 128 columns, ONE workgroup per CU) for the MFMA-bound layers: twice the columns per staged dY row -- 32 B of LDS-DMA per MFMA cycle and CU instead of
// 48 -- and 32 MFMAs per wave between two barriers instead of 16.  Its pixel splits ALWAYS go through the partial-tile
// workspace (a 256 x 256 fp32 tile per split through the ~1.3 TB/s atomic path would cost more than the tile's MFMAs).
// NS: ring slots (NS - 1 stages of LDS-DMA in flight beside the one being multiplied).  A workgroup has (NS - 1) x BUF bytes
// on the wire; with one or two workgroups per CU that -- not the MFMAs -- sets the rate of the HBM-bound layers: by Little's
// law 192 workgroups x 32 KB at ~2 us are 3 TB/s, which is what the 3-slot ring measured on slow res2 / res3 (3.0 .. 3.9).
template <int TCO, int NW, bool DG = false, int TCI = 128, int NS = 3>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 4 : 3)) void conv_wgrad_dma_kernel(const WgradK k) {
  constexpr int R = MK;
  constexpr int LO = TCO * 2, LI = TCI * 2;                 // tile row bytes
  constexpr int SO = LO / 16, SI = LI / 16;                 // 16-byte slots per row
  constexpr int NDO = R * LO / 1024 / NW, NDI = R * LI / 1024 / NW;   // DMA instructions per wave per stage
  constexpr int TILEO = R * LO, TILEI = R * LI, BUF = TILEO + TILEI;
  constexpr int WCO = TCO / 64;                             // waves along co (each wave 64 co)
  constexpr int WCI = NW / WCO;                             // waves along the columns
  constexpr int FI = TCI / WCI / 16;                        // 16-column fragments per wave (4 or 8)
  static_assert(NW == WCO * WCI && (FI == 4 || FI == 8) && NDO >= 1 && NDI >= 1 && (!DG || TCI == 128), "tile shape");
  static_assert(NS >= 3 && NS <= 6 && (NS - 2) * (NDO + NDI) <= 63 && (!DG || NS == 3), "ring depth");
  __shared__ __attribute__((aligned(16))) char smem[NS * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave % WCO, wci = wave / WCO;
  int tile_id, split_id;
  wg_block(k.ntiles, tile_id, split_id);
  const int cot = tile_id / k.citiles, cit = tile_id % k.citiles;
  const int stage0 = split_id * k.chunks_per_split;
  const int stage1 = min(stage0 + k.chunks_per_split, k.nchunks);
  if (stage0 >= stage1) return;
  constexpr uint32_t FAR = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xrs = sfk_make_rsrc(k.x, k.xbytes);
  const __amdgpu_buffer_rsrc_t drs = sfk_make_rsrc(k.dy, k.dbytes);

  // ---- dY slots of this lane: instruction j of this wave covers tile rows (wave*NDO + j)*(1024/LO) + lane/SO
  int d_row[NDO];
  uint32_t d_off[NDO];        // byte offset inside the stage: row*dld*2 + channel bytes; FAR when co is past cout
#pragma unroll
  for (int j = 0; j < NDO; ++j) {
    const int row = (wave * NDO + j) * (1024 / LO) + lane / SO;
    const int co = cot * TCO + ((lane % SO) ^ wg_swz(row)) * 8;
    d_row[j] = row;
    d_off[j] = co < k.cout ? (uint32_t)((row * k.dld + k.doff + co) * 2) : FAR;
  }
  // ---- X slots: the column block of a slot is one (tap, channel segment), fixed for the kernel
  int x_row[NDI], x_cb[NDI];
  sfk_tap x_tap[NDI];
  bool x_ok[NDI];
#pragma unroll
  for (int j = 0; j < NDI; ++j) {
    const int row = (wave * NDI + j) * (1024 / LI) + lane / SI;
    const int colseg = cit * SI + ((lane % SI) ^ wg_swz(row));      // 8-channel segment index on the (tap, cin) axis
    uint32_t tap, cseg;
    k.dspt.divmod((uint32_t)colseg, tap, cseg);
    x_row[j] = row;
    x_ok[j] = tap < (uint32_t)k.ntaps;
    x_tap[j] = k.taps[x_ok[j] ? tap : 0];
    x_cb[j] = ((int)cseg * 8 + k.xoff) * 2;
  }
  // pointwise stride-1 conv: X advances like dY
  const bool linear = k.ntaps == 1 && k.taps[0].dt == 0 && k.taps[0].dh == 0 && k.taps[0].dw == 0 && k.gst == 1 &&
                      k.gsh == 1 && k.gsw == 1 && k.xt == (int)k.drt.d && k.xh == (int)k.drh.d && k.xw == (int)k.drw.d;
  uint32_t x_lin[NDI];
#pragma unroll
  for (int j = 0; j < NDI; ++j) x_lin[j] = x_ok[j] ? (uint32_t)(x_row[j] * k.xld * 2 + x_cb[j]) : FAR;

  // The DMA instructions themselves sit in straight-line code (offsets are chosen by selects above them): with the
  // loads inside the linear / gathered branches the compiler lost track of which ring slot each one writes at the
  // control-flow merge and drained vmcnt(0) before every stage's fragment reads -- no DMA ever overlapped the MFMAs.
  auto dma = [&](int stage, int buf) __attribute__((always_inline)) {
    char* ds = smem + buf * BUF;
    char* xs = ds + TILEO;
    const int m0 = stage * R;
    const bool full = m0 + R <= k.M;                  // wave-uniform; the ragged last stage masks rows per lane
    const bool live = stage < stage1;                 // look-ahead past the block's range gathers nothing
    uint32_t dvo[NDO], xvo[NDI];
    const int dso = live ? m0 * k.dld * 2 : 0;
    const int xso = (live && linear) ? m0 * k.xld * 2 : 0;
#pragma unroll
    for (int j = 0; j < NDO; ++j) dvo[j] = (!live || (!full && m0 + d_row[j] >= k.M)) ? FAR : d_off[j];
#pragma unroll
    for (int j = 0; j < NDI; ++j) {
      uint32_t lin = (!live || (!full && m0 + x_row[j] >= k.M)) ? FAR : x_lin[j];
      uint32_t gat = FAR;
      if (!linear) {
        const int m = m0 + x_row[j];
        uint32_t q1, rw_, q2, rh_, n_, rt_;
        k.drw.divmod((uint32_t)m, q1, rw_);
        k.drh.divmod(q1, q2, rh_);
        k.drt.divmod(q2, n_, rt_);
        const int ti = (int)rt_ * k.gst + x_tap[j].dt, hi = (int)rh_ * k.gsh + x_tap[j].dh, wi = (int)rw_ * k.gsw + x_tap[j].dw;
        const bool ok = live && x_ok[j] && m < k.M && (unsigned)ti < (unsigned)k.xt && (unsigned)hi < (unsigned)k.xh &&
                        (unsigned)wi < (unsigned)k.xw;
        const uint32_t pix = (((uint32_t)n_ * k.xt + ti) * k.xh + hi) * k.xw + wi;
        gat = ok ? pix * (uint32_t)(k.xld * 2) + (uint32_t)x_cb[j] : FAR;
      }
      xvo[j] = linear ? lin : gat;
    }
#pragma unroll
    for (int j = 0; j < NDO; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void_t*)(ds + (wave * NDO + j) * 1024), 16, (int)dvo[j], dso, 0, 0);
#pragma unroll
    for (int j = 0; j < NDI; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t*)(xs + (wave * NDI + j) * 1024), 16, (int)xvo[j], xso, 0, 0);
  };

  // lgkmcnt(0): every fragment read of this stage has EXECUTED before the barrier lets other waves DMA into its slot
  // (see conv_igemm.hip ring_wait)
  auto ring_wait = [&]() {
    // retire the OLDEST stage in flight: the NS - 2 younger ones stay on the wire across the barrier
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * (NDO + NDI)) : "memory");
    __builtin_amdgcn_s_barrier();
  };

  if constexpr (DG) {
    static_assert(TCO == 256 && NW == 8, "fused data gradient: the 256-channel tile");
    if (wci == 1) {
      const int l15 = lane & 15, g4 = lane >> 4;
      const int pf = wco & 1, cp = wco >> 1;
      const bf16_t* wp = static_cast<const bf16_t*>(k.dgw);
      bf16x8 wdf[2][8];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
          wdf[i][ks] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)(32 * cp + 16 * i + l15) * k.cout + 32 * ks + 8 * g4);
      const int prow = 16 * pf + l15, psw = wg_swz(prow);
      bf16_t* yp = static_cast<bf16_t*>(k.dgy);
      auto dgc = [&](int slot_base, int st) __attribute__((always_inline)) {
        f32x4 da[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + slot_base + prow * LO + (((4 * ks + g4) ^ psw) << 4));
          da[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdf[0][ks], b, da[0], 0, 0, 0);
          da[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdf[1][ks], b, da[1], 0, 0, 0);
        }
        // rows = channels 4 g + r of fragment i, column = pixel l15: pair the fragments -> 8 consecutive channels per lane
        float v[8] = {da[0][0], da[0][1], da[0][2], da[0][3], da[1][0], da[1][1], da[1][2], da[1][3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) wg_swap16f(v[e], v[4 + e]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
        const int m = st * R + prow;
        if (m < k.M)
          *reinterpret_cast<bf16x8*>(yp + (int64_t)m * k.dgld + k.dgoff + 32 * cp + 16 * (g4 & 1) + 8 * (g4 >> 1)) = o;
      };
      dma(stage0, 0);
      dma(stage0 + 1, 1);
      ring_wait();
      for (int st = stage0;;) {
        dma(st + 2, 2); dgc(0 * BUF, st); ring_wait();
        if (++st >= stage1) break;
        dma(st + 2, 0); dgc(1 * BUF, st); ring_wait();
        if (++st >= stage1) break;
        dma(st + 2, 1); dgc(2 * BUF, st); ring_wait();
        if (++st >= stage1) break;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (k.ws) {                                       // this wave's (unused) partial-tile slots: zeros for the ordered reduce
        float4* wsp = k.ws + ((((int64_t)split_id * k.ntiles + tile_id) * NW + wave) * 16) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) wsp[i * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
  }

  f32x4 acc[4][FI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transpose-read addresses (loop-invariant): lane (g, q, p) reads row 8g+q (+4), channels c0+4p..+3
  const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  const int frow = 8 * g + q, fs = wg_swz(frow);
  int a_off[4], b_off[FI];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = frow * LO + (((wco * 8 + 2 * i + (p4 >> 1)) ^ fs) << 4) + (p4 & 1) * 8;
#pragma unroll
  for (int j = 0; j < FI; ++j) b_off[j] = TILEO + frow * LI + (((wci * (2 * FI) + 2 * j + (p4 >> 1)) ^ fs) << 4) + (p4 & 1) * 8;
  // The transposed reads go through inline asm: beside in-flight LDS-DMA, hipcc (ROCm 7.2) orders every
  // ds_read_tr intrinsic behind `s_waitcnt vmcnt(0)` -- it cannot tell which ring slot the read touches -- which drained
  // the look-ahead DMA before every stage (no overlap at all; the .s showed it).  As asm the reads are invisible to that
  // pass, so their completion is waited for by hand (lgkmcnt(0) before the MFMAs; the ring wait orders them against
  // the DMA that will overwrite the slot).
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)smem;
  auto rd = [&](int off, int rowb) __attribute__((always_inline)) {
    bf16x4 lo, hi;
    const uint32_t a0 = lds0 + (uint32_t)off, a1 = a0 + (uint32_t)(4 * rowb);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  };
  // column fragments in groups of 4: with FI = 8 the second group's reads are in flight while the first group's 16 MFMAs
  // issue (and the 256 x 256 tile stays under its 256-VGPR cap: 128 accumulators + 3 x 16 fragment registers)
  auto compute = [&](int slot_base) __attribute__((always_inline)) {
    bf16x8 a[4], b[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = rd(slot_base + a_off[i], LO);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = rd(slot_base + b_off[j], LI);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the MFMAs must not be scheduled above the wait: the asm results are only valid after it
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]));
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(b[j]));
    if constexpr (FI == 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] = rd(slot_base + b_off[4 + j], LI);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    if constexpr (FI == 8) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(c[j]));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], c[j], acc[i][4 + j], 0, 0, 0);
    }
  };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) dma(stage0 + i, i);
  ring_wait();
  for (int st = stage0;;) {
    bool done = false;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {              // unrolled: slot addresses are immediates
      if (!done) {
        dma(st + NS - 1, (sl + NS - 1) % NS); compute(sl * BUF); ring_wait();
        done = ++st >= stage1;
      }
    }
    if (done) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (k.ws) {
    float4* wp = k.ws + ((((int64_t)split_id * k.ntiles + tile_id) * NW + wave) * (4 * FI)) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < FI; ++j)
        wp[(i * FI + j) * 64] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    return;
  }
  // D[row = co][col]: lane holds co = 4*(lane>>4) + r and column lane & 15
  const int l15 = lane & 15;
#pragma unroll
  for (int j = 0; j < FI; ++j) {
    const int col = cit * TCI + wci * (16 * FI) + 16 * j + l15;
    uint32_t tap, ci;
    k.dcin.divmod((uint32_t)col, tap, ci);
    if (tap >= (uint32_t)k.ntaps) continue;
    const int widx = k.taps[tap].widx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * TCO + wco * 64 + 16 * i + 4 * g + r;
        if (co < k.cout) atomicAdd(k.dw + ((int64_t)co * k.wtaps + widx) * k.cin + ci, acc[i][j][r]);
      }
    }
  }
}

template <int TCO, int NW, bool DG = false, int TCI = 128, int NS = 3>
int launch_dma(WgradK& k, const sfk_wgrad_desc* d, hipStream_t s, int64_t* dry) {
  const int cols = d->ntaps * d->cin;
  const int cotiles = (d->cout + TCO - 1) / TCO;
  constexpr int FI = TCI / (NW / (TCO / 64)) / 16;
  k.citiles = (cols + TCI - 1) / TCI;
  k.nchunks = (k.M + MK - 1) / MK;
  const int base = cotiles * k.citiles;
  // pixel splits: one resident generation of workgroups (2 x 256 CUs for the 8-wave tile, 3 x 256 for the 4-wave one).
  // Every split adds a full copy of the tile to the fp32 atomic traffic (~1.3 TB/s chip-wide), so do not over-split.
  const int target8 = sfk_tune().wgrad_target_8w, target4 = sfk_tune().wgrad_target_4w;   // resident-block targets
  int splits = ((NW == 8 ? target8 : target4) + base - 1) / base;
  const int max_splits = (k.nchunks + 7) / 8;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  k.chunks_per_split = (k.nchunks + splits - 1) / splits;
  splits = (k.nchunks + k.chunks_per_split - 1) / k.chunks_per_split;
  k.ntiles = base;
  const int64_t need = (int64_t)splits * base * NW * (4 * FI) * 64 * 16;
  if (dry) { *dry = need; return SFK_OK; }
  if (k.ws && need > d->workspace_bytes) k.ws = nullptr;
  hipLaunchKernelGGL((conv_wgrad_dma_kernel<TCO, NW, DG, TCI, NS>), dim3((unsigned)(base * splits)), dim3(64 * NW), 0, s, k);
  SFK_CHECK_LAUNCH();
  if (k.ws) return launch_reduce<NW, TCO / 64, 4, FI>(k, splits, s);
  return SFK_OK;
}

// the fused data gradient rides on the 256 x 128 LDS-DMA tile with taps x cin = 64 (its second column half is idle)
bool dg_ok(const sfk_wgrad_desc* d) {
  if (!d->dg_w || d->x.dtype != SFK_BF16 || d->cout != 256 || d->cin != 64 || d->ntaps != 1) return false;
  if (d->taps[0].dt || d->taps[0].dh || d->taps[0].dw || d->gs[0] != 1 || d->gs[1] != 1 || d->gs[2] != 1) return false;
  if (d->x.t != d->dy.t || d->x.h != d->dy.h || d->x.w != d->dy.w) return false;
  const sfk_fmap* y = &d->dg_y;
  if (!sfk_fmap_ok(y) || y->dtype != SFK_BF16 || y->c != d->cin || y->n != d->x.n || y->t != d->x.t || y->h != d->x.h ||
      y->w != d->x.w || (y->ld % 8) || (y->c_off % 8) || (((uintptr_t)y->ptr) & 15) || (((uintptr_t)d->dg_w) & 15))
    return false;
  return sfk_fmap_bytes(&d->x) < 0x7FF00000ll && sfk_fmap_bytes(&d->dy) < 0x7FF00000ll;
}

int validate(const sfk_wgrad_desc* d) {
  if (!d || d->struct_size != sizeof(sfk_wgrad_desc)) return SFK_ERR_INVALID;   // ABI handshake (include/sfk.h)
  if (!d->dw) return SFK_ERR_INVALID;
  if (!sfk_fmap_ok(&d->x) || !sfk_fmap_ok(&d->dy)) return SFK_ERR_INVALID;
  if (d->x.dtype != d->dy.dtype || d->x.n != d->dy.n) return SFK_ERR_INVALID;
  if (d->cin != d->x.c || d->cout != d->dy.c) return SFK_ERR_INVALID;
  if (d->ntaps <= 0 || d->ntaps > SFK_MAX_TAPS || d->wtaps <= 0) return SFK_ERR_INVALID;
  for (int i = 0; i < d->ntaps; ++i)
    if (d->taps[i].widx >= d->wtaps) return SFK_ERR_INVALID;
  for (int a = 0; a < 3; ++a)
    if (d->gs[a] <= 0) return SFK_ERR_INVALID;
  if (sfk_fmap_pixels(&d->dy) >= (1ll << 31)) return SFK_ERR_UNSUPPORTED;
  if (!sfk_fmap_vec_ok(&d->x) || !sfk_fmap_vec_ok(&d->dy)) return SFK_ERR_UNSUPPORTED;
  if (sfk_fmap_bytes(&d->x) >= (1ll << 32) - 64 || sfk_fmap_bytes(&d->dy) >= (1ll << 32) - 64) return SFK_ERR_UNSUPPORTED;
  return SFK_OK;
}

template <typename T, int TCO, int TCI, int KS, bool GRAM = false>
int launch_cfg(WgradK& k, const sfk_wgrad_desc* d, hipStream_t s, int64_t* dry) {
  const int cols = d->ntaps * d->cin;
  const int cotiles = (d->cout + TCO - 1) / TCO;
  k.citiles = (cols + TCI - 1) / TCI;
  constexpr int R = MK * KS;
  k.nchunks = (k.M + R - 1) / R;
  // pixel splits: enough workgroups to cover the 256 CUs several times, at least 4 stages each
  const int base = cotiles * k.citiles;
  const int target = sfk_tune().wgrad_target_gen > 0 ? sfk_tune().wgrad_target_gen : 1024;
  int splits = (target + base - 1) / base;
  const int max_splits = (k.nchunks + 3) / 4;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  k.chunks_per_split = (k.nchunks + splits - 1) / splits;
  splits = (k.nchunks + k.chunks_per_split - 1) / k.chunks_per_split;
  k.ntiles = base;
  constexpr int FO = TCO / 32, FI = TCI / 32;
  if (dry) { *dry = (int64_t)splits * base * 4 * FO * FI * 64 * 16; return SFK_OK; }
  if (k.ws && (int64_t)splits * base * 4 * FO * FI * 64 * 16 > d->workspace_bytes) k.ws = nullptr;
  hipLaunchKernelGGL((conv_wgrad_kernel<T, TCO, TCI, KS, GRAM>), dim3((unsigned)(base * splits)), dim3(256), 0, s, k);
  SFK_CHECK_LAUNCH();
  if (k.ws) return launch_reduce<4, 2, FO, FI>(k, splits, s);
  return SFK_OK;
}

template <typename T>
int launch(const sfk_wgrad_desc* d, hipStream_t s, int64_t* dry = nullptr) {
  WgradK k;
  k.x = d->x.ptr; k.dy = d->dy.ptr; k.dw = d->dw;
  k.xt = d->x.t; k.xh = d->x.h; k.xw = d->x.w; k.xld = d->x.ld; k.xoff = d->x.c_off;
  k.dld = d->dy.ld; k.doff = d->dy.c_off;
  k.M = (int)sfk_fmap_pixels(&d->dy);
  k.drw.set(d->dy.w); k.drh.set(d->dy.h); k.drt.set(d->dy.t);
  k.gst = d->gs[0]; k.gsh = d->gs[1]; k.gsw = d->gs[2];
  k.cin = d->cin; k.cout = d->cout; k.wtaps = d->wtaps; k.ntaps = d->ntaps;
  k.dspt.set(d->cin / sfk_vec_of(d->x.dtype));
  k.dcin.set(d->cin);
  k.xbytes = (uint32_t)sfk_fmap_bytes(&d->x);
  k.dbytes = (uint32_t)sfk_fmap_bytes(&d->dy);
  for (int i = 0; i < SFK_MAX_TAPS; ++i) k.taps[i] = d->taps[i < d->ntaps ? i : 0];
  const int use_ws = sfk_tune().wgrad_use_workspace;
  k.ws = use_ws ? reinterpret_cast<float4*>(d->workspace) : nullptr;
  k.dgw = nullptr; k.dgy = nullptr; k.dgld = 0; k.dgoff = 0;
  const int cols = d->ntaps * d->cin;
  if constexpr (sizeof(T) == 2) {
    if ((dry || k.ws) && wgrad_p8_ok(d, k.M)) {      // the deep-pipelined 256 x 256 tile (conv_wgrad_p8.hip); workspace only
      const int r = launch_wgrad_p8(k, d, s, dry);
      if (r != SFK_ERR_UNSUPPORTED) return r;
    }
    if ((dry || k.ws) && wgrad_band_ok(d)) {
      const int r = launch_wgrad_band(k, d, s, dry);
      if (r != SFK_ERR_UNSUPPORTED) return r;
    }
  }
  if (sizeof(T) == 2 && cols >= 128 && d->cout >= 128 && k.xbytes < 0x7FF00000u && k.dbytes < 0x7FF00000u) {
    // wide layers: LDS-DMA ring; 256 output channels per tile once that still leaves enough workgroups
    const bool deep = (sfk_tune().wgrad_wide_co & 8) != 0;       // EXPERIMENT: 5-slot ring
    if (d->cout >= 256 && (int64_t)k.M * cols >= (1ll << 24)) return deep ? launch_dma<256, 8, false, 128, 5>(k, d, s, dry) : launch_dma<256, 8>(k, d, s, dry);
    return deep ? launch_dma<128, 4, false, 128, 5>(k, d, s, dry) : launch_dma<128, 4>(k, d, s, dry);
  }
  if constexpr (sizeof(T) == 2) {
    // wide output, narrow input (slow res2 conv_c: 64 -> 256): ONE tile holds all of dW, so x and dY are each read once
    const int wide_co = sfk_tune().wgrad_wide_co;
    if (d->dg_w) {                                      // fused data gradient: only on the tile below
      if (!dg_ok(d)) return SFK_ERR_UNSUPPORTED;
      k.dgw = d->dg_w; k.dgy = d->dg_y.ptr; k.dgld = d->dg_y.ld; k.dgoff = d->dg_y.c_off;
      return launch_dma<256, 8, true>(k, d, s, dry);
    }
    if ((wide_co & 2) && d->cout == 256 && cols == 64 && k.xbytes < 0x7FF00000u && k.dbytes < 0x7FF00000u)
      return launch_dma<256, 8>(k, d, s, dry);   // the LDS-DMA tile with its second column half idle still beats the
                                                 // register-staged 256 x 64 tile: 102..108 vs 133..136 us on slow res2's R
    if (wide_co && d->cout >= 256 && cols > 32 && cols <= 64) return launch_cfg<T, 256, 64, 1>(k, d, s, dry);
  }
  // Gram matrix (x and dy the same pointwise map, one tile): the tile is staged once for both operands
  const bool gram = d->x.ptr == d->dy.ptr && d->x.ld == d->dy.ld && d->x.c_off == d->dy.c_off && d->cin == d->cout && d->ntaps == 1 &&
                    d->taps[0].dt == 0 && d->taps[0].dh == 0 && d->taps[0].dw == 0 && d->gs[0] == 1 && d->gs[1] == 1 && d->gs[2] == 1 &&
                    (sfk_tune().wgrad_wide_co & 4);
  if (gram && cols <= 32) return launch_cfg<T, 32, 32, 4, true>(k, d, s, dry);
  if (gram && cols <= 64) return launch_cfg<T, 64, 64, 2, true>(k, d, s, dry);
  if (gram && cols > 64 && cols <= 128) return launch_cfg<T, 128, 128, 1, true>(k, d, s, dry);
  if (cols <= 32) {
    if (d->cout <= 32) return launch_cfg<T, 32, 32, 4>(k, d, s, dry);
    if (d->cout <= 64) return launch_cfg<T, 64, 32, 4>(k, d, s, dry);
    return launch_cfg<T, 128, 32, 2>(k, d, s, dry);
  }
  if (d->cout <= 32) return launch_cfg<T, 32, 128, 2>(k, d, s, dry);
  if (d->cout <= 64) return launch_cfg<T, 64, 128, 2>(k, d, s, dry);
  return launch_cfg<T, 128, 128, 1>(k, d, s, dry);
}

}  // namespace

extern "C" int64_t sfk_conv_wgrad_workspace_bytes(const sfk_wgrad_desc* d) {
  const int st = validate(d);
  if (st != SFK_OK) return st;
  int64_t bytes = 0;
  const int r = d->x.dtype == SFK_BF16 ? launch<bf16_t>(d, nullptr, &bytes) : launch<float>(d, nullptr, &bytes);
  return r != SFK_OK ? r : bytes;
}

extern "C" int sfk_conv_wgrad_wants_workspace(const sfk_wgrad_desc* d) {
  if (!d || validate(d) != SFK_OK || d->dg_w) return 0;
  const int M = (int)sfk_fmap_pixels(&d->dy);
  return (wgrad_p8_ok(d, M) || wgrad_band_ok(d)) ? 1 : 0;
}

extern "C" int sfk_conv_wgrad_dg_supported(const sfk_wgrad_desc* d) {
  return (d && validate(d) == SFK_OK && dg_ok(d)) ? 1 : 0;
}

extern "C" int sfk_conv_wgrad(const sfk_wgrad_desc* d, sfk_stream_t stream) {
  const int st = validate(d);
  if (st != SFK_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return d->x.dtype == SFK_BF16 ? launch<bf16_t>(d, s) : launch<float>(d, s);
}
