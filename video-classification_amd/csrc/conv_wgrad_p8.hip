// conv_wgrad_p8: the deep-pipelined filter-gradient kernel of the MFMA-bound layers (bf16, cin % 128 == 0, cout >= 256).
//
//   dW[co][col] = sum_pixel dY[pixel][co] * X[gather(pixel, tap(col))][ci(col)],   col = (tap, ci) flattened
//
// The structure of conv_igemm_p8.hip with the pixel axis as the GEMM's K: ONE workgroup per CU, 8 waves, a 256 (co) x 256
// (columns) tile, K-tiles of 64 pixels, all LDS in one array (2 K-tiles x 64 KB), LDS-DMA in flight across raw s_barriers behind
// a counted vmcnt, the two wave groups of a SIMD half a phase apart.  The pixel range is split over workgroups so that tiles x
// splits fills the chip once; a split leaves its partial tile in the caller's workspace and wgrad_reduce_kernel adds them in
// split order (conv_wgrad_common.h) -- deterministic, no atomics.
//
//   wave (g, wc): g = wave >> 2 owns output channels 128g .. 128g+127, wc = wave & 3 columns 64wc .. 64wc+63: 8 x 4 fragments.
//   Both operands lie [pixel][channel] in HBM and in LDS; the MFMA fragments (8 consecutive pixels of one channel per lane)
//   come out transposed by ds_read_b64_tr_b16, two reads per fragment and 32-pixel sub-step.
//   LDS per parity b: D_g = dY[64 pixels][128 co of group g] (256-byte rows) at b*64K + g*16K, X[64 pixels][256 columns]
//   (512-byte rows) at b*64K + 32K; 32-byte blocks XOR-swizzled inside each 256-byte window by (row & 3) | ((row >> 3) & 1) << 2,
//   on the source side of the DMA: the eight rows a 32-lane half of a transposed read touches land on eight bank windows.
//   A K-tile is four phases of 16 MFMAs:   A: pixels 0..31, co fragments 0-3   B: pixels 0..31, co 4-7 (column fragments kept)
//                                          C: pixels 32..63, co 0-3            D: pixels 32..63, co 4-7
//   with 16 / 8 / 16 / 8 transposed reads; a phase is  L: reads, 2 LDS-DMAs, s_waitcnt vmcnt(8), s_barrier
//                                                       M: s_waitcnt lgkmcnt(0), 16 MFMAs at raised priority, s_barrier
//   and group 1 runs one barrier behind group 0 (conv_igemm_p8.hip has the hazard argument; the same rules give this schedule):
//       units of K-tile t: X-p0 / X-p1 = pixels 0..31 / 32..63 of the X image (all 8 waves), D-p0 / D-p1 = the same of D_g (the
//       group's own 4 waves); reads: X-p0 in A, D-p0 in A + B, X-p1 in C, D-p1 in C + D
//       phase A of K-tile t issues X-p1(t+1), B issues D-p1(t+1), C issues X-p0(t+2), D issues D-p0(t+2):
//       every unit is issued >= 5 phases before its first read and, for the shared X units, >= 2 phases after the last read of
//       the slot's previous contents (>= 1 for the group-private D units); vmcnt(8) = all but the newest 4 phases have landed.
//
// Gather.  dY is linear in the pixel: per-lane offsets are loop-invariant, the K-tile rides the SGPR soffset, rows past M lie
// past the buffer and read zeros.  X is gathered: a DMA wave-instruction covers 2 pixel rows x 256 columns, and a 128-column
// half of the tile lies inside one tap (cin % 128 == 0), so a K-tile needs 16 distinct (row, half) source offsets per wave.
// They are computed ONCE per K-tile, at the top of phase B's M segment, under the latency of that phase's fragment reads
// (vector instructions issued from L compete with the partner wave's MFMAs): lane l does (instruction l & 3, row l >> 5, half (l >> 4) & 1), and ds_bpermute hands every DMA lane its value.
#include "conv_wgrad_common.h"

namespace sfk_wgrad {

typedef __attribute__((address_space(3))) void lds_void_w8_t;

__global__ __launch_bounds__(512, 2) void conv_wgrad_p8_kernel(const WgradK k) {
  constexpr int PAR = 65536, DG = 16384, XB = 32768, DROW = 256, XROW = 512;
  constexpr uint32_t FAR = 0x80000000u;
  __shared__ __attribute__((aligned(16))) char smem[2 * PAR];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;
  int tile_id, split_id;
  wg_block(k.ntiles, tile_id, split_id);
  const int cot = tile_id / k.citiles, cit = tile_id % k.citiles;
  const int kt0 = split_id * k.chunks_per_split;
  const int kt1 = min(kt0 + k.chunks_per_split, k.nchunks);
  if (kt0 >= kt1) return;                              // (whole workgroup: before any barrier)
  const int KT = kt1 - kt0;
  const __amdgpu_buffer_rsrc_t xrs = sfk_make_rsrc(k.x, k.xbytes);
  const __amdgpu_buffer_rsrc_t drs = sfk_make_rsrc(k.dy, k.dbytes);

  // ---- dY DMA lanes: instruction j of unit p covers rows 32p + 8wc + 4j + (lane >> 4) of D_g, 16-byte slot lane & 15
  uint32_t dvo[2];
  {
    const int r4 = lane >> 4, s = lane & 15;
    const int f = r4 | ((wc & 1) << 2);
    const int co = cot * 256 + g * 128 + (((s >> 1) ^ f) << 4) + (s & 1) * 8;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      dvo[j] = co < k.cout ? (uint32_t)(((8 * wc + 4 * j + r4) * k.dld + k.doff + co) * 2) : FAR;
  }
  // ---- X DMA lanes: instruction j of unit p covers rows 32p + 4*wave + 2j + (lane >> 5), 16-byte slot lane & 31
  uint32_t xcb[2];       // channel bytes of this lane's segment in instruction j (the row's XOR differs by 2 blocks = 32 channels)
  int tdt = 0, tdh = 0, tdw = 0;       // tap of this lane's column half
  bool tok;
  {
    const int rp = lane >> 5, s = lane & 31, hf = s >> 4;
    const int f = rp | (((wave >> 1) & 1) << 2);
    const int col = cit * 256 + hf * 128 + ((((s >> 1) & 7) ^ f) << 4) + (s & 1) * 8;
    uint32_t tap, ci;
    k.dcin.divmod((uint32_t)col, tap, ci);
    tok = tap < (uint32_t)k.ntaps;
    const sfk_tap tp = k.taps[tok ? tap : 0];
    tdt = tp.dt; tdh = tp.dh; tdw = tp.dw;
    xcb[0] = (uint32_t)((k.xoff + (int)ci) * 2);
    xcb[1] = (uint32_t)((k.xoff + (int)(ci ^ 32u)) * 2);
  }
  const int bp_base = (lane & 0x30) * 4;               // ds_bpermute byte address of lane (l & 0x30) | instruction

  // source offsets of the four X instructions of K-tile kt (absolute index): [unit p][instruction j]
  uint32_t xo[2][2];
  auto compute_xo = [&](const int kt) __attribute__((always_inline)) {
    const int jj = lane & 3;
    const int r = 32 * (jj >> 1) + 4 * wave + 2 * (jj & 1) + (lane >> 5);
    const int m = kt * 64 + r;
    uint32_t q1, rw_, q2, rh_, n_, rt_;
    k.drw.divmod((uint32_t)m, q1, rw_);
    k.drh.divmod(q1, q2, rh_);
    k.drt.divmod(q2, n_, rt_);
    const int ti = (int)rt_ * k.gst + tdt, hi = (int)rh_ * k.gsh + tdh, wi = (int)rw_ * k.gsw + tdw;
    const bool ok = tok && m < k.M && (unsigned)ti < (unsigned)k.xt && (unsigned)hi < (unsigned)k.xh && (unsigned)wi < (unsigned)k.xw;
    const uint32_t pix = (((uint32_t)n_ * k.xt + ti) * k.xh + hi) * k.xw + wi;
    const uint32_t po = ok ? pix * (uint32_t)(k.xld * 2) : FAR;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        xo[p][j] = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_base + (2 * p + j) * 4, (int)po) + xcb[j];
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) as an instruction hipcc's bookkeeping sees: the ds_bpermutes are retired
                                             // HERE, not behind the next phase's (untracked) transposed reads
  };
  // K-tiles past the split's range (the look-ahead of its last two) re-stage its first one: nobody reads those slots
  auto clampk = [&](const int t) __attribute__((always_inline)) { return kt0 + (t < KT ? t : 0); };
  auto issue_x = [&](const int b, const int p) __attribute__((always_inline)) {
    char* dst = smem + b * PAR + XB + (32 * p + 4 * wave) * XROW;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_w8_t*)(dst + 2 * j * XROW), 16, (int)xo[p][j], 0, 0, 0);
  };
  auto issue_d = [&](const int b, const int p, const int t) __attribute__((always_inline)) {
    char* dst = smem + b * PAR + g * DG + (32 * p + 8 * wc) * DROW;
    const int so = __builtin_amdgcn_readfirstlane((clampk(t) * 64 + 32 * p) * k.dld * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(drs, (lds_void_w8_t*)(dst + 4 * j * DROW), 16, (int)dvo[j], so, 0, 0);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- transposed-read addresses: lane (g4, q, p4) addresses row 8 g4 + q (+ 4) of a 32-pixel block, 4 channels at 8 p4 bytes
  // of the fragment's 32-byte block; the block index carries the row's XOR, so one address register per fragment and parity
  // (sub-step and the + 4 rows are immediates)
  uint32_t aa[2][8], ba[2][4];
  {
    const int g4 = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
    const int f = q | ((g4 & 1) << 2);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_w8_t*)smem;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int i = 0; i < 8; ++i) aa[b][i] = lds0 + (uint32_t)(b * PAR + g * DG + (8 * g4 + q) * DROW + ((i ^ f) << 5) + p4 * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        ba[b][j] = lds0 + (uint32_t)(b * PAR + XB + (8 * g4 + q) * XROW + (wc >> 1) * 256 + (((4 * (wc & 1) + j) ^ f) << 5) + p4 * 8);
    }
  }
  // the transposed reads go through inline asm (beside in-flight LDS-DMA hipcc orders the intrinsic behind vmcnt(0)); their
  // completion is waited for by hand at the top of M
  bf16x4 af[4][2], bfr[4][2];                            // [fragment][rows q / q + 4]
  auto read_a = [&](const int b, const int s, const int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(af[i][0]) : "v"(aa[b][i0 + i]), "i"(s * 32 * DROW));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(af[i][1]) : "v"(aa[b][i0 + i]), "i"(s * 32 * DROW + 4 * DROW));
    }
  };
  auto read_b = [&](const int b, const int s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bfr[j][0]) : "v"(ba[b][j]), "i"(s * 32 * XROW));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bfr[j][1]) : "v"(ba[b][j]), "i"(s * 32 * XROW + 4 * XROW));
    }
  };
  auto frag = [](const bf16x4& lo, const bf16x4& hi) __attribute__((always_inline)) {
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  };
  auto mfma = [&](const int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(af[i][0], af[i][1]), frag(bfr[j][0], bfr[j][1]), acc[i0 + i][j], 0, 0, 0);
  };
  // kx >= 0 (phase B): the X source offsets of K-tile kx are computed between the barrier and the wait for this phase's
  // fragment reads -- ~35 vector instructions that would otherwise sit in front of the first MFMA pass under the LDS latency
  auto end_l = [&](const int kx = -1) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kx >= 0) compute_xo(kx);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the asm reads' results are valid only behind the wait: no MFMA above it
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(af[i][0]), "+v"(af[i][1]), "+v"(bfr[i][0]), "+v"(bfr[i][1]));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
  };
  auto end_m = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // the four phases of K-tile t (relative to kt0; parity b = t & 1, a literal after unrolling)
  auto ktile = [&](const int b, const int t) __attribute__((always_inline)) {
    read_a(b, 0, 0); read_b(b, 0); issue_x(b ^ 1, 1);        end_l(); mfma(0);                         end_m();   // X-p1(t+1)
    read_a(b, 0, 4);               issue_d(b ^ 1, 1, t + 1); end_l(clampk(t + 2)); mfma(4);             end_m();   // D-p1(t+1)
    read_a(b, 1, 0); read_b(b, 1); issue_x(b, 0);            end_l(); mfma(0);                         end_m();   // X-p0(t+2)
    read_a(b, 1, 4);               issue_d(b, 0, t + 2);     end_l(); mfma(4);                         end_m();   // D-p0(t+2)
  };

  // prologue: K-tile 0 and pixels 0..31 of K-tile 1
  compute_xo(clampk(0));
  issue_x(0, 0); issue_d(0, 0, 0); issue_x(0, 1); issue_d(0, 1, 0);
  compute_xo(clampk(1));
  issue_x(1, 0); issue_d(1, 0, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
  for (int t = 0;;) {
    ktile(0, t);
    if (++t >= KT) break;
    ktile(1, t);
    if (++t >= KT) break;
  }
  if (g == 0) __builtin_amdgcn_s_barrier();      // ... and group 0 meets its last one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // partial tile as it lies in the accumulators (wgrad_reduce_kernel<8, 2, 8, 4>: wave slot = g + 2 wc, fragment = 4 i + j)
  float4* wp = k.ws + ((((int64_t)split_id * k.ntiles + tile_id) * 8 + (g + 2 * wc)) * 32) * 64 + lane;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      wp[(i * 4 + j) * 64] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
}

// MFMA-bound layers only: the tile needs 128-column halves inside one tap, a co tile that is mostly real, and enough pixels that
// one workgroup per CU still runs a dozen K-tiles
__attribute__((visibility("hidden"))) bool wgrad_p8_ok(const sfk_wgrad_desc* d, int M) {
  if (!sfk_tune().wgrad_p8 || d->x.dtype != SFK_BF16 || d->dg_w) return false;
  const int cols = d->ntaps * d->cin;
  if ((d->cin % 128) != 0 || cols < 256 || d->cout < 256 || (d->cout % 256 != 0 && d->cout < 512)) return false;
  if (sfk_fmap_bytes(&d->x) >= 0x7FF00000ll || sfk_fmap_bytes(&d->dy) >= 0x7FF00000ll) return false;
  const int base = ((d->cout + 255) / 256) * ((cols + 255) / 256);
  if (base > 256) return false;
  const int splits = 256 / base;
  const int nkt = (M + 63) / 64;
  return nkt / splits >= sfk_tune().wgrad_p8;            // (the knob is the minimum K-tile count per workgroup)
}

__attribute__((visibility("hidden"))) int launch_wgrad_p8(WgradK& k, const sfk_wgrad_desc* d, hipStream_t s, int64_t* dry) {
  const int cols = d->ntaps * d->cin;
  k.citiles = (cols + 255) / 256;
  k.ntiles = ((d->cout + 255) / 256) * k.citiles;
  k.nchunks = (k.M + 63) / 64;
  int splits = 256 / k.ntiles;
  if (splits < 1) splits = 1;
  k.chunks_per_split = (k.nchunks + splits - 1) / splits;
  splits = (k.nchunks + k.chunks_per_split - 1) / k.chunks_per_split;
  const int64_t need = (int64_t)splits * k.ntiles * 8 * 32 * 64 * 16;
  if (dry) { *dry = need; return SFK_OK; }
  if (!k.ws || need > d->workspace_bytes) return SFK_ERR_UNSUPPORTED;      // (the caller falls back to the ring kernels)
  hipLaunchKernelGGL(conv_wgrad_p8_kernel, dim3((unsigned)(k.ntiles * splits)), dim3(512), 0, s, k);
  SFK_CHECK_LAUNCH();
  return launch_reduce<8, 2, 8, 4>(k, splits, s);
}

}  // namespace sfk_wgrad
