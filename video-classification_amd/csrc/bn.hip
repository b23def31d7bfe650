// BatchNorm3d (+ReLU, +residual add) forward / backward on channels-last feature maps.
// All of these are HBM-bound streaming kernels: 16 B per lane, a thread owns ONE group of VEC channels for its
// whole lifetime (per-channel coefficients live in registers) and walks pixels; per-channel reductions end
// in a per-block partial row that a tiny finalize kernel folds in double precision (deterministic, no atomics).
#include "sfk_common.h"

// Cache hints of the three streaming kernels: `nt` bit 0 = non-temporal stores, bit 1 = non-temporal loads, chosen per
// launch from the map's size (nt_hint): a map far larger than the 256 MB Infinity Cache is streamed (its lines would only
// push other lanes' working sets out), a small one stays cached -- bn_bwd_apply re-reads what bn_bwd_reduce just read.

namespace {
// thresholds in MB of ONE map (pixels * channels * element size): sfk_tuning.nt_apply_mb / nt_reduce_mb / nt_bwd_apply_mb
inline int nt_hint(const sfk_fmap* f, int64_t mb, int bits) {
  const int64_t bytes = sfk_fmap_pixels(f) * f->c * (f->dtype == SFK_BF16 ? 2 : 4);
  return (mb >= 0 && bytes >= (mb << 20)) ? bits : 0;
}

struct FM {  // kernel-side feature map view
  void* p;
  int ld, off;
};
inline FM fm_of(const sfk_fmap* f) { return FM{f ? f->ptr : nullptr, f ? f->ld : 0, f ? f->c_off : 0}; }

// thread -> (channel group, first pixel, pixel step) ; blockDim = 256, grid = (pixel parts, channel-group chunks)
struct ChanMap {
  int cg, row, rows_b;
  bool active;
  __device__ __forceinline__ ChanMap(int cgs) {
    const int cgs_b = cgs < 256 ? cgs : 256;
    rows_b = 256 / cgs_b;
    const int tx = threadIdx.x % cgs_b, ty = threadIdx.x / cgs_b;
    cg = blockIdx.y * 256 + tx;
    row = ty;
    active = ty < rows_b && cg < cgs;
  }
};

// The streaming kernels walk the map in SPANS of SFK_BN_U * rows_b consecutive pixel rows: a thread issues the loads of its
// SFK_BN_U rows before the first use, and a block touches one contiguous stretch of memory (measured on bn_bwd_apply:
// 5.22 ms per step with one row per iteration and a 2048-block grid-stride loop, 4.93 with 4 rows per iteration, 4.61 with
// one span per thread and as many blocks as there are spans).
#ifndef SFK_BN_U_ROWS
#define SFK_BN_U_ROWS 4
#endif
constexpr int SFK_BN_U = SFK_BN_U_ROWS;
inline unsigned span_blocks(int cgs, int64_t pixels) {
  const int cgs_b = cgs < 256 ? cgs : 256, rows_b = 256 / cgs_b;
  const int64_t want = (pixels + (int64_t)rows_b * SFK_BN_U - 1) / ((int64_t)rows_b * SFK_BN_U);
  return (unsigned)(want < 65535 ? (want > 0 ? want : 1) : 65535);
}
inline dim3 chan_grid(int cgs, int64_t pixels, int max_parts, int* nparts) {
  const int cgs_b = cgs < 256 ? cgs : 256;
  const int rows_b = 256 / cgs_b;
  int64_t parts = (pixels + (int64_t)rows_b * 16 - 1) / ((int64_t)rows_b * 16);  // >= 16 pixels per thread
  const int cchunks = (cgs + 255) / 256;
  const int cap_all = sfk_tune().bn_parts;   // (engine.MAX_PARTS follows it)
  int64_t cap = cap_all / cchunks;
  if (cap < 1) cap = 1;
  if (parts > cap) parts = cap;
  if (max_parts > 0 && parts > max_parts) parts = max_parts;
  if (parts < 1) parts = 1;
  *nparts = (int)parts;
  return dim3((unsigned)parts, (unsigned)cchunks);
}

// COH: device-coherent loads (agent-scope atomics) -- the values were written by OTHER workgroups of this launch (the fused
// finalize prologue) and an acquire fence per workgroup would invalidate the L2 thousands of times per launch (first build of
// sfk_bn_finalize_apply: 900 us against 40)
template <int VEC, bool COH = false>
__device__ __forceinline__ void load_coef(float (&dst)[VEC], const float* src, int cg) {
#pragma unroll
  for (int i = 0; i < VEC; ++i)
    dst[i] = COH ? __hip_atomic_load(src + cg * VEC + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[cg * VEC + i];
}

// block-level reduction of per-thread (a[VEC], b[VEC]) over the threads that share a channel group.  The threads' values
// form a matrix [rows_b][W] in LDS, W = cgs_b * 2 * VEC columns (the layout of one partial row); a thread then owns ONE column
// and 1 / G of the rows (G = 256 / W row groups, in a fixed order: deterministic), and W threads add the G group sums.  One
// accumulator register per thread: the callers sit at their VGPR caps (bn_pool_bwd_kernel: 128), and the first version -- row
// lane 0 of every channel group walking all rows_b rows of 2 * VEC values -- was a serial tail of up to 256 x 16 LDS reads on
// the 8 .. 32-channel maps of the fast pathway (5 .. 8 us of a 17 us kernel).
template <int VEC>
__device__ __forceinline__ void block_reduce_store(const ChanMap& cm, int cgs, const float (&a)[VEC],
                                                   const float (&b)[VEC], float* partials, int c) {
  __shared__ float red[256 * 2 * VEC];
  float* mine = red + threadIdx.x * 2 * VEC;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    mine[2 * i] = cm.active ? a[i] : 0.f;
    mine[2 * i + 1] = cm.active ? b[i] : 0.f;
  }
  __syncthreads();
  const int cgs_b = cgs < 256 ? cgs : 256;
  const int W = cgs_b * 2 * VEC, rows = cm.rows_b, tid = threadIdx.x;
  float* out = partials + (int64_t)blockIdx.x * c * 2 + (int64_t)blockIdx.y * 256 * VEC * 2;
  const int ncol = (cgs - (int)blockIdx.y * 256 < cgs_b ? cgs - (int)blockIdx.y * 256 : cgs_b) * 2 * VEC;   // columns that exist
  if (W >= 256) {                 // rows <= 16: a thread adds its column(s) over all rows
    for (int col = tid; col < W; col += 256) {
      float sacc = 0.f;
      for (int r = 0; r < rows; ++r) sacc += red[r * W + col];
      if (col < ncol) out[col] = sacc;
    }
    return;
  }
  const int G = 256 / W;          // >= 2 row groups (W <= 128)
  const int g = tid / W, col = tid - g * W;
  float sacc = 0.f;
  if (g < G)
    for (int r = g; r < rows; r += G) sacc += red[r * W + col];
  __syncthreads();                // every first-level read is done before `red` is overwritten
  if (g < G) red[g * W + col] = sacc;
  __syncthreads();
  if (tid < W) {
    sacc = 0.f;
    for (int q = 0; q < G; ++q) sacc += red[q * W + tid];
    if (tid < ncol) out[tid] = sacc;
  }
}

// ------------------------------------------------------------------ forward statistics (stand-alone)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(FM y, int64_t pixels, int c, float* partials) {
  constexpr int VEC = DT<T>::VEC;
  const int cgs = c / VEC;
  const ChanMap cm(cgs);
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  if (cm.active) {
    const T* yp = static_cast<const T*>(y.p) + y.off + cm.cg * VEC;
    for (int64_t p = (int64_t)blockIdx.x * cm.rows_b + cm.row; p < pixels; p += (int64_t)gridDim.x * cm.rows_b) {
      Vec16<T> v;
      v.load(yp + p * y.ld);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float f = v.get(i);
        s1[i] += f;
        s2[i] += f * f;
      }
    }
  }
  block_reduce_store<VEC>(cm, cgs, s1, s2, partials, c);
}

// Folding the partial rows.  A conv over the benchmark's res2 maps leaves 3,136 rows x 256 channels (the fast stem
// 50,176 x 8), so the fold is split: bn_fold_kernel reduces `nparts` rows to <= SFK_BN_FOLD_ROWS rows with coalesced
// reads (64 adjacent channels per wave-row = 512 B), then one wave per channel folds those in the finalize kernel.
// Both levels accumulate in double and in a fixed order: deterministic, no atomics.
// block = (256 / cw) row lanes x cw channel lanes, cw = min(64, c rounded up to a power of two): the narrow maps of the
// fast pathway (8..32 channels, up to 50,176 rows) keep all 256 threads busy instead of c of every 64.
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ partials, int nparts, int c, int per,
                                                      float* __restrict__ out, int cw) {
  __shared__ double red[2][256];
  const int nrl = 256 / cw;
  const int tx = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int ch = blockIdx.y * cw + tx;
  const int r0 = blockIdx.x * per;
  int r1 = r0 + per;
  if (r1 > nparts) r1 = nparts;
  double s1 = 0.0, s2 = 0.0;
  if (ch < c) {
#pragma unroll 4
    for (int p = r0 + rl; p < r1; p += nrl) {
      const float2 v = *reinterpret_cast<const float2*>(partials + ((int64_t)p * c + ch) * 2);
      s1 += (double)v.x;
      s2 += (double)v.y;
    }
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  if (rl == 0 && ch < c) {
    double a = 0.0, b = 0.0;
    for (int r = 0; r < nrl; ++r) {          // fixed order: deterministic
      a += red[0][r * cw + tx];
      b += red[1][r * cw + tx];
    }
    *reinterpret_cast<float2*>(out + ((int64_t)blockIdx.x * c + ch) * 2) = make_float2((float)a, (float)b);
  }
}

// returns the rows left to fold (in `*rows_out`) and where they are; launches the first level when it pays: the finalize
// kernels give a whole 256-thread block to a channel pair, so up to SFK_BN_DIRECT_ROWS rows (16 loads per thread, all in
// flight at once) are folded there and the step's ~170 BatchNorms do without this launch (5.6 us each on the chain:
// finalize 10 -> 4..5 us).  Not for the widest tables (3,136 rows x 256 channels = 6.4 MB read through 16-byte segments of
// 64-byte sectors: 12.4 us in one launch against 10 in two)
constexpr int SFK_BN_DIRECT_ROWS = 4096;
inline const float* fold_partials(const float* partials, int nparts, int c, float* workspace, int* rows_out,
                                  hipStream_t s) {
  if (!workspace || (nparts <= SFK_BN_DIRECT_ROWS && (int64_t)nparts * c <= SFK_BN_DIRECT_ROWS * 128)) {
    *rows_out = nparts;
    return partials;
  }
  const int per = (nparts + SFK_BN_FOLD_ROWS - 1) / SFK_BN_FOLD_ROWS;
  const int rows = (nparts + per - 1) / per;
  int cw = 8;
  while (cw < c && cw < 64) cw <<= 1;
  hipLaunchKernelGGL(bn_fold_kernel, dim3(rows, (c + cw - 1) / cw), dim3(256), 0, s, partials, nparts, c, per, workspace, cw);
  *rows_out = rows;
  return workspace;
}

// one 256-thread block per channel PAIR: thread t folds rows t, t + 256, ... (one 16-byte load per row: (s1, s2) of both
// channels) in double, then a fixed-order tree over the block -- deterministic.  On return threads 0 and 1 hold the sums of
// channels ch0 and ch0 + 1.
__device__ __forceinline__ void block_sum_partials(const float* __restrict__ partials, int nparts, int c, int ch0,
                                                   double& s1, double& s2) {
  __shared__ double red[4][256];
  const int tid = threadIdx.x;
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  const float* base = partials + (int64_t)ch0 * 2;
#pragma unroll 4
  for (int p = tid; p < nparts; p += 256) {
    const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)p * c * 2);
    a0 += (double)v.x;
    a1 += (double)v.y;
    b0 += (double)v.z;
    b1 += (double)v.w;
  }
  red[0][tid] = a0;
  red[1][tid] = a1;
  red[2][tid] = b0;
  red[3][tid] = b1;
  __syncthreads();
#pragma unroll
  for (int st = 128; st >= 1; st >>= 1) {
    if (tid < st) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[j][tid] += red[j][tid + st];
    }
    __syncthreads();
  }
  s1 = red[(tid & 1) * 2][0];
  s2 = red[(tid & 1) * 2 + 1][0];
}

// what one 256-thread block does for channel pair `pair` (channels 2 pair, 2 pair + 1)
struct FinK {
  const float* partials;       // nullptr: no fused finalize
  int nparts, c;
  double count;
  const float* gamma; const float* beta;
  float eps, momentum;
  float* running_mean; float* running_var;
  int64_t* nbt;
  float* mean; float* invstd; float* scale; float* shift;
  int* sync;                   // [SFK_FIN_SYNC_INTS] zero before the first launch (counters of bn_fused_prologue / _epilogue)
};

// COH: the results go out as device-coherent (write-through) stores -- the fused launches' readers are other workgroups of the
// SAME launch, and a release FENCE per folding workgroup writes that XCD's whole L2 back (measured: 0.23 us per channel pair,
// serialised, while the apply's own output lines are dirty)
template <bool COH>
__device__ __forceinline__ void st_coef(float* p, float v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

template <bool COH = false>
__device__ __forceinline__ void bn_finalize_pair(const FinK& f, int pair) {
  const int ch = pair * 2 + (threadIdx.x & 1);         // c is a multiple of 2 (checked by the callers)
  double s1, s2;
  block_sum_partials(f.partials, f.nparts, f.c, pair * 2, s1, s2);
  if (threadIdx.x > 1) return;
  const double mu = s1 / f.count;
  double var = s2 / f.count - mu * mu;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)f.eps));
  const float sc = f.gamma[ch] * is;
  st_coef<COH>(f.mean + ch, (float)mu);
  st_coef<COH>(f.invstd + ch, is);
  st_coef<COH>(f.scale + ch, sc);
  st_coef<COH>(f.shift + ch, f.beta[ch] - (float)mu * sc);
  if (f.running_mean) {
    const double unb = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
    f.running_mean[ch] = (1.f - f.momentum) * f.running_mean[ch] + f.momentum * (float)mu;
    f.running_var[ch] = (1.f - f.momentum) * f.running_var[ch] + f.momentum * (float)unb;
  }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const FinK f) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && f.nbt) f.nbt[0] += 1;
  bn_finalize_pair(f, blockIdx.x);
}

// ---- the finalize as the PROLOGUE of its consumer (sfk_bn_finalize_apply, sfk_bn_bwd_finalize_apply).
// The step's 174 finalize launches are 8 us kernels on dependent chains, each behind a dispatch gap: with them skipped by the
// scheduler the step ran 26.4 instead of 27.5 ms.  Here the consumer's first workgroups to run fold the channel pairs -- the same
// deterministic block sums as the stand-alone kernel -- release their results (fence, then a counter), and every workgroup
// waits for the counter before it reads the coefficients: the fold runs under the dispatch ramp of the consumer's own grid.
// The last workgroup to leave zeroes the counters for the next launch.
constexpr int SFK_FIN_MAX_C = 512;                // widest BatchNorm the fused launches take (coefficients staged in LDS, pairs claimed one by one)
constexpr int SFK_FIN_GROUPS = 64;              // leave counters (sync[3 ..]): a read-modify-write on ONE address costs ~60 ns from any XCD
// every counter on its own 128-byte line: atomics on ONE line serialise (~25 ns each, whatever the address inside it)
constexpr int SFK_FIN_LINE = 32, SFK_FIN_DONE = 0, SFK_FIN_CLAIM = SFK_FIN_LINE, SFK_FIN_TOP = 2 * SFK_FIN_LINE, SFK_FIN_LEAVE = 3 * SFK_FIN_LINE;
constexpr int SFK_FIN_SYNC_INTS = SFK_FIN_LEAVE + SFK_FIN_GROUPS * SFK_FIN_LINE;
static_assert(SFK_FIN_SYNC_INTS == SFK_BN_SYNC_INTS, "include/sfk.h");

template <class Pair>
__device__ __forceinline__ void bn_fused_prologue(int* sync, int npairs, Pair&& do_pair) {
  // Channel pairs are CLAIMED (one counter) by running workgroups, not assigned by block id: with the other lanes' kernels on the
  // chip fewer workgroups than pairs may be resident, and a workgroup that waits for a pair owned by one that cannot be
  // dispatched until somebody leaves never leaves (first build: the step hung).  Only the first 2 npairs workgroups claim
  // (dispatch is in block order, so whenever anybody is resident some of them are running or done; 2,000 workgroups starting
  // together would otherwise queue 2,000 atomics on one address at ~60 ns each), and a workgroup only waits after the claim
  // counter passed npairs, i.e. every pair is in the hands of a RUNNING workgroup.
  __shared__ int s_pair;
  const int bid = blockIdx.y * gridDim.x + blockIdx.x;
  int mine = 0;
  if (bid < 2 * npairs) {
    for (;;) {
      if (threadIdx.x == 0)
        s_pair = __hip_atomic_load(&sync[SFK_FIN_CLAIM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= npairs ? npairs : atomicAdd(&sync[SFK_FIN_CLAIM], 1);
      __syncthreads();
      const int p = s_pair;
      __syncthreads();                                  // (s_pair and block_sum_partials' LDS are reused)
      if (p >= npairs) break;
      do_pair(p);
      ++mine;
    }
  }
  if (mine) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this thread's coherent stores are acknowledged (no fence: see st_coef) ...
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sync[SFK_FIN_DONE], mine);    // ... before the count says so
  }
  if (threadIdx.x == 0) {
    while (__hip_atomic_load(&sync[SFK_FIN_DONE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < npairs) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();                                      // (no acquire fence: the consumers read the coefficients with coherent loads)
}
// every workgroup reports that it is past the wait -- on one of 64 counters (6,000 workgroups on ONE counter cost 400 us); the
// workgroup that completes a counter reports to sync[1], the one that completes that zeroes everything for the next launch
__device__ __forceinline__ void bn_fused_epilogue(int* sync) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nblk = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int grp = bid % SFK_FIN_GROUPS, ngrp = nblk < SFK_FIN_GROUPS ? nblk : SFK_FIN_GROUPS;
    const int members = (nblk - grp + SFK_FIN_GROUPS - 1) / SFK_FIN_GROUPS;
    if (atomicAdd(&sync[SFK_FIN_LEAVE + grp * SFK_FIN_LINE], 1) == members - 1) {
      if (atomicAdd(&sync[SFK_FIN_TOP], 1) == ngrp - 1) {         // everybody is past the wait
        sync[SFK_FIN_DONE] = 0; sync[SFK_FIN_CLAIM] = 0; sync[SFK_FIN_TOP] = 0;
        for (int i = 0; i < SFK_FIN_GROUPS; ++i) sync[SFK_FIN_LEAVE + i * SFK_FIN_LINE] = 0;
        __threadfence();
      }
    }
  }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int c, float* scale, float* shift) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const float is = 1.f / sqrtf(rv[ch] + eps);
  const float sc = gamma[ch] * is;
  scale[ch] = sc;
  shift[ch] = beta[ch] - rm[ch] * sc;
}

// ------------------------------------------------------------------ forward apply
// RES: 0 none, 1 plain residual, 2 residual with its own scale/shift (projection shortcut's BN)
// relu_bits (optional, RELU only): one bit per output element, byte [pixel][channel group] = the VEC sign bits of the
// group -- 1/16 of the bf16 map.  The backward of act(bn(y) + shortcut) reads it instead of the activation itself.
// SUMS: the block also leaves the column sums of what it STORED as one partial row [c][2] = (sum a, 0) -- the fused block
// tail takes g = 1^T a from these (sfk_bn_tail_fwd) instead of a constant-1 channel group beside every pixel.
template <typename T, int RES, bool RELU, int NT, bool SUMS = false, bool COH = false>
__device__ __forceinline__ void bn_apply_body(FM y, FM res, FM out, int64_t pixels, int c,
                                              const float* scale, const float* shift,
                                              const float* rscale, const float* rshift, uint8_t* relu_bits,
                                              float* sums = nullptr) {
  constexpr int nt = NT;   // compile time: a run-time choice between the two access flavours is merged into plain accesses
  constexpr int VEC = DT<T>::VEC;
  const int cgs = c / VEC;
  const ChanMap cm(cgs);
  float asum[VEC], zsum[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { asum[i] = 0.f; zsum[i] = 0.f; }
  if (cm.active) {
  float sc[VEC], sh[VEC], rsc[VEC], rsh[VEC];
  load_coef<VEC, COH>(sc, scale, cm.cg);
  load_coef<VEC, COH>(sh, shift, cm.cg);
  if (RES == 2) {
    load_coef<VEC>(rsc, rscale, cm.cg);
    load_coef<VEC>(rsh, rshift, cm.cg);
  }
  const T* yp = static_cast<const T*>(y.p) + y.off + cm.cg * VEC;
  const T* rp = RES ? static_cast<const T*>(res.p) + res.off + cm.cg * VEC : nullptr;
  T* op = static_cast<T*>(out.p) + out.off + cm.cg * VEC;
  constexpr int U = SFK_BN_U;
  const int64_t step = cm.rows_b;
  for (int64_t p0 = (int64_t)blockIdx.x * cm.rows_b * U + cm.row; p0 < pixels; p0 += (int64_t)gridDim.x * cm.rows_b * U) {
    Vec16<T> v[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * step;
      const int64_t pc = p < pixels ? p : p0;          // past the end: re-read row p0 (not stored)
      if (nt & 2) {
        v[u].load_nt(yp + pc * y.ld);
        if (RES) r[u].load_nt(rp + pc * res.ld);
      } else {
        v[u].load(yp + pc * y.ld);
        if (RES) r[u].load(rp + pc * res.ld);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * step;
      if (p >= pixels) break;
      Vec16<T> o;
      uint32_t bits = 0;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float f = v[u].get(i) * sc[i] + sh[i];
        if (RES == 1) f += r[u].get(i);
        if (RES == 2) f += r[u].get(i) * rsc[i] + rsh[i];
        if (RELU) {
          bits |= (f > 0.f ? 1u : 0u) << i;
          f = f > 0.f ? f : 0.f;
        }
        o.set(i, f);
        if (SUMS) asum[i] += o.get(i);
      }
      if (nt & 1) o.store_nt(op + p * out.ld);
      else o.store(op + p * out.ld);
      if (RELU && relu_bits) relu_bits[p * cgs + cm.cg] = (uint8_t)bits;
    }
  }
  }
  if (SUMS) block_reduce_store<VEC>(cm, cgs, asum, zsum, sums, c);
}

template <typename T, int RES, bool RELU, int NT, bool SUMS = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(FM y, FM res, FM out, int64_t pixels, int c,
                                                       const float* scale, const float* shift,
                                                       const float* rscale, const float* rshift, uint8_t* relu_bits,
                                                       float* sums = nullptr) {
  bn_apply_body<T, RES, RELU, NT, SUMS>(y, res, out, pixels, c, scale, shift, rscale, rshift, relu_bits, sums);
}

// the apply with its BatchNorm's finalize as the prologue (scale / shift of `fin` are the ones the body reads)
template <typename T, int RES, bool RELU, int NT>
__global__ __launch_bounds__(256) void bn_apply_fin_kernel(FM y, FM res, FM out, int64_t pixels, int c,
                                                           const float* rscale, const float* rshift, uint8_t* relu_bits,
                                                           const FinK fin) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && fin.nbt) fin.nbt[0] += 1;
  bn_fused_prologue(fin.sync, fin.c / 2, [&](int p) { bn_finalize_pair<true>(fin, p); });
  // The coefficients were written by OTHER workgroups of this launch.  An acquire fence per workgroup invalidates the L2
  // thousands of times per launch (900 us against 40), device-coherent loads by every thread put 25 M requests on 128 addresses
  // (210 us): ONE coherent read of the c x 2 floats per workgroup, through LDS (c <= SFK_FIN_MAX_C).
  __shared__ float coefs[2 * SFK_FIN_MAX_C];
  for (int i = threadIdx.x; i < c; i += 256) {
    coefs[i] = __hip_atomic_load(fin.scale + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    coefs[SFK_FIN_MAX_C + i] = __hip_atomic_load(fin.shift + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  bn_apply_body<T, RES, RELU, NT, false>(y, res, out, pixels, c, coefs, coefs + SFK_FIN_MAX_C, rscale, rshift, relu_bits, nullptr);
  bn_fused_epilogue(fin.sync);
}

// ------------------------------------------------------------------ backward
// MASK: 0 none, 1 recompute ReLU mask from y*scale+shift, 2 mask = (mask_src > 0), 3 mask = relu_bits of bn_apply,
//       4 = relu_bits and NO y: only dz = da * mask and sum dz (the fused block tail, sfk_bn_tail_bwd)
template <typename T, int MASK, bool WRITE_DZ, int NT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(FM da, FM y, FM msrc, FM dzo, int64_t pixels, int c,
                                                            const float* mean, const float* invstd,
                                                            const float* scale, const float* shift,
                                                            float* partials, const uint8_t* relu_bits) {
  constexpr int nt = NT;
  constexpr int VEC = DT<T>::VEC;
  const int cgs = c / VEC;
  const ChanMap cm(cgs);
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  if (cm.active) {
    float mu[VEC], is[VEC], sc[VEC], sh[VEC];
    if (MASK != 4) {
      load_coef<VEC>(mu, mean, cm.cg);
      load_coef<VEC>(is, invstd, cm.cg);
    }
    if (MASK == 1) {
      load_coef<VEC>(sc, scale, cm.cg);
      load_coef<VEC>(sh, shift, cm.cg);
    }
    const T* dap = static_cast<const T*>(da.p) + da.off + cm.cg * VEC;
    const T* yp = MASK != 4 ? static_cast<const T*>(y.p) + y.off + cm.cg * VEC : nullptr;
    const T* mp = MASK == 2 ? static_cast<const T*>(msrc.p) + msrc.off + cm.cg * VEC : nullptr;
    T* zp = WRITE_DZ ? static_cast<T*>(dzo.p) + dzo.off + cm.cg * VEC : nullptr;
    constexpr int U = SFK_BN_U;                     // spans of U * rows_b consecutive rows, block b: spans b, b + G, ...
    const int64_t step = cm.rows_b;
    const int64_t pend = pixels;
    for (int64_t p0 = (int64_t)blockIdx.x * cm.rows_b * U + cm.row; p0 < pixels; p0 += (int64_t)gridDim.x * cm.rows_b * U) {
      Vec16<T> d[U], v[U], m[U];
      uint32_t bits[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t p = p0 + u * step;
        const int64_t pc = p < pend ? p : p0;          // past the end: re-read row p0 (neither summed nor stored)
        if (nt & 2) {
          d[u].load_nt(dap + pc * da.ld);
          if (MASK != 4) v[u].load_nt(yp + pc * y.ld);
        } else {
          d[u].load(dap + pc * da.ld);
          if (MASK != 4) v[u].load(yp + pc * y.ld);
        }
        if (MASK == 2) m[u].load(mp + pc * msrc.ld);
        bits[u] = 0;
        if (MASK >= 3) bits[u] = relu_bits[pc * cgs + cm.cg];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t p = p0 + u * step;
        if (p >= pend) break;
        Vec16<T> z;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float yv = MASK != 4 ? v[u].get(i) : 0.f;
          float dz = d[u].get(i);
          if (MASK == 1) dz = (yv * sc[i] + sh[i] > 0.f) ? dz : 0.f;
          if (MASK == 2) dz = (m[u].get(i) > 0.f) ? dz : 0.f;
          if (MASK >= 3) dz = ((bits[u] >> i) & 1u) ? dz : 0.f;
          if (WRITE_DZ) z.set(i, dz);
          s1[i] += dz;
          if (MASK != 4) s2[i] += dz * ((yv - mu[i]) * is[i]);
        }
        if (WRITE_DZ) {
          if (nt & 1) z.store_nt(zp + p * dzo.ld);
          else z.store(zp + p * dzo.ld);
        }
      }
    }
  }
  block_reduce_store<VEC>(cm, cgs, s1, s2, partials, c);
}

struct BFinK {
  const float* partials;
  int nparts, c;
  double count;
  const float* gamma; const float* invstd;
  float* dgamma; float* dbeta; float* coef;
  int* sync;
};

template <bool COH = false>
__device__ __forceinline__ void bn_bwd_finalize_pair(const BFinK& f, int pair) {
  const int ch = pair * 2 + (threadIdx.x & 1);
  double s1, s2;
  block_sum_partials(f.partials, f.nparts, f.c, pair * 2, s1, s2);
  if (threadIdx.x > 1) return;
  if (f.dgamma) f.dgamma[ch] += (float)s2;
  if (f.dbeta) f.dbeta[ch] += (float)s1;
  st_coef<COH>(f.coef + ch * 3 + 0, f.gamma[ch] * f.invstd[ch]);
  st_coef<COH>(f.coef + ch * 3 + 1, (float)(s1 / f.count));
  st_coef<COH>(f.coef + ch * 3 + 2, (float)(s2 / f.count));
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const BFinK f) { bn_bwd_finalize_pair(f, blockIdx.x); }

template <typename T, int MASK, int NT, bool COH = false>
__device__ __forceinline__ void bn_bwd_apply_body(FM da, FM y, FM msrc, FM dyo, int64_t pixels, int c,
                                                  const float* mean, const float* invstd,
                                                  const float* scale, const float* shift,
                                                  const float* coef) {
  constexpr int nt = NT;
  constexpr int VEC = DT<T>::VEC;
  const int cgs = c / VEC;
  const ChanMap cm(cgs);
  if (!cm.active) return;                      // (a thread, not the workgroup: the callers' barriers come before / after)
  float mu[VEC], is[VEC], sc[VEC], sh[VEC], c0[VEC], c1[VEC], c2[VEC];
  load_coef<VEC>(mu, mean, cm.cg);
  load_coef<VEC>(is, invstd, cm.cg);
  if (MASK == 1) {
    load_coef<VEC>(sc, scale, cm.cg);
    load_coef<VEC>(sh, shift, cm.cg);
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    if (COH) {        // written by other workgroups of this launch (fused finalize): device-coherent loads, see load_coef
      c0[i] = __hip_atomic_load(coef + (cm.cg * VEC + i) * 3 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      c1[i] = __hip_atomic_load(coef + (cm.cg * VEC + i) * 3 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      c2[i] = __hip_atomic_load(coef + (cm.cg * VEC + i) * 3 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      c0[i] = coef[(cm.cg * VEC + i) * 3 + 0];
      c1[i] = coef[(cm.cg * VEC + i) * 3 + 1];
      c2[i] = coef[(cm.cg * VEC + i) * 3 + 2];
    }
  }
  const T* dap = static_cast<const T*>(da.p) + da.off + cm.cg * VEC;
  const T* yp = static_cast<const T*>(y.p) + y.off + cm.cg * VEC;
  const T* mp = MASK == 2 ? static_cast<const T*>(msrc.p) + msrc.off + cm.cg * VEC : nullptr;
  T* op = static_cast<T*>(dyo.p) + dyo.off + cm.cg * VEC;
  // U pixel rows per iteration, all their loads issued before the first use (memory-level parallelism per thread)
  constexpr int U = SFK_BN_U;
  // a block owns U * rows_b CONSECUTIVE pixel rows per iteration (one contiguous span of the map)
  const int64_t step = cm.rows_b;
  for (int64_t p0 = (int64_t)blockIdx.x * cm.rows_b * U + cm.row; p0 < pixels; p0 += (int64_t)gridDim.x * cm.rows_b * U) {
    Vec16<T> d[U], v[U], m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * step;
      const int64_t pc = p < pixels ? p : p0;          // past the end: re-read row p0 (not stored)
      if (nt & 2) {
        d[u].load_nt(dap + pc * da.ld);
        v[u].load_nt(yp + pc * y.ld);
      } else {
        d[u].load(dap + pc * da.ld);
        v[u].load(yp + pc * y.ld);
      }
      if (MASK == 2) m[u].load(mp + pc * msrc.ld);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = p0 + u * step;
      if (p >= pixels) break;
      Vec16<T> o;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float yv = v[u].get(i);
        float dz = d[u].get(i);
        if (MASK == 1) dz = (yv * sc[i] + sh[i] > 0.f) ? dz : 0.f;
        if (MASK == 2) dz = (m[u].get(i) > 0.f) ? dz : 0.f;
        o.set(i, c0[i] * (dz - c1[i] - (yv - mu[i]) * is[i] * c2[i]));
      }
      if (nt & 1) o.store_nt(op + p * dyo.ld);
      else o.store(op + p * dyo.ld);
    }
  }
}

template <typename T, int MASK, int NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(FM da, FM y, FM msrc, FM dyo, int64_t pixels, int c,
                                                           const float* mean, const float* invstd,
                                                           const float* scale, const float* shift,
                                                           const float* coef) {
  bn_bwd_apply_body<T, MASK, NT>(da, y, msrc, dyo, pixels, c, mean, invstd, scale, shift, coef);
}

template <typename T, int MASK, int NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_fin_kernel(FM da, FM y, FM msrc, FM dyo, int64_t pixels, int c,
                                                               const float* mean, const float* scale, const float* shift,
                                                               const BFinK fin) {
  bn_fused_prologue(fin.sync, fin.c / 2, [&](int p) { bn_bwd_finalize_pair<true>(fin, p); });
  __shared__ float coefs[3 * SFK_FIN_MAX_C];            // one coherent read of the c x 3 coefficients per workgroup (see bn_apply_fin_kernel)
  for (int i = threadIdx.x; i < 3 * c; i += 256) coefs[i] = __hip_atomic_load(fin.coef + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  bn_bwd_apply_body<T, MASK, NT>(da, y, msrc, dyo, pixels, c, mean, fin.invstd, scale, shift, coefs);
  bn_fused_epilogue(fin.sync);
}

// ------------------------------------------------------------------ stem tail backward: MaxPool (1,3,3)/(1,2,2)/(0,1,1) + ReLU + BN
// The gradient of the activation a = relu(bn(y)) under the pool is never stored: both passes rebuild
//   da[p] = sum over the <= 2 x 2 windows that contain p of [argmax(window) == p] * d_out[window]     (rounded to T, as the
//   stand-alone sfk_maxpool_bwd would store it), dz = da * [y * scale + shift > 0]
// from d_out (1/4 of the map) and the argmax bytes; APPLY = false leaves the partial rows (sum dz, sum dz * x_hat),
// APPLY = true writes dy = c0 * (dz - c1 - x_hat * c2).  Replaces maxpool_bwd + bn_bwd_reduce + bn_bwd_apply:
// per input element 2 x (y read) + 1 write instead of 4 reads + 3 writes + the da map.
// A thread owns a 2 x 2 QUAD of input pixels (rows 2a, 2a+1; columns 2b, 2b+1): the quad meets exactly the four windows
// (a, b), (a, b+1), (a+1, b), (a+1, b+1) -- 4 gradient + 4 argmax loads for 4 pixels (per-pixel gathers would load 16).
struct PoolGeo {
  int H, W, Ho, Wo, qh, qw;     // input frame, pooled frame, quads per frame column / row
  FastDiv dqw, dqh;
};

template <typename T, bool APPLY>
__global__ __launch_bounds__(256, APPLY ? 3 : 4) void bn_pool_bwd_kernel(FM dout, const uint8_t* __restrict__ argmax, FM y, FM dyo,
                                                          int64_t quads, int c, PoolGeo g, const float* mean,
                                                          const float* invstd, const float* scale, const float* shift,
                                                          const float* coef, float* partials) {
  constexpr int VEC = DT<T>::VEC;
  const int cgs = c / VEC;
  const ChanMap cm(cgs);
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  if (cm.active) {
    float mu[VEC], is[VEC], sc[VEC], sh[VEC], c0[VEC], c1[VEC], c2[VEC];
    load_coef<VEC>(mu, mean, cm.cg);
    load_coef<VEC>(is, invstd, cm.cg);
    load_coef<VEC>(sc, scale, cm.cg);
    load_coef<VEC>(sh, shift, cm.cg);
    if (APPLY) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        c0[i] = coef[(cm.cg * VEC + i) * 3 + 0];
        c1[i] = coef[(cm.cg * VEC + i) * 3 + 1];
        c2[i] = coef[(cm.cg * VEC + i) * 3 + 2];
      }
    }
    const T* gp = static_cast<const T*>(dout.p) + dout.off + cm.cg * VEC;
    const uint8_t* ap = argmax + cm.cg * VEC;
    const T* yp = static_cast<const T*>(y.p) + y.off + cm.cg * VEC;
    T* op = APPLY ? static_cast<T*>(dyo.p) + dyo.off + cm.cg * VEC : nullptr;
    for (int64_t q = (int64_t)blockIdx.x * cm.rows_b + cm.row; q < quads; q += (int64_t)gridDim.x * cm.rows_b) {
      uint32_t q1, b, nt, a;
      g.dqw.divmod((uint32_t)q, q1, b);
      g.dqh.divmod(q1, nt, a);
      // windows (a + wa, b + wb), wa, wb in {0, 1}; pixels (2a + ra, 2b + rb)
      Vec16<T> gw[4], v[4];
      uint32_t alo[4], ahi[4];
      bool wok[4], pok[4];
#pragma unroll
      for (int wa = 0; wa < 2; ++wa)
#pragma unroll
        for (int wb = 0; wb < 2; ++wb) {
          const int ho = (int)a + wa, wo = (int)b + wb, k = 2 * wa + wb;
          wok[k] = ho < g.Ho && wo < g.Wo;
          const int hc = ho < g.Ho ? ho : g.Ho - 1, wc = wo < g.Wo ? wo : g.Wo - 1;
          const int64_t opix = ((int64_t)nt * g.Ho + hc) * g.Wo + wc;
          gw[k].load(gp + opix * dout.ld);
          const uint8_t* aq = ap + opix * c;
          if (VEC == 8) {
            const uint2 a2 = *reinterpret_cast<const uint2*>(aq);
            alo[k] = a2.x;
            ahi[k] = a2.y;
          } else {
            alo[k] = *reinterpret_cast<const uint32_t*>(aq);
            ahi[k] = 0;
          }
        }
      int64_t ipix[4];
#pragma unroll
      for (int ra = 0; ra < 2; ++ra)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int hi = 2 * (int)a + ra, wi = 2 * (int)b + rb, k = 2 * ra + rb;
          pok[k] = hi < g.H && wi < g.W;
          const int hc = hi < g.H ? hi : g.H - 1, wc = wi < g.W ? wi : g.W - 1;
          ipix[k] = ((int64_t)nt * g.H + hc) * g.W + wc;
          v[k].load(yp + ipix[k] * y.ld);
        }
#pragma unroll
      for (int ra = 0; ra < 2; ++ra)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const int k = 2 * ra + rb;
          if (!pok[k]) continue;
          Vec16<T> da, o;
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            float acc = 0.f;
            // pixel row 2a + ra lies in window row a + wa at kh = 1 + ra - 2 wa (ra = 0: only wa = 0); same for columns
#pragma unroll
            for (int wa = 0; wa <= ra; ++wa)
#pragma unroll
              for (int wb = 0; wb <= rb; ++wb) {
                const int w4 = 2 * wa + wb;
                const uint32_t code = (uint32_t)((1 + ra - 2 * wa) * 3 + (1 + rb - 2 * wb));
                const uint32_t am = ((i < 4 ? alo[w4] : ahi[w4]) >> (8 * (i & 3))) & 0xFFu;
                if (wok[w4] && am == code) acc += gw[w4].get(i);
              }
            da.set(i, acc);
          }
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float yv = v[k].get(i);
            const float dz = (yv * sc[i] + sh[i] > 0.f) ? da.get(i) : 0.f;
            const float xh = (yv - mu[i]) * is[i];
            if (APPLY) {
              o.set(i, c0[i] * (dz - c1[i] - xh * c2[i]));
            } else {
              s1[i] += dz;
              s2[i] += dz * xh;
            }
          }
          if (APPLY) o.store(op + ipix[k] * dyo.ld);
        }
    }
  }
  if (!APPLY) block_reduce_store<VEC>(cm, cgs, s1, s2, partials, c);
}

bool same_shape(const sfk_fmap* a, const sfk_fmap* b) {
  return a->n == b->n && a->t == b->t && a->h == b->h && a->w == b->w && a->c == b->c && a->dtype == b->dtype;
}

}  // namespace

extern "C" int sfk_bn_stats(const sfk_fmap* y, float* partials, int32_t max_parts, int32_t* nparts_out,
                            sfk_stream_t stream) {
  if (!sfk_fmap_ok(y) || !partials || !nparts_out || max_parts <= 0) return SFK_ERR_INVALID;
  if (!sfk_fmap_vec_ok(y)) return SFK_ERR_UNSUPPORTED;
  const int64_t px = sfk_fmap_pixels(y);
  int np;
  const dim3 grid = chan_grid(y->c / sfk_vec_of(y->dtype), px, max_parts, &np);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (y->dtype == SFK_BF16) hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, grid, dim3(256), 0, s, fm_of(y), px, y->c, partials);
  else hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(256), 0, s, fm_of(y), px, y->c, partials);
  SFK_CHECK_LAUNCH();
  *nparts_out = np;
  return SFK_OK;
}

extern "C" int sfk_bn_finalize(const float* partials, int32_t nparts, int32_t c, int64_t count, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                               float* scale, float* shift, float* workspace, sfk_stream_t stream) {
  if (!partials || nparts <= 0 || c <= 0 || count <= 0 || !gamma || !beta || !mean || !invstd || !scale || !shift)
    return SFK_ERR_INVALID;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SFK_ERR_INVALID;
  if (c & 1) return SFK_ERR_UNSUPPORTED;           // a block folds a channel PAIR (16-byte loads); maps have c % 4 == 0
  if ((((uintptr_t)partials) | ((uintptr_t)workspace)) & 15) return SFK_ERR_INVALID;   // rows are read as float4
  hipStream_t s = static_cast<hipStream_t>(stream);
  partials = fold_partials(partials, nparts, c, workspace, &nparts, s);
  const FinK f{partials, nparts, c, (double)count, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
               mean, invstd, scale, shift, nullptr};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(c / 2), dim3(256), 0, s, f);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int32_t c, float* scale, float* shift,
                                  sfk_stream_t stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || c <= 0) return SFK_ERR_INVALID;
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((c + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), gamma,
                     beta, running_mean, running_var, eps, c, scale, shift);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

namespace {
template <typename T>
int launch_apply(const sfk_fmap* y, const float* scale, const float* shift, const sfk_fmap* res,
                 const float* rs, const float* rb, int relu, const sfk_fmap* out, uint8_t* bits, float* sums, int max_parts,
                 int* nparts_out, hipStream_t s) {
  const int64_t px = sfk_fmap_pixels(y);
  int np;
  dim3 grid = chan_grid(y->c / DT<T>::VEC, px, sums ? max_parts : 0, &np);
  const dim3 blk(256);
  const FM fy = fm_of(y), fr = fm_of(res), fo = fm_of(out);
  const int mode = !res ? 0 : (rs ? 2 : 1);
  const int nt = nt_hint(y, sfk_tune().nt_apply_mb, 3);
  if (sums) {       // one partial row per block: the capped grid of the reductions (grid-stride loop over the spans)
    if (mode != 0 || !relu) return SFK_ERR_UNSUPPORTED;
    if (nt) hipLaunchKernelGGL((bn_apply_kernel<T, 0, true, 3, true>), grid, blk, 0, s, fy, fr, fo, px, y->c, scale, shift, rs, rb, bits, sums);
    else hipLaunchKernelGGL((bn_apply_kernel<T, 0, true, 0, true>), grid, blk, 0, s, fy, fr, fo, px, y->c, scale, shift, rs, rb, bits, sums);
    SFK_CHECK_LAUNCH();
    *nparts_out = np;
    return SFK_OK;
  }
  grid.x = span_blocks(y->c / DT<T>::VEC, px);      // one span per thread: no grid-stride loop
#define SFK_APPLY(R, A)                                                                                                      \
  do {                                                                                                                     \
    if (nt) hipLaunchKernelGGL((bn_apply_kernel<T, R, A, 3>), grid, blk, 0, s, fy, fr, fo, px, y->c, scale, shift, rs, rb, bits); \
    else hipLaunchKernelGGL((bn_apply_kernel<T, R, A, 0>), grid, blk, 0, s, fy, fr, fo, px, y->c, scale, shift, rs, rb, bits);    \
  } while (0)
  if (relu) {
    if (mode == 0) SFK_APPLY(0, true); else if (mode == 1) SFK_APPLY(1, true); else SFK_APPLY(2, true);
  } else {
    if (mode == 0) SFK_APPLY(0, false); else if (mode == 1) SFK_APPLY(1, false); else SFK_APPLY(2, false);
  }
#undef SFK_APPLY
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}
}  // namespace

extern "C" int sfk_bn_apply(const sfk_fmap* y, const float* scale, const float* shift, const sfk_fmap* res,
                            const float* res_scale, const float* res_shift, int32_t relu, const sfk_fmap* out,
                            uint8_t* relu_bits, float* out_sums, int32_t max_parts, int32_t* nparts_out,
                            sfk_stream_t stream) {
  if (relu_bits && !relu) return SFK_ERR_INVALID;
  if (out_sums && (max_parts <= 0 || !nparts_out)) return SFK_ERR_INVALID;
  if (!sfk_fmap_ok(y) || !sfk_fmap_ok(out) || !scale || !shift || !same_shape(y, out)) return SFK_ERR_INVALID;
  if (res && (!sfk_fmap_ok(res) || !same_shape(y, res))) return SFK_ERR_INVALID;
  if ((res_scale == nullptr) != (res_shift == nullptr) || (res_scale && !res)) return SFK_ERR_INVALID;
  if (!sfk_fmap_vec_ok(y) || !sfk_fmap_vec_ok(out) || (res && !sfk_fmap_vec_ok(res))) return SFK_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return y->dtype == SFK_BF16
             ? launch_apply<bf16_t>(y, scale, shift, res, res_scale, res_shift, relu, out, relu_bits, out_sums, max_parts, nparts_out, s)
             : launch_apply<float>(y, scale, shift, res, res_scale, res_shift, relu, out, relu_bits, out_sums, max_parts, nparts_out, s);
}

namespace {
template <typename T>
int launch_apply_fin(const sfk_fmap* y, const sfk_fmap* res, const float* rs, const float* rb, int relu, const sfk_fmap* out,
                     uint8_t* bits, const FinK& fin, hipStream_t s) {
  const int64_t px = sfk_fmap_pixels(y);
  int np;
  dim3 grid = chan_grid(y->c / DT<T>::VEC, px, 0, &np);
  const dim3 blk(256);
  grid.x = span_blocks(y->c / DT<T>::VEC, px);
  const FM fy = fm_of(y), fr = fm_of(res), fo = fm_of(out);
  const int mode = !res ? 0 : (rs ? 2 : 1);
  const int nt = nt_hint(y, sfk_tune().nt_apply_mb, 3);
#define SFK_APPLY_FIN(R, A)                                                                                              \
  do {                                                                                                                   \
    if (nt) hipLaunchKernelGGL((bn_apply_fin_kernel<T, R, A, 3>), grid, blk, 0, s, fy, fr, fo, px, y->c, rs, rb, bits, fin); \
    else hipLaunchKernelGGL((bn_apply_fin_kernel<T, R, A, 0>), grid, blk, 0, s, fy, fr, fo, px, y->c, rs, rb, bits, fin);    \
  } while (0)
  if (relu) {
    if (mode == 0) SFK_APPLY_FIN(0, true); else if (mode == 1) SFK_APPLY_FIN(1, true); else SFK_APPLY_FIN(2, true);
  } else {
    if (mode == 0) SFK_APPLY_FIN(0, false); else if (mode == 1) SFK_APPLY_FIN(1, false); else SFK_APPLY_FIN(2, false);
  }
#undef SFK_APPLY_FIN
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}
}  // namespace

extern "C" int sfk_bn_finalize_apply(const float* partials, int32_t nparts, int64_t count, const float* gamma, const float* beta,
                                     float eps, float momentum, float* running_mean, float* running_var,
                                     int64_t* num_batches_tracked, float* mean, float* invstd, float* workspace, int32_t* sync,
                                     const sfk_fmap* y, float* scale, float* shift, const sfk_fmap* res, const float* res_scale,
                                     const float* res_shift, int32_t relu, const sfk_fmap* out, uint8_t* relu_bits,
                                     sfk_stream_t stream) {
  if (!sfk_fmap_ok(y) || !partials || nparts <= 0 || count <= 0 || !gamma || !beta || !mean || !invstd || !scale || !shift || !sync)
    return SFK_ERR_INVALID;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SFK_ERR_INVALID;
  if (relu_bits && !relu) return SFK_ERR_INVALID;
  if (!sfk_fmap_ok(out) || !same_shape(y, out)) return SFK_ERR_INVALID;
  if (res && (!sfk_fmap_ok(res) || !same_shape(y, res))) return SFK_ERR_INVALID;
  if ((res_scale == nullptr) != (res_shift == nullptr) || (res_scale && !res)) return SFK_ERR_INVALID;
  if (!sfk_fmap_vec_ok(y) || !sfk_fmap_vec_ok(out) || (res && !sfk_fmap_vec_ok(res)) || (y->c & 1) || y->c > SFK_FIN_MAX_C) return SFK_ERR_UNSUPPORTED;
  if (((((uintptr_t)partials) | ((uintptr_t)workspace)) & 15) || (((uintptr_t)sync) & 3)) return SFK_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rows = nparts;
  partials = fold_partials(partials, nparts, y->c, workspace, &rows, s);
  const FinK fin{partials, rows, y->c, (double)count, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked,
                 mean, invstd, scale, shift, sync};
  return y->dtype == SFK_BF16 ? launch_apply_fin<bf16_t>(y, res, res_scale, res_shift, relu, out, relu_bits, fin, s)
                              : launch_apply_fin<float>(y, res, res_scale, res_shift, relu, out, relu_bits, fin, s);
}

namespace {
int check_bwd(const sfk_fmap* da, const sfk_fmap* y, const sfk_fmap* mask_src, const float* mean,
              const float* invstd, const float* scale, const float* shift, int relu) {
  if (!sfk_fmap_ok(da) || !sfk_fmap_ok(y) || !same_shape(da, y) || !mean || !invstd) return SFK_ERR_INVALID;
  if (mask_src && (!sfk_fmap_ok(mask_src) || !same_shape(da, mask_src))) return SFK_ERR_INVALID;
  if (!mask_src && relu && (!scale || !shift)) return SFK_ERR_INVALID;
  if (!sfk_fmap_vec_ok(da) || !sfk_fmap_vec_ok(y) || (mask_src && !sfk_fmap_vec_ok(mask_src))) return SFK_ERR_UNSUPPORTED;
  return SFK_OK;
}

template <typename T>
int launch_bwd_reduce(const sfk_fmap* da, const sfk_fmap* y, const sfk_fmap* ms, const float* mean,
                      const float* invstd, const float* scale, const float* shift, int relu,
                      const sfk_fmap* dzo, float* partials, int max_parts, int* nparts_out, const uint8_t* bits,
                      hipStream_t s) {
  const int64_t px = sfk_fmap_pixels(da);
  int np;
  const dim3 grid = chan_grid(da->c / DT<T>::VEC, px, max_parts, &np), blk(256);
  const FM a = fm_of(da), b = fm_of(y), m = fm_of(ms), z = fm_of(dzo);
  const int mask = bits ? (y ? 3 : 4) : (ms ? 2 : (relu ? 1 : 0));
  const int nt = nt_hint(da, sfk_tune().nt_reduce_mb, 2);
#define SFK_RED(M, W)                                                                                                        \
  do {                                                                                                                     \
    if (nt) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, M, W, 2>), grid, blk, 0, s, a, b, m, z, px, da->c, mean, invstd, scale, shift, partials, bits); \
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, M, W, 0>), grid, blk, 0, s, a, b, m, z, px, da->c, mean, invstd, scale, shift, partials, bits);    \
  } while (0)
  if (mask == 4) {
    if (dzo) SFK_RED(4, true); else SFK_RED(4, false);
  } else if (dzo) {
    if (mask == 0) SFK_RED(0, true); else if (mask == 1) SFK_RED(1, true); else if (mask == 2) SFK_RED(2, true); else SFK_RED(3, true);
  } else {
    if (mask == 0) SFK_RED(0, false); else if (mask == 1) SFK_RED(1, false); else if (mask == 2) SFK_RED(2, false); else SFK_RED(3, false);
  }
#undef SFK_RED
  SFK_CHECK_LAUNCH();
  *nparts_out = np;
  return SFK_OK;
}

template <typename T>
int launch_bwd_apply(const sfk_fmap* da, const sfk_fmap* y, const sfk_fmap* ms, const float* mean,
                     const float* invstd, const float* scale, const float* shift, int relu, const float* coef,
                     const sfk_fmap* dy, hipStream_t s) {
  const int64_t px = sfk_fmap_pixels(y);
  int np;
  dim3 grid = chan_grid(y->c / DT<T>::VEC, px, 0, &np);
  const dim3 blk(256);
  grid.x = span_blocks(y->c / DT<T>::VEC, px);      // one span per thread: no grid-stride loop
  const FM a = fm_of(da), b = fm_of(y), m = fm_of(ms), o = fm_of(dy);
  const int mask = ms ? 2 : (relu ? 1 : 0);
  const int nt = nt_hint(y, sfk_tune().nt_bwd_apply_mb, 3);
#define SFK_APP(M)                                                                                                           \
  do {                                                                                                                     \
    if (nt) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, M, 3>), grid, blk, 0, s, a, b, m, o, px, y->c, mean, invstd, scale, shift, coef); \
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, M, 0>), grid, blk, 0, s, a, b, m, o, px, y->c, mean, invstd, scale, shift, coef);    \
  } while (0)
  if (mask == 0) SFK_APP(0); else if (mask == 1) SFK_APP(1); else SFK_APP(2);
#undef SFK_APP
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}
}  // namespace

extern "C" int sfk_bn_bwd_reduce(const sfk_fmap* da, const sfk_fmap* y, const sfk_fmap* mask_src, const float* mean,
                                 const float* invstd, const float* scale, const float* shift, int32_t relu,
                                 const sfk_fmap* dz_out, float* partials, int32_t max_parts, int32_t* nparts_out,
                                 const uint8_t* relu_bits, sfk_stream_t stream) {
  if (relu_bits && mask_src) return SFK_ERR_INVALID;
  if (!y) {        // mask + sum dz only (the fused block tail): needs the bitmap
    if (!relu_bits || !sfk_fmap_ok(da)) return SFK_ERR_INVALID;
    if (!sfk_fmap_vec_ok(da)) return SFK_ERR_UNSUPPORTED;
  } else {
    const int st = check_bwd(da, y, mask_src, mean, invstd, (relu_bits ? nullptr : scale), (relu_bits ? nullptr : shift),
                             relu_bits ? 0 : relu);
    if (st != SFK_OK) return st;
  }
  if (!partials || max_parts <= 0 || !nparts_out) return SFK_ERR_INVALID;
  if (dz_out && (!sfk_fmap_ok(dz_out) || !same_shape(da, dz_out))) return SFK_ERR_INVALID;
  if (dz_out && !sfk_fmap_vec_ok(dz_out)) return SFK_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return da->dtype == SFK_BF16
             ? launch_bwd_reduce<bf16_t>(da, y, mask_src, mean, invstd, scale, shift, relu, dz_out, partials, max_parts, nparts_out, relu_bits, s)
             : launch_bwd_reduce<float>(da, y, mask_src, mean, invstd, scale, shift, relu, dz_out, partials, max_parts, nparts_out, relu_bits, s);
}

extern "C" int sfk_bn_bwd_finalize(const float* partials, int32_t nparts, int32_t c, int64_t count, const float* gamma,
                                   const float* invstd, float* dgamma, float* dbeta, float* coef,
                                   float* workspace, sfk_stream_t stream) {
  if (!partials || nparts <= 0 || c <= 0 || count <= 0 || !gamma || !invstd || !coef) return SFK_ERR_INVALID;
  if (c & 1) return SFK_ERR_UNSUPPORTED;
  if ((((uintptr_t)partials) | ((uintptr_t)workspace)) & 15) return SFK_ERR_INVALID;   // rows are read as float4
  hipStream_t s = static_cast<hipStream_t>(stream);
  partials = fold_partials(partials, nparts, c, workspace, &nparts, s);
  const BFinK f{partials, nparts, c, (double)count, gamma, invstd, dgamma, dbeta, coef, nullptr};
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c / 2), dim3(256), 0, s, f);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_bn_bwd_apply(const sfk_fmap* da, const sfk_fmap* y, const sfk_fmap* mask_src, const float* mean,
                                const float* invstd, const float* scale, const float* shift, int32_t relu,
                                const float* coef, const sfk_fmap* dy, sfk_stream_t stream) {
  const int st = check_bwd(da, y, mask_src, mean, invstd, scale, shift, relu);
  if (st != SFK_OK) return st;
  if (!coef || !sfk_fmap_ok(dy) || !same_shape(da, dy)) return SFK_ERR_INVALID;
  if (!sfk_fmap_vec_ok(dy)) return SFK_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return y->dtype == SFK_BF16
             ? launch_bwd_apply<bf16_t>(da, y, mask_src, mean, invstd, scale, shift, relu, coef, dy, s)
             : launch_bwd_apply<float>(da, y, mask_src, mean, invstd, scale, shift, relu, coef, dy, s);
}

namespace {
template <typename T>
int launch_bwd_apply_fin(const sfk_fmap* da, const sfk_fmap* y, const sfk_fmap* ms, const float* mean, const float* scale,
                         const float* shift, int relu, const sfk_fmap* dy, const BFinK& fin, hipStream_t s) {
  const int64_t px = sfk_fmap_pixels(y);
  int np;
  dim3 grid = chan_grid(y->c / DT<T>::VEC, px, 0, &np);
  const dim3 blk(256);
  grid.x = span_blocks(y->c / DT<T>::VEC, px);
  const FM a = fm_of(da), b = fm_of(y), m = fm_of(ms), o = fm_of(dy);
  const int mask = ms ? 2 : (relu ? 1 : 0);
  const int nt = nt_hint(y, sfk_tune().nt_bwd_apply_mb, 3);
#define SFK_APP_FIN(M)                                                                                                     \
  do {                                                                                                                   \
    if (nt) hipLaunchKernelGGL((bn_bwd_apply_fin_kernel<T, M, 3>), grid, blk, 0, s, a, b, m, o, px, y->c, mean, scale, shift, fin); \
    else hipLaunchKernelGGL((bn_bwd_apply_fin_kernel<T, M, 0>), grid, blk, 0, s, a, b, m, o, px, y->c, mean, scale, shift, fin);    \
  } while (0)
  if (mask == 0) SFK_APP_FIN(0); else if (mask == 1) SFK_APP_FIN(1); else SFK_APP_FIN(2);
#undef SFK_APP_FIN
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}
}  // namespace

extern "C" int sfk_bn_bwd_finalize_apply(const float* partials, int32_t nparts, int64_t count, const float* gamma, float* dgamma,
                                         float* dbeta, float* coef, float* workspace, int32_t* sync, const sfk_fmap* da,
                                         const sfk_fmap* y, const sfk_fmap* mask_src, const float* mean, const float* invstd,
                                         const float* scale, const float* shift, int32_t relu, const sfk_fmap* dy,
                                         sfk_stream_t stream) {
  const int st = check_bwd(da, y, mask_src, mean, invstd, scale, shift, relu);
  if (st != SFK_OK) return st;
  if (!partials || nparts <= 0 || count <= 0 || !gamma || !coef || !sync || !sfk_fmap_ok(dy) || !same_shape(da, dy)) return SFK_ERR_INVALID;
  if (!sfk_fmap_vec_ok(dy) || (y->c & 1) || y->c > SFK_FIN_MAX_C) return SFK_ERR_UNSUPPORTED;
  if (((((uintptr_t)partials) | ((uintptr_t)workspace)) & 15) || (((uintptr_t)sync) & 3)) return SFK_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rows = nparts;
  partials = fold_partials(partials, nparts, y->c, workspace, &rows, s);
  const BFinK fin{partials, rows, y->c, (double)count, gamma, invstd, dgamma, dbeta, coef, sync};
  return y->dtype == SFK_BF16 ? launch_bwd_apply_fin<bf16_t>(da, y, mask_src, mean, scale, shift, relu, dy, fin, s)
                              : launch_bwd_apply_fin<float>(da, y, mask_src, mean, scale, shift, relu, dy, fin, s);
}

// ---- stem tail backward (see bn_pool_bwd_kernel): only the stems' (3, 2, 1) pool
namespace {
int bn_pool_check(const sfk_fmap* d_out, const uint8_t* argmax, const sfk_fmap* y, const float* mean, const float* invstd,
                  const float* scale, const float* shift) {
  if (!sfk_fmap_ok(d_out) || !sfk_fmap_ok(y) || !argmax || !mean || !invstd || !scale || !shift) return SFK_ERR_INVALID;
  if (d_out->dtype != y->dtype || d_out->n != y->n || d_out->t != y->t || d_out->c != y->c) return SFK_ERR_INVALID;
  if (d_out->h != (y->h + 2 - 3) / 2 + 1 || d_out->w != (y->w + 2 - 3) / 2 + 1) return SFK_ERR_INVALID;
  if (!sfk_fmap_vec_ok(d_out) || !sfk_fmap_vec_ok(y)) return SFK_ERR_UNSUPPORTED;
  if (sfk_fmap_pixels(y) >= (1ll << 31)) return SFK_ERR_UNSUPPORTED;
  return SFK_OK;
}
inline PoolGeo pool_geo(const sfk_fmap* d_out, const sfk_fmap* y) {
  PoolGeo g;
  g.H = y->h; g.W = y->w; g.Ho = d_out->h; g.Wo = d_out->w;
  g.qh = (y->h + 1) / 2; g.qw = (y->w + 1) / 2;
  g.dqw.set(g.qw); g.dqh.set(g.qh);
  return g;
}
inline int64_t pool_quads(const sfk_fmap* y, const PoolGeo& g) { return (int64_t)y->n * y->t * g.qh * g.qw; }
}  // namespace

extern "C" int sfk_bn_maxpool_bwd_reduce(const sfk_fmap* d_out, const uint8_t* argmax, const sfk_fmap* y, const float* mean,
                                         const float* invstd, const float* scale, const float* shift, float* partials,
                                         int32_t max_parts, int32_t* nparts_out, sfk_stream_t stream) {
  const int st = bn_pool_check(d_out, argmax, y, mean, invstd, scale, shift);
  if (st != SFK_OK) return st;
  if (!partials || !nparts_out || max_parts <= 0) return SFK_ERR_INVALID;
  const int64_t px = sfk_fmap_pixels(y);
  int np;
  const dim3 grid = chan_grid(y->c / sfk_vec_of(y->dtype), px, max_parts, &np);   // rows as sfk_bn_bwd_reduce would leave
  hipStream_t s = static_cast<hipStream_t>(stream);
  const PoolGeo g = pool_geo(d_out, y);
  const int64_t quads = pool_quads(y, g);
  if (y->dtype == SFK_BF16)
    hipLaunchKernelGGL((bn_pool_bwd_kernel<bf16_t, false>), grid, dim3(256), 0, s, fm_of(d_out), argmax, fm_of(y), FM{nullptr, 0, 0},
                       quads, y->c, g, mean, invstd, scale, shift, nullptr, partials);
  else
    hipLaunchKernelGGL((bn_pool_bwd_kernel<float, false>), grid, dim3(256), 0, s, fm_of(d_out), argmax, fm_of(y), FM{nullptr, 0, 0},
                       quads, y->c, g, mean, invstd, scale, shift, nullptr, partials);
  SFK_CHECK_LAUNCH();
  *nparts_out = np;
  return SFK_OK;
}

extern "C" int sfk_bn_maxpool_bwd_apply(const sfk_fmap* d_out, const uint8_t* argmax, const sfk_fmap* y, const float* mean,
                                        const float* invstd, const float* scale, const float* shift, const float* coef,
                                        const sfk_fmap* dy, sfk_stream_t stream) {
  const int st = bn_pool_check(d_out, argmax, y, mean, invstd, scale, shift);
  if (st != SFK_OK) return st;
  if (!coef || !sfk_fmap_ok(dy) || !same_shape(y, dy)) return SFK_ERR_INVALID;
  if (!sfk_fmap_vec_ok(dy)) return SFK_ERR_UNSUPPORTED;
  const int cgs = y->c / sfk_vec_of(y->dtype);
  const int cgs_b = cgs < 256 ? cgs : 256, rows_b = 256 / cgs_b;
  const PoolGeo g = pool_geo(d_out, y);
  const int64_t quads = pool_quads(y, g);
  int64_t want = (quads + rows_b - 1) / rows_b;                               // one quad per thread
  if (want > 65535 * 16) want = 65535 * 16;
  const dim3 grid((unsigned)(want > 0 ? want : 1), (unsigned)((cgs + 255) / 256));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (y->dtype == SFK_BF16)
    hipLaunchKernelGGL((bn_pool_bwd_kernel<bf16_t, true>), grid, dim3(256), 0, s, fm_of(d_out), argmax, fm_of(y), fm_of(dy), quads,
                       y->c, g, mean, invstd, scale, shift, coef, nullptr);
  else
    hipLaunchKernelGGL((bn_pool_bwd_kernel<float, true>), grid, dim3(256), 0, s, fm_of(d_out), argmax, fm_of(y), fm_of(dy), quads,
                       y->c, g, mean, invstd, scale, shift, coef, nullptr);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}
