// The bottleneck tail  a -> conv_c (1x1x1) -> norm_c [-> + shortcut -> ReLU]  without the conv output in HBM
// (include/sfk.h, sfk_bn_tail_fwd / sfk_bn_tail_bwd): the small per-layer algebra between the big kernels.
//
// What the big kernels deliver:
//     gram [c][c] fp32     : G = a^T a  (an ordinary sfk_conv_wgrad call with x = dy = a)
//     a_sums [rows][c][2]  : partial column sums of a, left by the sfk_bn_apply pass that WROTE a; folded here into g = 1^T a
//     r    [cout][c] fp32  : R = dz^T a;  s = sum dz arrives as the partial rows of the kernel that wrote dz
// What this file computes from them -- everything is O(cout * c^2) or less, i.e. independent of the pixel count:
//     forward : T = W G, batch mean / variance of y = a W^T per output channel, running statistics, scale / shift, and
//               wd = (diag(A) W)^T, A = gamma * invstd: the filter of the backward's FIRST data-gradient pass dz (A W) is
//               known once the statistics are, so that pass can run beside R = dz^T a instead of behind it
//     backward: dgamma, dbeta, dW = diag(A) R + diag(B) T + C (x) g, and the operands of the second data-gradient pass
//               m = W^T diag(B) W (c x c, in the filter's precision), bias = C W
// Sums over channels / rows run in double and in a fixed order: deterministic, no atomics.
#include "sfk_common.h"

namespace {

template <typename D> __device__ __forceinline__ float wload(const void* w, int64_t i) {
  return (float)static_cast<const D*>(w)[i];
}

constexpr int TF_CO = 8;      // output channels per block of the forward kernel
constexpr int TF_MAXC = 512;  // widest conv_c input of the ResNet-50/101/152 SlowFast family (2048 / 4)

// g[ci] = sum over the partial rows of a_sums[row][ci][0]: one wave per channel, double, fixed order
__global__ __launch_bounds__(256) void bn_tail_fold_g_kernel(const float* __restrict__ parts, int nparts, int c, float* __restrict__ g) {
  const int ci = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ci >= c) return;
  double s = 0.0;
  for (int p = lane; p < nparts; p += 64) s += (double)parts[((int64_t)p * c + ci) * 2];
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) s += __shfl_xor(s, sft);
  if (lane == 0) g[ci] = (float)s;
}

// T[co][:] = W[co][:] G,  mean = W[co] . g / n,  var = W[co] Gc W[co]^T (Gc = the centred Gram matrix)  for TF_CO channels per
// block; thread <-> column ci
template <typename D>
__global__ __launch_bounds__(256) void bn_tail_fwd_kernel(const float* __restrict__ gram, const float* __restrict__ gvec,
                                                          double n, int c, const void* w,
                                                          int cout, const float* gamma, const float* beta, float eps,
                                                          float momentum, float* running_mean, float* running_var,
                                                          int64_t* nbt, float* mean, float* invstd, float* scale,
                                                          float* shift, float* __restrict__ t, D* __restrict__ wd) {
  __shared__ float wl[TF_CO][TF_MAXC];
  __shared__ float gl[TF_MAXC];        // column means g / n
  __shared__ double red[4 * TF_CO][2];
  __shared__ float acoef[TF_CO];
  const int co0 = blockIdx.x * TF_CO, tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0 && nbt) nbt[0] += 1;
  const double inv_n = 1.0 / n;
  for (int i = tid; i < TF_CO * c; i += 256) {
    const int o = i / c, j = i % c;
    wl[o][j] = co0 + o < cout ? wload<D>(w, (int64_t)(co0 + o) * c + j) : 0.f;
  }
  for (int j = tid; j < c; j += 256) gl[j] = (float)((double)gvec[j] * inv_n);
  __syncthreads();
  constexpr int NR = TF_MAXC / 256;
  // acc = T = W G (the backward's operand).  The VARIANCE is not taken from it: E[y^2] - E[y]^2 cancels mean^2 / var digits of an
  // fp32 T (1e-4 relative at mean^2 / var = 1e4; a near-constant channel clamps to var = 0, invstd = 1 / sqrt(eps)).  accc sums the
  // CENTRED matrix  Gc = G / n - (g / n)(g / n)^T  -- each entry formed in double, so the products are covariances, small
  // against nothing -- and var = W Gc W^T needs no subtraction.
  float acc[TF_CO][NR], accc[TF_CO][NR];
  double gbar[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int ci = tid + 256 * r;
    gbar[r] = ci < c ? (double)gvec[ci] * inv_n : 0.0;
  }
#pragma unroll
  for (int o = 0; o < TF_CO; ++o)
#pragma unroll
    for (int r = 0; r < NR; ++r) { acc[o][r] = 0.f; accc[o][r] = 0.f; }
#pragma unroll 8                       // 8 rows of G in flight: the loop is bound by the latency of these loads
  for (int j = 0; j < c; ++j) {
    float gj[NR], gc[NR];
    const double mj = (double)gl[j];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int ci = tid + 256 * r;
      gj[r] = ci < c ? gram[(int64_t)j * c + ci] : 0.f;
      gc[r] = (float)((double)gj[r] * inv_n - mj * gbar[r]);
    }
#pragma unroll
    for (int o = 0; o < TF_CO; ++o) {
      const float wv = wl[o][j];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        acc[o][r] += wv * gj[r];
        accc[o][r] += wv * gc[r];
      }
    }
  }
  // per channel: p1 = W[co] . g, p2 = (W Gc)[co] . W[co]; wave butterflies in double, then 4 waves through LDS (fixed order)
  double p1[TF_CO], p2[TF_CO];
#pragma unroll
  for (int o = 0; o < TF_CO; ++o) {
    p1[o] = 0.0;
    p2[o] = 0.0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int ci = tid + 256 * r;
      if (ci < c) {
        if (co0 + o < cout) t[(int64_t)(co0 + o) * c + ci] = acc[o][r];
        p1[o] += (double)wl[o][ci] * (double)gvec[ci];
        p2[o] += (double)accc[o][r] * (double)wl[o][ci];
      }
    }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
      p1[o] += __shfl_xor(p1[o], sft);
      p2[o] += __shfl_xor(p2[o], sft);
    }
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int o = 0; o < TF_CO; ++o) {
      red[(tid >> 6) * TF_CO + o][0] = p1[o];
      red[(tid >> 6) * TF_CO + o][1] = p2[o];
    }
  }
  __syncthreads();
  if (tid < TF_CO && co0 + tid < cout) {
    const int ch = co0 + tid;
    const double s1 = red[tid][0] + red[TF_CO + tid][0] + red[2 * TF_CO + tid][0] + red[3 * TF_CO + tid][0];
    const double s2 = red[tid][1] + red[TF_CO + tid][1] + red[2 * TF_CO + tid][1] + red[3 * TF_CO + tid][1];
    const double mu = s1 / n;
    double var = s2;                    // = W Gc W^T: already centred (see accc)
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[ch] * is;
    mean[ch] = (float)mu;
    invstd[ch] = is;
    scale[ch] = sc;
    shift[ch] = beta[ch] - (float)mu * sc;
    acoef[tid] = sc;
    if (running_mean) {
      const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)mu;
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unb;
    }
  }
  if (!wd) return;
  __syncthreads();
  // wd[ci][co0 .. co0+7] = A[co] W[co][ci]: TF_CO adjacent output channels per input channel
  for (int ci = tid; ci < c; ci += 256)
#pragma unroll
    for (int o = 0; o < TF_CO; ++o)
      if (co0 + o < cout) wd[(int64_t)ci * cout + co0 + o] = (D)(acoef[o] * wl[o][ci]);
}

// one wave per output channel: sum dz*y = W[co] . R[co]; coefficients of dy = A dz + B y + C; dgamma, dbeta
template <typename D>
__global__ __launch_bounds__(256) void bn_tail_coef_kernel(const float* __restrict__ rx, const float* __restrict__ parts,
                                                           int nparts, double n, int c,
                                                           const void* w, int cout, const float* gamma,
                                                           const float* mean, const float* invstd, float* dgamma,
                                                           float* dbeta, float* coef) {
  const int co = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (co >= cout) return;
  double sdy = 0.0;
#pragma unroll 4
  for (int ci = lane; ci < c; ci += 64) sdy += (double)wload<D>(w, (int64_t)co * c + ci) * (double)rx[(int64_t)co * c + ci];
  double s = 0.0;                                          // sum dz: the partial rows [nparts][cout][2], component 0
#pragma unroll 8
  for (int p = lane; p < nparts; p += 64) s += (double)parts[((int64_t)p * cout + co) * 2];
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) {
    sdy += __shfl_xor(sdy, sft);
    s += __shfl_xor(s, sft);
  }
  if (lane != 0) return;
  const double is = (double)invstd[co], mu = (double)mean[co];
  const double sxh = is * (sdy - mu * s);                 // sum dz * x_hat
  dgamma[co] += (float)sxh;
  dbeta[co] += (float)s;
  const double A = (double)gamma[co] * is, c1 = s / n, c2 = sxh / n;
  coef[co * 4 + 0] = (float)A;
  coef[co * 4 + 1] = (float)(-A * c2 * is);               // B
  coef[co * 4 + 2] = (float)(A * (c2 * is * mu - c1));    // C
  coef[co * 4 + 3] = 0.f;
}

// 32 x 32 (co, ci) tiles: dW += A R + B T + C g
template <typename D>
__device__ __forceinline__ void bn_tail_apply_part(int bx, int by, const float* __restrict__ rx, const float* __restrict__ gvec,
                                                   const float* __restrict__ t, int c, int cout,
                                                   const float* __restrict__ coef, float* dw) {
  const int co0 = by * 32, ci0 = bx * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + ty + 8 * i, ci = ci0 + tx;
    if (co < cout && ci < c) {
      const float A = coef[co * 4], B = coef[co * 4 + 1], Cc = coef[co * 4 + 2];
      const int64_t idx = (int64_t)co * c + ci;
      dw[idx] += A * rx[idx] + B * t[idx] + Cc * gvec[ci];
    }
  }
}

// m[i][j] = sum_co B[co] W[co][i] W[co][j]  (c x c, symmetric): 16 x 16 outputs per block, one per thread; the co axis is
// walked in order through LDS slabs of 16 output channels (fixed summation order)
template <typename D>
__device__ __forceinline__ void bn_tail_m_part(int bx, int by, const void* w, int c, int cout, const float* __restrict__ coef, D* m) {
  __shared__ float wi[16][17], wj[16][17], bco[16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i = by * 16 + ty, j = bx * 16 + tx;
  float acc = 0.f;
  for (int co0 = 0; co0 < cout; co0 += 16) {
    // stage 16 output channels x (16 i-columns, 16 j-columns) of W and their B
    const int co = co0 + ty;
    const int ci = by * 16 + tx, cj = bx * 16 + tx;
    wi[ty][tx] = (co < cout && ci < c) ? wload<D>(w, (int64_t)co * c + ci) : 0.f;
    wj[ty][tx] = (co < cout && cj < c) ? wload<D>(w, (int64_t)co * c + cj) : 0.f;
    if (threadIdx.x < 16) bco[threadIdx.x] = co0 + threadIdx.x < cout ? coef[(co0 + threadIdx.x) * 4 + 1] : 0.f;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += bco[q] * wi[q][ty] * wj[q][tx];
    __syncthreads();
  }
  if (i < c && j < c) m[(int64_t)i * c + j] = (D)acc;
}

// bias[ci] = sum_co C[co] W[co][ci]: 16 columns x 16 row partitions per block, fixed summation order
template <typename D>
__device__ __forceinline__ void bn_tail_bias_part(int bx, const void* w, int c, int cout, const float* __restrict__ coef,
                                                  float* bias) {
  __shared__ double red[16][17];
  const int col = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int ci = bx * 16 + col;
  double s = 0.0;
  if (ci < c) {
#pragma unroll 4
    for (int co = part; co < cout; co += 16) s += (double)coef[co * 4 + 2] * (double)wload<D>(w, (int64_t)co * c + ci);
  }
  red[part][col] = s;
  __syncthreads();
  if (part == 0 && ci < c) {
    double a = 0.0;
#pragma unroll
    for (int p = 0; p < 16; ++p) a += red[p][col];
    bias[ci] = (float)a;
  }
}

// ONE launch for everything that only needs the coefficients: block ranges [dW tiles | m tiles | bias columns]
template <typename D>
__global__ __launch_bounds__(256) void bn_tail_post_kernel(const float* rx, const float* gvec, const float* t, int c,
                                                           const void* w, int cout, const float* coef, float* dw, D* m,
                                                           float* bias, int ax, int nap, int mx, int nm) {
  int b = blockIdx.x;
  if (b < nap) { bn_tail_apply_part<D>(b % ax, b / ax, rx, gvec, t, c, cout, coef, dw); return; }
  b -= nap;
  if (b < nm) { bn_tail_m_part<D>(b % mx, b / mx, w, c, cout, coef, m); return; }
  bn_tail_bias_part<D>(b - nm, w, c, cout, coef, bias);
}

inline bool tail_args_ok(int c, int cout, int dtype) {
  return c > 0 && cout > 0 && (dtype == SFK_F32 || dtype == SFK_BF16);
}

}  // namespace

extern "C" int sfk_bn_tail_fwd(const float* gram, const float* a_sums, int32_t a_nparts, int64_t count, float* g, int32_t c,
                               const void* w, int32_t w_dtype, int32_t cout,
                               const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, int64_t* nbt, float* mean, float* invstd, float* scale, float* shift,
                               float* t, void* wd, sfk_stream_t stream) {
  if (!gram || !a_sums || a_nparts <= 0 || count <= 0 || !g || !w || !gamma || !beta || !mean || !invstd || !scale || !shift || !t)
    return SFK_ERR_INVALID;
  if (!tail_args_ok(c, cout, w_dtype) || (!running_mean) != (!running_var)) return SFK_ERR_INVALID;
  if (c > TF_MAXC) return SFK_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bn_tail_fold_g_kernel, dim3((c + 3) / 4), dim3(256), 0, s, a_sums, a_nparts, c, g);
  const dim3 grid((cout + TF_CO - 1) / TF_CO), blk(256);
  if (w_dtype == SFK_BF16)
    hipLaunchKernelGGL(bn_tail_fwd_kernel<bf16_t>, grid, blk, 0, s, gram, g, (double)count, c, w, cout, gamma, beta, eps, momentum,
                       running_mean, running_var, nbt, mean, invstd, scale, shift, t, static_cast<bf16_t*>(wd));
  else
    hipLaunchKernelGGL(bn_tail_fwd_kernel<float>, grid, blk, 0, s, gram, g, (double)count, c, w, cout, gamma, beta, eps, momentum,
                       running_mean, running_var, nbt, mean, invstd, scale, shift, t, static_cast<float*>(wd));
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

template <typename D>
static int tail_bwd_launch(const float* rx, const float* parts, int nparts, const float* g, int64_t count, const float* t, int c, const void* w, int cout,
                           const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                           float* dw, void* m, float* bias, float* coef, hipStream_t s) {
  hipLaunchKernelGGL(bn_tail_coef_kernel<D>, dim3((cout + 3) / 4), dim3(256), 0, s, rx, parts, nparts, (double)count, c, w, cout, gamma, mean,
                     invstd, dgamma, dbeta, coef);
  const int ax = (c + 31) / 32, ay = (cout + 31) / 32, mx = (c + 15) / 16, nb = (c + 15) / 16;
  hipLaunchKernelGGL(bn_tail_post_kernel<D>, dim3(ax * ay + mx * mx + nb), dim3(256), 0, s, rx, g, t, c, w, cout, coef, dw,
                     static_cast<D*>(m), bias, ax, ax * ay, mx, mx * mx);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_bn_tail_bwd(const float* rx, const float* parts, int32_t nparts, const float* g, int64_t count, const float* t,
                               int32_t c, const void* w, int32_t w_dtype, int32_t cout, const float* gamma, const float* mean, const float* invstd,
                               float* dgamma, float* dbeta, float* dw, void* m, float* bias, float* coef,
                               sfk_stream_t stream) {
  if (!rx || !parts || nparts <= 0 || !g || count <= 0 || !t || !w || !gamma || !mean || !invstd || !dgamma || !dbeta || !dw || !m || !bias || !coef)
    return SFK_ERR_INVALID;
  if (!tail_args_ok(c, cout, w_dtype)) return SFK_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return w_dtype == SFK_BF16
             ? tail_bwd_launch<bf16_t>(rx, parts, nparts, g, count, t, c, w, cout, gamma, mean, invstd, dgamma, dbeta, dw, m, bias, coef, s)
             : tail_bwd_launch<float>(rx, parts, nparts, g, count, t, c, w, cout, gamma, mean, invstd, dgamma, dbeta, dw, m, bias, coef, s);
}
