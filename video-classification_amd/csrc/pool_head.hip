// Stem MaxPool3d, head (AvgPool3d windows -> dropout -> position mean), Linear, softmax cross-entropy.
// Small HBM-bound kernels; channels-last, 16 B per lane where the data is a feature map.
#include "sfk_common.h"

namespace {

struct FM {
  void* p;
  int t, h, w, ld, off;
};
inline FM fm_of(const sfk_fmap* f) { return FM{f->ptr, f->t, f->h, f->w, f->ld, f->c_off}; }

// ------------------------------------------------------------------ MaxPool (1,k,k)/(1,s,s)/(0,p,p)
// KC/SC/PC > 0: window / stride / padding known at compile time (the stems' (3, 2, 1)): the window loops unroll and the
// backward's divisibility tests fold to parity tests
// BN: the input is a raw conv output and the pooled quantity is relu(x * scale + shift) rounded to T -- BatchNorm + ReLU +
// MaxPool of a stem in one pass (sfk_bn_maxpool_fwd); the activation map itself never exists in HBM
template <typename T, int KC = 0, int SC = 0, int PC = 0, bool BN = false>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(FM x, FM y, uint8_t* argmax, int n, int c, int k_, int s_,
                                                          int p_, FastDiv dcg, FastDiv dwo, FastDiv dho,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift) {
  const int k = KC ? KC : k_, s = KC ? SC : s_, p = KC ? PC : p_;
  constexpr int VEC = DT<T>::VEC;
  const int cgs = c / VEC;
  const int64_t total = (int64_t)n * y.t * y.h * y.w * cgs;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    uint32_t pix, cg, q1, wo, nt, ho;
    dcg.divmod((uint32_t)idx, pix, cg);
    dwo.divmod(pix, q1, wo);
    dho.divmod(q1, nt, ho);  // nt = n*T + t (pooling is per frame)
    float best[VEC];
    int arg[VEC];
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      best[i] = -INFINITY;
      arg[i] = 0;
      sc[i] = BN ? scale[cg * VEC + i] : 1.f;
      sh[i] = BN ? shift[cg * VEC + i] : 0.f;
    }
    bool first = true;
    if (KC) {
      // compile-time window: every tap is loaded (coordinates clamped into the frame, so no branch sits around a load and
      // all k*k loads are in flight together) and taps outside the frame are skipped afterwards
      Vec16<T> v[(KC ? KC : 1) * (KC ? KC : 1)];
#pragma unroll
      for (int kh = 0; kh < (KC ? KC : 1); ++kh)
#pragma unroll
        for (int kw = 0; kw < (KC ? KC : 1); ++kw) {
          int hi = (int)ho * s - p + kh, wi = (int)wo * s - p + kw;
          hi = hi < 0 ? 0 : (hi >= x.h ? x.h - 1 : hi);
          wi = wi < 0 ? 0 : (wi >= x.w ? x.w - 1 : wi);
          v[kh * (KC ? KC : 1) + kw].load(static_cast<const T*>(x.p) + (((int64_t)nt * x.h + hi) * x.w + wi) * x.ld + x.off + cg * VEC);
        }
#pragma unroll
      for (int kh = 0; kh < (KC ? KC : 1); ++kh)
#pragma unroll
        for (int kw = 0; kw < (KC ? KC : 1); ++kw) {
          const int hi = (int)ho * s - p + kh, wi = (int)wo * s - p + kw;
          const bool in = (unsigned)hi < (unsigned)x.h && (unsigned)wi < (unsigned)x.w;
          Vec16<T> act;
          if (BN) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) act.set(i, fmaxf(v[kh * (KC ? KC : 1) + kw].get(i) * sc[i] + sh[i], 0.f));
          }
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float f = BN ? act.get(i) : v[kh * (KC ? KC : 1) + kw].get(i);
            if (in && (first || f > best[i] || f != f)) { best[i] = f; arg[i] = kh * k + kw; }
          }
          first = first && !in;
        }
    } else
    for (int kh = 0; kh < k; ++kh) {
      const int hi = (int)ho * s - p + kh;
      if ((unsigned)hi >= (unsigned)x.h) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int wi = (int)wo * s - p + kw;
        if ((unsigned)wi >= (unsigned)x.w) continue;
        Vec16<T> v;
        v.load(static_cast<const T*>(x.p) + (((int64_t)nt * x.h + hi) * x.w + wi) * x.ld + x.off + cg * VEC);
        if (BN) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) v.set(i, fmaxf(v.get(i) * sc[i] + sh[i], 0.f));
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float f = v.get(i);
          // torch: take the first element, then replace on (val > max) or isnan(val)
          if (first || f > best[i] || f != f) { best[i] = f; arg[i] = kh * k + kw; }
        }
        first = false;
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.set(i, best[i]);
    o.store(static_cast<T*>(y.p) + (int64_t)pix * y.ld + y.off + cg * VEC);
    // the VEC argmax bytes of this channel group leave as ONE store (8 B for bf16, 4 B for f32)
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if (i < 4) lo |= (uint32_t)arg[i] << (8 * i);
      else hi |= (uint32_t)arg[i] << (8 * (i - 4));
    }
    uint8_t* ap = argmax + (int64_t)pix * c + cg * VEC;
    if (VEC == 8) *reinterpret_cast<uint2*>(ap) = make_uint2(lo, hi);
    else *reinterpret_cast<uint32_t*>(ap) = lo;
  }
}

template <typename T, int KC = 0, int SC = 0, int PC = 0>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(FM dy, const uint8_t* argmax, FM dx, int n, int c, int k_,
                                                          int s_, int p_, FastDiv dcg, FastDiv dwi, FastDiv dhi) {
  const int k = KC ? KC : k_, s = KC ? SC : s_, p = KC ? PC : p_;
  constexpr int VEC = DT<T>::VEC;
  const int cgs = c / VEC;
  const int64_t total = (int64_t)n * dx.t * dx.h * dx.w * cgs;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    uint32_t pix, cg, q1, wi, nt, hi;
    dcg.divmod((uint32_t)idx, pix, cg);
    dwi.divmod(pix, q1, wi);
    dhi.divmod(q1, nt, hi);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    if (KC == 3 && SC == 2 && PC == 1) {
      // (3, 2, 1): an input row belongs to window rows ho0 = (hi+1-kh0)/2 with kh0 = (hi+1)&1 and, when kh0 == 0, also to
      // ho0-1 with kh = 2 -- at most 2 x 2 windows.  All four (gradient, argmax) pairs are loaded from clamped coordinates,
      // invalid ones are dropped afterwards: no branch around a load
      const int kh0 = ((int)hi + 1) & 1, kw0 = ((int)wi + 1) & 1;
      const int ho0 = ((int)hi + 1 - kh0) >> 1, wo0 = ((int)wi + 1 - kw0) >> 1;
      Vec16<T> g[4];
      uint32_t alo[4], ahi[4];
      bool ok[4];
      int code[4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int ho = ho0 - a, wo = wo0 - b, kh = kh0 + 2 * a, kw = kw0 + 2 * b;
          ok[2 * a + b] = kh < 3 && kw < 3 && ho >= 0 && wo >= 0 && ho < dy.h && wo < dy.w;
          code[2 * a + b] = kh * 3 + kw;
          const int hc = ho < 0 ? 0 : (ho >= dy.h ? dy.h - 1 : ho), wc = wo < 0 ? 0 : (wo >= dy.w ? dy.w - 1 : wo);
          const int64_t opix = ((int64_t)nt * dy.h + hc) * dy.w + wc;
          g[2 * a + b].load(static_cast<const T*>(dy.p) + opix * dy.ld + dy.off + cg * VEC);
          const uint8_t* ap = argmax + opix * c + cg * VEC;
          if (VEC == 8) {
            const uint2 a2 = *reinterpret_cast<const uint2*>(ap);
            alo[2 * a + b] = a2.x; ahi[2 * a + b] = a2.y;
          } else {
            alo[2 * a + b] = *reinterpret_cast<const uint32_t*>(ap); ahi[2 * a + b] = 0;
          }
        }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const uint32_t am = ((i < 4 ? alo[q] : ahi[q]) >> (8 * (i & 3))) & 0xFFu;
          if (ok[q] && am == (uint32_t)code[q]) acc[i] += g[q].get(i);
        }
    } else
    // windows (ho, wo) that contain (hi, wi): ho*s - p + kh == hi
    for (int kh = 0; kh < k; ++kh) {
      const int num_h = (int)hi + p - kh;
      if (num_h < 0 || num_h % s) continue;
      const int ho = num_h / s;
      if (ho >= dy.h) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int num_w = (int)wi + p - kw;
        if (num_w < 0 || num_w % s) continue;
        const int wo = num_w / s;
        if (wo >= dy.w) continue;
        const int64_t opix = ((int64_t)nt * dy.h + ho) * dy.w + wo;
        Vec16<T> g;
        g.load(static_cast<const T*>(dy.p) + opix * dy.ld + dy.off + cg * VEC);
        const uint8_t* ap = argmax + opix * c + cg * VEC;
        uint32_t lo, hi = 0;
        if (VEC == 8) {
          const uint2 a2 = *reinterpret_cast<const uint2*>(ap);
          lo = a2.x; hi = a2.y;
        } else {
          lo = *reinterpret_cast<const uint32_t*>(ap);
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const uint32_t a = ((i < 4 ? lo : hi) >> (8 * (i & 3))) & 0xFFu;
          if (a == (uint32_t)(kh * k + kw)) acc[i] += g.get(i);
        }
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o.set(i, acc[i]);
    o.store(static_cast<T*>(dx.p) + (int64_t)pix * dx.ld + dx.off + cg * VEC);
  }
}

// ------------------------------------------------------------------ head pooling with counter-based dropout
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// keep decision for (sample n, feature f, position p): uniform 24-bit draw >= rate
__device__ __forceinline__ bool keep_of(uint64_t seed, int n, int f, int p, float rate) {
  const uint64_t key = (((uint64_t)(uint32_t)n << 40) ^ ((uint64_t)(uint32_t)f << 20) ^ (uint64_t)(uint32_t)p);
  const uint32_t r = (uint32_t)(splitmix64(seed ^ splitmix64(key)) >> 40);
  return (float)r * (1.0f / 16777216.0f) >= rate;
}

template <typename T>
__global__ __launch_bounds__(256) void head_pool_fwd_kernel(FM x, int n, int c, int kt, int kh, int kw, float rate,
                                                            const uint64_t* seedp, float* feat, int feat_ld,
                                                            int f_off) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * c) return;
  const int ni = idx / c, ch = idx % c;
  const int pt = x.t - kt + 1, ph = x.h - kh + 1, pw = x.w - kw + 1;
  const float inv_win = 1.f / (float)(kt * kh * kw), inv_keep = rate > 0.f ? 1.f / (1.f - rate) : 1.f;
  const uint64_t seed = rate > 0.f ? seedp[0] : 0ull;
  const T* base = static_cast<const T*>(x.p) + (int64_t)ni * x.t * x.h * x.w * x.ld + x.off + ch;
  float total = 0.f;
  int pos = 0;
  for (int a = 0; a < pt; ++a)
    for (int b = 0; b < ph; ++b)
      for (int d = 0; d < pw; ++d, ++pos) {
        if (rate > 0.f && !keep_of(seed, ni, f_off + ch, pos, rate)) continue;
        float sum = 0.f;
        for (int i = 0; i < kt; ++i)
          for (int j = 0; j < kh; ++j)
            for (int l = 0; l < kw; ++l)
              sum += (float)base[(((int64_t)(a + i) * x.h + (b + j)) * x.w + (d + l)) * x.ld];
        total += sum * inv_win * inv_keep;
      }
  feat[(int64_t)ni * feat_ld + f_off + ch] = total / (float)(pt * ph * pw);
}

// Single-position head (the window covers the whole map: canonical (8,7,7) / (32,7,7) pools of the 8x8 model): the
// dropout decision depends on (sample, feature) only, so it is drawn once per channel and the map is reduced by 32
// pixel slices of 16-byte loads per workgroup -- the per-(n, channel) thread of the general kernel walks 392..1568
// two-byte loads serially.  grid = (c / 256 rounded up, n), block = 1024 = 32 channel groups of 8 x 32 slices.
template <typename T>
__global__ __launch_bounds__(1024) void head_pool1_fwd_kernel(FM x, int c, float rate, const uint64_t* seedp, float* feat,
                                                              int feat_ld, int f_off) {
  constexpr int VEC = DT<T>::VEC;
  constexpr int CG = 256 / VEC;                 // channel groups per block (256 channels)
  constexpr int SL = 1024 / CG;                 // pixel slices
  __shared__ float red[SL][256 + 8];
  const int ni = blockIdx.y;
  const int cgl = threadIdx.x % CG, sl = threadIdx.x / CG;
  const int ch0 = blockIdx.x * 256 + cgl * VEC;
  const int pixels = x.t * x.h * x.w;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  if (ch0 < c) {
    const T* base = static_cast<const T*>(x.p) + (int64_t)ni * pixels * x.ld + x.off + ch0;
    for (int px = sl; px < pixels; px += SL) {
      Vec16<T> v;
      v.load(base + (int64_t)px * x.ld);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += v.get(i);
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) red[sl][cgl * VEC + i] = acc[i];
  __syncthreads();
  if (threadIdx.x < 256) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch < c) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < SL; ++r) s += red[r][threadIdx.x];
      float scale = 1.f / (float)pixels;
      if (rate > 0.f) scale = keep_of(seedp[0], ni, f_off + ch, 0, rate) ? scale / (1.f - rate) : 0.f;
      feat[(int64_t)ni * feat_ld + f_off + ch] = s * scale;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void head_pool1_bwd_kernel(const float* dfeat, int feat_ld, int f_off, FM dx, int c,
                                                             float rate, const uint64_t* seedp, int px_per_block) {
  constexpr int VEC = DT<T>::VEC;
  const int ni = blockIdx.z;
  const int cgs = c / VEC, cgs_b = cgs < 256 ? cgs : 256, rows_b = 256 / cgs_b;
  const int cg = blockIdx.y * 256 + threadIdx.x % cgs_b, row = threadIdx.x / cgs_b;
  if (cg >= cgs || row >= rows_b) return;
  const int pixels = dx.t * dx.h * dx.w;
  float scale = 1.f / (float)pixels;
  Vec16<T> o;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int ch = cg * VEC + i;
    float g = dfeat[(int64_t)ni * feat_ld + f_off + ch] * scale;
    if (rate > 0.f) g = keep_of(seedp[0], ni, f_off + ch, 0, rate) ? g / (1.f - rate) : 0.f;
    o.set(i, g);
  }
  T* base = static_cast<T*>(dx.p) + (int64_t)ni * pixels * dx.ld + dx.off + cg * VEC;
  const int p0 = blockIdx.x * px_per_block, p1 = min(p0 + px_per_block, pixels);
  for (int px = p0 + row; px < p1; px += rows_b) o.store(base + (int64_t)px * dx.ld);
}

template <typename T>
__global__ __launch_bounds__(256) void head_pool_bwd_kernel(const float* dfeat, int feat_ld, int f_off, FM dx, int n,
                                                            int c, int kt, int kh, int kw, float rate,
                                                            const uint64_t* seedp) {
  const int64_t total = (int64_t)n * dx.t * dx.h * dx.w * c;
  const int pt = dx.t - kt + 1, ph = dx.h - kh + 1, pw = dx.w - kw + 1;
  const float inv = 1.f / ((float)(kt * kh * kw) * (float)(pt * ph * pw)) * (rate > 0.f ? 1.f / (1.f - rate) : 1.f);
  const uint64_t seed = rate > 0.f ? seedp[0] : 0ull;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int ch = (int)(idx % c);
    int64_t pix = idx / c;
    const int wi = (int)(pix % dx.w); pix /= dx.w;
    const int hi = (int)(pix % dx.h); pix /= dx.h;
    const int ti = (int)(pix % dx.t);
    const int ni = (int)(pix / dx.t);
    int cnt = 0;
    for (int a = max(0, ti - kt + 1); a <= min(ti, pt - 1); ++a)
      for (int b = max(0, hi - kh + 1); b <= min(hi, ph - 1); ++b)
        for (int d = max(0, wi - kw + 1); d <= min(wi, pw - 1); ++d) {
          const int pos = (a * ph + b) * pw + d;
          if (rate > 0.f && !keep_of(seed, ni, f_off + ch, pos, rate)) continue;
          ++cnt;
        }
    const float g = dfeat[(int64_t)ni * feat_ld + f_off + ch] * inv * (float)cnt;
    static_cast<T*>(dx.p)[(idx / c) * dx.ld + dx.off + ch] = (T)g;
  }
}

__global__ void head_mask_kernel(int n, int c, int f_off, int positions, float rate, const uint64_t* seedp,
                                 uint8_t* mask) {
  const int64_t total = (int64_t)n * c * positions;
  const uint64_t seed = seedp[0];
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int pos = (int)(idx % positions);
    const int ch = (int)((idx / positions) % c);
    const int ni = (int)(idx / ((int64_t)positions * c));
    mask[idx] = (rate <= 0.f || keep_of(seed, ni, f_off + ch, pos, rate)) ? 1 : 0;
  }
}

// ------------------------------------------------------------------ Linear (fp32)
__global__ __launch_bounds__(256) void fc_fwd_kernel(const float* feat, const float* w, const float* b, float* out,
                                                     int n, int f, int k) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= n * k) return;
  const int ni = o / k, ki = o % k;
  const float* fp = feat + (int64_t)ni * f;
  const float* wp = w + (int64_t)ki * f;
  float s = 0.f;
#pragma unroll 8                                    // (the forward's last kernel: keep 16 loads in flight per lane)
  for (int i = lane; i < f; i += 64) s += fp[i] * wp[i];
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) s += __shfl_xor(s, sft);
  if (lane == 0) out[o] = s + (b ? b[ki] : 0.f);
}

__global__ __launch_bounds__(256) void fc_bwd_dfeat_kernel(const float* dl, const float* w, float* dfeat, int n,
                                                           int f, int k) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * f) return;
  const int ni = idx / f, fi = idx % f;
  // the whole backward waits for this kernel: keep 8 independent loads in flight per thread (sum order unchanged)
  float s = 0.f;
  int ki = 0;
  for (; ki + 8 <= k; ki += 8) {
    float wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wv[u] = w[(int64_t)(ki + u) * f + fi];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += dl[ni * k + ki + u] * wv[u];
  }
  for (; ki < k; ++ki) s += dl[ni * k + ki] * w[(int64_t)ki * f + fi];
  dfeat[idx] = s;
}

__global__ __launch_bounds__(256) void fc_bwd_dw_kernel(const float* dl, const float* feat, float* dw, float* db,
                                                        int n, int f, int k) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)k * f) return;
  const int ki = (int)(idx / f), fi = (int)(idx % f);
  float s = 0.f, sb = 0.f;
  for (int ni = 0; ni < n; ++ni) {
    const float d = dl[ni * k + ki];
    s += d * feat[(int64_t)ni * f + fi];
    sb += d;
  }
  dw[idx] += s;
  if (fi == 0 && db) db[ki] += sb;
}

// ------------------------------------------------------------------ softmax cross-entropy (one block per sample)
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* logits, const int64_t* labels, int n, int k,
                                                         float gscale, float* dlogits, float* loss_out,
                                                         float* loss_sum, int* correct) {
  __shared__ float sred[256];
  __shared__ int sidx[256];
  const int ni = blockIdx.x, tid = threadIdx.x;
  const float* lp = logits + (int64_t)ni * k;
  float mx = -INFINITY;
  int mi = 0x7fffffff;
  for (int i = tid; i < k; i += 256) {
    const float v = lp[i];
    if (v > mx) { mx = v; mi = i; }  // strided scan keeps the first index per thread
  }
  sred[tid] = mx;
  sidx[tid] = mi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const float o = sred[tid + s];
      const int oi = sidx[tid + s];
      if (o > sred[tid] || (o == sred[tid] && oi < sidx[tid])) { sred[tid] = o; sidx[tid] = oi; }
    }
    __syncthreads();
  }
  mx = sred[0];
  const int amax = sidx[0];
  __syncthreads();
  float se = 0.f;
  for (int i = tid; i < k; i += 256) se += expf(lp[i] - mx);
  sred[tid] = se;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) sred[tid] += sred[tid + s];
    __syncthreads();
  }
  se = sred[0];
  const int lab = (int)labels[ni];
  const float inv_n = 1.f / (float)n;
  if (dlogits)
    for (int i = tid; i < k; i += 256)
      dlogits[(int64_t)ni * k + i] = (expf(lp[i] - mx) / se - (i == lab ? 1.f : 0.f)) * inv_n * gscale;
  if (tid == 0) {
    const float loss = (logf(se) + mx - lp[lab]) * inv_n;
    if (loss_out) atomicAdd(loss_out, loss);
    if (loss_sum) atomicAdd(loss_sum, loss);
    if (correct && amax == lab) atomicAdd(correct, 1);
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = (total + 255) / 256;
  const int64_t cap = sfk_tune().pool_blocks;   // one pass per thread (a grid-stride loop over 8192 blocks: +8 % on the max-pool backward)
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

static int maxpool_fwd_launch(const sfk_fmap* x, const float* scale, const float* shift, const sfk_fmap* y, uint8_t* argmax,
                              int32_t k, int32_t s, int32_t p, sfk_stream_t stream) {
  if (!sfk_fmap_ok(x) || !sfk_fmap_ok(y) || !argmax || k <= 0 || k > 15 || s <= 0 || p < 0) return SFK_ERR_INVALID;
  if (x->dtype != y->dtype || x->n != y->n || x->t != y->t || x->c != y->c) return SFK_ERR_INVALID;
  if (y->h != (x->h + 2 * p - k) / s + 1 || y->w != (x->w + 2 * p - k) / s + 1) return SFK_ERR_INVALID;
  if (!sfk_fmap_vec_ok(x) || !sfk_fmap_vec_ok(y)) return SFK_ERR_UNSUPPORTED;
  const int cgs = x->c / sfk_vec_of(x->dtype);
  const int64_t total = sfk_fmap_pixels(y) * cgs;
  if (total >= (1ll << 31)) return SFK_ERR_UNSUPPORTED;
  FastDiv dcg, dwo, dho;
  dcg.set(cgs); dwo.set(y->w); dho.set(y->h);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(total)), blk(256);
  const bool k321 = k == 3 && s == 2 && p == 1;
#define SFK_POOL(T, K, S, P, BN) \
  hipLaunchKernelGGL((maxpool_fwd_kernel<T, K, S, P, BN>), grid, blk, 0, st, fm_of(x), fm_of(y), argmax, x->n, x->c, k, s, p, dcg, dwo, dho, scale, shift)
  if (scale) {
    if (x->dtype == SFK_BF16) { if (k321) SFK_POOL(bf16_t, 3, 2, 1, true); else SFK_POOL(bf16_t, 0, 0, 0, true); }
    else SFK_POOL(float, 0, 0, 0, true);
  } else {
    if (x->dtype == SFK_BF16) { if (k321) SFK_POOL(bf16_t, 3, 2, 1, false); else SFK_POOL(bf16_t, 0, 0, 0, false); }
    else SFK_POOL(float, 0, 0, 0, false);
  }
#undef SFK_POOL
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_maxpool_fwd(const sfk_fmap* x, const sfk_fmap* y, uint8_t* argmax, int32_t k, int32_t s, int32_t p,
                               sfk_stream_t stream) {
  return maxpool_fwd_launch(x, nullptr, nullptr, y, argmax, k, s, p, stream);
}

extern "C" int sfk_bn_maxpool_fwd(const sfk_fmap* y, const float* scale, const float* shift, const sfk_fmap* out,
                                  uint8_t* argmax, int32_t k, int32_t s, int32_t p, sfk_stream_t stream) {
  if (!scale || !shift) return SFK_ERR_INVALID;
  return maxpool_fwd_launch(y, scale, shift, out, argmax, k, s, p, stream);
}

extern "C" int sfk_maxpool_bwd(const sfk_fmap* dy, const uint8_t* argmax, const sfk_fmap* dx, int32_t k, int32_t s,
                               int32_t p, sfk_stream_t stream) {
  if (!sfk_fmap_ok(dx) || !sfk_fmap_ok(dy) || !argmax || k <= 0 || k > 15 || s <= 0 || p < 0) return SFK_ERR_INVALID;
  if (dx->dtype != dy->dtype || dx->n != dy->n || dx->t != dy->t || dx->c != dy->c) return SFK_ERR_INVALID;
  if (dy->h != (dx->h + 2 * p - k) / s + 1 || dy->w != (dx->w + 2 * p - k) / s + 1) return SFK_ERR_INVALID;
  if (!sfk_fmap_vec_ok(dx) || !sfk_fmap_vec_ok(dy)) return SFK_ERR_UNSUPPORTED;
  const int cgs = dx->c / sfk_vec_of(dx->dtype);
  const int64_t total = sfk_fmap_pixels(dx) * cgs;
  if (total >= (1ll << 31)) return SFK_ERR_UNSUPPORTED;
  FastDiv dcg, dwi, dhi;
  dcg.set(cgs); dwi.set(dx->w); dhi.set(dx->h);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dx->dtype == SFK_BF16)
    if (k == 3 && s == 2 && p == 1)
      hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 3, 2, 1>), dim3(grid_for(total)), dim3(256), 0, st, fm_of(dy), argmax, fm_of(dx), dx->n, dx->c, k, s, p, dcg, dwi, dhi);
    else
      hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, fm_of(dy), argmax, fm_of(dx), dx->n, dx->c, k, s, p, dcg, dwi, dhi);
  else
    hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, fm_of(dy), argmax, fm_of(dx), dx->n, dx->c, k, s, p, dcg, dwi, dhi);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_head_pool_fwd(const sfk_fmap* x, int32_t kt, int32_t kh, int32_t kw, float rate,
                                 const uint64_t* seed, float* feat, int32_t feat_ld, int32_t f_off,
                                 sfk_stream_t stream) {
  if (!sfk_fmap_ok(x) || !feat || kt <= 0 || kh <= 0 || kw <= 0 || kt > x->t || kh > x->h || kw > x->w)
    return SFK_ERR_INVALID;
  if (rate < 0.f || rate >= 1.f || (rate > 0.f && !seed) || f_off < 0 || feat_ld < f_off + x->c) return SFK_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (kt == x->t && kh == x->h && kw == x->w && sfk_fmap_vec_ok(x)) {   // one position: the window is the whole map
    const dim3 g1((unsigned)((x->c + 255) / 256), (unsigned)x->n);
    if (x->dtype == SFK_BF16)
      hipLaunchKernelGGL(head_pool1_fwd_kernel<bf16_t>, g1, dim3(1024), 0, st, fm_of(x), x->c, rate, seed, feat, feat_ld, f_off);
    else
      hipLaunchKernelGGL(head_pool1_fwd_kernel<float>, g1, dim3(1024), 0, st, fm_of(x), x->c, rate, seed, feat, feat_ld, f_off);
    SFK_CHECK_LAUNCH();
    return SFK_OK;
  }
  const unsigned g = (unsigned)((x->n * x->c + 255) / 256);
  if (x->dtype == SFK_BF16)
    hipLaunchKernelGGL(head_pool_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, st, fm_of(x), x->n, x->c, kt, kh, kw, rate, seed, feat, feat_ld, f_off);
  else
    hipLaunchKernelGGL(head_pool_fwd_kernel<float>, dim3(g), dim3(256), 0, st, fm_of(x), x->n, x->c, kt, kh, kw, rate, seed, feat, feat_ld, f_off);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_head_pool_bwd(const float* dfeat, int32_t feat_ld, int32_t f_off, int32_t kt, int32_t kh, int32_t kw,
                                 float rate, const uint64_t* seed, const sfk_fmap* dx, sfk_stream_t stream) {
  if (!sfk_fmap_ok(dx) || !dfeat || kt <= 0 || kh <= 0 || kw <= 0 || kt > dx->t || kh > dx->h || kw > dx->w)
    return SFK_ERR_INVALID;
  if (rate < 0.f || rate >= 1.f || (rate > 0.f && !seed) || f_off < 0 || feat_ld < f_off + dx->c) return SFK_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (kt == dx->t && kh == dx->h && kw == dx->w && sfk_fmap_vec_ok(dx)) {
    const int vec = sfk_vec_of(dx->dtype), cgs = dx->c / vec;
    const int pixels = dx->t * dx->h * dx->w;
    int pxb = (pixels + 7) / 8;                     // 8 pixel blocks per (sample, channel chunk)
    if (pxb < 1) pxb = 1;
    const dim3 g1((unsigned)((pixels + pxb - 1) / pxb), (unsigned)((cgs + 255) / 256), (unsigned)dx->n);
    if (dx->dtype == SFK_BF16)
      hipLaunchKernelGGL(head_pool1_bwd_kernel<bf16_t>, g1, dim3(256), 0, st, dfeat, feat_ld, f_off, fm_of(dx), dx->c, rate, seed, pxb);
    else
      hipLaunchKernelGGL(head_pool1_bwd_kernel<float>, g1, dim3(256), 0, st, dfeat, feat_ld, f_off, fm_of(dx), dx->c, rate, seed, pxb);
    SFK_CHECK_LAUNCH();
    return SFK_OK;
  }
  const int64_t total = sfk_fmap_pixels(dx) * dx->c;
  if (dx->dtype == SFK_BF16)
    hipLaunchKernelGGL(head_pool_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, dfeat, feat_ld, f_off, fm_of(dx), dx->n, dx->c, kt, kh, kw, rate, seed);
  else
    hipLaunchKernelGGL(head_pool_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, dfeat, feat_ld, f_off, fm_of(dx), dx->n, dx->c, kt, kh, kw, rate, seed);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_head_dropout_mask(int32_t n, int32_t c, int32_t f_off, int32_t positions, float rate,
                                     const uint64_t* seed, uint8_t* mask, sfk_stream_t stream) {
  if (n <= 0 || c <= 0 || positions <= 0 || !seed || !mask || rate < 0.f || rate >= 1.f) return SFK_ERR_INVALID;
  hipLaunchKernelGGL(head_mask_kernel, dim3(grid_for((int64_t)n * c * positions)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n, c, f_off, positions, rate, seed, mask);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_fc_fwd(const float* feat, const float* w, const float* b, float* logits, int32_t n, int32_t f,
                          int32_t k, sfk_stream_t stream) {
  if (!feat || !w || !logits || n <= 0 || f <= 0 || k <= 0) return SFK_ERR_INVALID;
  hipLaunchKernelGGL(fc_fwd_kernel, dim3((unsigned)((n * k + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     feat, w, b, logits, n, f, k);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_fc_bwd(const float* dlogits, const float* feat, const float* w, float* dfeat, float* dw, float* db,
                          int32_t n, int32_t f, int32_t k, sfk_stream_t stream) {
  if (!dlogits || !feat || !w || n <= 0 || f <= 0 || k <= 0) return SFK_ERR_INVALID;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dfeat) hipLaunchKernelGGL(fc_bwd_dfeat_kernel, dim3((unsigned)((n * f + 255) / 256)), dim3(256), 0, st, dlogits, w, dfeat, n, f, k);
  if (dw) hipLaunchKernelGGL(fc_bwd_dw_kernel, dim3((unsigned)(((int64_t)k * f + 255) / 256)), dim3(256), 0, st, dlogits, feat, dw, db, n, f, k);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_softmax_ce(const float* logits, const int64_t* labels, int32_t n, int32_t k, float gscale,
                              float* dlogits, float* loss_out, float* loss_sum, int32_t* correct,
                              sfk_stream_t stream) {
  if (!logits || !labels || n <= 0 || k <= 0) return SFK_ERR_INVALID;
  hipLaunchKernelGGL(softmax_ce_kernel, dim3((unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream), logits,
                     labels, n, k, gscale, dlogits, loss_out, loss_sum, correct);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}
