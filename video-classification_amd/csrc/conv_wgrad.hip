// sfk_conv_wgrad: Conv3d filter gradient on gfx950 MFMA.
//
//   dW[co][tap][ci] += sum_{pixel} dY[pixel][co] * X[gather(pixel, tap)][ci]
//
// The reduction runs over PIXELS, which is the strided dimension of both channels-last operands.  Tiles are
// staged as they lie in HBM ([pixel][channel] rows, 16 B per lane) and the MFMA fragments (8 consecutive pixels
// of one channel per lane) come out of LDS already transposed:
//   bf16: ds_read_b64_tr_b16 (two per fragment)          f32: one ds_read_b32 per MFMA (one k per lane)
// Grid = (co-tile x ci-tile, tap, pixel-split); partial tiles are combined with fp32 atomics into dW.
#include "sfk_common.h"

namespace {

struct WgradK {
  const void* x;
  const void* dy;
  float* dw;
  int xt, xh, xw, xld, xoff;
  int dld, doff;
  int M;
  FastDiv drw, drh, drt;
  int gst, gsh, gsw;
  int cin, cout, wtaps;
  int citiles;
  int chunks_per_split, nchunks;
  sfk_tap taps[SFK_MAX_TAPS];
};

constexpr int MK = 32;  // pixels per K-step

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <typename T, int TC> struct WT;
template <int TC> struct WT<bf16_t, TC> {
  static constexpr int VEC = 8, SEGS = TC / 8, ROWB = TC * 2 + 16;
  typedef bf16x8 frag;
  // fragment for channels c0..c0+15: lane (g = lane>>4, q = (lane&15)>>2, p = lane&3) addresses row 8g+q (+4),
  // columns c0+4p..c0+4p+3; the transpose read hands lane i of the group column c0+i of those 4 rows.
  static __device__ __forceinline__ frag load(const char* tile, int c0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const char* a = tile + (8 * g + q) * ROWB + (c0 + 4 * p) * 2;
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
template <int TC> struct WT<float, TC> {
  static constexpr int VEC = 4, SEGS = TC / 4, ROWB = TC * 4 + 16;
  struct frag { float v[8]; };
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

// TC = tile edge in channels (both co and ci); 4 waves as 2 x 2, each (TC/2) x (TC/2)
template <typename T, int TC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradK k) {
  using W = WT<T, TC>;
  constexpr int VEC = W::VEC, SEGS = W::SEGS, ROWB = W::ROWB;
  constexpr int F = TC / 2 / 16;                 // fragments per wave per side
  constexpr int NL = (MK * SEGS + 255) / 256;    // 16-byte loads per thread per operand per K-step
  constexpr int BUF = 2 * MK * ROWB;             // dY tile + X tile
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave & 1, wci = wave >> 1;
  const int cot = blockIdx.x / k.citiles, cit = blockIdx.x % k.citiles;
  const sfk_tap tp = k.taps[blockIdx.y];
  const int chunk0 = blockIdx.z * k.chunks_per_split;
  const int chunk1 = min(chunk0 + k.chunks_per_split, k.nchunks);

  const T* __restrict__ xp = static_cast<const T*>(k.x);
  const T* __restrict__ dp = static_cast<const T*>(k.dy);

  uint4 dr[NL], xr[NL];
  auto gload = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / SEGS, seg = idx % SEGS;
      const int m = chunk * MK + row;
      const bool rok = (MK * SEGS >= 256 || idx < MK * SEGS) && m < k.M;
      uint32_t q1, rw_, q2, rh_, n_, rt_;
      k.drw.divmod((uint32_t)m, q1, rw_);
      k.drh.divmod(q1, q2, rh_);
      k.drt.divmod(q2, n_, rt_);
      const int co = cot * TC + seg * VEC;
      dr[i] = (rok && co < k.cout) ? *reinterpret_cast<const uint4*>(dp + (int64_t)m * k.dld + k.doff + co)
                                   : make_uint4(0, 0, 0, 0);
      const int ci = cit * TC + seg * VEC;
      const int ti = (int)rt_ * k.gst + tp.dt, hi = (int)rh_ * k.gsh + tp.dh, wi = (int)rw_ * k.gsw + tp.dw;
      const bool xok = rok && ci < k.cin && (unsigned)ti < (unsigned)k.xt && (unsigned)hi < (unsigned)k.xh &&
                       (unsigned)wi < (unsigned)k.xw;
      const int64_t off = ((((int64_t)n_ * k.xt + ti) * k.xh + hi) * k.xw + wi) * k.xld + k.xoff + ci;
      xr[i] = xok ? *reinterpret_cast<const uint4*>(xp + off) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
    char* ds = smem + buf * BUF;
    char* xs = ds + MK * ROWB;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = tid + i * 256;
      if (MK * SEGS >= 256 || idx < MK * SEGS) {
        const int row = idx / SEGS, seg = idx % SEGS;
        *reinterpret_cast<uint4*>(ds + row * ROWB + seg * 16) = dr[i];
        *reinterpret_cast<uint4*>(xs + row * ROWB + seg * 16) = xr[i];
      }
    }
  };

  f32x4 acc[F][F];
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (chunk0 < chunk1) {
    gload(chunk0);
    lstore(0);
    __syncthreads();
    for (int ch = chunk0; ch < chunk1; ++ch) {
      const int buf = (ch - chunk0) & 1;
      const bool more = ch + 1 < chunk1;
      if (more) gload(ch + 1);
      const char* ds = smem + buf * BUF;
      const char* xs = ds + MK * ROWB;
      typename W::frag a[F], b[F];
#pragma unroll
      for (int i = 0; i < F; ++i) a[i] = W::load(ds, wco * (TC / 2) + 16 * i, lane);
#pragma unroll
      for (int j = 0; j < F; ++j) b[j] = W::load(xs, wci * (TC / 2) + 16 * j, lane);
#pragma unroll
      for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < F; ++j) W::mma(acc[i][j], a[i], b[j]);
      if (more) lstore(buf ^ 1);
      __syncthreads();
    }
  }

  // D[row = co][col = ci]: lane holds co = 4*(lane>>4) + r, ci = lane & 15 -> 16 lanes add 16 consecutive floats
  const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < F; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cot * TC + wco * (TC / 2) + 16 * i + 4 * g + r;
      if (co >= k.cout) continue;
      float* rowp = k.dw + ((int64_t)co * k.wtaps + tp.widx) * k.cin;
#pragma unroll
      for (int j = 0; j < F; ++j) {
        const int ci = cit * TC + wci * (TC / 2) + 16 * j + l15;
        if (ci < k.cin) atomicAdd(rowp + ci, acc[i][j][r]);
      }
    }
  }
}

int validate(const sfk_wgrad_desc* d) {
  if (!d || !d->dw) return SFK_ERR_INVALID;
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
  return SFK_OK;
}

template <typename T>
int launch(const sfk_wgrad_desc* d, hipStream_t s) {
  WgradK k;
  k.x = d->x.ptr; k.dy = d->dy.ptr; k.dw = d->dw;
  k.xt = d->x.t; k.xh = d->x.h; k.xw = d->x.w; k.xld = d->x.ld; k.xoff = d->x.c_off;
  k.dld = d->dy.ld; k.doff = d->dy.c_off;
  k.M = (int)sfk_fmap_pixels(&d->dy);
  k.drw.set(d->dy.w); k.drh.set(d->dy.h); k.drt.set(d->dy.t);
  k.gst = d->gs[0]; k.gsh = d->gs[1]; k.gsw = d->gs[2];
  k.cin = d->cin; k.cout = d->cout; k.wtaps = d->wtaps;
  for (int i = 0; i < SFK_MAX_TAPS; ++i) k.taps[i] = d->taps[i < d->ntaps ? i : 0];
  const int tc = (d->cin >= 128 && d->cout >= 128) ? 128 : 64;
  const int cotiles = (d->cout + tc - 1) / tc;
  k.citiles = (d->cin + tc - 1) / tc;
  k.nchunks = (k.M + MK - 1) / MK;
  // pixel splits: enough workgroups to cover the 256 CUs a few times, at least 8 K-steps each
  const int base = cotiles * k.citiles * d->ntaps;
  int splits = (1024 + base - 1) / base;
  const int max_splits = (k.nchunks + 7) / 8;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  k.chunks_per_split = (k.nchunks + splits - 1) / splits;
  splits = (k.nchunks + k.chunks_per_split - 1) / k.chunks_per_split;
  const dim3 grid((unsigned)(cotiles * k.citiles), (unsigned)d->ntaps, (unsigned)splits), block(256);
  if (tc == 128) hipLaunchKernelGGL((conv_wgrad_kernel<T, 128>), grid, block, 0, s, k);
  else hipLaunchKernelGGL((conv_wgrad_kernel<T, 64>), grid, block, 0, s, k);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

}  // namespace

extern "C" int sfk_conv_wgrad(const sfk_wgrad_desc* d, sfk_stream_t stream) {
  const int st = validate(d);
  if (st != SFK_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return d->x.dtype == SFK_BF16 ? launch<bf16_t>(d, s) : launch<float>(d, s);
}
