// Fused Adam over the flat parameter arena, filter re-layout for the data-gradient pass, casts, the tuning table.
#include <string.h>

#include "sfk_common.h"

namespace {

__global__ void adam_step_inc_kernel(int64_t* step) { step[0] += 1; }

template <typename S>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t count,
                                                   float lr, float b1, float b2, float eps, float gscale,
                                                   const int64_t* step, S* __restrict__ shadow) {
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const double t = (double)step[0];
    bc[0] = (float)(1.0 - pow((double)b1, t));
    bc[1] = (float)(1.0 - pow((double)b2, t));
  }
  __syncthreads();
  const float step_size = lr / bc[0];
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc[1]);
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < count; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < count) {
      float4 pv = *reinterpret_cast<float4*>(p + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      float4 mv = *reinterpret_cast<float4*>(m + i);
      float4 vv = *reinterpret_cast<float4*>(v + i);
      float* pp = reinterpret_cast<float*>(&pv);
      const float* gp = reinterpret_cast<const float*>(&gv);
      float* mp = reinterpret_cast<float*>(&mv);
      float* vp = reinterpret_cast<float*>(&vv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gj = gp[j] * gscale;
        mp[j] = b1 * mp[j] + (1.f - b1) * gj;
        vp[j] = b2 * vp[j] + (1.f - b2) * gj * gj;
        pp[j] -= step_size * mp[j] / (sqrtf(vp[j]) * inv_sqrt_bc2 + eps);
      }
      *reinterpret_cast<float4*>(p + i) = pv;
      *reinterpret_cast<float4*>(m + i) = mv;
      *reinterpret_cast<float4*>(v + i) = vv;
      if (shadow) {
#pragma unroll
        for (int j = 0; j < 4; ++j) shadow[i + j] = (S)pp[j];
      }
    } else {
      for (int64_t e = i; e < count; ++e) {
        const float gj = g[e] * gscale;
        m[e] = b1 * m[e] + (1.f - b1) * gj;
        v[e] = b2 * v[e] + (1.f - b2) * gj * gj;
        p[e] -= step_size * m[e] / (sqrtf(v[e]) * inv_sqrt_bc2 + eps);
        if (shadow) shadow[e] = (S)p[e];
      }
    }
  }
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    dst[i] = (D)(float)src[i];
}

// dst[ci][widx][co] = src[co][widx][ci]
template <typename S, typename D>
__global__ __launch_bounds__(256) void filter_transpose_kernel(const S* __restrict__ src, D* __restrict__ dst,
                                                               int cout, int wtaps, int cin) {
  const int64_t total = (int64_t)cout * wtaps * cin;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int co = (int)(i % cout);
    const int64_t r = i / cout;
    const int wi = (int)(r % wtaps);
    const int ci = (int)(r / wtaps);
    dst[i] = (D)(float)src[((int64_t)co * wtaps + wi) * cin + ci];
  }
}

// One launch for ALL convs: s[off + i] = cast(master[off + i]) in the forward layout [co][tap][ci] and, for layers with
// a data-gradient pass, st[off + (ci*wtaps + tap)*cout + co] = the same value.  A block owns one 32 x 32 (co, ci) tile of
// one tap of one layer (found by bisection of the table's first_block column) and transposes it through LDS, so the
// fp32 read and both bf16 writes are contiguous row segments (a thread-per-element transpose read 4-byte words
// wtaps*cin floats apart: 0.46 ms per step for 34.5 M parameters).
template <typename D>
__global__ __launch_bounds__(256) void filter_refresh_kernel(const float* __restrict__ master, D* __restrict__ s,
                                                             D* __restrict__ st, const sfk_filter_ent* __restrict__ table,
                                                             int n) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const sfk_filter_ent e = table[lo];
  const int tiles_ci = (e.cin + 31) / 32;
  int lb = b - e.first_block;
  const int tap = lb % e.wtaps;
  lb /= e.wtaps;
  const int ci0 = (lb % tiles_ci) * 32, co0 = (lb / tiles_ci) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + ty + 8 * i, ci = ci0 + tx;
    float v = 0.f;
    if (co < e.cout && ci < e.cin) {
      const int64_t idx = e.off + ((int64_t)co * e.wtaps + tap) * e.cin + ci;
      v = master[idx];
      if (s) s[idx] = (D)v;
    }
    tile[ty + 8 * i][tx] = v;
  }
  if (!st || !e.transpose) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ci = ci0 + ty + 8 * i, co = co0 + tx;
    if (co < e.cout && ci < e.cin) st[e.off + ((int64_t)ci * e.wtaps + tap) * e.cout + co] = (D)tile[tx][ty + 8 * i];
  }
}

inline unsigned grid_for(int64_t total, int per_thread = 1) {
  int64_t b = (total + 256ll * per_thread - 1) / (256ll * per_thread);
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int sfk_adam(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1,
                        float beta2, float eps, float grad_scale, int64_t* step, void* shadow,
                        int32_t shadow_dtype, sfk_stream_t stream) {
  if (!p || !g || !m || !v || !step || count <= 0) return SFK_ERR_INVALID;
  if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return SFK_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(1), 0, s, step);
  const dim3 grid(grid_for(count, 4)), blk(256);
  if (shadow && shadow_dtype == SFK_BF16)
    hipLaunchKernelGGL(adam_kernel<bf16_t>, grid, blk, 0, s, p, g, m, v, count, lr, beta1, beta2, eps, grad_scale, step, static_cast<bf16_t*>(shadow));
  else
    hipLaunchKernelGGL(adam_kernel<float>, grid, blk, 0, s, p, g, m, v, count, lr, beta1, beta2, eps, grad_scale, step, static_cast<float*>(shadow));
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}


static inline bool dtype_ok(int d) { return d == SFK_F32 || d == SFK_BF16; }

extern "C" int sfk_filter_transpose(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int32_t cout,
                                    int32_t wtaps, int32_t cin, sfk_stream_t stream) {
  if (!src || !dst || cout <= 0 || wtaps <= 0 || cin <= 0 || !dtype_ok(src_dtype) || !dtype_ok(dst_dtype))
    return SFK_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t total = (int64_t)cout * wtaps * cin;
  if (src_dtype == SFK_F32 && dst_dtype == SFK_F32)
    hipLaunchKernelGGL((filter_transpose_kernel<float, float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)src, (float*)dst, cout, wtaps, cin);
  else if (src_dtype == SFK_F32 && dst_dtype == SFK_BF16)
    hipLaunchKernelGGL((filter_transpose_kernel<float, bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, cout, wtaps, cin);
  else if (src_dtype == SFK_BF16 && dst_dtype == SFK_F32)
    hipLaunchKernelGGL((filter_transpose_kernel<bf16_t, float>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, cout, wtaps, cin);
  else
    hipLaunchKernelGGL((filter_transpose_kernel<bf16_t, bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, cout, wtaps, cin);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_filter_refresh(const float* master, void* s, void* st, int32_t dtype, const sfk_filter_ent* table,
                                  int32_t n_layers, int32_t total_blocks, sfk_stream_t stream) {
  if (!master || (!s && !st) || !table || n_layers <= 0 || total_blocks <= 0 || !dtype_ok(dtype)) return SFK_ERR_INVALID;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (dtype == SFK_BF16)
    hipLaunchKernelGGL((filter_refresh_kernel<bf16_t>), dim3((unsigned)total_blocks), dim3(256), 0, hs, master, (bf16_t*)s, (bf16_t*)st, table, n_layers);
  else
    hipLaunchKernelGGL((filter_refresh_kernel<float>), dim3((unsigned)total_blocks), dim3(256), 0, hs, master, (float*)s, (float*)st, table, n_layers);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_cast(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t count,
                        sfk_stream_t stream) {
  if (!src || !dst || count <= 0 || !dtype_ok(src_dtype) || !dtype_ok(dst_dtype)) return SFK_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (src_dtype == SFK_F32 && dst_dtype == SFK_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid_for(count, 4)), dim3(256), 0, s, (const float*)src, (float*)dst, count);
  else if (src_dtype == SFK_F32 && dst_dtype == SFK_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid_for(count, 4)), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, count);
  else if (src_dtype == SFK_BF16 && dst_dtype == SFK_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid_for(count, 4)), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, count);
  else
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(grid_for(count, 4)), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, count);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_fill_zero(void* p, size_t bytes, sfk_stream_t stream) {
  if (!p) return SFK_ERR_INVALID;
  if (bytes == 0) return SFK_OK;
  return hipMemsetAsync(p, 0, bytes, static_cast<hipStream_t>(stream)) == hipSuccess ? SFK_OK : SFK_ERR_LAUNCH;
}

// ------------------------------------------------------------------ tuning table (sfk_init)
namespace {
constexpr sfk_tuning kDefaults = {(uint32_t)sizeof(sfk_tuning), 5, 0, 1, 192, 256, 1, 7, 1024, 0, 48, 150, 1, 1ll << 20, 3, 512, 0, 48, 1, 16, 1, 3, 3};
sfk_tuning g_tuning = kDefaults;
bool g_tuning_set = false;
}  // namespace
const sfk_tuning& sfk_tune() { return g_tuning; }

// struct_size is the caller's sizeof(sfk_tuning): a binding compiled against another layout is refused before a byte moves
extern "C" int sfk_default_tuning(sfk_tuning* out) {
  if (!out || out->struct_size != sizeof(sfk_tuning)) return SFK_ERR_INVALID;
  *out = kDefaults;
  return SFK_OK;
}

extern "C" int sfk_get_tuning(sfk_tuning* out) {
  if (!out || out->struct_size != sizeof(sfk_tuning)) return SFK_ERR_INVALID;
  *out = g_tuning;
  return SFK_OK;
}

extern "C" int sfk_init(const sfk_tuning* t) {
  sfk_tuning want = kDefaults;
  if (t) {
    if (t->struct_size != sizeof(sfk_tuning)) return SFK_ERR_INVALID;
    want = *t;
  }
  if (want.bn_parts < 1 || want.wgrad_target_8w < 1 || want.wgrad_target_4w < 1 || want.pool_blocks < 1 ||
      want.igemm_short_k < 0)
    return SFK_ERR_INVALID;
  if (g_tuning_set) return memcmp(&want, &g_tuning, sizeof(want)) == 0 ? SFK_OK : SFK_ERR_INVALID;   // write-once
  g_tuning = want;
  g_tuning_set = true;
  return SFK_OK;
}

extern "C" int sfk_abi_version(void) { return SFK_ABI_VERSION; }

extern "C" const char* sfk_status_string(int status) {
  switch (status) {
    case SFK_OK: return "ok";
    case SFK_ERR_INVALID: return "invalid argument";
    case SFK_ERR_UNSUPPORTED: return "unsupported shape or alignment";
    case SFK_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown status";
  }
}
