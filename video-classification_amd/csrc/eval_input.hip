// Kernels either side of the training hot path:
//   sfk_u8_normalize_crop   the dataset's ToTensor + Normalize + RandomCrop on device, from uint8 frames
//   sfk_eval_aggregate      run_eval's softmax -> per-video mean -> argmax -> accuracy, without per-batch D2H
//   sfk_sparse_fusion_*     the late-fusion SparseModel (one Linear(num_part, 1) per class)
// All of them are small HBM-/latency-bound byte movers: coalesced 16-byte accesses, one pass, no atomics except the
// single correct-count word.
#include "sfk_common.h"

namespace {

// ------------------------------------------------------------------ uint8 HWC frames -> normalised CHW clip, cropped
// out[n][t][ch][y][x] = lut[ src[n][t][y + oy - pad][x + ox - pad][ch] ]   (0 outside the frame: RandomCrop pads the
// NORMALISED tensor with zeros), (oy, ox) = crop[n] or (pad, pad) when crop == NULL (no augmentation).
// A block stages one source row segment (all channels) through LDS so that both the HWC read and the CHW write are
// contiguous; 8 output pixels per thread and channel.
template <typename T>
__global__ __launch_bounds__(256) void u8_normalize_crop_kernel(const uint8_t* __restrict__ src, const float* __restrict__ lut,
                                                                const int32_t* __restrict__ crop, int pad, T* __restrict__ out,
                                                                int n, int t, int c, int h, int w) {
  extern __shared__ __attribute__((aligned(16))) uint8_t row[];       // [w][c] source bytes of one output row
  __shared__ float s_lut[256];
  s_lut[threadIdx.x] = lut[threadIdx.x];
  const int y = blockIdx.x % h;
  const int nt = blockIdx.x / h;            // n*t + frame
  const int ni = nt / t;
  const int oy = crop ? crop[2 * ni] : pad, ox = crop ? crop[2 * ni + 1] : pad;
  const int sy = y + oy - pad;
  const bool row_ok = sy >= 0 && sy < h;
  const int64_t rbytes = (int64_t)w * c;
  if (row_ok) {
    const uint8_t* sp = src + ((int64_t)nt * h + sy) * rbytes;
    for (int64_t i = (int64_t)threadIdx.x * 16; i < rbytes; i += 256 * 16) {
      if (i + 16 <= rbytes && ((reinterpret_cast<uintptr_t>(sp + i) & 15) == 0)) {
        *reinterpret_cast<uint4*>(row + i) = *reinterpret_cast<const uint4*>(sp + i);
      } else {
        for (int64_t j = i; j < i + 16 && j < rbytes; ++j) row[j] = sp[j];
      }
    }
  }
  __syncthreads();
  // (channel, x) pairs of this output row; x fastest so the stores of a channel plane are contiguous
  for (int e = threadIdx.x; e < c * w; e += 256) {
    const int ch = e / w, x = e % w;
    const int sx = x + ox - pad;
    float v = 0.f;
    if (row_ok && sx >= 0 && sx < w) v = s_lut[row[sx * c + ch]];
    out[((((int64_t)nt * c + ch) * h + y) * w) + x] = (T)v;
  }
}

// ------------------------------------------------------------------ eval aggregation
// one block per video: ps = softmax(logits) per clip (optional, written out), mean over the video's clips, argmax
// (first maximum, as numpy), compared with the label of its first clip.
__global__ __launch_bounds__(256) void eval_aggregate_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                             const int32_t* __restrict__ seg_off, int c, int softmax,
                                                             float* __restrict__ ps_out, int32_t* __restrict__ pred,
                                                             int32_t* __restrict__ correct) {
  extern __shared__ float mean[];            // [c]
  __shared__ float s_red[256];
  __shared__ int s_idx[256];
  const int v = blockIdx.x;
  const int r0 = seg_off[v], r1 = seg_off[v + 1];
  for (int j = threadIdx.x; j < c; j += 256) mean[j] = 0.f;
  __syncthreads();
  for (int r = r0; r < r1; ++r) {
    const float* lp = logits + (int64_t)r * c;
    float inv = 1.f, mx = 0.f;
    if (softmax) {
      // exp(x - max) / sum: the reference's np.exp(x) / np.exp(x).sum() without its overflow
      float m = -INFINITY;
      for (int j = threadIdx.x; j < c; j += 256) m = fmaxf(m, lp[j]);
      s_red[threadIdx.x] = m;
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) s_red[threadIdx.x] = fmaxf(s_red[threadIdx.x], s_red[threadIdx.x + s]);
        __syncthreads();
      }
      mx = s_red[0];
      __syncthreads();
      float sum = 0.f;
      for (int j = threadIdx.x; j < c; j += 256) sum += expf(lp[j] - mx);
      s_red[threadIdx.x] = sum;
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) s_red[threadIdx.x] += s_red[threadIdx.x + s];
        __syncthreads();
      }
      inv = 1.f / s_red[0];
      __syncthreads();
    }
    for (int j = threadIdx.x; j < c; j += 256) {
      const float p = softmax ? expf(lp[j] - mx) * inv : lp[j];
      if (ps_out) ps_out[(int64_t)r * c + j] = p;
      mean[j] += p;
    }
  }
  __syncthreads();
  if (r1 <= r0) {                             // an empty video is skipped by the reference loop
    if (threadIdx.x == 0) pred[v] = -1;
    return;
  }
  const float cnt = (float)(r1 - r0);
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = threadIdx.x; j < c; j += 256) {
    const float m = mean[j] / cnt;
    if (m > best) { best = m; bi = j; }       // ascending j per thread: keeps the first maximum
  }
  s_red[threadIdx.x] = best;
  s_idx[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float o = s_red[threadIdx.x + s];
      const int oi = s_idx[threadIdx.x + s];
      if (o > s_red[threadIdx.x] || (o == s_red[threadIdx.x] && oi < s_idx[threadIdx.x])) {
        s_red[threadIdx.x] = o;
        s_idx[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pred[v] = s_idx[0];
    if (correct && (int64_t)s_idx[0] == labels[r0]) atomicAdd(correct, 1);
  }
}

// ------------------------------------------------------------------ SparseModel
// y[n][k] = b[k] + sum_p w[k][p] * x[n][p][k]
__global__ __launch_bounds__(256) void sparse_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int n, int p,
                                                         int c) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)n * c) return;
  const int ni = (int)(i / c), k = (int)(i % c);
  float acc = b[k];
  for (int q = 0; q < p; ++q) acc += w[k * p + q] * x[((int64_t)ni * p + q) * c + k];
  y[i] = acc;
}
// dw[k][q] += sum_n dy[n][k] * x[n][q][k] ; db[k] += sum_n dy[n][k]        one wave per (k, q | bias)
__global__ __launch_bounds__(256) void sparse_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dw, float* __restrict__ db, int n, int p, int c) {
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (item >= c * (p + 1)) return;
  const int k = item / (p + 1), q = item % (p + 1);
  float acc = 0.f;
  for (int ni = lane; ni < n; ni += 64) {
    const float g = dy[(int64_t)ni * c + k];
    acc += q < p ? g * x[((int64_t)ni * p + q) * c + k] : g;
  }
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) acc += __shfl_xor(acc, s);
  if (lane == 0) {
    if (q < p) dw[k * p + q] += acc;
    else db[k] += acc;
  }
}

}  // namespace

extern "C" int sfk_u8_normalize_crop(const uint8_t* src, const float* lut, const int32_t* crop, int32_t pad, void* out,
                                     int32_t out_dtype, int32_t n, int32_t t, int32_t c, int32_t h, int32_t w,
                                     sfk_stream_t stream) {
  if (!src || !lut || !out || n <= 0 || t <= 0 || c <= 0 || h <= 0 || w <= 0 || pad < 0) return SFK_ERR_INVALID;
  if (out_dtype != SFK_F32 && out_dtype != SFK_BF16) return SFK_ERR_INVALID;
  const size_t lds = (((size_t)w * c) + 15) / 16 * 16;
  if (lds > 60 * 1024 || (int64_t)n * t * h >= (1ll << 31)) return SFK_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(n * t * h));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (out_dtype == SFK_BF16)
    hipLaunchKernelGGL(u8_normalize_crop_kernel<bf16_t>, grid, dim3(256), lds, s, src, lut, crop, pad, (bf16_t*)out, n, t, c, h, w);
  else
    hipLaunchKernelGGL(u8_normalize_crop_kernel<float>, grid, dim3(256), lds, s, src, lut, crop, pad, (float*)out, n, t, c, h, w);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_eval_aggregate(const float* logits, const int64_t* labels, const int32_t* seg_off, int32_t nvideos,
                                  int32_t c, int32_t softmax, float* ps_out, int32_t* pred, int32_t* correct,
                                  sfk_stream_t stream) {
  if (!logits || !labels || !seg_off || !pred || nvideos <= 0 || c <= 0) return SFK_ERR_INVALID;
  if ((size_t)c * 4 > 48 * 1024) return SFK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(eval_aggregate_kernel, dim3((unsigned)nvideos), dim3(256), (size_t)c * 4,
                     static_cast<hipStream_t>(stream), logits, labels, seg_off, c, softmax, ps_out, pred, correct);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_sparse_fusion_fwd(const float* x, const float* w, const float* b, float* y, int32_t n, int32_t p,
                                     int32_t c, sfk_stream_t stream) {
  if (!x || !w || !b || !y || n <= 0 || p <= 0 || c <= 0) return SFK_ERR_INVALID;
  hipLaunchKernelGGL(sparse_fwd_kernel, dim3((unsigned)(((int64_t)n * c + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, w, b, y, n, p, c);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_sparse_fusion_bwd(const float* x, const float* dy, float* dw, float* db, int32_t n, int32_t p, int32_t c,
                                     sfk_stream_t stream) {
  if (!x || !dy || !dw || !db || n <= 0 || p <= 0 || c <= 0) return SFK_ERR_INVALID;
  hipLaunchKernelGGL(sparse_bwd_kernel, dim3((unsigned)((c * (p + 1) + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, dy, dw, db, n, p, c);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}
