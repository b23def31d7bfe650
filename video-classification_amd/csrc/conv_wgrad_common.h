// Shared pieces of the filter-gradient kernels (conv_wgrad.hip, conv_wgrad_p8.hip): the kernel argument block, the XCD-aware
// block order, and the second pass of the workspace path (ordered sum of the pixel splits' partial tiles into dW).
#pragma once
#include "sfk_common.h"
#include <stdlib.h>

namespace sfk_wgrad {


struct WgradK {
  const void* x;
  const void* dy;
  float* dw;
  int xt, xh, xw, xld, xoff;
  int dld, doff;
  int M;
  FastDiv drw, drh, drt;
  int gst, gsh, gsw;
  int cin, cout, wtaps, ntaps;
  int citiles;                    // column tiles over the flattened (tap, cin) axis
  int chunks_per_split, nchunks;  // in stages of KS*32 pixels
  FastDiv dspt, dcin;             // 16-byte segments per tap; channels per tap
  uint32_t xbytes, dbytes;        // extents of the buffer resources
  float4* ws;                     // partial-tile workspace (NULL: fp32 atomics straight into dw)
  int ntiles;                     // cotiles * citiles
  const void* dgw;                // fused data gradient (DG kernels): [cin][cout] matrix, output map
  void* dgy;
  int dgld, dgoff;
  sfk_tap taps[SFK_MAX_TAPS];
};

constexpr int MK = 32;  // pixels per K-step

// XCD-aware block order (as conv_igemm): blocks b, b+8, ... share an L2, so consecutive LOGICAL ids go to one XCD and the
// tiles of one pixel split (which read the same dY rows and overlapping X rows) are neighbours there.
__device__ __forceinline__ void wg_block(int ntiles, int& tile, int& split) {
  const int nblk = gridDim.x, b = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = b & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  tile = logical % ntiles;
  split = logical / ntiles;
}


// Second pass of the workspace path: dw[co][widx][ci] += sum over pixel splits of the partial tiles, in split order
// (deterministic; fp32 atomics moved ~1.3 TB/s chip-wide and every split re-adds the whole tile).
// NWV waves per block laid out as WCO x (NWV/WCO), FO x FI accumulator fragments per wave.
// A block owns 256 / ZG consecutive float4 of the tile image and ZG split groups: group zg adds splits zg, zg + ZG, ... in
// order, then the ZG group sums are added in order -- the summation tree is fixed by (splits, ZG), whatever the launch does.
// ZG follows the split count (launch_reduce): with 16 groups and 5 splits (res5: 48 tiles x 5) eleven of sixteen threads had
// nothing to read and the pass took 39 us for 74 MB.
template <int NWV, int WCO, int FO, int FI, int ZG>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradK k, int splits) {
  constexpr int PER_TILE = NWV * FO * FI * 64;          // float4 per tile
  constexpr int TCO = WCO * 16 * FO, TCI = (NWV / WCO) * 16 * FI;
  constexpr int E = 256 / ZG;
  __shared__ float4 red[ZG][E];
  const int le = threadIdx.x % E, zg = threadIdx.x / E;
  const int64_t idx = (int64_t)blockIdx.x * E + le;
  const int64_t total = (int64_t)k.ntiles * PER_TILE;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < total) {
    const float4* p = k.ws + idx;
    int z = zg;
    for (; z + 3 * ZG < splits; z += 4 * ZG) {          // four loads in flight, added in order
      const float4 v0 = p[(int64_t)z * total], v1 = p[(int64_t)(z + ZG) * total], v2 = p[(int64_t)(z + 2 * ZG) * total],
                   v3 = p[(int64_t)(z + 3 * ZG) * total];
      sum.x += v0.x; sum.y += v0.y; sum.z += v0.z; sum.w += v0.w;
      sum.x += v1.x; sum.y += v1.y; sum.z += v1.z; sum.w += v1.w;
      sum.x += v2.x; sum.y += v2.y; sum.z += v2.z; sum.w += v2.w;
      sum.x += v3.x; sum.y += v3.y; sum.z += v3.z; sum.w += v3.w;
    }
    for (; z < splits; z += ZG) {
      const float4 v = p[(int64_t)z * total];
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
  }
  if constexpr (ZG > 1) {
    red[zg][le] = sum;
    __syncthreads();
    if (zg != 0) return;
#pragma unroll
    for (int z = 1; z < ZG; ++z) {
      const float4 v = red[z][le];
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
  }
  if (idx >= total) return;
  const int tile = (int)(idx / PER_TILE), e = (int)(idx % PER_TILE);
  const int lane = e & 63, frag = (e >> 6) % (FO * FI), wave = (e >> 6) / (FO * FI);
  const int i = frag / FI, j = frag % FI, wco = wave % WCO, wci = wave / WCO;
  const int cot = tile / k.citiles, cit = tile % k.citiles;
  const int col = cit * TCI + wci * 16 * FI + 16 * j + (lane & 15);
  uint32_t tap, ci;
  k.dcin.divmod((uint32_t)col, tap, ci);
  if (tap >= (uint32_t)k.ntaps) return;
  const int widx = k.taps[tap].widx;
  const int co0 = cot * TCO + wco * 16 * FO + 16 * i + 4 * (lane >> 4);
  const float v4[4] = {sum.x, sum.y, sum.z, sum.w};
  float* dp = k.dw + ((int64_t)co0 * k.wtaps + widx) * k.cin + ci;
  const int64_t rs = (int64_t)k.wtaps * k.cin;
  float old[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) old[r] = co0 + r < k.cout ? dp[r * rs] : 0.f;     // all four in flight before the first store
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (co0 + r < k.cout) dp[r * rs] = old[r] + v4[r];
}

template <int NWV, int WCO, int FO, int FI>
int launch_reduce(const WgradK& k, int splits, hipStream_t s) {
  constexpr int PER_TILE = NWV * FO * FI * 64;
  const int64_t total = (int64_t)k.ntiles * PER_TILE;
  if (splits >= 32)
    hipLaunchKernelGGL((wgrad_reduce_kernel<NWV, WCO, FO, FI, 16>), dim3((unsigned)((total + 15) / 16)), dim3(256), 0, s, k, splits);
  else if (splits >= 8)
    hipLaunchKernelGGL((wgrad_reduce_kernel<NWV, WCO, FO, FI, 4>), dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, k, splits);
  else
    hipLaunchKernelGGL((wgrad_reduce_kernel<NWV, WCO, FO, FI, 1>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k, splits);
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

}  // namespace sfk_wgrad
