// sfk_stem_conv_fwd / sfk_stem_conv_wgrad: the (kt,7,7) stride-(1,2,2) stem convolutions, directly from the clip.
//
// Why not the implicit GEMM: with cin = 3 and 49 (x5) window taps every output pixel would pull K*2 B = 1.5 KB out of
// L2 (19 GB per step for the fast pathway).  Here a block stages the INPUT PATCH of a 16x16 output tile once in LDS
// ((2*16+5) rows x 40 cols per (frame, channel) plane), and all MFMA operands are built from it:
//   K order = ((f*cin + ci)*7 + kh)*8 + kw  (kw padded 7 -> 8, filter zero there): the 8 consecutive k of one lane
//   are 8 consecutive input columns 2*wo .. 2*wo+7 of one patch row -> one 16-byte (4-byte aligned) LDS run.
//   forward : A = filter rows (co), B = patch runs (pixel on the lane)        -> D[co][pixel], epilogue = conv_igemm's
//   wgrad   : pixels are the K dim: A = dY^T via ds_read_b64_tr_b16, B = 8 pixels (stride-2 columns) of one (kh,kw)
// The clip is read in place through element strides (N,T,C,H,W dataset memory or N,C,T,H,W) with an optional frame
// index (PackPathway), as f32 or bf16.
#include "sfk_common.h"

namespace {

constexpr int TS = 16;            // output tile edge (pixels)
constexpr int PR = 2 * TS + 5;    // patch rows  (37)
constexpr int PC = 40;            // patch cols  (2*15 + 7 + 1 = 38 used, padded)
constexpr int KH = 7;

struct StemK {
  const void* src;
  int64_t sn, sc, st, sh, sw;
  int cin, t_in, h_in, w_in;
  const int32_t* t_index;
  int t_log;      // logical clip length (after frame selection)
  int kt, pt;
  int krows;      // kt*cin*7
  int kp;         // padded K = roundup4(krows)*8
  int planes;     // kt*cin
  int cout;
  int ho, wo, t_out;
  int tiles_h, tiles_w;
  FastDiv dtw, dth, dt, d7, dcin, dpc;
  const void* w;  // [cout][kp]
  void* y;        // forward output / wgrad dY
  int yld, yoff;
  float* stats;
  float* dw;
  int tiles_per_block, ntiles;
  uint32_t src_bytes, y_bytes;   // extents for the buffer resources of the v2 kernels (fit 32 bits, checked at launch)
};

template <typename S> __device__ __forceinline__ float ldsrc(const void* p, int64_t off);
template <> __device__ __forceinline__ float ldsrc<float>(const void* p, int64_t off) { return static_cast<const float*>(p)[off]; }
template <> __device__ __forceinline__ float ldsrc<bf16_t>(const void* p, int64_t off) { return (float)static_cast<const bf16_t*>(p)[off]; }

// The patch of one (frame, channel) plane is PR*PC = 1480 elements = NSLOT slots per thread.  A thread's slots have the
// same (row, col) for every plane and tile, so their source offsets are computed once per tile and the loads of a plane
// (or several planes) are issued back to back -- the loader is latency-bound otherwise.
constexpr int NSLOT = (PR * PC + 255) / 256;   // 6

struct PatchSlots {
  int64_t off[NSLOT];   // hi*sh + wi*sw of the current tile
  bool ok[NSLOT];
  __device__ __forceinline__ void set_tile(const StemK& k, int ho0, int wo0) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int e = threadIdx.x + 256 * i;
      uint32_t r, c;
      k.dpc.divmod((uint32_t)e, r, c);
      const int hi = 2 * ho0 - 3 + (int)r, wi = 2 * wo0 - 3 + (int)c;
      ok[i] = e < PR * PC && (unsigned)hi < (unsigned)k.h_in && (unsigned)wi < (unsigned)k.w_in;
      off[i] = (int64_t)hi * k.sh + (int64_t)wi * k.sw;
    }
  }
};

// source base offset of plane pl for output frame `to` of clip n; <0 when the frame is temporal padding
__device__ __forceinline__ int64_t plane_base(const StemK& k, int pl, int n, int to) {
  uint32_t f, ci;
  k.dcin.divmod((uint32_t)pl, f, ci);
  const int tt = to + (int)f - k.pt;
  int frame = -1;
  if (tt >= 0 && tt < k.t_log) frame = k.t_index ? k.t_index[tt] : tt;
  if (frame < 0 || frame >= k.t_in) return -1;
  return (int64_t)n * k.sn + (int64_t)ci * k.sc + (int64_t)frame * k.st;
}

template <typename S>
__device__ __forceinline__ void plane_fetch(const StemK& k, const PatchSlots& ps, int64_t base, float (&v)[NSLOT]) {
  // branch-free: padding slots read element 0 of the plane and are zeroed by a select (a load under a branch makes
  // hipcc drain vmcnt behind it -- NSLOT serial round trips per plane instead of one)
  const bool bok = base >= 0;
  const int64_t b = bok ? base : 0;
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const float x = ldsrc<S>(k.src, b + (ps.ok[i] ? ps.off[i] : 0));
    v[i] = (bok && ps.ok[i]) ? x : 0.f;
  }
}

template <typename T>
__device__ __forceinline__ void plane_store(T* patch, const float (&v)[NSLOT]) {
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int e = threadIdx.x + 256 * i;
    if (e < PR * PC) patch[e] = (T)v[i];
  }
}

__device__ __forceinline__ void tile_coords(const StemK& k, int tile, int& n, int& to, int& ho0, int& wo0) {
  uint32_t q1, tw, q2, th, n_, t_;
  k.dtw.divmod((uint32_t)tile, q1, tw);
  k.dth.divmod(q1, q2, th);
  k.dt.divmod(q2, n_, t_);
  n = (int)n_; to = (int)t_; ho0 = (int)th * TS; wo0 = (int)tw * TS;
}

// ------------------------------------------------------------------------------------------ forward
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  typedef bf16x8 ab;
  static __device__ __forceinline__ ab run8(const bf16_t* p) {   // 8 consecutive elements, 4-byte aligned
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
    uint4 v = make_uint4(q[0], q[1], q[2], q[3]);
    return *reinterpret_cast<ab*>(&v);
  }
  static __device__ __forceinline__ ab ld16(const bf16_t* p) { return *reinterpret_cast<const ab*>(p); }
  static __device__ __forceinline__ void mma(f32x4& acc, const ab& a, const ab& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
};
struct F8 { float v[8]; };
template <> struct Frag<float> {
  typedef F8 ab;
  static __device__ __forceinline__ ab run8(const float* p) {    // 8 consecutive floats, 8-byte aligned
    ab r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 t = *reinterpret_cast<const float2*>(p + 2 * i);
      r.v[2 * i] = t.x; r.v[2 * i + 1] = t.y;
    }
    return r;
  }
  static __device__ __forceinline__ ab ld16(const float* p) { return run8(p); }
  static __device__ __forceinline__ void mma(f32x4& acc, const ab& a, const ab& b) {
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[s], b.v[s], acc, 0, 0, 0);
  }
};

__device__ __forceinline__ void store4(float* p, const f32x4& v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void store4(bf16_t* p, const f32x4& v) {
  bf16x4 o;
  o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
  *reinterpret_cast<bf16x4*>(p) = o;
}

// FN = co fragments (cout <= 16*FN).  A block walks tiles blockIdx.x, +gridDim.x, ... (filters are staged once);
// one tile = 16x16 output pixels, wave w owns output rows 4w..4w+3.
template <typename T, typename S, int FN>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const StemK k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* wl = reinterpret_cast<T*>(smem);                         // [16*FN][kp + 8]
  const int wrow = k.kp + 8;                                  // +16 B: spreads ds_read_b128 rows over banks
  T* patch = wl + 16 * FN * wrow;                             // [planes][PR][PC]
  float* red = reinterpret_cast<float*>(patch + k.planes * PR * PC);   // [4 waves][16*FN][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;

  // filters -> LDS (rows >= cout are zero)
  const T* wp = static_cast<const T*>(k.w);
  const int segs = k.kp / 8;
#pragma unroll 4
  for (int e = tid; e < 16 * FN * segs; e += 256) {
    const int row = e / segs, seg = e % segs;
    typename Frag<T>::ab v;
    if (row < k.cout) v = Frag<T>::ld16(wp + (int64_t)row * k.kp + seg * 8);
    else {
      for (int i = 0; i < 8; ++i) reinterpret_cast<T*>(&v)[i] = (T)0.f;
    }
    *reinterpret_cast<typename Frag<T>::ab*>(wl + row * wrow + seg * 8) = v;
  }
  T* yp = static_cast<T*>(k.y);
  const int ksteps = k.kp / 32;
  PatchSlots ps;

  for (int tile = blockIdx.x; tile < k.ntiles; tile += gridDim.x) {
    int n, to, ho0, wo0;
    tile_coords(k, tile, n, to, ho0, wo0);
    ps.set_tile(k, ho0, wo0);
    __syncthreads();                                          // previous tile's patch / red reads are done
    int pl = 0;
    for (; pl + 3 <= k.planes; pl += 3) {                     // 18 loads in flight per thread
      float v0[NSLOT], v1[NSLOT], v2[NSLOT];
      // the three frame lookups first (one wait), then all 18 element loads back to back
      const int64_t b0 = plane_base(k, pl, n, to), b1 = plane_base(k, pl + 1, n, to), b2 = plane_base(k, pl + 2, n, to);
      plane_fetch<S>(k, ps, b0, v0);
      plane_fetch<S>(k, ps, b1, v1);
      plane_fetch<S>(k, ps, b2, v2);
      plane_store<T>(patch + pl * PR * PC, v0);
      plane_store<T>(patch + (pl + 1) * PR * PC, v1);
      plane_store<T>(patch + (pl + 2) * PR * PC, v2);
    }
    for (; pl < k.planes; ++pl) {
      float v0[NSLOT];
      plane_fetch<S>(k, ps, plane_base(k, pl, n, to), v0);
      plane_store<T>(patch + pl * PR * PC, v0);
    }
    __syncthreads();

    f32x4 acc[FN][4];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int s = 0; s < ksteps; ++s) {
      int rk = 4 * s + g;                      // K row of this lane group: (plane, kh)
      if (rk >= k.krows) rk = k.krows - 1;     // padded rows multiply zero filter columns; keep the read in bounds
      uint32_t pln, kh;
      k.d7.divmod((uint32_t)rk, pln, kh);
      const T* prow = patch + (int)pln * PR * PC + (int)kh * PC + 2 * l15;
      typename Frag<T>::ab a[FN], b[4];
#pragma unroll
      for (int i = 0; i < FN; ++i) a[i] = Frag<T>::ld16(wl + (16 * i + l15) * wrow + 32 * s + 8 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Frag<T>::run8(prow + 2 * (4 * wave + j) * PC);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Frag<T>::mma(acc[i][j], a[i], b[j]);
    }

    // epilogue: lane holds co = 16i + 4g + r for pixel (row 4*wave + j, col l15)
    const int wo = wo0 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ho = ho0 + 4 * wave + j;
      const bool pok = ho < k.ho && wo < k.wo;
      if (!pok) {
#pragma unroll
        for (int i = 0; i < FN; ++i) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};   // outside the map: not stored, not counted
        continue;
      }
      const int64_t poff = ((((int64_t)n * k.t_out + to) * k.ho + ho) * k.wo + wo) * k.yld + k.yoff;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int co = 16 * i + 4 * g;
        if (co < k.cout) store4(yp + poff + co, acc[i][j]);
      }
    }
    if (k.stats) {
#pragma unroll
      for (int i = 0; i < FN; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = acc[i][j][r];
            s1 += v;
            s2 += v * v;
          }
#pragma unroll
          for (int sft = 1; sft < 16; sft <<= 1) {
            s1 += __shfl_xor(s1, sft);
            s2 += __shfl_xor(s2, sft);
          }
          if (l15 == 0) {
            const int col = 16 * i + 4 * g + r;
            red[(wave * 16 * FN + col) * 2 + 0] = s1;
            red[(wave * 16 * FN + col) * 2 + 1] = s2;
          }
        }
      }
      __syncthreads();
      if (tid < k.cout) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < 4; ++w_) {
          s1 += red[(w_ * 16 * FN + tid) * 2 + 0];
          s2 += red[(w_ * 16 * FN + tid) * 2 + 1];
        }
        float* o = k.stats + ((int64_t)tile * k.cout + tid) * 2;
        o[0] = s1;
        o[1] = s2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ filter gradient
// grid = (plane, tile split).  A block keeps dW[cout][plane][7][8] in registers over its range of tiles: wave w owns
// column fragment w = filter rows kh = 2w, 2w+1 (x 8 kw).  Per tile: dY tile [256 pixels][16*FN] and the plane's patch
// are staged; K-step s = output rows 2s, 2s+1 (32 pixels).
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <typename T, typename S, int FN>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const StemK k) {
  constexpr int DC = 16 * FN;                  // dY tile columns
  constexpr int DROW = DC + 8;                 // elements per dY tile row (+16 B pad for bf16)
  constexpr int VEC = DT<T>::VEC;
  constexpr int DSEG = DC / VEC;               // 16-byte segments per dY row
  constexpr int NDL = DSEG;                    // 256 rows * DSEG segments / 256 threads
  __shared__ __attribute__((aligned(16))) T dyt[256 * DROW];
  __shared__ __attribute__((aligned(16))) T patch[PR * PC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int pl = blockIdx.x;
  const int tile0 = blockIdx.y * k.tiles_per_block;
  const int tile1 = min(tile0 + k.tiles_per_block, k.ntiles);
  if (tile0 >= tile1) return;
  const T* dyp = static_cast<const T*>(k.y);

  f32x4 acc[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int khh = l15 >> 3, kw = l15 & 7;
  const int kh = 2 * wave + khh;                 // kh == 7 (wave 3, khh 1) is padding: its column is never stored

  PatchSlots ps;
  float pv[NSLOT];
  uint4 dv[NDL];
  // dY tile: pixel p = th*16 + tw -> row p, channels [0, DC) (zero outside the map / beyond cout)
  auto fetch = [&](int tile) {
    int n, to, ho0, wo0;
    tile_coords(k, tile, n, to, ho0, wo0);
    ps.set_tile(k, ho0, wo0);
    plane_fetch<S>(k, ps, plane_base(k, pl, n, to), pv);
#pragma unroll
    for (int i = 0; i < NDL; ++i) {
      const int e = tid + 256 * i;
      const int p = e / DSEG, seg = e % DSEG;
      const int ho = ho0 + (p >> 4), wo = wo0 + (p & 15);
      // branch-free (see plane_fetch): slots outside the map read the first 16 bytes of dY and are zeroed by a select
      const bool ok = ho < k.ho && wo < k.wo && seg * VEC < k.cout;
      const int64_t off = ok ? ((((int64_t)n * k.t_out + to) * k.ho + ho) * k.wo + wo) * k.yld + k.yoff + seg * VEC : 0;
      const uint4 x = *reinterpret_cast<const uint4*>(dyp + off);
      dv[i] = ok ? x : make_uint4(0, 0, 0, 0);
    }
  };
  auto stage = [&]() {
    plane_store<T>(patch, pv);
#pragma unroll
    for (int i = 0; i < NDL; ++i) {
      const int e = tid + 256 * i;
      *reinterpret_cast<uint4*>(dyt + (e / DSEG) * DROW + (e % DSEG) * VEC) = dv[i];
    }
  };

  fetch(tile0);
  for (int tile = tile0; tile < tile1; ++tile) {
    __syncthreads();                              // previous tile's LDS reads are done
    stage();
    __syncthreads();
    if (tile + 1 < tile1) fetch(tile + 1);        // next tile's loads fly while this tile runs on the matrix cores
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
      if constexpr (sizeof(T) == 2) {
        // A = dY^T: 8 consecutive pixels (k = 8g + e) of channel 16i + l15 via the transpose read
        const int q = l15 >> 2, p4 = lane & 3;
        bf16x8 b;
        // B: pixels (row 2s + (g>>1), tw = (g&1)*8 + e), column (kh, kw): patch[2*th + kh][2*tw + kw]
        const bf16_t* pr = patch + (2 * (2 * s + (g >> 1)) + kh) * PC + 2 * ((g & 1) * 8) + kw;
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = (kh < KH) ? pr[2 * e] : (bf16_t)0.f;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          const bf16_t* ap = dyt + (32 * s + 8 * g + q) * DROW + 16 * i + 4 * p4;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(ap));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(ap + 4 * DROW));
          bf16x8 a;
          a[0] = lo[0]; a[1] = lo[1]; a[2] = lo[2]; a[3] = lo[3];
          a[4] = hi[0]; a[5] = hi[1]; a[6] = hi[2]; a[7] = hi[3];
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
      } else {
        // f32: MFMA sub-step e takes pixel 4e + g of the 32
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int p = 32 * s + 4 * e + g;
          const int th = p >> 4, tw = p & 15;
          const float bv = (kh < KH) ? (float)patch[(2 * th + kh) * PC + 2 * tw + kw] : 0.f;
#pragma unroll
          for (int i = 0; i < FN; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)dyt[p * DROW + 16 * i + l15], bv, acc[i], 0, 0, 0);
        }
      }
    }
  }
  // D[row = co][col = (khh, kw)]
  if (kh < KH && kw < 7) {   // kw == 7 and kh == 7 are padding of the stem layout: never written
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 16 * i + 4 * g + r;
        if (co < k.cout) atomicAdd(k.dw + (int64_t)co * k.kp + (pl * KH + kh) * 8 + kw, acc[i][r]);
      }
  }
}

// ------------------------------------------------------------------------------------------ forward, v2
// Canonical fast stem (kt = 5 or 3, cin = 3, cout <= 8), bf16, 16-byte addressable rows.  Two changes of structure
// against stem_fwd_kernel:
//   * temporal PAIRS fill the MFMA: M = 16 rows = (jt, co) for output frames to0, to0+1; the K axis walks the
//     NF = kt+1 input frames both outputs see (row (jt, co) carries filter tap f = f' - jt, zero outside 0..kt-1),
//     so the 8-channel stem uses all 16 rows instead of 8;
//   * a block keeps a ROLLING ring of NF input frames of its 16x16 output tile in LDS and walks the clip in time: a
//     pair loads 2 new frames (16-byte loads) instead of re-staging kt frames per output frame element by element.
// K order inside a frame: (ci, kh padded to 8, k' = kw + 1) = 24 rows of 8 = 6 MFMA chunks per frame, so the ring
// rotation is a wave-uniform choice of the A chunk block (scalar arithmetic), and the kh rows of a chunk are taken in
// the order {0,2,1,3} so the two 32-lane halves of a ds_read_b32 hit disjoint banks (row pitch 96 B).
constexpr int F2_PITCH = 96;                 // patch row pitch, bytes (48 columns from wi = 2*wo0 - 8)
constexpr int F2_PR = 38;                    // patch rows (37 + the row the padded kh = 7 reads)
constexpr int F2_PLANE = F2_PR * F2_PITCH;   // 3648 B

template <int CIN, int KT>
__global__ __launch_bounds__(512, 2) void stem_fwd_v2_kernel(const StemK k, int pairs_per_unit, int nunits) {
  constexpr int NF = KT + 1, PT = KT / 2;
  constexpr int SLOT = CIN * F2_PLANE;
  constexpr int CPF = CIN * 2;                                   // MFMA chunks per frame
  constexpr int NCH = NF * CPF;                                  // chunks of the A matrix
  constexpr int FRAME_CHUNKS = CIN * F2_PR * 6;                  // 16-byte chunks of one frame patch
  constexpr int NXC = (2 * FRAME_CHUNKS + 511) / 512;            // per thread, two frames
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                             // [NF][CIN][38][96 B]
  char* amat = smem + NF * SLOT;                                 // [NCH][16 rows][64 B], 16-B slots XOR-swizzled
  float* red = reinterpret_cast<float*>(amat + NCH * 1024);      // [8 waves][16 rows][2]
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bf16_t* src = static_cast<const bf16_t*>(k.src);
  bf16_t* yp = static_cast<bf16_t*>(k.y);
  auto a_off = [](int r, int sg) { return r * 64 + ((sg ^ ((4 - ((r >> 2) & 3)) & 3)) << 4); };
  const int perm_g = ((g & 1) << 1) | (g >> 1);                  // {0,2,1,3}

  // ---- A matrix: amat[chunk f'*CPF + c6][row (jt,co)][k = 8g + k'] = w[co][f'-jt][ci = c6>>1][kh][k'-1]
  {
    const bf16_t* wp = static_cast<const bf16_t*>(k.w);
    for (int e = tid; e < NCH * 64; e += 512) {
      const int ch = e >> 6, row = (e >> 2) & 15, gg = e & 3;
      const int fp = ch / CPF, c6 = ch % CPF;
      const int jt = row >> 3, co = row & 7, f = fp - jt;
      const int kh = (c6 & 1) * 4 + (((gg & 1) << 1) | (gg >> 1)), ci = c6 >> 1;
      bf16x8 v;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.f;
      if (f >= 0 && f < KT && co < k.cout && kh < KH) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wp + (int64_t)co * k.kp + ((f * CIN + ci) * KH + kh) * 8);
#pragma unroll
        for (int i = 1; i < 8; ++i) v[i] = wv[i - 1];
      }
      *reinterpret_cast<bf16x8*>(amat + ch * 1024 + a_off(row, gg)) = v;
    }
  }

  // ---- staging slots of this thread within a two-frame load: (which of the two frames, ci, row, 16-B chunk)
  int x_fr[NXC], x_loc[NXC], x_r[NXC], x_j[NXC];
#pragma unroll
  for (int i = 0; i < NXC; ++i) {
    const int e = tid + 512 * i;
    x_fr[i] = e < 2 * FRAME_CHUNKS ? e / FRAME_CHUNKS : -1;
    const int rem = e % FRAME_CHUNKS;
    const int ci = rem / (F2_PR * 6), rr = rem % (F2_PR * 6);
    x_r[i] = rr / 6;
    x_j[i] = rr % 6;
    x_loc[i] = ci * F2_PLANE + x_r[i] * F2_PITCH + x_j[i] * 16;
    x_j[i] |= ci << 8;                                           // pack ci
  }
  uint4 xr[NXC];
  int n = 0, ho0 = 0, wo0 = 0;
  // frames F0, F0+1 (logical, may lie outside the clip = temporal zero padding)
  // branch-free: a chunk outside the clip gets an out-of-range offset and the buffer load returns zeros.  With a branch
  // around every load (`if (inside) x = load`) hipcc waited vmcnt(0) after each one: three serial round trips per pair
  // in front of the MFMAs instead of one batch behind them.
  const __amdgpu_buffer_rsrc_t srs = sfk_make_rsrc(k.src, k.src_bytes);
  auto fetch = [&](int F0) {
    // the two frames of this load, resolved through the frame index ONCE and up front (wave-uniform): a lookup inside
    // the chunk loop is an ordinary load whose wait drains the chunk loads issued before it
    int fr[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int F = F0 + u;
      int f = (F >= 0 && F < k.t_log) ? F : -1;
      if (f >= 0 && k.t_index) f = k.t_index[f];
      fr[u] = (f >= 0 && f < k.t_in) ? f : -1;
    }
#pragma unroll
    for (int i = 0; i < NXC; ++i) {
      const int frame = x_fr[i] == 0 ? fr[0] : fr[1];
      const bool fok = x_fr[i] >= 0 && frame >= 0;
      const int hi = 2 * ho0 - 3 + x_r[i], wi = 2 * wo0 - 8 + 8 * (x_j[i] & 255), ci = x_j[i] >> 8;
      const bool ok = fok && (unsigned)hi < (unsigned)k.h_in && wi >= 0 && wi + 8 <= k.w_in;
      const int64_t off = (int64_t)n * k.sn + (int64_t)ci * k.sc + (int64_t)frame * k.st + (int64_t)hi * k.sh + wi;
      xr[i] = sfk_buffer_load16(srs, ok ? (uint32_t)(off * 2) : SFK_OOB);
    }
  };
  auto stage = [&](int F0) {
#pragma unroll
    for (int i = 0; i < NXC; ++i) {
      if (x_fr[i] < 0) continue;
      const int sl = (F0 + x_fr[i] + 4 * NF) % NF;
      *reinterpret_cast<uint4*>(ring + sl * SLOT + x_loc[i]) = xr[i];
    }
  };

  const int b_lane = (4 * wave + perm_g) * F2_PITCH + (2 * l15 + 4) * 2;
  const int a_lane = a_off(l15, g);
  const int tpairs = (k.t_log + 1) / 2;
  const int tchunks = (tpairs + pairs_per_unit - 1) / pairs_per_unit;

  for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    const int tc = unit % tchunks;
    int tile = unit / tchunks, th, tw;
    tw = tile % k.tiles_w; tile /= k.tiles_w;
    th = tile % k.tiles_h;
    n = tile / k.tiles_h;
    ho0 = th * TS; wo0 = tw * TS;
    const int p0 = tc * pairs_per_unit, p1 = min(p0 + pairs_per_unit, tpairs);
    __syncthreads();                                   // the previous unit's ring reads (and the A build) are done
    for (int f2 = 0; f2 < NF; f2 += 2) {               // NF is even: fill the window of the first pair
      fetch(2 * p0 - PT + f2);
      stage(2 * p0 - PT + f2);
    }
    __syncthreads();
    for (int p = p0; p < p1; ++p) {
      const int to0 = 2 * p;
      if (p + 1 < p1) fetch(to0 + PT + 2);             // the two frames the next pair adds fly during this pair's MFMAs
      f32x4 acc[2];
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int base_f = (to0 - PT + 4 * NF) % NF;     // physical slot of f' = 0
#pragma unroll
      for (int sl = 0; sl < NF; ++sl) {
        int fp = sl - base_f;
        if (fp < 0) fp += NF;                          // wave-uniform
        const char* ab = amat + fp * CPF * 1024 + a_lane;
        const char* bb = ring + sl * SLOT + b_lane;
#pragma unroll
        for (int c6 = 0; c6 < CPF; ++c6) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(ab + c6 * 1024);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(bb + (c6 >> 1) * F2_PLANE + (2 * j + 4 * (c6 & 1)) * F2_PITCH);
            uint4 bv = make_uint4(q[0], q[1], q[2], q[3]);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, *reinterpret_cast<const bf16x8*>(&bv), acc[j], 0, 0, 0);
          }
        }
      }
      // ---- epilogue: lane holds rows 4g..4g+3 = (jt = g>>1, co = 4*(g&1) + r) of pixel (2*wave + j, l15)
      const int jt = g >> 1, co0 = 4 * (g & 1), to = to0 + jt;
      const int wo = wo0 + l15;
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ho = ho0 + 2 * wave + j;
        if (to < k.t_out && ho < k.ho && wo < k.wo && co0 < k.cout) {
          store4(yp + ((((int64_t)n * k.t_out + to) * k.ho + ho) * k.wo + wo) * k.yld + k.yoff + co0, acc[j]);
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[r] += acc[j][r]; s2[r] += acc[j][r] * acc[j][r]; }
        }
      }
      if (k.stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int sft = 1; sft < 16; sft <<= 1) {
            s1[r] += __shfl_xor(s1[r], sft);
            s2[r] += __shfl_xor(s2[r], sft);
          }
          if (l15 == 0) {
            red[(wave * 16 + 4 * g + r) * 2 + 0] = s1[r];
            red[(wave * 16 + 4 * g + r) * 2 + 1] = s2[r];
          }
        }
      }
      __syncthreads();                                 // ring reads of this pair are done; red is complete
      if (k.stats && tid < 16) {
        const int sjt = tid >> 3, sco = tid & 7, sto = to0 + sjt;
        if (sto < k.t_out && sco < k.cout) {
          float a1 = 0.f, a2 = 0.f;
#pragma unroll
          for (int w_ = 0; w_ < 8; ++w_) { a1 += red[(w_ * 16 + tid) * 2]; a2 += red[(w_ * 16 + tid) * 2 + 1]; }
          const int64_t trow = (((int64_t)n * k.t_out + sto) * k.tiles_h + th) * k.tiles_w + tw;
          k.stats[(trow * k.cout + sco) * 2 + 0] = a1;
          k.stats[(trow * k.cout + sco) * 2 + 1] = a2;
        }
      }
      if (p + 1 < p1) stage(to0 + PT + 2);
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------ forward, v3 (the design; the kernels follow)
// stem_fwd_v2_kernel is LDS-bound: every MFMA takes its own 16-byte pixel run (4-byte aligned: four ds_read_b32) from the ring,
// 8 LDS cycles per 16-cycle MFMA and wave, four waves' worth per CU -- 2.5 x the matrix pipe's time (560 us for the metric
// geometry against ~110 us of MFMAs).  The v3 design is INPUT-FRAME stationary: an input frame (2m + par) feeds every output pair
// it reaches at once -- pairs m - 1, m, m + 1 (kt = 5) -- so one pixel run is read ONCE for up to three MFMAs (18 instead of 6
// per 16 bytes), the three pairs' accumulators roll through registers (pair m - 1 is complete after input pair m and is
// stored), and the filter lives in REGISTERS: the 36 A fragments (f', chunk) of the v2 matrix, where an input frame of
// parity par meets pair m - 1 + r at window position f' = par + 2 - 2 r + PT.  LDS holds only the current and the next
// input pair (4 frame patches); the ring of kt + 1 frames and its rotation are gone.
// A unit is (clip, output tile, chunk of output pairs p0 .. p1 - 1) and walks input pairs p0 - 1 .. p1 (the two boundary
// pairs feed one output pair each; pairs outside the unit are multiplied too and roll out unstored -- a branch per pair made
// hipcc shuffle all 24 accumulator registers around every MFMA pair: 1,137 us); frames outside the clip are skipped.
// Staging: 16-byte chunks, each load instruction within ONE frame so that the (clip, frame) base rides the instruction's scalar
// offset (not range-checked) and the per-thread part is a 32-bit voffset computed once per unit (a 64-bit address per chunk
// cost ~100 vector instructions per pair and wave); padding = voffset 2^31, a frame outside the clip reads through a
// zero-sized resource; the physical frames of a unit are resolved ONCE into an LDS table (a frame-index load in the loop
// would drain the loads in flight); a pair is fetched two iterations ahead.
// First build (512 threads, 16 x 16 tiles): 560 -> 365 us alone; its phases ran one after the other (MFMAs 122 us, fetch /
// stage 92, epilogue 56, pixel-run reads 48 by removal) -- the half-tile kernels below overlap them across workgroups.

// 16-lane row sum with DPP shifts (4 vector instructions; __shfl_xor goes through ds_bpermute): the total ends in lane 15
__device__ __forceinline__ float stem_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
  return v;
}

// ------------------------------------------------------------------------------------------ forward, slow stem (kt = 1, 64 channels)
// The canonical slow stem through stem_fwd_kernel stages its patch element by element (18 two-byte loads in flight per thread),
// re-reads the filter from LDS every K-step, reduces its statistics with 128 ds_bpermute shuffles per tile and stores 8 bytes
// per lane: 300 us alone for 32 us of MFMAs and 61 us of HBM traffic.  The slow-stem kernel below (stem_fwd_s4_kernel) is the v3 machinery with the
// rows of the A fragments = 16 output channels (4 fragments, no temporal window): two frames per 16-byte-chunk fetch, two
// fetches ahead, the 24 filter fragments in registers, one pixel run per FOUR MFMAs, DPP row sums, v_permlane16_swap between
// channel fragments so that a lane stores 16 bytes (8 consecutive channels of a pixel).
__device__ __forceinline__ void stem_swap16(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// ------------------------------------------------------------------------------------------ forward, half tiles (256 threads)
// The first builds of the v3 design were 512-thread workgroups with ~235 registers per lane: ONE workgroup owns a CU's whole
// register file, so (a) its phases -- fetch, MFMAs, epilogue, barrier -- run strictly one after the other (measured: the parts
// add up), and (b) while one stem's persistent workgroups hold the CUs nothing else becomes resident: the two stems ran one after
// the other and the slow pathway's small kernels waited (bn_finalize 139 us in the step's timeline).  The kernels below are the
// same machines on HALF tiles (16 x 8 output pixels) with 4 waves: two workgroups -- of either kernel -- share a CU, one
// multiplies while the other fetches / stores.  BatchNorm partial sums: a workgroup adds up its whole unit and writes ONE
// statistics row; the rows of the unit's other frames are zeros (the consumer folds all rows).  The two halves of a 16 x 16
// statistics tile write disjoint rows: half h the frames of parity h (a unit has at least two frames).
constexpr int H2_ROWS = 8;                      // output rows of a half tile
constexpr int H2_PR = 2 * H2_ROWS + 6;          // 22 patch rows (21 + the row the padded kh = 7 reads)
constexpr int H2_PLANE = H2_PR * F2_PITCH;      // 2112 B

// staging of input PAIRS for a 256-thread workgroup: four 16-byte chunks per thread and pair (slot s: frame s & 1, chunk
// tid + 256 (s >> 1) of the frame's CIN x 22 x 6), the (clip, frame) base on the scalar offset, padding = voffset 2^31
template <int CIN>
struct H2Stage {
  static constexpr int SLOT = CIN * H2_PLANE, FC = CIN * H2_PR * 6;
  static_assert(FC > 256 && FC <= 512, "two loads per thread and frame");
  uint32_t desc[2];      // per chunk index (s >> 1): bits 0..15 byte offset inside the frame patch, 16..21 row, 22..24 chunk, 31 unused
  int chan[2];
  uint32_t xv[2];        // voffsets of the current unit
  __device__ __forceinline__ void init(const StemK& k, int tid) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = tid + 256 * h;
      const int ci = e / (H2_PR * 6), rr = e % (H2_PR * 6);
      const int row = rr / 6, jc = rr % 6;
      desc[h] = (uint32_t)(ci * H2_PLANE + row * F2_PITCH + jc * 16) | (uint32_t)row << 16 | (uint32_t)jc << 22 | (uint32_t)(e >= FC) << 31;
      chan[h] = (int)(ci * k.sc * 2);
    }
  }
  __device__ __forceinline__ void unit(const StemK& k, int ho0, int wo0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int hi = 2 * ho0 - 3 + (int)((desc[h] >> 16) & 63), wi = 2 * wo0 - 8 + 8 * (int)((desc[h] >> 22) & 7);
      const bool ok = !(desc[h] >> 31) && (unsigned)hi < (unsigned)k.h_in && wi >= 0 && wi + 8 <= k.w_in;
      xv[h] = ok ? (uint32_t)(chan[h] + (hi * (int)k.sh + wi) * 2) : 0x80000000u;
    }
  }
  // frames f0 / f1 = physical frames (-1: zeros) of the pair
  __device__ __forceinline__ void fetch(const StemK& k, uint4 (&xr)[4], __amdgpu_buffer_rsrc_t rs, __amdgpu_buffer_rsrc_t rs0, int n, int f0,
                                        int f1) const {
    const uint32_t so0 = f0 >= 0 ? (uint32_t)(((int64_t)n * k.sn + (int64_t)f0 * k.st) * 2) : 0u;
    const uint32_t so1 = f1 >= 0 ? (uint32_t)(((int64_t)n * k.sn + (int64_t)f1 * k.st) * 2) : 0u;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      const bool second = s_ & 1;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128((second ? f1 : f0) >= 0 ? rs : rs0, (int)xv[s_ >> 1], (int)(second ? so1 : so0), 0);
      xr[s_] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  }
  __device__ __forceinline__ void stage(char* pair_buf, const uint4 (&xr)[4]) const {
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
      if (!(desc[s_ >> 1] >> 31)) *reinterpret_cast<uint4*>(pair_buf + (s_ & 1) * SLOT + (desc[s_ >> 1] & 0xFFFFu)) = xr[s_];
  }
};

// unit decomposition shared by the two kernels: (clip, half tile, chunk of pairs); the last chunk takes a trailing half pair
struct H2Unit {
  int n, th2, tw, p0, p1;
  __device__ __forceinline__ void set(const StemK& k, int unit, int ppu, int tchunks, int tiles_h2, int tpairs) {
    const int tc = unit % tchunks;
    int tile = unit / tchunks;
    tw = tile % k.tiles_w; tile /= k.tiles_w;
    th2 = tile % tiles_h2;
    n = tile / tiles_h2;
    p0 = tc * ppu;
    p1 = tc == tchunks - 1 ? tpairs : p0 + ppu;
  }
};

// fast stem
template <int CIN, int KT>
__global__ __launch_bounds__(256, 2) void stem_fwd_v4_kernel(const StemK k, int pairs_per_unit, int nunits, int tchunks) {
  constexpr int NF = KT + 1, PT = KT / 2;
  constexpr int SLOT = CIN * H2_PLANE;
  constexpr int CPF = CIN * 2;
  constexpr int NCH = NF * CPF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* pb = smem;                                               // [2 pair buffers][2 frames][CIN][22][96 B]
  char* amat = smem + 4 * SLOT;                                  // [NCH][16 rows][64 B]
  float* red = reinterpret_cast<float*>(amat + NCH * 1024);      // [4 waves][16 rows][2]
  int* fidx = reinterpret_cast<int*>(red + 4 * 16 * 2);          // [32]
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bf16_t* yp = static_cast<bf16_t*>(k.y);
  auto a_off = [](int r, int sg) { return r * 64 + ((sg ^ ((4 - ((r >> 2) & 3)) & 3)) << 4); };
  const int perm_g = ((g & 1) << 1) | (g >> 1);
  {
    const bf16_t* wp = static_cast<const bf16_t*>(k.w);
    for (int e = tid; e < NCH * 64; e += 256) {
      const int ch = e >> 6, row = (e >> 2) & 15, gg = e & 3;
      const int fp = ch / CPF, c6 = ch % CPF;
      const int jt = row >> 3, co = row & 7, f = fp - jt;
      const int kh = (c6 & 1) * 4 + (((gg & 1) << 1) | (gg >> 1)), ci = c6 >> 1;
      bf16x8 v;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.f;
      if (f >= 0 && f < KT && co < k.cout && kh < KH) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wp + (int64_t)co * k.kp + ((f * CIN + ci) * KH + kh) * 8);
#pragma unroll
        for (int i = 1; i < 8; ++i) v[i] = wv[i - 1];
      }
      *reinterpret_cast<bf16x8*>(amat + ch * 1024 + a_off(row, gg)) = v;
    }
  }
  __syncthreads();
  bf16x8 afr[NCH];
  {
    const int a_lane = a_off(l15, g);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      afr[ch] = *reinterpret_cast<const bf16x8*>(amat + ch * 1024 + a_lane);
      asm volatile("" : "+v"(afr[ch]));
    }
  }
  H2Stage<CIN> stg;
  stg.init(k, tid);
  uint4 xra[4], xrb[4];
  const __amdgpu_buffer_rsrc_t srs = sfk_make_rsrc(k.src, k.src_bytes);
  const __amdgpu_buffer_rsrc_t srs0 = sfk_make_rsrc(k.src, 0);
  const int b_lane = (4 * wave + perm_g) * F2_PITCH + (2 * l15 + 4) * 2;
  const int tpairs = (k.t_log + 1) / 2;
  const int tiles_h2 = (k.ho + H2_ROWS - 1) / H2_ROWS;
  H2Unit u;
  // XCD-aware order: workgroups b, b + 8, ... share an L2, so consecutive LOGICAL ids -- neighbouring tiles and temporal chunks,
  // which read overlapping patch rows / columns / frames -- go to one XCD (round-robin units fetched 4.4 x the clip from HBM)
  const int lb = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  for (int unit = lb; unit < nunits; unit += gridDim.x) {
    u.set(k, unit, pairs_per_unit, tchunks, tiles_h2, tpairs);
    const int n = u.n, p0 = u.p0, p1 = u.p1;
    const int ho0 = u.th2 * H2_ROWS, wo0 = u.tw * TS;
    stg.unit(k, ho0, wo0);
    const int fbase = 2 * (p0 - 1);
    if (tid < 2 * (p1 - p0 + 2)) {
      const int F = fbase + tid;
      int f = (F >= 0 && F < k.t_log) ? F : -1;
      if (f >= 0 && k.t_index) f = k.t_index[f];
      fidx[tid] = (f >= 0 && f < k.t_in) ? f : -1;
    }
    __syncthreads();                                   // the previous unit's patch / red reads are done; fidx is visible
    auto frame_of = [&](int F) { return __builtin_amdgcn_readfirstlane(fidx[F - fbase]); };
    stg.fetch(k, xra, srs, srs0, n, frame_of(2 * (p0 - 1)), frame_of(2 * (p0 - 1) + 1));
    stg.stage(pb, xra);
    stg.fetch(k, xrb, srs, srs0, n, frame_of(2 * p0), frame_of(2 * p0 + 1));
    __syncthreads();
    f32x4 acc[3][2];
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[r][0] = acc[r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};      // the unit's sums of this lane's rows (jt, co)
    auto pair_iter = [&](const int m, const int cur, uint4 (&xf)[4], const uint4 (&xs)[4]) __attribute__((always_inline)) {
      if (m + 2 <= p1) stg.fetch(k, xf, srs, srs0, n, frame_of(2 * (m + 2)), frame_of(2 * (m + 2) + 1));
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int F = 2 * m + par;
        if (F < 0 || F >= k.t_log) continue;           // temporal padding (wave-uniform)
        const char* bb = pb + (cur * 2 + par) * SLOT + b_lane;
#pragma unroll
        for (int c6 = 0; c6 < CPF; ++c6) {
          bf16x8 bv[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(bb + (c6 >> 1) * H2_PLANE + (2 * j + 4 * (c6 & 1)) * F2_PITCH);
            uint4 t4 = make_uint4(q[0], q[1], q[2], q[3]);
            bv[j] = *reinterpret_cast<const bf16x8*>(&t4);
          }
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const int fp = par + 2 - 2 * r + PT;       // (pairs outside the unit are multiplied too and roll out unstored)
            if (fp >= 0 && fp < NF) {
              acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[fp * CPF + c6], bv[0], acc[r][0], 0, 0, 0);
              acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[fp * CPF + c6], bv[1], acc[r][1], 0, 0, 0);
            }
          }
        }
      }
      // ---- output pair m - 1 is complete: lane holds rows 4g..4g+3 = (jt = g>>1, co = 4*(g&1) + r) of pixel (2*wave + j, l15)
      if (m - 1 >= p0) {
        const int jt = g >> 1, co0 = 4 * (g & 1), to = 2 * (m - 1) + jt;
        const int wo = wo0 + l15;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ho = ho0 + 2 * wave + j;
          if (to < k.t_out && ho < k.ho && wo < k.wo && co0 < k.cout) {
            store4(yp + ((((int64_t)n * k.t_out + to) * k.ho + ho) * k.wo + wo) * k.yld + k.yoff + co0, acc[0][j]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[r] += acc[0][j][r]; s2[r] += acc[0][j][r] * acc[0][j][r]; }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) { acc[0][j] = acc[1][j]; acc[1][j] = acc[2][j]; acc[2][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      if (m < p1) stg.stage(pb + (cur ^ 1) * 2 * SLOT, xs);
      __syncthreads();                                 // this pair's patch reads are done, the next pair is staged
    };
    for (int m = p0 - 1;;) {
      pair_iter(m, 0, xra, xrb);
      if (++m > p1) break;
      pair_iter(m, 1, xrb, xra);
      if (++m > p1) break;
    }
    if (k.stats) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = stem_row16_sum(s1[r]), c = stem_row16_sum(s2[r]);
        if (l15 == 15) {
          red[(wave * 16 + 4 * g + r) * 2 + 0] = a;
          red[(wave * 16 + 4 * g + r) * 2 + 1] = c;
        }
      }
      __syncthreads();
      const int half = u.th2 & 1, th = u.th2 >> 1;
      const bool alone = (u.th2 | 1) >= tiles_h2;          // the statistics tile has no second half: all its rows are ours
      const int nfr = min(2 * p1, k.t_out) - 2 * p0;       // output frames of this unit (>= 2)
      for (int e = tid; e < nfr * k.cout; e += 256) {
        const int fr = e / k.cout, co = e - fr * k.cout;
        if (!alone && (fr & 1) != half) continue;            // the other half tile's rows
        float a1 = 0.f, a2 = 0.f;
        if (fr == half && co < 8) {
#pragma unroll
          for (int w_ = 0; w_ < 4; ++w_)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) { a1 += red[(w_ * 16 + jt * 8 + co) * 2]; a2 += red[(w_ * 16 + jt * 8 + co) * 2 + 1]; }
        }
        const int64_t trow = (((int64_t)n * k.t_out + 2 * p0 + fr) * k.tiles_h + th) * k.tiles_w + u.tw;
        k.stats[(trow * k.cout + co) * 2 + 0] = a1;
        k.stats[(trow * k.cout + co) * 2 + 1] = a2;
      }
    }
  }
}

// slow stem (kt = 1, 64 output channels = 4 A fragments per chunk, no temporal window)
template <int CIN>
__global__ __launch_bounds__(256, 2) void stem_fwd_s4_kernel(const StemK k, int pairs_per_unit, int nunits, int tchunks) {
  constexpr int SLOT = CIN * H2_PLANE;
  constexpr int CPF = CIN * 2, NCF = 4, NCH = NCF * CPF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* pb = smem;
  char* amat = smem + 4 * SLOT;
  float* red = reinterpret_cast<float*>(amat + NCH * 1024);      // [4 waves][64 co][2]
  int* fidx = reinterpret_cast<int*>(red + 4 * 64 * 2);          // [32]
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bf16_t* yp = static_cast<bf16_t*>(k.y);
  auto a_off = [](int r, int sg) { return r * 64 + ((sg ^ ((4 - ((r >> 2) & 3)) & 3)) << 4); };
  const int perm_g = ((g & 1) << 1) | (g >> 1);
  {
    const bf16_t* wp = static_cast<const bf16_t*>(k.w);
    for (int e = tid; e < NCH * 64; e += 256) {
      const int ch = e >> 6, row = (e >> 2) & 15, gg = e & 3;
      const int cf = ch / CPF, c6 = ch % CPF;
      const int co = 16 * cf + row;
      const int kh = (c6 & 1) * 4 + (((gg & 1) << 1) | (gg >> 1)), ci = c6 >> 1;
      bf16x8 v;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.f;
      if (co < k.cout && kh < KH) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wp + (int64_t)co * k.kp + (ci * KH + kh) * 8);
#pragma unroll
        for (int i = 1; i < 8; ++i) v[i] = wv[i - 1];
      }
      *reinterpret_cast<bf16x8*>(amat + ch * 1024 + a_off(row, gg)) = v;
    }
  }
  __syncthreads();
  bf16x8 afr[NCH];
  {
    const int a_lane = a_off(l15, g);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      afr[ch] = *reinterpret_cast<const bf16x8*>(amat + ch * 1024 + a_lane);
      asm volatile("" : "+v"(afr[ch]));
    }
  }
  H2Stage<CIN> stg;
  stg.init(k, tid);
  uint4 xra[4], xrb[4];
  const __amdgpu_buffer_rsrc_t srs = sfk_make_rsrc(k.src, k.src_bytes);
  const __amdgpu_buffer_rsrc_t srs0 = sfk_make_rsrc(k.src, 0);
  const int b_lane = (4 * wave + perm_g) * F2_PITCH + (2 * l15 + 4) * 2;
  const int tpairs = (k.t_log + 1) / 2;
  const int tiles_h2 = (k.ho + H2_ROWS - 1) / H2_ROWS;
  H2Unit u;
  // XCD-aware order: workgroups b, b + 8, ... share an L2, so consecutive LOGICAL ids -- neighbouring tiles and temporal chunks,
  // which read overlapping patch rows / columns / frames -- go to one XCD (round-robin units fetched 4.4 x the clip from HBM)
  const int lb = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  for (int unit = lb; unit < nunits; unit += gridDim.x) {
    u.set(k, unit, pairs_per_unit, tchunks, tiles_h2, tpairs);
    const int n = u.n, p0 = u.p0, p1 = u.p1;                 // input = output pairs p0 .. p1 - 1
    const int ho0 = u.th2 * H2_ROWS, wo0 = u.tw * TS;
    stg.unit(k, ho0, wo0);
    const int fbase = 2 * p0;
    if (tid < 2 * (p1 - p0)) {
      const int F = fbase + tid;
      int f = (F >= 0 && F < k.t_log) ? F : -1;
      if (f >= 0 && k.t_index) f = k.t_index[f];
      fidx[tid] = (f >= 0 && f < k.t_in) ? f : -1;
    }
    __syncthreads();
    auto frame_of = [&](int F) { return __builtin_amdgcn_readfirstlane(fidx[F - fbase]); };
    stg.fetch(k, xra, srs, srs0, n, frame_of(2 * p0), frame_of(2 * p0 + 1));
    stg.stage(pb, xra);
    if (p0 + 1 < p1) stg.fetch(k, xrb, srs, srs0, n, frame_of(2 * (p0 + 1)), frame_of(2 * (p0 + 1) + 1));
    __syncthreads();
    float s1[NCF][4], s2[NCF][4];
#pragma unroll
    for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
      for (int r = 0; r < 4; ++r) s1[cf][r] = s2[cf][r] = 0.f;
    auto pair_iter = [&](const int m, const int cur, uint4 (&xf)[4], const uint4 (&xs)[4]) __attribute__((always_inline)) {
      if (m + 2 < p1) stg.fetch(k, xf, srs, srs0, n, frame_of(2 * (m + 2)), frame_of(2 * (m + 2) + 1));
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int to = 2 * m + par;
        if (to >= k.t_out) continue;                   // odd clip length (wave-uniform)
        const char* bb = pb + (cur * 2 + par) * SLOT + b_lane;
        f32x4 acc[NCF][2];
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf) acc[cf][0] = acc[cf][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c6 = 0; c6 < CPF; ++c6) {
          bf16x8 bv[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(bb + (c6 >> 1) * H2_PLANE + (2 * j + 4 * (c6 & 1)) * F2_PITCH);
            uint4 t4 = make_uint4(q[0], q[1], q[2], q[3]);
            bv[j] = *reinterpret_cast<const bf16x8*>(&t4);
          }
#pragma unroll
          for (int cf = 0; cf < NCF; ++cf) {
            acc[cf][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[cf * CPF + c6], bv[0], acc[cf][0], 0, 0, 0);
            acc[cf][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[cf * CPF + c6], bv[1], acc[cf][1], 0, 0, 0);
          }
        }
        const int wo = wo0 + l15;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ho = ho0 + 2 * wave + j;
          const bool pok = ho < k.ho && wo < k.wo;
          if (pok) {
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
              for (int r = 0; r < 4; ++r) { s1[cf][r] += acc[cf][j][r]; s2[cf][r] += acc[cf][j][r] * acc[cf][j][r]; }
          }
          bf16_t* pix = yp + ((((int64_t)n * k.t_out + to) * k.ho + ho) * k.wo + wo) * k.yld + k.yoff;
#pragma unroll
          for (int cp = 0; cp < NCF; cp += 2) {
            float v[8] = {acc[cp][j][0], acc[cp][j][1], acc[cp][j][2], acc[cp][j][3],
                          acc[cp + 1][j][0], acc[cp + 1][j][1], acc[cp + 1][j][2], acc[cp + 1][j][3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) stem_swap16(v[e], v[4 + e]);
            const int co = 16 * cp + 16 * (g & 1) + 8 * (g >> 1);
            if (pok && co < k.cout) {
              bf16x8 o;
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
              *reinterpret_cast<bf16x8*>(pix + co) = o;
            }
          }
        }
      }
      if (m + 1 < p1) stg.stage(pb + (cur ^ 1) * 2 * SLOT, xs);
      __syncthreads();
    };
    for (int m = p0;;) {
      pair_iter(m, 0, xra, xrb);
      if (++m >= p1) break;
      pair_iter(m, 1, xrb, xra);
      if (++m >= p1) break;
    }
    if (k.stats) {
#pragma unroll
      for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = stem_row16_sum(s1[cf][r]), c = stem_row16_sum(s2[cf][r]);
          if (l15 == 15) {
            red[(wave * 64 + 16 * cf + 4 * g + r) * 2 + 0] = a;
            red[(wave * 64 + 16 * cf + 4 * g + r) * 2 + 1] = c;
          }
        }
      __syncthreads();
      const int half = u.th2 & 1, th = u.th2 >> 1;
      const bool alone = (u.th2 | 1) >= tiles_h2;
      const int nfr = min(2 * p1, k.t_out) - 2 * p0;       // output frames of this unit (>= 2)
      for (int e = tid; e < nfr * k.cout; e += 256) {
        const int fr = e / k.cout, co = e - fr * k.cout;
        if (!alone && (fr & 1) != half) continue;
        float a1 = 0.f, a2 = 0.f;
        if (fr == half) {
#pragma unroll
          for (int w_ = 0; w_ < 4; ++w_) { a1 += red[(w_ * 64 + co) * 2]; a2 += red[(w_ * 64 + co) * 2 + 1]; }
        }
        const int64_t trow = (((int64_t)n * k.t_out + 2 * p0 + fr) * k.tiles_h + th) * k.tiles_w + u.tw;
        k.stats[(trow * k.cout + co) * 2 + 0] = a1;
        k.stats[(trow * k.cout + co) * 2 + 1] = a2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ filter gradient, v2
// The canonical fast stem (5x7x7, 3 -> 8 channels) is where the generic kernel above loses: one MFMA per B fragment
// built from 8 two-byte LDS reads, the dY tile re-staged once per (frame, channel) plane, element-wise global loads.
// v2 is INPUT-FRAME stationary: a work item is (clip n, input frame tf, 16x16 output tile).  Its x patch (cin planes)
// is staged once with 16-byte loads; the dY tiles of the kt output frames that see this input frame
// (to = tf + kt/2 - f) are staged side by side as [256 pixels][kt*8 channels], so the GEMM is
//     dW[(f, co)][(ci, kh, kw)] += sum_pixels dYt[pixel][(f, co)] * patch[ci][2*ho + kh][2*wo + kw]
// with M = kt*8 <= 48 rows (3 MFMA row fragments per B fragment instead of 1).  A B fragment (16 columns = 2 kh x 8 kw,
// 8 consecutive output pixels per lane) is 8 aligned dword reads + 4 v_perm (even / odd halves = the stride-2 walk).
// Blocks are persistent over a contiguous range of items and keep dW in registers; one atomic flush at the end.
// Requires: bf16 clip with unit W stride, 16-byte aligned rows/planes, W % 8 == 0, cout <= 8, kt <= 5, cin == CIN.
constexpr int V2_PC = 56;                   // patch row pitch in elements (48 used; 112 B keeps the b128 stores aligned)
constexpr int V2_PR = 38;                   // patch rows (37 + the row the padded kh = 7 column reads)
constexpr int V2_CH = 6;                    // 16-byte chunks per patch row (48 columns from wi = 2*wo0 - 8)
constexpr int V2_DROW = 128;                // dY tile row pitch in bytes: 4 blocks of 32 B, block index XOR-swizzled

__device__ __forceinline__ int v2_swz(int px) { return ((px >> 1) & 1) | (((px >> 3) & 1) << 1); }

template <int CIN>
__global__ __launch_bounds__(256, 2) void stem_wgrad_v2_kernel(const StemK k) {
  constexpr int PLANE = V2_PR * V2_PC * 2;                       // bytes
  constexpr int NXC = (CIN * V2_PR * V2_CH + 255) / 256;         // patch chunks per thread
  constexpr int NDC = 5;                                         // dY chunks per thread (one per f; kt <= 5)
  __shared__ __attribute__((aligned(16))) char smem[CIN * PLANE + 256 * V2_DROW];
  char* patch = smem;
  char* dyt = smem + CIN * PLANE;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware ranges (workgroups b, b + 8, ... share an L2): consecutive LOGICAL workgroups walk neighbouring frames of one clip
  // at the same tile at the same time, and the five input frames that see one dY tile then find it in their XCD's L2
  const int lb = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int item0 = lb * k.tiles_per_block;
  const int item1 = min(item0 + k.tiles_per_block, k.ntiles);
  if (item0 >= item1) return;
  const bf16_t* src = static_cast<const bf16_t*>(k.src);
  const bf16_t* dyp = static_cast<const bf16_t*>(k.y);

  // ---- staging slots of this thread (fixed): patch chunk (ci, row, j) and dY chunk (pixel, f)
  int x_ci[NXC], x_r[NXC], x_j[NXC];
#pragma unroll
  for (int i = 0; i < NXC; ++i) {
    const int e = tid + 256 * i;
    x_ci[i] = e / (V2_PR * V2_CH);
    const int rem = e % (V2_PR * V2_CH);
    x_r[i] = rem / V2_CH;
    x_j[i] = rem % V2_CH;
    if (e >= CIN * V2_PR * V2_CH) x_ci[i] = -1;
  }
  uint4 xr[NXC], dr[NDC];
  const __amdgpu_buffer_rsrc_t srs = sfk_make_rsrc(k.src, k.src_bytes);
  const __amdgpu_buffer_rsrc_t yrs = sfk_make_rsrc(k.y, k.y_bytes);
  // Addresses: the per-thread part (channel / patch row / chunk, or the pixel of the dY tile) is a 32-bit voffset, the
  // (clip, frame) base rides the instruction's scalar offset (not range-checked: an out-of-range voffset still reads zeros);
  // a frame outside the clip reads through a zero-sized resource.  The first version built a 64-bit address per chunk: ~150
  // vector instructions per item and wave, as many issue cycles as the item's 72 MFMAs.
  const __amdgpu_buffer_rsrc_t srs0 = sfk_make_rsrc(k.src, 0);
  const __amdgpu_buffer_rsrc_t yrs0 = sfk_make_rsrc(k.y, 0);
  int x_chan[NXC];
#pragma unroll
  for (int i = 0; i < NXC; ++i) x_chan[i] = x_ci[i] < 0 ? 0 : (int)(x_ci[i] * k.sc * 2);
  auto fetch = [&](int item) {
    int n, tf, ho0, wo0;
    tile_coords(k, item, n, tf, ho0, wo0);
    int frame = k.t_index ? __builtin_amdgcn_readfirstlane(k.t_index[tf]) : tf;
    const bool fok = frame >= 0 && frame < k.t_in;
    const uint32_t xso = fok ? (uint32_t)(((int64_t)n * k.sn + (int64_t)frame * k.st) * 2) : 0u;
    // branch-free buffer loads (out-of-range offset -> zeros): all of an item's loads are issued in one batch
#pragma unroll
    for (int i = 0; i < NXC; ++i) {
      const int hi = 2 * ho0 - 3 + x_r[i], wi = 2 * wo0 - 8 + 8 * x_j[i];
      const bool ok = x_ci[i] >= 0 && (unsigned)hi < (unsigned)k.h_in && wi >= 0 && wi + 8 <= k.w_in;
      const uint32_t vo = ok ? (uint32_t)(x_chan[i] + (hi * (int)k.sh + wi) * 2) : 0x80000000u;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(fok ? srs : srs0, (int)vo, (int)xso, 0);
      xr[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    const int ho = ho0 + (tid >> 4), wo = wo0 + (tid & 15);
    const bool pok = ho < k.ho && wo < k.wo;
    const uint32_t yvo = pok ? (uint32_t)(((ho * k.wo + wo) * k.yld + k.yoff) * 2) : 0x80000000u;
    const int64_t fstride = (int64_t)k.ho * k.wo * k.yld * 2;          // bytes per output frame
#pragma unroll
    for (int f = 0; f < NDC; ++f) {
      const int to = tf + k.pt - f;
      const bool ok = f < k.kt && to >= 0 && to < k.t_out;               // (wave-uniform)
      const uint32_t yso = ok ? (uint32_t)(((int64_t)n * k.t_out + to) * fstride) : 0u;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ok ? yrs : yrs0, (int)yvo, (int)yso, 0);
      dr[f] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NXC; ++i)
      if (x_ci[i] >= 0)
        *reinterpret_cast<uint4*>(patch + x_ci[i] * PLANE + x_r[i] * (V2_PC * 2) + x_j[i] * 16) = xr[i];
    const int sw = v2_swz(tid);
#pragma unroll
    for (int f = 0; f < NDC; ++f)
      *reinterpret_cast<uint4*>(dyt + tid * V2_DROW + (((f >> 1) ^ sw) << 5) + (f & 1) * 16) = dr[f];
    // channel block 5 (f = 5) of M fragment 2 is never a real filter tap, but it feeds the MFMA: keep it zero
    *reinterpret_cast<uint4*>(dyt + tid * V2_DROW + ((2 ^ sw) << 5) + 16) = make_uint4(0, 0, 0, 0);
  };

  // ---- compute coordinates: this wave owns column fragment cf = wave (kh = 2*wave + khh, kw = l15 & 7)
  const int khh = l15 >> 3, kw = l15 & 7;
  const uint32_t sel = (kw & 1) ? 0x05040100u : 0x07060302u;     // odd kw: window column is even -> low halves
  const int cbe = (kw & 1) ? 5 + kw : 4 + kw;                    // first (even) element of the dword walk
  const int b_lane = ((g >> 1) * 2 + 2 * wave + khh) * (V2_PC * 2) + (2 * (g & 1) * 8 + cbe) * 2;
  const int q = l15 >> 2, p4 = lane & 3;
  int a_lane[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int row = 8 * g + q;                                   // low 4 bits of the pixel index decide the swizzle
    a_lane[i] = row * V2_DROW + ((i ^ v2_swz(row)) << 5) + p4 * 8;
  }
  f32x4 acc[CIN][3];
#pragma unroll
  for (int c = 0; c < CIN; ++c)
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  fetch(item0);
  for (int item = item0; item < item1; ++item) {
    __syncthreads();
    stage();
    __syncthreads();
    if (item + 1 < item1) fetch(item + 1);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      bf16x8 a[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const char* ap = dyt + s * 32 * V2_DROW + a_lane[i];
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(ap));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(ap + 4 * V2_DROW));
        a[i][0] = lo[0]; a[i][1] = lo[1]; a[i][2] = lo[2]; a[i][3] = lo[3];
        a[i][4] = hi[0]; a[i][5] = hi[1]; a[i][6] = hi[2]; a[i][7] = hi[3];
      }
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        const uint32_t* bp = reinterpret_cast<const uint32_t*>(patch + c * PLANE + s * 4 * (V2_PC * 2) + b_lane);
        uint32_t d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = bp[e];
        uint4 bv;
        bv.x = __builtin_amdgcn_perm(d[1], d[0], sel);
        bv.y = __builtin_amdgcn_perm(d[3], d[2], sel);
        bv.z = __builtin_amdgcn_perm(d[5], d[4], sel);
        bv.w = __builtin_amdgcn_perm(d[7], d[6], sel);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(&bv);
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[c][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[c][i], 0, 0, 0);
      }
    }
  }
  // D[row = (f, co)][col = (khh, kw)]: lane holds rows 16i + 4g + r
  const int kh = 2 * wave + khh;
  if (kh < KH && kw < 7) {
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * i + 4 * g + r, f = row >> 3, co = row & 7;
          if (f < k.kt && co < k.cout)
            atomicAdd(k.dw + (int64_t)co * k.kp + ((f * CIN + c) * KH + kh) * 8 + kw, acc[c][i][r]);
        }
  }
}

// bytes from s->src to one past the last element any (n, ci, t, h, w) index reaches (positive strides)
inline int64_t stem_src_extent(const sfk_stem_src* s, int n) {
  const int64_t esz = s->src_dtype == SFK_BF16 ? 2 : 4;
  if (s->sn < 0 || s->sc < 0 || s->st < 0 || s->sh < 0 || s->sw < 0) return (int64_t)1 << 40;
  return esz * ((int64_t)(n - 1) * s->sn + (int64_t)(s->cin - 1) * s->sc + (int64_t)(s->t_in - 1) * s->st +
                (int64_t)(s->h_in - 1) * s->sh + (int64_t)(s->w_in - 1) * s->sw + 1);
}

int fill(const sfk_stem_src* s, int cout, int t_out, int ho, int wo, StemK& k) {
  if (!s || !s->src || s->cin <= 0 || s->kt <= 0 || !(s->kt & 1) || s->t_in <= 0 || s->h_in <= 0 || s->w_in <= 0)
    return SFK_ERR_INVALID;
  if (s->src_dtype != SFK_F32 && s->src_dtype != SFK_BF16) return SFK_ERR_INVALID;
  const int t_log = s->t_index ? s->t_len : s->t_in;
  if (t_log <= 0 || t_out != t_log) return SFK_ERR_INVALID;
  if (ho != (s->h_in + 6 - 7) / 2 + 1 || wo != (s->w_in + 6 - 7) / 2 + 1) return SFK_ERR_INVALID;
  if (cout <= 0 || cout > 64 || (cout % 4)) return SFK_ERR_UNSUPPORTED;
  k.src = s->src; k.sn = s->sn; k.sc = s->sc; k.st = s->st; k.sh = s->sh; k.sw = s->sw;
  k.cin = s->cin; k.t_in = s->t_in; k.h_in = s->h_in; k.w_in = s->w_in; k.t_index = s->t_index; k.t_log = t_log;
  k.kt = s->kt; k.pt = s->kt / 2;
  k.planes = s->kt * s->cin;
  k.krows = k.planes * KH;
  k.kp = ((k.krows + 3) / 4) * 4 * 8;
  k.cout = cout; k.ho = ho; k.wo = wo; k.t_out = t_out;
  k.tiles_h = (ho + TS - 1) / TS; k.tiles_w = (wo + TS - 1) / TS;
  k.dtw.set(k.tiles_w); k.dth.set(k.tiles_h); k.dt.set(t_out); k.d7.set(KH); k.dcin.set(s->cin); k.dpc.set(PC);
  return SFK_OK;
}

}  // namespace

extern "C" int sfk_stem_kp(int32_t cin, int32_t kt) { return ((kt * cin * KH + 3) / 4) * 4 * 8; }

extern "C" int sfk_stem_conv_tiles(const sfk_stem_src* s, const sfk_fmap* y) {
  if (!s || !sfk_fmap_ok(y)) return SFK_ERR_INVALID;
  return y->n * y->t * ((y->h + TS - 1) / TS) * ((y->w + TS - 1) / TS);
}

extern "C" int sfk_stem_conv_fwd(const sfk_stem_src* s, const void* w, const sfk_fmap* y, float* stats,
                                 sfk_stream_t stream) {
  if (!w || !sfk_fmap_ok(y)) return SFK_ERR_INVALID;
  StemK k;
  const int st = fill(s, y->c, y->t, y->h, y->w, k);
  if (st != SFK_OK) return st;
  if ((y->ld % 4) || (y->c_off % 4) || (((uintptr_t)y->ptr) & 15) || (((uintptr_t)w) & 15)) return SFK_ERR_UNSUPPORTED;
  k.w = w; k.y = y->ptr; k.yld = y->ld; k.yoff = y->c_off; k.stats = stats; k.dw = nullptr;
  k.ntiles = y->n * y->t * k.tiles_h * k.tiles_w;
  // canonical fast stem geometry in bf16 with 16-byte addressable rows: temporal pairs over a rolling frame ring
  if (y->dtype == SFK_BF16 && s->src_dtype == SFK_BF16 && s->cin == 3 && (s->kt == 5 || s->kt == 3) && y->c <= 8 &&
      s->sw == 1 && (s->w_in % 8) == 0 && !((s->sn | s->sc | s->st | s->sh) & 7) && !(((uintptr_t)s->src) & 15) &&
      stem_src_extent(s, y->n) < (1ll << 32) - 64) {
    k.src_bytes = (uint32_t)stem_src_extent(s, y->n);
    k.y_bytes = 0;
    hipStream_t hs2 = static_cast<hipStream_t>(stream);
    const int nf = s->kt + 1;
    const int lds2 = nf * 3 * F2_PLANE + nf * 6 * 1024 + 8 * 16 * 2 * 4;
    const int tpairs = (k.t_log + 1) / 2;
    const int ppu = tpairs < 8 ? tpairs : 8;
    const int nunits = y->n * k.tiles_h * k.tiles_w * ((tpairs + ppu - 1) / ppu);
    const int grid2 = nunits < 256 ? nunits : 256;
    const bool v3 = (sfk_tune().stem_v3 & 1) != 0 && stem_src_extent(s, y->n) < (1ll << 31) && k.t_log >= 2;      // (padding = voffset 2^31)
    if (v3) {                                                      // half tiles, two workgroups per CU
      const int nfull = k.t_log / 2, tch = nfull <= ppu ? 1 : (nfull + ppu - 1) / ppu;
      const int nu = y->n * ((y->h + H2_ROWS - 1) / H2_ROWS) * k.tiles_w * tch;
      const int grid4 = nu < 512 ? nu : 512;
      const int lds4 = 4 * 3 * H2_PLANE + nf * 6 * 1024 + 4 * 16 * 2 * 4 + 32 * 4;
      if (s->kt == 5) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd_v4_kernel<3, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        hipLaunchKernelGGL((stem_fwd_v4_kernel<3, 5>), dim3((unsigned)grid4), dim3(256), lds4, hs2, k, ppu, nu, tch);
      } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd_v4_kernel<3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        hipLaunchKernelGGL((stem_fwd_v4_kernel<3, 3>), dim3((unsigned)grid4), dim3(256), lds4, hs2, k, ppu, nu, tch);
      }
    } else if (s->kt == 5) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd_v2_kernel<3, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
      hipLaunchKernelGGL((stem_fwd_v2_kernel<3, 5>), dim3((unsigned)grid2), dim3(512), lds2, hs2, k, ppu, nunits);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd_v2_kernel<3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
      hipLaunchKernelGGL((stem_fwd_v2_kernel<3, 3>), dim3((unsigned)grid2), dim3(512), lds2, hs2, k, ppu, nunits);
    }
    SFK_CHECK_LAUNCH();
    return SFK_OK;
  }
  // canonical slow stem geometry in bf16: the register-filter kernel above
  if ((sfk_tune().stem_v3 & 2) && y->dtype == SFK_BF16 && s->src_dtype == SFK_BF16 && s->cin == 3 && s->kt == 1 && y->c == 64 &&
      s->sw == 1 && (s->w_in % 8) == 0 && !((s->sn | s->sc | s->st | s->sh) & 7) && !(((uintptr_t)s->src) & 15) &&
      !(y->ld % 8) && !(y->c_off % 8) && stem_src_extent(s, y->n) < (1ll << 31) && k.t_log >= 2) {
    k.src_bytes = (uint32_t)stem_src_extent(s, y->n);
    k.y_bytes = 0;
    hipStream_t hs3 = static_cast<hipStream_t>(stream);
    const int ppu = 2;
    const int nfull = k.t_log / 2, tch = nfull <= ppu ? 1 : (nfull + ppu - 1) / ppu;
    const int nu = y->n * ((y->h + H2_ROWS - 1) / H2_ROWS) * k.tiles_w * tch;
    const int grid4 = nu < 512 ? nu : 512;
    const int lds4 = 4 * 3 * H2_PLANE + 24 * 1024 + 4 * 64 * 2 * 4 + 32 * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd_s4_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
    hipLaunchKernelGGL((stem_fwd_s4_kernel<3>), dim3((unsigned)grid4), dim3(256), lds4, hs3, k, ppu, nu, tch);
    SFK_CHECK_LAUNCH();
    return SFK_OK;
  }
  const int fn = (y->c + 15) / 16;
  const size_t esz = y->dtype == SFK_BF16 ? 2 : 4;
  const size_t lds = esz * ((size_t)16 * fn * (k.kp + 8) + (size_t)k.planes * PR * PC) + 4 * 16 * fn * 2 * sizeof(float);
  if (lds > 160 * 1024) return SFK_ERR_UNSUPPORTED;
  k.ntiles = y->n * y->t * k.tiles_h * k.tiles_w;
  const int resident = 256 * (int)((160 * 1024) / lds < 1 ? 1 : ((160 * 1024) / lds > 4 ? 4 : (160 * 1024) / lds));
  const dim3 grid((unsigned)(k.ntiles < 4 * resident ? k.ntiles : 4 * resident)), blk(256);
  hipStream_t hs = static_cast<hipStream_t>(stream);
#define SFK_STEM_FWD(T, S, FN)                                                                              \
  do {                                                                                                      \
    if (lds > 64 * 1024)                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd_kernel<T, S, FN>),                      \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
    hipLaunchKernelGGL((stem_fwd_kernel<T, S, FN>), grid, blk, lds, hs, k);                                 \
  } while (0)
#define SFK_STEM_FWD_FN(T, S)                                       \
  do {                                                              \
    if (fn == 1) SFK_STEM_FWD(T, S, 1);                             \
    else if (fn == 2) SFK_STEM_FWD(T, S, 2);                        \
    else SFK_STEM_FWD(T, S, 4);                                     \
  } while (0)
  if (fn == 3) return SFK_ERR_UNSUPPORTED;
  if (y->dtype == SFK_BF16) {
    if (s->src_dtype == SFK_BF16) SFK_STEM_FWD_FN(bf16_t, bf16_t); else SFK_STEM_FWD_FN(bf16_t, float);
  } else {
    if (s->src_dtype == SFK_BF16) SFK_STEM_FWD_FN(float, bf16_t); else SFK_STEM_FWD_FN(float, float);
  }
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}

extern "C" int sfk_stem_conv_wgrad(const sfk_stem_src* s, const sfk_fmap* dy, float* dw, sfk_stream_t stream) {
  if (!dw || !sfk_fmap_ok(dy)) return SFK_ERR_INVALID;
  StemK k;
  const int st = fill(s, dy->c, dy->t, dy->h, dy->w, k);
  if (st != SFK_OK) return st;
  if (!sfk_fmap_vec_ok(dy)) return SFK_ERR_UNSUPPORTED;
  k.w = nullptr; k.y = dy->ptr; k.yld = dy->ld; k.yoff = dy->c_off; k.stats = nullptr; k.dw = dw;
  k.ntiles = dy->n * dy->t * k.tiles_h * k.tiles_w;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  // canonical fast stem geometry in bf16 with 16-byte addressable rows: the input-frame-stationary kernel
  if (dy->dtype == SFK_BF16 && s->src_dtype == SFK_BF16 && s->cin == 3 && s->kt <= 5 && dy->c <= 8 && s->sw == 1 &&
      (s->w_in % 8) == 0 && !((s->sn | s->sc | s->st | s->sh) & 7) && !(((uintptr_t)s->src) & 15) &&
      stem_src_extent(s, dy->n) < (1ll << 31) && sfk_fmap_bytes(dy) < (1ll << 31)) {      // (padding = voffset 2^31)
    k.src_bytes = (uint32_t)stem_src_extent(s, dy->n);
    k.y_bytes = (uint32_t)sfk_fmap_bytes(dy);
    int blocks = 256 * 3;                       // 3 resident workgroups per CU (LDS 44 KB each)
    if (blocks > k.ntiles) blocks = k.ntiles;
    k.tiles_per_block = (k.ntiles + blocks - 1) / blocks;
    if (sfk_tune().stem_v3 & 4) {
      // whole FRAMES per workgroup: neighbouring workgroups (= neighbouring frames of a clip, one XCD) are then at the same tile
      // at the same time, and the kt input frames that read one dY tile find it in the L2 (ranges of 66 tiles drift by 17 tiles
      // per workgroup: 2.04 GB from HBM for 0.51 GB of operands)
      const int tpf = k.tiles_h * k.tiles_w;
      k.tiles_per_block = (k.tiles_per_block + tpf - 1) / tpf * tpf;
    }
    blocks = (k.ntiles + k.tiles_per_block - 1) / k.tiles_per_block;
    blocks = (blocks + 7) & ~7;                 // a multiple of 8 for the XCD-aware range order (surplus workgroups find no items)
    hipLaunchKernelGGL((stem_wgrad_v2_kernel<3>), dim3((unsigned)blocks), dim3(256), 0, hs, k);
    SFK_CHECK_LAUNCH();
    return SFK_OK;
  }
  int splits = (2048 + k.planes - 1) / k.planes;
  if (splits > k.ntiles) splits = k.ntiles;
  if (splits > 65535) splits = 65535;
  k.tiles_per_block = (k.ntiles + splits - 1) / splits;
  splits = (k.ntiles + k.tiles_per_block - 1) / k.tiles_per_block;
  const int fn = (dy->c + 15) / 16;
  if (fn == 3) return SFK_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)k.planes, (unsigned)splits), blk(256);
#define SFK_STEM_WG(T, S)                                                                   \
  do {                                                                                      \
    if (fn == 1) hipLaunchKernelGGL((stem_wgrad_kernel<T, S, 1>), grid, blk, 0, hs, k);     \
    else if (fn == 2) hipLaunchKernelGGL((stem_wgrad_kernel<T, S, 2>), grid, blk, 0, hs, k); \
    else hipLaunchKernelGGL((stem_wgrad_kernel<T, S, 4>), grid, blk, 0, hs, k);             \
  } while (0)
  if (dy->dtype == SFK_BF16) {
    if (s->src_dtype == SFK_BF16) SFK_STEM_WG(bf16_t, bf16_t); else SFK_STEM_WG(bf16_t, float);
  } else {
    if (s->src_dtype == SFK_BF16) SFK_STEM_WG(float, bf16_t); else SFK_STEM_WG(float, float);
  }
  SFK_CHECK_LAUNCH();
  return SFK_OK;
}
