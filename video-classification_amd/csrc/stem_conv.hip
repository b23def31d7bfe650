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
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) v[i] = (base >= 0 && ps.ok[i]) ? ldsrc<S>(k.src, base + ps.off[i]) : 0.f;
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
      plane_fetch<S>(k, ps, plane_base(k, pl, n, to), v0);
      plane_fetch<S>(k, ps, plane_base(k, pl + 1, n, to), v1);
      plane_fetch<S>(k, ps, plane_base(k, pl + 2, n, to), v2);
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
      dv[i] = make_uint4(0, 0, 0, 0);
      if (ho < k.ho && wo < k.wo && seg * VEC < k.cout)
        dv[i] = *reinterpret_cast<const uint4*>(dyp + ((((int64_t)n * k.t_out + to) * k.ho + ho) * k.wo + wo) * k.yld + k.yoff + seg * VEC);
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
  int splits = (2048 + k.planes - 1) / k.planes;
  if (splits > k.ntiles) splits = k.ntiles;
  if (splits > 65535) splits = 65535;
  k.tiles_per_block = (k.ntiles + splits - 1) / splits;
  splits = (k.ntiles + k.tiles_per_block - 1) / k.tiles_per_block;
  const int fn = (dy->c + 15) / 16;
  if (fn == 3) return SFK_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)k.planes, (unsigned)splits), blk(256);
  hipStream_t hs = static_cast<hipStream_t>(stream);
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
