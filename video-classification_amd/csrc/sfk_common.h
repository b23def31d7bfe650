// Shared device/host helpers for libsfk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sfk.h"

// the process-wide tuning table of sfk_init (optim_misc.hip); read-only after the first launch
__attribute__((visibility("hidden"))) const sfk_tuning& sfk_tune();
// conv_pw.hip: streaming pointwise conv with the fused output transform (dispatched to by sfk_conv_igemm)
__attribute__((visibility("hidden"))) int sfk_conv_pw_fused(const sfk_conv_desc* d, hipStream_t s);
__attribute__((visibility("hidden"))) int sfk_conv_pw_dgrad(const sfk_conv_desc* d, hipStream_t s);
__attribute__((visibility("hidden"))) int sfk_conv_pw_dgrad_rows(const sfk_conv_desc* d);

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define SFK_CHECK_LAUNCH()                                  \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return SFK_ERR_LAUNCH; \
  } while (0)

// q = n / d for 0 <= n < 2^31, d >= 1, via one mul-high (divisions by runtime extents are hot in the gathers)
struct FastDiv {
  uint32_t mul, shr, d;
  __host__ void set(int32_t denom) {
    d = (uint32_t)denom;
    if (denom <= 1) { mul = 0; shr = 0; return; }
    uint32_t lg = 0;
    while ((1u << lg) < (uint32_t)denom) ++lg;
    uint32_t p = 31 + lg;
    uint64_t m = ((1ull << p) + (uint64_t)denom - 1) / (uint64_t)denom;
    mul = (uint32_t)m;
    shr = p - 32;
  }
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return d <= 1 ? n : (__umulhi(n, mul) >> shr); }
  __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const {
    q = div(n);
    r = n - q * d;
  }
};

// 128-bit buffer resource over [base, base+bytes): raw buffer loads with a per-lane BYTE offset; lanes whose offset is
// >= bytes (we pass 0xFFFFFFFF for "this slot is conv padding / past the tile") read zeros -- no branch, no select,
// and the compiler can count the loads exactly for s_waitcnt vmcnt(N).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sfk_make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 sfk_buffer_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
constexpr uint32_t SFK_OOB = 0xFFFFFFFFu;

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int code = SFK_F32;
  static constexpr int VEC = 4;  // elements per 16 bytes
};
template <> struct DT<bf16_t> {
  static constexpr int code = SFK_BF16;
  static constexpr int VEC = 8;
};

// 16-byte vector of T <-> float[VEC]
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  float4 v;
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
  __device__ __forceinline__ void load_nt(const float* p) { load(p); }      // parity precision: plain accesses
  __device__ __forceinline__ void store_nt(float* p) const { store(p); }
  __device__ __forceinline__ float get(int i) const { return reinterpret_cast<const float*>(&v)[i]; }
  __device__ __forceinline__ void set(int i, float f) { reinterpret_cast<float*>(&v)[i] = f; }
  __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <> struct Vec16<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<bf16x8*>(p) = v; }
  // streaming variants (non-temporal hint): for tensors the kernel touches once and nobody re-reads soon
  __device__ __forceinline__ void load_nt(const bf16_t* p) { v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p)); }
  __device__ __forceinline__ void store_nt(bf16_t* p) const { __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p)); }
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float f) { v[i] = (bf16_t)f; }
  __device__ __forceinline__ void zero() {
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.f;
  }
};

static inline bool sfk_fmap_ok(const sfk_fmap* f) {
  return f && f->ptr && f->n > 0 && f->t > 0 && f->h > 0 && f->w > 0 && f->c > 0 && f->ld >= f->c_off + f->c &&
         f->c_off >= 0 && (f->dtype == SFK_F32 || f->dtype == SFK_BF16);
}
static inline int sfk_vec_of(int dtype) { return dtype == SFK_BF16 ? 8 : 4; }
static inline bool sfk_fmap_vec_ok(const sfk_fmap* f) {
  const int v = sfk_vec_of(f->dtype);
  return (f->c % v) == 0 && (f->ld % v) == 0 && (f->c_off % v) == 0 && (((uintptr_t)f->ptr) & 15) == 0;
}
static inline int64_t sfk_fmap_pixels(const sfk_fmap* f) { return (int64_t)f->n * f->t * f->h * f->w; }
// bytes from f->ptr to the end of the map's pixel records (the extent a buffer resource must cover)
static inline int64_t sfk_fmap_bytes(const sfk_fmap* f) {
  return sfk_fmap_pixels(f) * f->ld * (f->dtype == SFK_BF16 ? 2 : 4);
}
