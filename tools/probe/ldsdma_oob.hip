// Hardware probe: what does an out-of-range lane of `buffer_load_dwordx4 ... lds` do to its LDS slot?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const void* src, unsigned* out, unsigned nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned s[64 * 4 * 4];
  for (int i = threadIdx.x; i < 64 * 4 * 4; i += 256) s[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, (int)nbytes, 0x00020000);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // odd lanes are out of range; even lanes read 16 B at lane*16
  const unsigned off = (lane & 1) ? 0xFFFFFFFFu : (wave * 64 + lane) * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(s + wave * 256), 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 4 * 4; i += 256) out[i] = s[i];
}
int main() {
  unsigned *src, *out;
  const int n = 256 * 4;
  hipMalloc(&src, n * 4); hipMalloc(&out, n * 4);
  unsigned h[n]; for (int i = 0; i < n; ++i) h[i] = 0x1000 + i;
  hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, src, out, (unsigned)(n * 4));
  hipMemcpy(h, out, n * 4, hipMemcpyDeviceToHost);
  int zeros = 0, untouched = 0, good = 0, other = 0;
  for (int t = 0; t < 256; ++t) {
    unsigned v = h[t * 4];
    if (t & 1) { if (v == 0) ++zeros; else if (v == 0xDEADBEEFu) ++untouched; else ++other; }
    else { if (v == 0x1000u + t * 4) ++good; else ++other; }
  }
  printf("in-range lanes correct: %d/128 ; OOB lanes: zero-filled %d, untouched %d, other %d\n", good, zeros, untouched, other);
  printf("lane0..3 first words: %08x %08x %08x %08x\n", h[0], h[4], h[8], h[12]);
  return 0;
}
