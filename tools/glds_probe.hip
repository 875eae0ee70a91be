// Probe: does `buffer_load_dwordx4 ... offen lds` (LDS-DMA) write zeros for lanes whose offset is out of range?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned U4;
__global__ void k(const void* x, unsigned bytes, U4* out, const unsigned* offs) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  *reinterpret_cast<U4*>(sm + threadIdx.x * 16) = U4{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + wave * 1024), 16, offs[threadIdx.x], 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = *reinterpret_cast<U4*>(sm + threadIdx.x * 16);
}
int main() {
  const int n = 256;
  std::vector<unsigned> src(n * 4), offs(n);
  for (int i = 0; i < n * 4; ++i) src[i] = 1000 + i;
  for (int i = 0; i < n; ++i) offs[i] = (i % 5 == 3) ? 0xFFFFFFF0u : (unsigned)(((i * 7) % n) * 16);
  unsigned *dsrc, *doffs; U4* dout;
  hipMalloc(&dsrc, n * 16); hipMalloc(&doffs, n * 4); hipMalloc(&dout, n * 16);
  hipMemcpy(dsrc, src.data(), n * 16, hipMemcpyHostToDevice);
  hipMemcpy(doffs, offs.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, dsrc, (unsigned)(n * 16), dout, doffs);
  std::vector<unsigned> out(n * 4);
  hipMemcpy(out.data(), dout, n * 16, hipMemcpyDeviceToHost);
  int bad = 0, oobzero = 0, oobother = 0;
  for (int i = 0; i < n; ++i) {
    if (offs[i] == 0xFFFFFFF0u) {
      bool z = out[4*i]==0 && out[4*i+1]==0 && out[4*i+2]==0 && out[4*i+3]==0;
      if (z) ++oobzero; else { ++oobother; if (oobother < 4) printf("oob lane %d -> %08x %08x\n", i, out[4*i], out[4*i+1]); }
    } else {
      int s = (i * 7) % n;
      for (int j = 0; j < 4; ++j) if (out[4*i+j] != src[4*s+j]) ++bad;
    }
  }
  printf("inrange mismatches=%d oob_zero=%d oob_other=%d\n", bad, oobzero, oobother);
  return 0;
}
