// HBM streaming ceilings for the access mixes of the pointwise kernels (development aid; build: hipcc --offload-arch=gfx950 -O3
// tools/stream_probe.hip -o tools/_bin/stream_probe).  Each kernel moves 16 B per lane per access and does the bf16 unpack /
// fma / pack arithmetic of bn_act_fwd, so the numbers are what a perfectly simple kernel of that shape reaches on this chip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned U4;

__device__ __forceinline__ U4 xform(U4 v, float s, float b) {
  unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
    lo = lo * s + b;
    hi = hi * s + b;
    lo = lo >= 0.f ? lo : 0.2f * lo;
    hi = hi >= 0.f ? hi : 0.2f * hi;
    __bf16 l = (__bf16)lo, h = (__bf16)hi;
    unsigned short ls, hs;
    __builtin_memcpy(&ls, &l, 2);
    __builtin_memcpy(&hs, &h, 2);
    w[i] = (unsigned)ls | ((unsigned)hs << 16);
  }
  return U4{w[0], w[1], w[2], w[3]};
}

template <int READS, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const U4* __restrict__ a, const U4* __restrict__ b, U4* __restrict__ out,
                                                     size_t nvec, float s, float sh) {
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (size_t i0 = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i0 < nvec; i0 += stride) {
    U4 va[UNROLL], vb[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t i = i0 + (size_t)u * 256;
      if (i < nvec) {
        va[u] = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (READS == 2) vb[u] = NT ? __builtin_nontemporal_load(b + i) : b[i];
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t i = i0 + (size_t)u * 256;
      if (i < nvec) {
        U4 r = xform(va[u], s, sh);
        if (READS == 2) {
          r.x ^= vb[u].x & 1u;   // keep the second stream live (cheap)
          r.y ^= vb[u].y & 1u;
        }
        if (NT) __builtin_nontemporal_store(r, out + i); else out[i] = r;
      }
    }
  }
}

template <int READS, int UNROLL, bool NT>
static void run(const char* name, const U4* a, const U4* b, U4* out, size_t nvec, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream_kernel<READS, UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, a, b, out, nvec, 1.1f, 0.1f);
  hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stream_kernel<READS, UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, a, b, out, nvec, 1.1f, 0.1f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nvec * 16 * (READS + 1) * reps;
  printf("%-34s blocks %6d : %6.2f TB/s\n", name, blocks, bytes / (ms * 1e-3) / 1e12);
}

int main() {
  for (size_t mb : {67, 537}) {   // one tensor of 67 MB (trunk: everything fits the 256 MB Infinity Cache) / 537 MB (D.b1)
    const size_t nvec = mb * 1000 * 1000 / 16;
    U4 *a, *b, *o;
    hipMalloc(&a, nvec * 16);
    hipMalloc(&b, nvec * 16);
    hipMalloc(&o, nvec * 16);
    hipMemset(a, 1, nvec * 16);
    hipMemset(b, 1, nvec * 16);
    printf("== tensor %zu MB\n", mb);
    for (int blocks : {2048, 4096, 16384, 65536}) {
      run<1, 1, false>("1R1W unroll1", a, b, o, nvec, blocks);
      run<1, 2, false>("1R1W unroll2", a, b, o, nvec, blocks);
      run<1, 4, false>("1R1W unroll4", a, b, o, nvec, blocks);
      run<1, 2, true>("1R1W unroll2 nontemporal", a, b, o, nvec, blocks);
      run<2, 1, false>("2R1W unroll1", a, b, o, nvec, blocks);
      run<2, 2, false>("2R1W unroll2", a, b, o, nvec, blocks);
      run<2, 2, true>("2R1W unroll2 nontemporal", a, b, o, nvec, blocks);
    }
    hipFree(a);
    hipFree(b);
    hipFree(o);
  }
  return 0;
}
