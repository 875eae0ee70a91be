// Shader clock under load (development aid): hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/_bin/clock_probe
// A kernel runs `iters` rounds of 16 independent bf16 MFMAs per wave (8 waves per CU on every CU, or ONE wave on one CU) and
// stamps s_memtime (shader-clock cycles, CDNA ISA) around the loop; the host times the same launch with HIP events.
// cycles / seconds = the clock the chip actually holds under that load; 16 * iters MFMAs * 16 cycles / cycles = MFMA issue rate.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void mfma_spin(int iters, unsigned long long* out, float* sink) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(0.001f * (threadIdx.x + i));
    b[i] = (__bf16)(0.002f * (threadIdx.x - i));
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (s == 123.456f) sink[0] = s;
}

int main() {
  unsigned long long* d;
  float* sink;
  hipMalloc(&d, 4096 * 8);
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 200000;
  struct Cfg { int blocks, threads; const char* name; } cfgs[] = {{1, 64, "one wave on one CU"}, {256, 64, "one wave per CU"},
                                                                   {256, 512, "8 waves per CU, all CUs"}, {512, 512, "16 waves per CU"}};
  for (auto c : cfgs) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_spin, dim3(c.blocks), dim3(c.threads), 0, 0, iters, d, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(c.blocks);
      hipMemcpy(h.data(), d, c.blocks * 8, hipMemcpyDeviceToHost);
      double cyc = 0;
      for (auto v : h) cyc += (double)v;
      cyc /= c.blocks;
      const double waves_per_simd = c.threads / 64 / 4.0 * (c.blocks > 256 ? 2 : 1);
      const double tflops = (double)c.blocks * (c.threads / 64) * iters * 16.0 * 16 * 16 * 32 * 2 / (ms * 1e-3) / 1e12;
      printf("%-28s rep %d: %8.3f ms  %12.0f cycles/wave  => %.3f GHz   %7.1f TFLOP/s  (MFMA issue: %.2f of cycles per SIMD, %.2f waves/SIMD)\n",
             c.name, rep, ms, cyc, cyc / (ms * 1e6), tflops,
             16.0 * iters * 16 * (waves_per_simd < 1 ? 1 : waves_per_simd) / cyc, waves_per_simd);
    }
  }
  return 0;
}
