// Diagnostic harness (not part of the library): conv_first2_kernel built with -DDSR_F2_STAMPS at 32 x 512 x 512 on random data;
// prints the launch time and, per phase of a tile, the cycles wave 0 of a block spends there (median over blocks).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDSR_F2_STAMPS -I deep-super-resolution_amd/csrc -I include tools/diag_first2.cpp -o tools/_bin/diag_first2
//   tools/_bin/diag_first2 [keep: 1 writes the first layer's activation, 0 does not]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../deep-super-resolution_amd/csrc/conv_first2.hip"

int dsr_launch_status(const char*) { return 0; }
int dsr_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
  return code;
}

int main(int argc, char** argv) {
  const int keep = argc > 1 ? atoi(argv[1]) : 1;
  const int N = 32, H = 512, W = 512;
  const size_t nx = (size_t)N * H * W * 8, na = (size_t)N * H * W * 64, ny = (size_t)N * (H / 2) * (W / 2) * 64;
  std::vector<unsigned short> h(nx);
  unsigned x = 12345u;
  for (size_t i = 0; i < nx; ++i) {
    x = x * 1664525u + 1013904223u;
    h[i] = (i & 7) < 3 ? (unsigned short)(0x3c00u + ((x >> 9) & 0x3ffu) + ((x >> 31) << 15)) : (unsigned short)0;
  }
  void *dx, *w0, *w1, *a0, *y1;
  float* stats;
  hipMalloc(&dx, nx * 2);
  hipMalloc(&w0, 1 << 20);
  hipMalloc(&w1, 1 << 20);
  hipMalloc(&a0, na * 2);
  hipMalloc(&y1, ny * 2);
  hipMalloc((void**)&stats, 512 * 2 * 64 * 4);
  hipMemcpy(dx, h.data(), nx * 2, hipMemcpyHostToDevice);
  hipMemcpy(w0, h.data(), 1 << 20, hipMemcpyHostToDevice);
  hipMemcpy(w1, h.data(), 1 << 20, hipMemcpyHostToDevice);
  dsr_conv_desc d0{DSR_BF16, N, H, W, 3, 64, 3, 3, 1, 1, DSR_PAD_ZERO}, d1{DSR_BF16, N, H, W, 64, 64, 3, 3, 2, 1, DSR_PAD_ZERO};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) dsr_conv_first2_fwd(&d0, &d1, dx, w0, nullptr, 0.2f, w1, nullptr, keep ? a0 : nullptr, y1, stats, 0);
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) dsr_conv_first2_fwd(&d0, &d1, dx, w0, nullptr, 0.2f, w1, nullptr, keep ? a0 : nullptr, y1, stats, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("32 x 512 x 512, activation %s: %.3f ms\n", keep ? "kept" : "not kept", ms / reps);
#ifdef DSR_F2_STAMPS
  unsigned long long st[256][12];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(f2_stamps), sizeof(st));
  const char* names[12] = {"wait for the image halo (+ stores)", "barrier at the top", "stage 1 (first layer -> LDS)", "barrier after stage 1",
                           "activation -> HBM", "stage 2 (64 -> 64, stride 2)", "barriers around the epilogue", "epilogue: C tile + statistics",
                           "(loop overhead)", "statistics fold + y1 stores", "tiles", "whole block"};
  for (int k = 0; k < 12; ++k) {
    std::vector<unsigned long long> v;
    for (int b = 0; b < 256; ++b) v.push_back(st[b][k]);
    std::sort(v.begin(), v.end());
    printf("  %-38s median %10llu   min %10llu   max %10llu\n", names[k], v[128], v[0], v[255]);
  }
#endif
  return 0;
}
