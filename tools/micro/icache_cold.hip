// Microbenchmark: how fast does a wave get through STRAIGHT-LINE code it has never executed (instruction cache cold at the start
// of a launch), against the same code the second time round?  One kernel = a 2-trip loop around a block of N independent 8-byte
// VALU instructions; thread 0 of every workgroup stamps s_memtime before and after the block on both trips.  256 workgroups of
// 256 threads (one per CU), launched back to back like a decode GEMM.
//   hipcc --offload-arch=gfx950 -O3 -o icache_cold tools/micro/icache_cold.hip && ./icache_cold
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define BLOCK_OF(N) ".rept " #N "\n v_add_u32 %0, 0x12345678, %0\n v_add_u32 %1, 0x12345678, %1\n v_add_u32 %2, 0x12345678, %2\n v_add_u32 %3, 0x12345678, %3\n.endr\n"

template <int N>
__global__ __launch_bounds__(256) void code_walk(unsigned long long* stamps, int* sink) {
  int a = threadIdx.x, b = blockIdx.x, c = 3, d = 4;
  unsigned long long t[4];
#pragma unroll 1
  for (int trip = 0; trip < 2; ++trip) {
    t[2 * trip] = __builtin_amdgcn_s_memtime();
    if constexpr (N == 64) asm volatile(BLOCK_OF(64) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 256) asm volatile(BLOCK_OF(256) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if constexpr (N == 1024) asm volatile(BLOCK_OF(1024) : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    t[2 * trip + 1] = __builtin_amdgcn_s_memtime();
  }
  if (threadIdx.x == 0)
    for (int i = 0; i < 4; ++i) stamps[blockIdx.x * 4 + i] = t[i];
  if (a + b + c + d == 0x7fffffff) *sink = 1;
}

template <int N>
void run() {
  unsigned long long* d_stamps;
  int* d_sink;
  hipMalloc(&d_stamps, 256 * 4 * sizeof(unsigned long long));
  hipMalloc(&d_sink, 4);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(code_walk<N>, dim3(256), dim3(256), 0, 0, d_stamps, d_sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256 * 4);
  hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cold, warm;
  for (int b = 0; b < 256; ++b) {
    cold.push_back((double)(h[b * 4 + 1] - h[b * 4]));
    warm.push_back((double)(h[b * 4 + 3] - h[b * 4 + 2]));
  }
  std::sort(cold.begin(), cold.end());
  std::sort(warm.begin(), warm.end());
  const int instrs = N * 4;
  printf("%5d instructions (%6d bytes): first trip %8.0f clocks (%5.1f per instruction, %5.2f bytes/clock), second trip %7.0f clocks (%4.1f per instruction)\n",
         instrs, instrs * 8, cold[128], cold[128] / instrs, instrs * 8 / cold[128], warm[128], warm[128] / instrs);
  hipFree(d_stamps);
  hipFree(d_sink);
}

int main() {
  run<64>();
  run<256>();
  run<1024>();
  run<64>();
  return 0;
}
