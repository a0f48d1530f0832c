// Microbenchmark: which fp8 MFMA shape delivers more FLOP/s on a power-limited MI355X?  Bare loops, operands in registers
// (random e4m3 bit patterns from the benchmark's distribution: bytes of (0.25 * rand) cast to e4m3), one or two waves per SIMD,
// every CU busy, same output tile per wave (128 x 64: 32 accumulators of 16x16 or 8 of 32x32).
//   v_mfma_f32_16x16x128_f8f6f4 (what the scaled GEMM uses)  vs  v_mfma_f32_32x32x64_f8f6f4
// Reports TFLOP/s and the clock held (s_memtime / s_memrealtime).  MI355X_MICROARCH.md "DVFS give-back" item 7 has the bf16
// answer (16x16x32 ahead by 1.15x); this is the fp8 one.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape tools/micro/mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef int i32x4 __attribute__((ext_vector_type(4)));

// LDS = true: the twelve fragments are re-read from LDS (random bytes, another 24 KiB window every iteration, conflict-free
// ds_read_b128) before every 32 MFMAs, as a GEMM K loop does: the operands CHANGE, which is what the matrix pipe's power depends on
template <int SHAPE, bool LDS>
__global__ __launch_bounds__(512, 2) void mfma_kernel(const int* __restrict__ src, float* __restrict__ out, unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(1024))) char lds[128 * 1024];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (LDS) {
    for (int i = threadIdx.x; i < 128 * 1024 / 16; i += blockDim.x)
      ((i32x4*)lds)[i] = i32x4{src[(i * 4 + blockIdx.x * 131) & 0xfffff], src[(i * 4 + 1 + blockIdx.x * 131) & 0xfffff],
                               src[(i * 4 + 2 + blockIdx.x * 131) & 0xfffff], src[(i * 4 + 3 + blockIdx.x * 131) & 0xfffff]};
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // fragments: 8 row fragments + 4 column fragments of 32 bytes per lane (16x16x128) -- or 4 + 2 of 32 bytes per lane (32x32x64,
  // two K halves) -- the registers a wave of the GEMM holds for one K step
  i32x8 fa[8], fb[4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[i][j] = src[(tid * 97 + i * 8 + j) & 0xfffff];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[i][j] = src[(tid * 53 + 4096 + i * 8 + j) & 0xfffff];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
      if (LDS) {
        const int base = ((it * 24576 + wave * 3072) & (128 * 1024 - 1)) & ~1023;
#pragma unroll
        for (int f = 0; f < 12; ++f) {
          const int off = (base + f * 2048 + lane * 16) & (128 * 1024 - 1);
          const i32x4 lo = *(const i32x4*)(lds + off), hi = *(const i32x4*)(lds + (off ^ 1024));
          i32x8& dst = f < 8 ? fa[f] : fb[f - 8];
          dst = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i], acc[i][j], 0, 0, 0, 0, 0, 0);
      // rotate the fragments so that the data keeps changing like a K loop's (cheap VALU, once per 32 MFMAs)
      fa[0][0] ^= it;  // (a static index: a runtime-indexed register array would go to scratch)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (LDS) {
        const int base = ((it * 24576 + wave * 3072) & (128 * 1024 - 1)) & ~1023;
#pragma unroll
        for (int f = 0; f < 12; ++f) {
          const int off = (base + f * 2048 + lane * 16) & (128 * 1024 - 1);
          const i32x4 lo = *(const i32x4*)(lds + off), hi = *(const i32x4*)(lds + (off ^ 1024));
          i32x8& dst = f < 8 ? fa[f] : fb[f - 8];
          dst = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
      // 128 x 64 outputs x 128 K bytes = 4 x 2 tiles x 2 K halves: the same FLOPs as the 32 MFMAs above
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[2 * h + j], fa[4 * h + i], acc[i][j], 0, 0, 0, 0, 0, 0);
      fa[0][0] ^= it;  // (a static index: a runtime-indexed register array would go to scratch)
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][15];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  out[tid] = sum;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t1 - t0;
    stamps[blockIdx.x * 2 + 1] = c1 - c0;
  }
}

// The same loop with a 128 x 128 output tile per wave and ONE wave per SIMD (256 threads, 64 accumulators = 256 AGPRs): 16
// fragments re-read per 64 MFMAs instead of 12 per 32 -- a third fewer LDS bytes per FLOP at the same workgroup tile (256 x 256)
template <bool LDS>
__global__ __launch_bounds__(256, 1) void mfma_big_kernel(const int* __restrict__ src, float* __restrict__ out, unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(1024))) char lds[128 * 1024];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (LDS) {
    for (int i = threadIdx.x; i < 128 * 1024 / 16; i += blockDim.x)
      ((i32x4*)lds)[i] = i32x4{src[(i * 4 + blockIdx.x * 131) & 0xfffff], src[(i * 4 + 1 + blockIdx.x * 131) & 0xfffff],
                               src[(i * 4 + 2 + blockIdx.x * 131) & 0xfffff], src[(i * 4 + 3 + blockIdx.x * 131) & 0xfffff]};
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  i32x8 fa[8], fb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      fa[i][j] = src[(tid * 97 + i * 8 + j) & 0xfffff];
      fb[i][j] = src[(tid * 53 + 4096 + i * 8 + j) & 0xfffff];
    }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
      const int base = ((it * 32768 + wave * 4096) & (128 * 1024 - 1)) & ~1023;
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        const int off = (base + f * 2048 + lane * 16) & (128 * 1024 - 1);
        const i32x4 lo = *(const i32x4*)(lds + off), hi = *(const i32x4*)(lds + (off ^ 1024));
        i32x8& dst = f < 8 ? fa[f] : fb[f - 8];
        dst = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i], acc[i][j], 0, 0, 0, 0, 0, 0);
    fa[0][0] ^= it;
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][3];
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  out[tid] = sum;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t1 - t0;
    stamps[blockIdx.x * 2 + 1] = c1 - c0;
  }
}

// ... and software-pipelined, as a one-wave-per-SIMD K loop has to be: while the eight MFMAs of column fragment j run, the wave
// fetches column fragment j + 1 and row fragment j of the NEXT K step (3 KiB of the 32 KiB per step), so no MFMA waits for LDS
__global__ __launch_bounds__(256, 1) void mfma_big_pipelined_kernel(const int* __restrict__ src, float* __restrict__ out, unsigned long long* stamps,
                                                                     int iters) {
  __shared__ __attribute__((aligned(1024))) char lds[128 * 1024];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = threadIdx.x; i < 128 * 1024 / 16; i += blockDim.x)
    ((i32x4*)lds)[i] = i32x4{src[(i * 4 + blockIdx.x * 131) & 0xfffff], src[(i * 4 + 1 + blockIdx.x * 131) & 0xfffff],
                             src[(i * 4 + 2 + blockIdx.x * 131) & 0xfffff], src[(i * 4 + 3 + blockIdx.x * 131) & 0xfffff]};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto frag = [&](int off) {
    off &= 128 * 1024 - 1;
    const i32x4 lo = *(const i32x4*)(lds + off), hi = *(const i32x4*)(lds + (off ^ 1024));
    return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  i32x8 fa[8], fa_next[8], fb, fb_next;
#pragma unroll
  for (int i = 0; i < 8; ++i) fa[i] = frag(wave * 4096 + i * 2048 + lane * 16);
  fb = frag(wave * 4096 + 16384 + lane * 16);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const int base = (((it + 1) * 32768 + wave * 4096) & (128 * 1024 - 1)) & ~1023;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      fb_next = frag((j < 7 ? base - 32768 : base) + 16384 + ((j + 1) & 7) * 2048 + lane * 16);
      fa_next[j] = frag(base + j * 2048 + lane * 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa[i], acc[i][j], 0, 0, 0, 0, 0, 0);
      fb = fb_next;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = fa_next[i];
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][3];
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  out[tid] = sum;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t1 - t0;
    stamps[blockIdx.x * 2 + 1] = c1 - c0;
  }
}

template <int MODE>
void run_big(const int* src, float* out, unsigned long long* d_st, const char* what) {
  const int blocks = 256, iters = 20000, threads = 256;
  std::vector<unsigned long long> st(blocks * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best_ms = 1e30f;
  double clk = 0;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    if (MODE == 2) hipLaunchKernelGGL(mfma_big_pipelined_kernel, dim3(blocks), dim3(threads), 0, 0, src, out, d_st, iters);
    else hipLaunchKernelGGL((mfma_big_kernel<MODE == 1>), dim3(blocks), dim3(threads), 0, 0, src, out, d_st, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 3 && ms < best_ms) best_ms = ms;
    hipMemcpy(st.data(), d_st, blocks * 16, hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < blocks; ++b) s += (double)st[2 * b + 1] / st[2 * b] * 0.1;
    clk = s / blocks;
  }
  const double flops = blocks * 4.0 * iters * 64.0 * 2 * 16 * 16 * 128;
  printf("%-28s 1 wave  per SIMD: %7.2f ms  %7.1f TFLOP/s  (%.3f of 5 PF)  clock %.2f GHz\n", what, best_ms, flops / best_ms / 1e9,
         flops / best_ms / 1e9 / 5000, clk);
}

template <int SHAPE, bool LDS>
void run(const int* src, float* out, unsigned long long* d_st, int threads, const char* what) {
  const int blocks = 256, iters = 40000;
  std::vector<unsigned long long> st(blocks * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best_ms = 1e30f;
  double clk = 0;
  for (int rep = 0; rep < 6; ++rep) {  // the last repetitions run on a chip that has been loaded for a second
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_kernel<SHAPE, LDS>), dim3(blocks), dim3(threads), 0, 0, src, out, d_st, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 3 && ms < best_ms) best_ms = ms;
    hipMemcpy(st.data(), d_st, blocks * 16, hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < blocks; ++b) s += (double)st[2 * b + 1] / st[2 * b] * 0.1;
    clk = s / blocks;
  }
  const double waves = blocks * (threads / 64.0);
  const double flops = waves * iters * 32.0 * 2 * 16 * 16 * 128;
  printf("%-28s %d waves per SIMD: %7.2f ms  %7.1f TFLOP/s  (%.3f of 5 PF)  clock %.2f GHz\n", what, threads / 256, best_ms, flops / best_ms / 1e9,
         flops / best_ms / 1e9 / 5000, clk);
}

int main() {
  const int n = 1 << 20;
  std::vector<int> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) {  // four e4m3 codes of 0.25 * U[0,1): sign 0, exponent <= 0b0101 (0.25 = 2^-2 -> biased 5), random mantissa
    unsigned w = 0;
    for (int b = 0; b < 4; ++b) {
      const double v = 0.25 * (rand() / (RAND_MAX + 1.0));
      int e = 0;
      double m = v;
      unsigned code = 0;
      if (v >= 0.001953125) {  // normal range of e4m3 starts at 2^-6
        e = 0;
        while (m < 1.0) { m *= 2; --e; }
        const int be = e + 7;
        const int mant = (int)((m - 1.0) * 8 + 0.5);
        code = be > 0 ? (unsigned)((be << 3) + (mant > 7 ? 7 : mant)) : (unsigned)(v / 0.001953125 + 0.5);
      } else {
        code = (unsigned)(v / 0.001953125 + 0.5);
      }
      w |= (code & 0x7f) << (8 * b);
    }
    h[i] = (int)w;
  }
  int* src;
  float* out;
  unsigned long long* d_st;
  hipMalloc(&src, n * 4);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&d_st, 256 * 16);
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int round = 0; round < 2; ++round) {
    run<16, false>(src, out, d_st, 512, "16x16x128, operands fixed");
    run<32, false>(src, out, d_st, 512, "32x32x64, operands fixed");
    run<16, true>(src, out, d_st, 512, "16x16x128, LDS re-read");
    run<32, true>(src, out, d_st, 512, "32x32x64, LDS re-read");
    run_big<0>(src, out, d_st, "16x16x128 128x128/wave fixed");
    run_big<1>(src, out, d_st, "16x16x128 128x128/wave LDS");
    run_big<2>(src, out, d_st, "  ... software-pipelined");
  }
  hipMemset(src, 0, n * 4);
  run<16, true>(src, out, d_st, 512, "16x16x128 LDS, ALL-ZERO data");
  run<32, true>(src, out, d_st, 512, "32x32x64 LDS, ALL-ZERO data");
  return 0;
}
