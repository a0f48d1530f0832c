// Measurement helper (NOT a product path): what this GPU, on this data, sustains in a bare MFMA loop fed from LDS -- the ceiling
// beside the datasheet peak that bench.py reports as `roofline.measured_peak` ("builder must cite and also
// measure an achievable-peak microbenchmark").  Built into libconch_micro.so, loaded only by bench.py.
//
// The loop is the register / LDS traffic of one K step of the product kernels and nothing else: a wave owns a 128 x 64 output
// tile (32 accumulators of 16 x 16), re-reads its twelve 32-byte fragments from LDS with conflict-free ds_read_b128 (another 24 KiB
// window every iteration, so the operands CHANGE as a K loop's do: the matrix pipe's power depends on it) and issues the MFMAs
// of the step: 32 x v_mfma_f32_16x16x128_f8f6f4 (mode 0, scaled GEMM) or 64 x v_mfma_f32_16x16x32_f16 (mode 1, mixed GEMM).  No
// global traffic, no barrier, no epilogue.  512-thread workgroups, one per CU, two waves per SIMD -- the product kernels' shape.
// LDS is filled from the caller's buffer: the benchmark's own operand bytes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kLds = 128 * 1024;

template <int MODE>
__global__ __launch_bounds__(512, 2) void micro_peak_kernel(const int* __restrict__ src, uint32_t mask, float* __restrict__ out,
                                                            unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(1024))) char lds[kLds];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = threadIdx.x; i < kLds / 16; i += blockDim.x) {
    const uint32_t w = (uint32_t)(i * 4 + blockIdx.x * 8191 * 4);
    ((i32x4*)lds)[i] = i32x4{src[w & mask], src[(w + 1) & mask], src[(w + 2) & mask], src[(w + 3) & mask]};
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  i32x8 fa[8], fb[4];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int base = ((it * 24576 + wave * 3072) & (kLds - 1)) & ~1023;
#pragma unroll
    for (int f = 0; f < 12; ++f) {
      const int off = (base + f * 2048 + lane * 16) & (kLds - 1);
      const i32x4 lo = *(const i32x4*)(lds + off), hi = *(const i32x4*)(lds + (off ^ 1024));
      i32x8& dst = f < 8 ? fa[f] : fb[f - 8];
      dst = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (MODE == 0) {
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[j], fa[i], acc[i][j], 0, 0, 0, 0, 0, 0);
        } else {
          const i32x4 alo = {fa[i][0], fa[i][1], fa[i][2], fa[i][3]}, ahi = {fa[i][4], fa[i][5], fa[i][6], fa[i][7]};
          const i32x4 blo = {fb[j][0], fb[j][1], fb[j][2], fb[j][3]}, bhi = {fb[j][4], fb[j][5], fb[j][6], fb[j][7]};
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, blo), __builtin_bit_cast(f16x8, alo), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bhi), __builtin_bit_cast(f16x8, ahi), acc[i][j], 0, 0, 0);
        }
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
  out[tid] = sum;
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t1 - t0;
    stamps[blockIdx.x * 2 + 1] = c1 - c0;
  }
}

struct Scratch {
  float* out = nullptr;
  unsigned long long* stamps = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int blocks = 0;
};

Scratch& scratch(int blocks) {
  static Scratch s;
  if (s.blocks < blocks) {
    if (s.out) (void)hipFree(s.out);
    if (s.stamps) (void)hipFree(s.stamps);
    if (hipMalloc(&s.out, (size_t)blocks * 512 * 4) != hipSuccess || hipMalloc(&s.stamps, (size_t)blocks * 16) != hipSuccess) {
      s.out = nullptr;
      s.stamps = nullptr;
      s.blocks = 0;
      return s;
    }
    s.blocks = blocks;
  }
  if (!s.e0) {
    (void)hipEventCreate(&s.e0);
    (void)hipEventCreate(&s.e1);
  }
  return s;
}

}  // namespace

// mode 0: fp8 e4m3 16x16x128 (scaled GEMM's MFMA), mode 1: fp16 16x16x32 (mixed GEMM's).  `src`: device buffer of `words`
// 4-byte words (a power of two) holding operand bytes of the benchmark's distribution.  One launch of `iters` loop iterations on
// every CU; returns its duration (HIP events on `stream`), the FLOPs it performed and the median in-loop clock
// (d s_memtime / d s_memrealtime x 100 MHz over the workgroups).  0 = ok, else a hipError_t.
extern "C" int conch_micro_mfma_peak(const void* src, int64_t words, int mode, int iters, void* stream, float* ms, double* flops,
                                     float* clock_mhz) {
  if (!src || words < 1024 || (words & (words - 1)) || iters < 1 || (mode != 0 && mode != 1)) return (int)hipErrorInvalidValue;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return (int)hipErrorInvalidDevice;
  Scratch& s = scratch(cus);
  if (!s.out) return (int)hipErrorOutOfMemory;
  hipStream_t st = (hipStream_t)stream;
  (void)hipEventRecord(s.e0, st);
  if (mode == 0) hipLaunchKernelGGL(micro_peak_kernel<0>, dim3(cus), dim3(512), 0, st, (const int*)src, (uint32_t)(words - 1), s.out, s.stamps, iters);
  else hipLaunchKernelGGL(micro_peak_kernel<1>, dim3(cus), dim3(512), 0, st, (const int*)src, (uint32_t)(words - 1), s.out, s.stamps, iters);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  (void)hipEventRecord(s.e1, st);
  e = hipEventSynchronize(s.e1);
  if (e != hipSuccess) return (int)e;
  (void)hipEventElapsedTime(ms, s.e0, s.e1);
  std::vector<unsigned long long> h((size_t)cus * 2);
  e = hipMemcpy(h.data(), s.stamps, (size_t)cus * 16, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return (int)e;
  std::vector<double> clk;
  for (int b = 0; b < cus; ++b)
    if (h[2 * b]) clk.push_back((double)h[2 * b + 1] / (double)h[2 * b] * 100.0);
  std::sort(clk.begin(), clk.end());
  *clock_mhz = clk.empty() ? 0.f : (float)clk[clk.size() / 2];
  // per wave and iteration: 32 accumulator tiles x (16 x 16 outputs x 128 K bytes of fp8 | 64 K elements of fp16) x 2
  const double per_wave_iter = 32.0 * 2 * 16 * 16 * (mode == 0 ? 128 : 64);
  *flops = (double)cus * 8.0 * iters * per_wave_iter;
  return 0;
}
