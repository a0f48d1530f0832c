// Microbenchmark: what bounds the C-tile store burst of the 256x256-tile GEMM epilogue -- the CU's own store path, the XCD's
// L2 write ingest, or the chip?  Every ACTIVE workgroup (512 threads, one per CU: 100 KiB of LDS) stores one 256 x 256 bf16
// tile (128 KiB) of a [4096][11008] matrix from registers, 16 stores of 16 bytes per lane, in one of two patterns, and stamps
// s_memrealtime (100 MHz) before the first store, after the last store is ISSUED and after s_waitcnt vmcnt(0).
// Activity masks: all 256 workgroups; one per XCD; n of the 32 on one XCD; n on every XCD.
//   hipcc --offload-arch=gfx950 -O3 -o store_rate tools/micro/store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kN = 11008, kM = 4096;

// flavour: 0 plain, 1 sc1, 2 nt, 3 sc0 sc1, 4 sc1 nt
// pattern 0: accumulator layout of gemm_mfma.hip's epilogue (a 16-lane quarter writes 64 B in each of 4 rows ... per instruction
//            16 rows x 64 B); pattern 1: row-major, 8 rows x 128 B per instruction (whole lines)
template <int PATTERN, int FLAVOUR>
__global__ __launch_bounds__(512) void store_kernel(uint16_t* c, unsigned long long* stamps, int per_xcd, int xcd_mask, int reps) {
  __shared__ char lds[100 * 1024];
  if (threadIdx.x == 0) lds[0] = 0;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, idx = bid >> 3;
  const bool active = ((xcd_mask >> xcd) & 1) && idx < per_xcd;
  if (!active) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wr = wave >> 2, wc = wave & 3;
  i32x4 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = i32x4{lane + i, wave, bid, i};
  for (int r = 0; r < reps; ++r) {
    // a fresh tile every repetition, like the GEMM's rounds (16 x 43 tiles; lines that are not in L2)
    const int tile = (bid + 256 * r) % 688;
    const int tm = tile % 16, tn = tile / 16;
    uint16_t* base = c + (int64_t)(tm * 256 + wr * 128) * kN + tn * 256 + wc * 64;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      uint16_t* dst;
      if (PATTERN == 0) {
        const int g = lane >> 4, jm = lane & 15;
        const int nh = i >> 3, mt = i & 7;
        dst = base + (int64_t)(mt * 16 + jm) * kN + nh * 32 + 8 * g;
      } else {
        const int rr = lane >> 3, ch = lane & 7;
        dst = base + (int64_t)(i * 8 + rr) * kN + ch * 8;
      }
      if (FLAVOUR == 0) *(i32x4*)dst = v[i];
      else if (FLAVOUR == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v[i]) : "memory");
      else if (FLAVOUR == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v[i]) : "memory");
      else if (FLAVOUR == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v[i]) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(dst), "v"(v[i]) : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c2 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && r == reps - 1) {
      stamps[bid * 4 + 0] = t1 - t0;
      stamps[bid * 4 + 1] = t2 - t0;
      stamps[bid * 4 + 2] = t3 - t0;
      stamps[bid * 4 + 3] = c2 - c0;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i][0] += r;
  }
}

template <int PATTERN, int FLAVOUR>
void run(uint16_t* c, unsigned long long* d_st, const char* what, int per_xcd, int xcd_mask) {
  std::vector<unsigned long long> st(256 * 4, 0);
  hipMemset(d_st, 0, 256 * 4 * 8);
  hipLaunchKernelGGL((store_kernel<PATTERN, FLAVOUR>), dim3(256), dim3(512), 0, 0, c, d_st, per_xcd, xcd_mask, 20);
  hipDeviceSynchronize();
  hipMemcpy(st.data(), d_st, 256 * 4 * 8, hipMemcpyDeviceToHost);
  std::vector<double> issue, done, all, clk;
  for (int b = 0; b < 256; ++b)
    if (st[b * 4 + 1]) {
      issue.push_back(st[b * 4 + 0] / 100.0);
      done.push_back(st[b * 4 + 1] / 100.0);
      all.push_back(st[b * 4 + 2] / 100.0);
      clk.push_back((double)st[b * 4 + 3] / (st[b * 4 + 1] / 100.0) / 1000.0);
    }
  auto med = [](std::vector<double>& x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
  const double bytes = 128.0 * 1024;
  const double t = med(all);
  printf("%-34s pattern %d flavour %d: active %3zu  wave0 issue %.2f us  wave0 drained %.2f us  workgroup %.2f us (max %.2f)  %.1f B/clk/CU at %.2f GHz  %.2f TB/s total\n",
         what, PATTERN, FLAVOUR, all.size(), med(issue), med(done), t, all.back(), bytes / (t * 1e-6) / (med(clk) * 1e9), med(clk),
         bytes * all.size() / (t * 1e-6) / 1e12);
}

int main() {
  uint16_t* c;
  unsigned long long* d_st;
  hipMalloc(&c, (size_t)kM * kN * 2);
  hipMalloc(&d_st, 256 * 4 * 8);
  hipMemset(c, 0, (size_t)kM * kN * 2);
#define ALL(P, F)                                                \
  run<P, F>(c, d_st, "all 256 CUs", 32, 0xff);                   \
  run<P, F>(c, d_st, "16 per XCD, all XCDs", 16, 0xff);          \
  run<P, F>(c, d_st, "8 per XCD, all XCDs", 8, 0xff);            \
  run<P, F>(c, d_st, "2 per XCD, all XCDs", 2, 0xff);            \
  run<P, F>(c, d_st, "1 per XCD, all XCDs", 1, 0xff);            \
  run<P, F>(c, d_st, "32 on ONE XCD", 32, 0x01);                 \
  run<P, F>(c, d_st, "8 on ONE XCD", 8, 0x01);                   \
  run<P, F>(c, d_st, "1 on ONE XCD", 1, 0x01);
  for (int warm = 0; warm < 2; ++warm) run<0, 0>(c, d_st, "(warm-up)", 32, 0xff);
  ALL(0, 0)
  ALL(1, 0)
  ALL(0, 1)
  ALL(0, 2)
  ALL(1, 1)
  ALL(1, 2)
  ALL(1, 3)
  ALL(1, 4)
  return 0;
}
