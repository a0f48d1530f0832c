// mixed_precision_gemm for decode batches (M <= 64): packed int4 / int8 weights straight from global memory into the
// MFMA operand registers, activations staged once per workgroup, K split across workgroups.
//
// The LDS-tiled kernel (gemm_mixed.hip) walks K in 64-element steps over a 256-row tile: for a handful of rows it is a
// latency chain of K/64 steps on N/192 CUs (~55 us at K = 4096 whatever M is) while the weights -- 22.5 MB at C4's N, K --
// could stream in a few microseconds.  Here:
//   * workgroup = (64-column block, 1024-element K slice), 4 waves x 16 columns; grid = N/64 x K/1024 workgroups;
//   * a lane of the MFMA's weight operand is (column n0 + lane%16, k-group lane/16): its 8 consecutive k are exactly ONE
//     packed int4 word (two int8 words) of the [K/pf][N] tensor -- 16 lanes read 64 contiguous bytes of a word row -- so
//     the weights need no LDS and no transposition: load, dequantise (mixed_dequant.hpp, bit-identical to w_ref), multiply;
//   * the whole slice is requested up front (2 x 16 words, scales / zero points per K step, and the slice of X by LDS-DMA
//     into the same swizzled 8-row x 128-byte image the other kernels use), then consumed;
//   * partial sums go to fp32 slabs [slice][M][N] in library scratch; mixed_skinny_reduce_kernel adds the slices in a
//     fixed order and casts (the reference's result type, kernels/quantization/gemm.py:482-545).
// ROWS (16 / 32 / 64) = rows of X a workgroup stages and multiplies.
#include <algorithm>

#include "common.hpp"
#include "gemm.hpp"
#include "mfma_tile.hpp"
#include "mixed_dequant.hpp"

namespace conch {
namespace {

using namespace tile;
using mixed::ChunkDequant;

constexpr int kMsThreads = 256;
constexpr int kMsN = 64;                  // columns per workgroup
constexpr int kMsStepK = 64;              // k elements per step (128 bytes of fp16 / bf16)
#ifdef CONCH_EXP_MS_STEPS
constexpr int kMsSteps = CONCH_EXP_MS_STEPS;  // (experiment builds: tools/ab_strip_parts.py)
#else
constexpr int kMsSteps = 16;              // steps per slice
#endif
constexpr int kMsSliceK = kMsSteps * kMsStepK;
constexpr int kMsMaxM = 256;
constexpr int kMsMaxTiles = 16384;  // arrival counters per (device, stream): the 64 KiB slot gemm_skinny.hip uses too

template <int X_DT, int BITS, int ZP>
__device__ __forceinline__ i32x4 dequant8(uint32_t w0, uint32_t w1, int off, uint32_t scale_bits, uint32_t mask, uint32_t magic, uint32_t mask_hi) {
  ChunkDequant<X_DT, BITS, (X_DT == CONCH_DT_BF16 && BITS == 8 && ZP != CONCH_ZP_NONE)> cv;
#pragma unroll
  for (int s = 0; s + 1 < ChunkDequant<X_DT, BITS>::kSlices; ++s) cv.slice(s, w0, w1, off, scale_bits, nullptr, mask, magic, mask_hi);
  return cv.out;
}

// FUSED = the reduce of the slices runs in the same launch (same hand-off as gemm_skinny.hip's one-launch form, the first row of
// MI355X_MICROARCH.md's table of measured hand-offs): every wave stores its partial tile write-through (sc1) and drains its own
// stores, one lane draws a ticket from the (tile, wave) agent-scope counter -- per wave, no workgroup barrier: wave w of every
// slice's workgroup owns the same 16 columns --; the wave that draws the last ticket re-reads all slices with sc1 loads, adds them IN SLICE ORDER -- the order of the reduce kernel: bit-identical results --
// casts, stores C and puts the counter back to zero.  At GEMV sizes the slab tile of a workgroup is 4 KiB: the second launch
// costs more than the last arriver's pass over the slices.
#ifdef CONCH_CLOCK_PROBE
__device__ unsigned long long g_probe_mixed_skinny[kProbeBlocks * 8];
#define MS_STAMP(slot) CONCH_STAMP(g_probe_mixed_skinny, slot, (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)))
#define MS_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define MS_STAMP(slot) do { } while (0)
#define MS_DRAIN() do { } while (0)
#endif

// WLDS (int4 words, 16-byte aligned word rows): the packed words take the LDS-DMA path too.  A 4-byte load per lane moves 256
// bytes per wave instruction and the address path charges a wave instruction 16 cycles whatever its width (profiles/r03/
// probe_mixed_skinny.txt: 56 vector-memory instructions per wave, 224 per CU = 1.7 of the 2.8 us the slice's loads take to issue).
// Lane L fetches word row L >> 2 of a PAIR of steps, columns 4 (L & 3) .. + 3 (16 bytes; 8 instructions per wave instead of 32)
// into the wave's 8 KiB of LDS -- the image is row-major [16 word rows][16 columns], so operand lane (column r, k-group g) reads
// dword 16 (8 (s & 1) + 4 h + g) + r: 64 different banks.
template <int X_DT, int BITS, int ZP, int ROWS, bool FUSED = false, int OUT_DT = CONCH_DT_FP16, bool WLDS = false>
__global__ __launch_bounds__(kMsThreads) void mixed_skinny_kernel(MixedGemmArgs p, float* __restrict__ slabs, unsigned* __restrict__ counters) {
  static_assert(!WLDS || BITS == 4, "the word staging is built for int4 (one word per lane and half step)");
  constexpr int kUnit = ROWS * 128;            // one K step of X
  constexpr int kWpc = BITS == 4 ? 1 : 2;      // 32-bit words per 8-k chunk
  constexpr int kWordRows = kMsStepK * BITS / 32;  // word rows per K step (8 / 16)
  constexpr int kWBytes = WLDS ? 4 * (kMsSteps / 2) * 1024 : 0;  // per wave: one KiB per pair of steps
  __shared__ __attribute__((aligned(1024))) char lds[kMsSteps * kUnit + kWBytes];
  MS_STAMP(0);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const int nw = blockIdx.x * kMsN + wave * 16;       // first column of this wave
  const int n = min(nw + r, (int)p.n - 1);            // this lane's weight column
  const int k0 = blockIdx.y * kMsSliceK;
  const int m0 = blockIdx.z * ROWS;

  // A ragged LAST slice (K % 1024 != 0; K % 64 == 0 always): the steps past K load X, the packed words, the scale and the zero
  // point from an out-of-range buffer offset (the range check is on the VGPR offset) -- zeros in LDS, and a zero SCALE makes
  // the dequantised weights zero whatever (0 - bias - zp) is: their MFMAs add 0 x 0.
  // Everything up to the last load is 32-bit and branch-free.  Round 3 (profiles/r03/probe_mixed_skinny.txt): the prologue had
  // a 64-bit division (K / group_size), a 32-bit one per step (the step's group), and walked the X pieces with a branch per
  // (step, piece) -- ~1100 instructions ahead of the last load, 4 of the launch's 7.6 us.
  const int kk = (int)p.k;
  const int valid_steps = min(kMsSteps, (kk - k0) / kMsStepK);  // workgroup-uniform
  // (a macro, not a lambda: a lambda in this kernel template makes hipcc's host pass drop the kernel's launch stub)
#define CONCH_KILL(s) ((s) < valid_steps ? 0 : (int)0x80000000)

  // ---- weights, scales, zero points of the whole slice -> registers (first: they come from HBM, the X slice from L2)
  const uint32_t steps_per_group = (uint32_t)p.group_size / kMsStepK;                 // group_size % 64 == 0 (contract)
  const uint32_t first_step = (uint32_t)blockIdx.y * kMsSteps;
  const uint32_t first_group = first_step / steps_per_group;                          // the slice's ONE division
  const uint32_t groups = (uint32_t)kk / (uint32_t)p.group_size;
  const uint32_t word_rows = (uint32_t)kk * BITS / 32;
  const __amdgpu_buffer_rsrc_t rq =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.w_q, 0, ((word_rows - 1) * (uint32_t)p.wq_stride_k + (uint32_t)p.n) * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.w_s, 0, ((groups - 1) * (uint32_t)p.ws_stride_g + (uint32_t)p.n) * 2, 0x00020000);
  __amdgpu_buffer_rsrc_t rz = rs;
  if constexpr (ZP == CONCH_ZP_TENSOR)
    rz = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_zp, 0, ((groups - 1) * (uint32_t)p.wzp_stride_g + (uint32_t)p.n) * 4, 0x00020000);
  const int q_row = (int)p.wq_stride_k * 4;                         // bytes per word row
  const int vq = (g * kWpc) * q_row + n * 4;                        // this lane's k-group inside a half step
  const int q_base = (k0 * BITS / 32) * q_row;
  uint32_t wq[kMsSteps][2][kWpc];  // register path
  char* wl = lds + kMsSteps * kUnit + wave * (kWBytes / 4);  // this wave's word staging (WLDS)
  if constexpr (WLDS) {
    const int vw = (lane >> 2) * q_row + (nw + 4 * (lane & 3)) * 4;  // columns past N: never stored; past the tensor: zeros
#pragma unroll
    for (int i = 0; i < kMsSteps / 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void_t*)(wl + i * 1024), 16, vw | CONCH_KILL(2 * i), q_base + i * 16 * q_row, 0, 0);
  }
  uint32_t sc[kMsSteps];
  int zp[kMsSteps];
  int s_off = (int)first_group * (int)p.ws_stride_g * 2, z_off = (int)first_group * (int)p.wzp_stride_g * 4;
  uint32_t left = steps_per_group - (first_step - first_group * steps_per_group);    // steps before the group changes
#pragma unroll
  for (int s = 0; s < kMsSteps; ++s) {
    if constexpr (!WLDS) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < kWpc; ++i)
          wq[s][h][i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rq, vq | CONCH_KILL(s), q_base + (s * kWordRows + h * (kWordRows / 2) + i) * q_row, 0);
    }
    // (one scale load per step although a group spans two or more: fetching the slice's group rows in two loads and handing
    // them out with a ds_bpermute per step behind the barrier took 1 us off the ISSUE of the loads and made the launch 3-8 %
    // SLOWER -- the gather sits on the critical path, the loads it saves were issued under the latency of earlier ones:
    // profiles/r03/ms_scale_ab.txt; and skipping the load of a step that shares its group with the step before it, by a
    // workgroup-uniform branch, is 30-80 % SLOWER: hipcc drains the loads in flight at every branch of the issue loop)
    sc[s] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (n * 2) | CONCH_KILL(s), s_off, 0);
    if constexpr (ZP == CONCH_ZP_TENSOR) zp[s] = (int)__builtin_amdgcn_raw_buffer_load_b32(rz, (n * 4) | CONCH_KILL(s), z_off, 0);
    else zp[s] = 0;
    const bool wrap = left == 1;
    left = wrap ? steps_per_group : left - 1;
    s_off += wrap ? (int)p.ws_stride_g * 2 : 0;
    z_off += wrap ? (int)p.wzp_stride_g * 4 : 0;
  }

  // ---- X slice -> LDS (8-row x 128-byte subtiles, source-side swizzle); piece q = rows 8q..8q+7 of a step.  The
  // kMsSteps x kPieces (step, piece) items are dealt round-robin to the four waves: item t = wave + 4 j is step t / kPieces,
  // piece t % kPieces -- a wave meets one piece (kPieces <= 4) or two (8), whose lane offsets are computed once.
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.x, 0, (((uint32_t)p.m - 1) * (uint32_t)p.x_stride_m + (uint32_t)kk) * 2, 0x00020000);
  constexpr int kPieces = ROWS / 8;
  constexpr int kMine = kPieces > 4 ? 2 : 1;
  int voff_x[2];  // the first kMine are used (a dependent array bound here makes hipcc's host pass drop the kernel's stub)
#pragma unroll
  for (int v = 0; v < kMine; ++v) {
    const int q = (int)(((unsigned)wave + 4u * v) % kPieces);
    const int row = 8 * q + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    voff_x[v] = min(m0 + row, (int)p.m - 1) * (int)p.x_stride_m * 2 + chunk * 16;
  }
#pragma unroll
  for (int j = 0; j < kMsSteps * kPieces / 4; ++j) {
    const unsigned t = (unsigned)wave + 4u * j;
    const int s = (int)(t / kPieces), q = (int)(t % kPieces);  // wave-uniform
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(lds + s * kUnit + q * 1024), 16, voff_x[kMine == 2 ? (j & 1) : 0] | CONCH_KILL(s),
                                             (k0 + s * kMsStepK) * 2, 0, 0);
  }
#undef CONCH_KILL
  const int off_base = p.weight_bias + (ZP == CONCH_ZP_SCALAR ? p.w_zp[0] : 0);
  uint32_t and_mask = BITS == 4 ? 0x000f000fu : 0x00ff00ffu, or_magic = 0x64006400u;
  uint32_t and_mask_hi = and_mask << 4;  // int4: the pairs four bits up, taken without a shift (ChunkDequant)
  asm volatile("" : "+v"(and_mask), "+s"(or_magic), "+v"(and_mask_hi));

  f32x4 acc[ROWS / 16];
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) acc[i] = f32x4{0, 0, 0, 0};

  MS_STAMP(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA pieces (and its register loads) have landed
  __builtin_amdgcn_s_barrier();                     // ... and so have the other waves' pieces
  MS_STAMP(2);

  const int lane_off = (r >> 3) * 1024 + (r & 7) * 128 + ((g ^ ((r >> 1) & 7)) * 16);
  const int w_read = 64 * g + 4 * r;  // WLDS: this lane's word inside a half step's four word rows
#pragma unroll
  for (int s = 0; s < kMsSteps; ++s) {
    // the lane's 8 + 8 weights of this step: k = 8g..8g+7 (first half step) and 32 + 8g.. (second), column n
    uint32_t w00, w01, w10, w11;
    if constexpr (WLDS) {
      w00 = w01 = *(const uint32_t*)(wl + (s >> 1) * 1024 + (s & 1) * 512 + w_read);
      w10 = w11 = *(const uint32_t*)(wl + (s >> 1) * 1024 + (s & 1) * 512 + 256 + w_read);
    } else {
      w00 = wq[s][0][0], w01 = wq[s][0][kWpc - 1], w10 = wq[s][1][0], w11 = wq[s][1][kWpc - 1];
    }
    const i32x4 w_lo = dequant8<X_DT, BITS, ZP>(w00, w01, off_base + zp[s], sc[s], and_mask, or_magic, and_mask_hi);
    const i32x4 w_hi = dequant8<X_DT, BITS, ZP>(w10, w11, off_base + zp[s], sc[s], and_mask, or_magic, and_mask_hi);
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
      const Frag fx = read_frag(lds, s * kUnit + lane_off + i * 2048);
      if constexpr (X_DT == CONCH_DT_FP16) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w_lo), __builtin_bit_cast(f16x8, fx.lo), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w_hi), __builtin_bit_cast(f16x8, fx.hi), acc[i], 0, 0, 0);
      } else {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w_lo), __builtin_bit_cast(bf16x8, fx.lo), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w_hi), __builtin_bit_cast(bf16x8, fx.hi), acc[i], 0, 0, 0);
      }
    }
  }

  MS_STAMP(3);
  // D rows = n (4g + e), D columns = m (lane % 16): four consecutive n of one row per lane
  const int nn = nw + 4 * g;
  if constexpr (!FUSED) {
    float* slab = slabs + (int64_t)blockIdx.y * p.m * p.n;
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
      const int m = m0 + i * 16 + r;
      if (m < p.m && nn + 4 <= p.n) *(f32x4*)(slab + (int64_t)m * p.n + nn) = acc[i];
    }
  } else {
    const int slices = (int)gridDim.y;
    const uint32_t slab_bytes = (uint32_t)(p.m * p.n * 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, 0, slab_bytes * (uint32_t)slices, 0x00020000);
    int voff_s[ROWS / 16];
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) voff_s[i] = (min(m0 + i * 16 + r, (int)p.m - 1) * (int)p.n + min(nn, (int)p.n - 4)) * 4;
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
      const int m = m0 + i * 16 + r;
      if (m < p.m && nn + 4 <= p.n)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i]), rs, voff_s[i] + (int)blockIdx.y * (int)slab_bytes, 0, 16);  // sc1; constant soffset: see gemm_mfma.hip epilogue_rows_body
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its own write-through stores
    MS_STAMP(4);
    // the hand-off is per WAVE (gemm_skinny.hip, skinny_splitk_kernel): wave w of every slice's workgroup owns the same 16 columns
    unsigned* cnt = counters + 4 * (blockIdx.z * gridDim.x + blockIdx.x) + wave;
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
    MS_STAMP(5);
    if (ticket != (unsigned)(slices - 1)) return;  // wave-uniform
    f32x4 sum[ROWS / 16];
    for (int sb = 0; sb < slices; sb += 4) {
      u32x4 part[4][ROWS / 16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int soff = min(sb + j, slices - 1) * (int)slab_bytes;
#pragma unroll
        for (int i = 0; i < ROWS / 16; ++i) part[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_s[i], soff, 16);  // sc1
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool first = sb + j == 0, live = sb + j < slices;
#pragma unroll
        for (int i = 0; i < ROWS / 16; ++i) {
          const f32x4 v = __builtin_bit_cast(f32x4, part[j][i]);
          sum[i] = first ? v : live ? sum[i] + v : sum[i];
        }
      }
    }
    MS_DRAIN();
    MS_STAMP(6);
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
      const int m = m0 + i * 16 + r;
      if (m < p.m && nn + 4 <= p.n) {
        i32x2 pk;
        pk[0] = (int)pack2_bits16<OUT_DT>(f32x2{sum[i][0], sum[i][1]});
        pk[1] = (int)pack2_bits16<OUT_DT>(f32x2{sum[i][2], sum[i][3]});
        uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + nn;
        if ((((uintptr_t)dst) & 7) == 0) {
          *(i32x2*)dst = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e] = (uint16_t)((uint32_t)pk[e >> 1] >> (16 * (e & 1)));
        }
      }
    }
    MS_DRAIN();
    MS_STAMP(7);
    if (lane == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }
}

// Grid of the slab reduce kernels: blockIdx.y walks the rows, blockIdx.x x 256 + thread the 4-column quads of a row.  (Round 3:
// the kernels had taken a flat quad index and split it with a 64-bit division per thread -- ~200 instructions ahead of the first
// load in a kernel that lives 3-4 us -- and the fp32 ones added their slices one dependent L2 round trip at a time.)
inline dim3 reduce_grid(int64_t m, int64_t quads_per_row) {
  return dim3((unsigned)((quads_per_row + 255) / 256), (unsigned)std::min<int64_t>(m, 65535));
}

// four slabs of loads in flight, added in slice order (deterministic)
__device__ __forceinline__ f32x4 sum_slabs(const float* __restrict__ base, int64_t slab_stride, int slices) {
  f32x4 sum = f32x4{0, 0, 0, 0};
  for (int sb = 0; sb < slices; sb += 4) {
    f32x4 part[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) part[j] = *(const f32x4*)(base + (int64_t)min(sb + j, slices - 1) * slab_stride);
#pragma unroll
    for (int j = 0; j < 4; ++j) sum = (sb + j == 0) ? part[j] : (sb + j < slices) ? sum + part[j] : sum;
  }
  return sum;
}

// out[m][n..n+3] = cast( sum over slices, in slice order )
template <int OUT_DT>
__global__ __launch_bounds__(256) void mixed_skinny_reduce_kernel(MixedGemmArgs p, const float* __restrict__ slabs, int slices) {
  const int q = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (q >= (int)(p.n / 4)) return;
  const int n = q * 4;
  for (int m = blockIdx.y; m < (int)p.m; m += gridDim.y) {
    const f32x4 sum = sum_slabs(slabs + (int64_t)m * p.n + n, p.m * p.n, slices);
    i32x2 pk;
    pk[0] = (int)pack2_bits16<OUT_DT>(f32x2{sum[0], sum[1]});
    pk[1] = (int)pack2_bits16<OUT_DT>(f32x2{sum[2], sum[3]});
    uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n;
    if ((((uintptr_t)dst) & 7) == 0) {
      *(i32x2*)dst = pk;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = (uint16_t)((uint32_t)pk[e >> 1] >> (16 * (e & 1)));
    }
  }
}

// The same with the gate/up FFN fusion (conch_mixed_precision_gemm_silu_and_mul at decode batch sizes): the slabs hold the
// plain product for n = 2d columns [gate | up]; g, u, silu(g) and the product are rounded like silu_and_mul_kernel does.
template <int OUT_DT>
__global__ __launch_bounds__(256) void mixed_skinny_reduce_silu_kernel(MixedGemmArgs p, const float* __restrict__ slabs, int slices) {
  const int64_t d = p.n / 2;
  const int q = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (q >= (int)(d / 4)) return;
  const int j0 = q * 4;
  for (int m = blockIdx.y; m < (int)p.m; m += gridDim.y) {
    const f32x4 g = sum_slabs(slabs + (int64_t)m * p.n + j0, p.m * p.n, slices);
    const f32x4 u = sum_slabs(slabs + (int64_t)m * p.n + d + j0, p.m * p.n, slices);
    uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + j0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = bits16_to_float<OUT_DT>(float_to_bits16<OUT_DT>(g[e]));
      const float ur = bits16_to_float<OUT_DT>(float_to_bits16<OUT_DT>(u[e]));
      const float sr = bits16_to_float<OUT_DT>(float_to_bits16<OUT_DT>(pin_f32(act_f32(gr, p.fuse_silu))));
      dst[e] = float_to_bits16<OUT_DT>(pin_f32(sr * ur));
    }
  }
}

// WL = the int4 words through LDS-DMA (see mixed_skinny_kernel): 16-byte aligned word rows, CONCH_TUNE_SKINNY_GATHER != 1
template <int X_DT, int BITS, int ZP, bool WL>
void launch_rows_wl(const MixedGemmArgs& p, int rows, dim3 grid, float* ws, unsigned* counters, hipStream_t stream) {
  const dim3 block(kMsThreads);
  if (counters) {  // one launch (rows <= 32 only: the GEMV sizes it is for); output dtype = the activation dtype or the other one
#define CONCH_ONE(R)                                                                                                                   \
  do {                                                                                                                                 \
    if (p.out_dtype == CONCH_DT_FP16)                                                                                                  \
      hipLaunchKernelGGL((mixed_skinny_kernel<X_DT, BITS, ZP, R, true, CONCH_DT_FP16, WL>), grid, block, 0, stream, p, ws, counters);   \
    else                                                                                                                               \
      hipLaunchKernelGGL((mixed_skinny_kernel<X_DT, BITS, ZP, R, true, CONCH_DT_BF16, WL>), grid, block, 0, stream, p, ws, counters);   \
  } while (0)
    if (rows == 16) CONCH_ONE(16);
    else CONCH_ONE(32);
#undef CONCH_ONE
    return;
  }
  if (rows == 16) hipLaunchKernelGGL((mixed_skinny_kernel<X_DT, BITS, ZP, 16, false, CONCH_DT_FP16, WL>), grid, block, 0, stream, p, ws, counters);
  else if (rows == 32) hipLaunchKernelGGL((mixed_skinny_kernel<X_DT, BITS, ZP, 32, false, CONCH_DT_FP16, WL>), grid, block, 0, stream, p, ws, counters);
  else hipLaunchKernelGGL((mixed_skinny_kernel<X_DT, BITS, ZP, 64, false, CONCH_DT_FP16, WL>), grid, block, 0, stream, p, ws, counters);
}

template <int X_DT, int BITS, int ZP>
void launch_rows(const MixedGemmArgs& p, int rows, dim3 grid, float* ws, unsigned* counters, hipStream_t stream) {
  if constexpr (BITS == 4) {
    // measured (profiles/r03/int4_decode_wlds.txt): 3-5 % ahead while the launch is at most one workgroup per CU (16 x 4096 x 4096
    // 7.0 -> 6.7 us); with several workgroups per CU the 32 KiB of staging cost residency -- 16 x 4096 x 11008 9.1 -> 10.4 us,
    // 32 x 4096 x 11008 12.6 -> 16.7 us -- so only then, and for the 64-row form, whose 128 KiB of X hold a CU to one workgroup
    // anyway (1-3.5 % ahead on nine shapes: profiles/r03/int4_decode_wlds.txt); 2 = always
    const int mode = tuning(CONCH_TUNE_SKINNY_GATHER);
    const bool one_round = (size_t)grid.x * grid.y * grid.z <= (size_t)device_cu_count();
    if ((((uintptr_t)p.w_q) & 15) == 0 && p.wq_stride_k % 4 == 0 && (mode == 2 || (mode == 0 && (one_round || rows == 64))))
      return launch_rows_wl<X_DT, BITS, ZP, true>(p, rows, grid, ws, counters, stream);
  }
  launch_rows_wl<X_DT, BITS, ZP, false>(p, rows, grid, ws, counters, stream);
}

template <int X_DT, int BITS>
void launch_zp_mode(const MixedGemmArgs& p, int rows, dim3 grid, float* ws, unsigned* counters, hipStream_t stream) {
  if (p.zp_mode == CONCH_ZP_NONE) launch_rows<X_DT, BITS, CONCH_ZP_NONE>(p, rows, grid, ws, counters, stream);
  else if (p.zp_mode == CONCH_ZP_SCALAR) launch_rows<X_DT, BITS, CONCH_ZP_SCALAR>(p, rows, grid, ws, counters, stream);
  else launch_rows<X_DT, BITS, CONCH_ZP_TENSOR>(p, rows, grid, ws, counters, stream);
}

}  // namespace

// out[m][n] = cast( sum over slices, in slice order ) of fp32 slabs [slice][M][N] (also the reduce of bnb.hip's 4-bit decode GEMM)
int launch_f32_slab_reduce(void* c, const float* slabs, int slices, int64_t m, int64_t n, int64_t c_stride_m, int out_dtype, hipStream_t stream) {
  MixedGemmArgs p{};
  p.c = c;
  p.m = m;
  p.n = n;
  p.c_stride_m = c_stride_m;
  if (out_dtype == CONCH_DT_FP16)
    hipLaunchKernelGGL((mixed_skinny_reduce_kernel<CONCH_DT_FP16>), reduce_grid(m, n / 4), dim3(256), 0, stream, p, slabs, slices);
  else
    hipLaunchKernelGGL((mixed_skinny_reduce_kernel<CONCH_DT_BF16>), reduce_grid(m, n / 4), dim3(256), 0, stream, p, slabs, slices);
  return check_launch("f32_slab_reduce");
}

bool mixed_gemm_skinny_supported(const MixedGemmArgs& p) {
  if (!mixed_gemm_mfma_supported(p)) return false;  // dtypes, bits in {4, 8}, alignment, 32-bit buffer offsets
  if (p.fuse_silu) return false;
  if (p.m > kMsMaxM || p.n % 4) return false;        // up to four 64-row blocks (the dispatcher decides above 64 rows)
  // K: any multiple of 64 (the base contract); the last 1024-element slice may be shorter (K = 11008: 10 slices + 12 steps)
  if ((((uintptr_t)p.c) & 1) || p.c_stride_m < p.n) return false;
  return true;
}

int launch_mixed_gemm_skinny(const MixedGemmArgs& p, hipStream_t stream) {
  const int slices = (int)((p.k + kMsSliceK - 1) / kMsSliceK);
  const int rows = p.m <= 16 ? 16 : p.m <= 32 ? 32 : 64;
  void* ws = nullptr;
  const size_t bytes = (size_t)slices * p.m * p.n * 4;
  if (int rc = get_scratch(stream, kScratchMixedSplitK, bytes, &ws)) return rc;
  const dim3 grid((unsigned)((p.n + kMsN - 1) / kMsN), (unsigned)slices, (unsigned)((p.m + rows - 1) / rows));
  // CONCH_TUNE_SKINNY_MODE (shared with the scaled split-K kernel): 0 = auto, 1 = two launches, 2 / 3 = one launch.  Auto: one
  // launch for up to 32 rows (GEMV sizes: the slab tile of a workgroup is 2-8 KiB, the last arriver's pass over the slices is
  // cheaper than a second launch: profiles/r02/mixed_decode_one_launch.txt)
  const int mode = tuning(CONCH_TUNE_SKINNY_MODE);
  const bool one_launch = !p.fuse_silu && rows <= 32 && slices >= 2 && mode != 1 && bytes < ((size_t)1 << 31) &&
                          (size_t)grid.x * grid.z * 4 <= (size_t)kMsMaxTiles;  // one counter per (tile, wave)
  unsigned* counters = nullptr;
  if (one_launch) {
    void* cbuf = nullptr;
    if (int rc = get_scratch(stream, kScratchCounters, (size_t)kMsMaxTiles * 4, &cbuf, /*zero_on_alloc=*/true)) return rc;
    counters = (unsigned*)cbuf;
  }
  if (p.x_dtype == CONCH_DT_FP16) {
    if (p.bits == 4) launch_zp_mode<CONCH_DT_FP16, 4>(p, rows, grid, (float*)ws, counters, stream);
    else launch_zp_mode<CONCH_DT_FP16, 8>(p, rows, grid, (float*)ws, counters, stream);
  } else {
    if (p.bits == 4) launch_zp_mode<CONCH_DT_BF16, 4>(p, rows, grid, (float*)ws, counters, stream);
    else launch_zp_mode<CONCH_DT_BF16, 8>(p, rows, grid, (float*)ws, counters, stream);
  }
  if (one_launch) return check_launch("mixed_gemm_skinny_one_launch");
  if (p.fuse_silu) {  // p.n = 2d columns [gate | up], C has d (caller: mixed_gemm_skinny_supported(wide) && d % 4 == 0)
    if (p.out_dtype == CONCH_DT_FP16)
      hipLaunchKernelGGL((mixed_skinny_reduce_silu_kernel<CONCH_DT_FP16>), reduce_grid(p.m, p.n / 8), dim3(256), 0, stream, p, (const float*)ws, slices);
    else
      hipLaunchKernelGGL((mixed_skinny_reduce_silu_kernel<CONCH_DT_BF16>), reduce_grid(p.m, p.n / 8), dim3(256), 0, stream, p, (const float*)ws, slices);
    return check_launch("mixed_gemm_skinny_silu");
  }
  if (p.out_dtype == CONCH_DT_FP16)
    hipLaunchKernelGGL((mixed_skinny_reduce_kernel<CONCH_DT_FP16>), reduce_grid(p.m, p.n / 4), dim3(256), 0, stream, p, (const float*)ws, slices);
  else
    hipLaunchKernelGGL((mixed_skinny_reduce_kernel<CONCH_DT_BF16>), reduce_grid(p.m, p.n / 4), dim3(256), 0, stream, p, (const float*)ws, slices);
  return check_launch("mixed_gemm_skinny");
}

}  // namespace conch

#ifdef CONCH_CLOCK_PROBE
CONCH_PROBE_READER(conch_debug_probe_mixed_skinny, conch::g_probe_mixed_skinny)
#endif
