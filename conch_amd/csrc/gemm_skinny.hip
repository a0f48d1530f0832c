// Skinny-M scaled GEMM (M <= 256): the decode-shaped case, e.g. BASELINE config C2
// (int8 -> bf16, M=128, K=4096, N=4096), which is HBM / latency bound (18 MB, AI = 233 FLOP/B).
//
// The 256x256-tile kernels would put such a problem on N/256 = 16 CUs.  Here every workgroup owns a
// 128(M) x 16(N) block of C, so N/16 = 256 workgroups stream B^T once, each 64 KiB slice by exactly
// one CU; A (<= 1 MB) is re-read by every workgroup but stays L2-resident.  Inside a workgroup the K
// range is split over the four waves (no barrier in the main loop, each wave is an independent
// stream): per 128-byte K step a wave loads 8 A fragments and 1 B^T fragment straight from global
// memory into registers (no LDS round trip: nothing is shared between waves), double-buffered one
// step ahead, and issues 8 (fp8) / 16 (int8) 16x16 MFMAs.  The four partial accumulators are summed
// through LDS in a fixed order (exact for int32, deterministic for fp32) and the usual fused
// epilogue (sb * (sa * acc), cast, + bias) writes 8-byte pieces of C rows.
#include "common.hpp"
#include "gemm.hpp"
#include "mfma_tile.hpp"
#include "quant_common.hpp"

namespace conch {
namespace {

using namespace tile;

constexpr int kSkThreads = 256;
constexpr int kSkM = 128;   // rows per workgroup (8 MFMA tiles)
constexpr int kSkN = 16;    // columns per workgroup (1 MFMA tile)
constexpr int kSkWaves = 4; // K split

struct SkFrags {
  Frag a[8];
  Frag b;
};

__device__ __forceinline__ i32x4 ld16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
  return __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

__device__ __forceinline__ void sk_load(SkFrags& f, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                                        const int (&voff_a)[8], int voff_b, int k) {
  f.b.lo = ld16(rb, voff_b, k);
  f.b.hi = ld16(rb, voff_b, k + 64);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f.a[i].lo = ld16(ra, voff_a[i], k);
    f.a[i].hi = ld16(ra, voff_a[i], k + 64);
  }
}

template <int MMA>
__device__ __forceinline__ void sk_mma(typename AccT<MMA>::type (&acc)[8], const SkFrags& f) {
#pragma unroll
  for (int i = 0; i < 8; ++i) mma_step<MMA>(acc[i], f.b, f.a[i]);  // D rows = n, D cols = m
}

template <int MMA, int OUT_DT>
__global__ __launch_bounds__(kSkThreads, 1) void scaled_gemm_skinny_kernel(ScaledGemmArgs p) {
  __shared__ __attribute__((aligned(16))) int red[kSkWaves][8][64][4];  // 32 KiB of partial accumulators
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * kSkN, m0 = blockIdx.y * kSkM;

  const uint32_t a_bytes = (uint32_t)((p.m - 1) * p.a_stride_m + p.k);
  const uint32_t b_bytes = (uint32_t)((p.n - 1) * p.b_stride_n + p.k);
  const __amdgpu_buffer_rsrc_t ra = make_uniform_rsrc(p.a, a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_uniform_rsrc(p.b, b_bytes);
  // lane (r, g) holds bytes [16g,16g+16) and [64+16g, ...) of row r of every fragment's 128-byte K step
  int voff_a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) voff_a[i] = min(m0 + i * 16 + r, (int)p.m - 1) * (int)p.a_stride_m + 16 * g;
  const int voff_b = min(n0 + r, (int)p.n - 1) * (int)p.b_stride_n + 16 * g;

  typename AccT<MMA>::type acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = typename AccT<MMA>::type{0, 0, 0, 0};

  const int k_per_wave = (int)(p.k / kSkWaves);  // multiple of 256 (dispatcher): an even number of steps
  const int k_begin = wave * k_per_wave, k_end = k_begin + k_per_wave;
  SkFrags f0, f1;
  sk_load(f0, ra, rb, voff_a, voff_b, k_begin);
  for (int k = k_begin; k < k_end; k += 2 * kStepBytes) {
    sk_load(f1, ra, rb, voff_a, voff_b, k + kStepBytes);
    sk_mma<MMA>(acc, f0);
    if (k + 2 * kStepBytes < k_end) sk_load(f0, ra, rb, voff_a, voff_b, k + 2 * kStepBytes);
    sk_mma<MMA>(acc, f1);
  }

  // cross-wave reduction through LDS, fixed order 0+1+2+3
#pragma unroll
  for (int i = 0; i < 8; ++i) *(i32x4*)&red[wave][i][lane][0] = __builtin_bit_cast(i32x4, acc[i]);
  __syncthreads();
  const bool has_bias = p.bias != nullptr;
  const bool sa_vec = p.scale_a_numel != 1, sb_vec = p.scale_b_numel != 1;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int mt = wave * 2 + j;
    typename AccT<MMA>::type sum = __builtin_bit_cast(typename AccT<MMA>::type, *(const i32x4*)&red[0][mt][lane][0]);
#pragma unroll
    for (int w = 1; w < kSkWaves; ++w)
      sum += __builtin_bit_cast(typename AccT<MMA>::type, *(const i32x4*)&red[w][mt][lane][0]);
    const int m = m0 + mt * 16 + r;
    const int nb = n0 + 4 * g;
    if (m >= p.m) continue;
    const float sa = p.scale_a[sa_vec ? m : 0];
    uint16_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = min(nb + e, (int)p.n - 1);
      const float sb = p.scale_b[sb_vec ? n : 0];
      float v = sa * (float)sum[e];   // scaled_gemm.py:21
      v = pin_f32(sb * v);            // :22
      uint16_t h = float_to_bits16<OUT_DT>(v);  // :23
      if (has_bias)                   // :24-25
        h = float_to_bits16<OUT_DT>(pin_f32(bits16_to_float<OUT_DT>(h) + bits16_to_float<OUT_DT>(((const uint16_t*)p.bias)[n])));
      o[e] = h;
    }
    uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + nb;
    if (nb + 4 <= p.n && (((uintptr_t)dst) & 7) == 0) {
      i32x2 pk;
      pk[0] = (int)((uint32_t)o[0] | ((uint32_t)o[1] << 16));
      pk[1] = (int)((uint32_t)o[2] | ((uint32_t)o[3] << 16));
      *(i32x2*)dst = pk;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (nb + e < p.n) dst[e] = o[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Split-K form.  The in-workgroup K split above makes every workgroup read ALL of A (M x K bytes from
// L2): 256 x 512 KB = 128 MB for C2, and the per-CU load path (~30 GB/s of 64-byte row pieces into
// registers) becomes the limit.  Here the K range is split ACROSS workgroups instead: workgroup
// (n-block of 64 columns, k-slice) reads only its K slice of A -- once, through a 4-slot LDS-DMA ring
// shared by its four waves (full 128-byte lines, same swizzled image as the tiled kernels) -- and each
// wave streams its own 16 columns of B^T into registers.  Partial accumulators go to fp32 / int32
// slabs [slice][M][N] in library-owned scratch; a second tiny kernel adds the slices in a fixed
// order (exact for int32, deterministic for fp32) and applies the fused epilogue.  C2: 64 n-blocks x
// 4 slices = 256 workgroups, 192 KB of operands each.
// ---------------------------------------------------------------------------------------------
constexpr int kSpN = 64;          // columns per workgroup (16 per wave)
constexpr int kSpSteps = 8;       // 128-byte K steps per slice: a slice is 1024 bytes of K
constexpr int kSpSliceK = kSpSteps * kStepBytes;

// ROWS = rows of A a workgroup handles (32, 64 or 128): a decode batch of 32 rows stages, multiplies and writes a quarter
// of what the 128-row form does.  Per step and wave: 2 register loads of B^T + ROWS/32 LDS-DMA pieces of A.
template <int MMA, int ROWS, int STEPS, int S>
__device__ __forceinline__ void sp_consume(typename AccT<MMA>::type (&acc)[ROWS / 16], const Frag (&fb)[STEPS], const char* lds,
                                           int lane_off) {
  constexpr int kOps = 2 + ROWS / 32;
  if constexpr (S < STEPS) {
    wait_vmcnt_n<kOps * (STEPS - 1 - S)>();  // step S of this wave has landed ...
    __builtin_amdgcn_s_barrier();            // ... and so has every other wave's quarter of unit S
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
      const Frag fa = read_frag(lds, S * (ROWS * kStepBytes) + lane_off + i * 2048);
      mma_step<MMA>(acc[i], fb[S], fa);  // D rows = n, D cols = m
    }
    sp_consume<MMA, ROWS, STEPS, S + 1>(acc, fb, lds, lane_off);
  }
}

// scale / cast / bias of four consecutive columns of one row (scaled_gemm.py:21-25), packed as 2 dwords
template <int OUT_DT, class ACC>
__device__ __forceinline__ i32x2 sp_epilogue4(const ScaledGemmArgs& p, const ACC& sum, int m, int n) {
  const float sa = p.scale_a[p.scale_a_numel != 1 ? m : 0];
  uint16_t o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float sb = p.scale_b[p.scale_b_numel != 1 ? n + e : 0];
    float v = sa * (float)sum[e];   // scaled_gemm.py:21
    v = pin_f32(sb * v);            // :22
    uint16_t h = float_to_bits16<OUT_DT>(v);  // :23
    if (p.bias)                     // :24-25
      h = float_to_bits16<OUT_DT>(pin_f32(bits16_to_float<OUT_DT>(h) + bits16_to_float<OUT_DT>(((const uint16_t*)p.bias)[n + e])));
    o[e] = h;
  }
  i32x2 pk;
  pk[0] = (int)((uint32_t)o[0] | ((uint32_t)o[1] << 16));
  pk[1] = (int)((uint32_t)o[2] | ((uint32_t)o[3] << 16));
  return pk;
}

// The whole slice is put in flight at once -- STEPS A units (ROWS x 128 bytes each, LDS-DMA) and STEPS B^T fragments
// (8 VGPRs each) per wave -- so the slice costs one memory latency plus its transfer time instead of a
// latency per K step; the steps are then consumed in issue order with counted vmcnt waits.
//
// FUSED = the reduce of the slices runs in the same launch (BASELINE config C2: one launch instead of two, no second
// pass over the slabs by a second grid).  Protocol (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the
// table of measured hand-offs; cdna_hip_programming.md section 5, in-launch split-K reduction): every wave stores its
// partial tile WRITE-THROUGH (sc1) and drains its own stores (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane draws a
// ticket from the tile's agent-scope counter; the workgroup that draws the last ticket re-reads the other slices' slabs
// with sc1 loads (they bypass this CU's L1, so no acquire is needed) and adds them IN SLICE ORDER -- its own slice from
// registers, bit-identical to what it stored -- then runs the fused epilogue and puts the counter back to zero.
// Placement-independent: nothing here depends on which XCD or CU a slice runs on.  The LDS request is kept above half
// of the CU's 160 KiB so that one workgroup runs per CU (the regime the hand-off was measured in).
// A_SRC != 0 (CONCH_DT_FP16 / BF16; two-launch form only): A arrives as 16-bit activations and is quantised on the way into
// LDS with the static per-tensor scale scale_a[0] -- the arithmetic of quant.hip (x * (1 / scale), clamp, convert), so the bytes
// in LDS are the ones static_scaled_{int8,fp8}_quant would have written to HBM and the product is bit-identical to the unfused
// pair.  The B^T fragments are put in flight first; the A slice (ROWS x STEPS x 128 elements) then passes through registers:
// 16 elements per thread and item, laid down where the LDS-DMA would have put them (unit row rho, 16-byte position pos holds
// source chunk pos ^ ((rho >> 1) & 7)).  At decode sizes the separate quantiser is a launch of the GEMM's own order.
template <int MMA, int OUT_DT, int ROWS, int STEPS, bool FUSED, int A_SRC = 0>
__global__ __launch_bounds__(kSkThreads, 1) void skinny_splitk_kernel(ScaledGemmArgs p, int* __restrict__ slabs,
                                                                       unsigned* __restrict__ counters) {
  static_assert(A_SRC == 0 || !FUSED, "the quantising prologue is built for the two-launch form");
  constexpr int kUnit = ROWS * kStepBytes;  // one K step of A
  constexpr int kPieces = ROWS / 32;        // 8-row x 128-byte subtiles a wave feeds per step
  // (Round 2 first padded the one-launch form's LDS request above half a CU, to stay in the one-workgroup-per-CU regime the
  // guide's hand-off table was measured in.  The hand-off does not need it -- sc1 stores are at L2 once drained, sc1 loads come
  // from L2 whatever else the CU runs; stress tests with two to four workgroups per CU -- and the padding cost up to 20 % at
  // decode sizes: profiles/r02/splitk_one_launch_nopad.txt.)
  constexpr int kLds = STEPS * kUnit;
  __shared__ __attribute__((aligned(1024))) char lds[kLds];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * kSpN + wave * 16, m0 = blockIdx.z * ROWS;
  const int slice = blockIdx.y;
  const int k_begin = slice * (STEPS * kStepBytes);

  const uint32_t a_bytes = (uint32_t)(((p.m - 1) * p.a_stride_m + p.k) * (A_SRC ? 2 : 1));
  const uint32_t b_bytes = (uint32_t)((p.n - 1) * p.b_stride_n + p.k);
  const __amdgpu_buffer_rsrc_t ra = make_uniform_rsrc(p.a, a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_uniform_rsrc(p.b, b_bytes);

  // A unit staging: wave w feeds unit rows [ROWS/4 * w, ROWS/4 * (w + 1)) = kPieces subtiles of 8 rows x 128 bytes
  int voff_a[4];  // the first kPieces are used (a dependent array bound here makes hipcc's host pass drop the kernel's stub)
#pragma unroll
  for (int j = 0; j < kPieces; ++j) {
    const int rho = (ROWS / 4) * wave + 8 * j + (lane >> 3);
    const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
    voff_a[j] = min(m0 + rho, (int)p.m - 1) * (int)p.a_stride_m + chunk * 16;
  }
  const int voff_b = min(n0 + r, (int)p.n - 1) * (int)p.b_stride_n + 16 * g;
  const int lane_off = (r >> 3) * 1024 + (r & 7) * 128 + ((g ^ ((r >> 1) & 7)) * 16);

  // A ragged LAST slice (K % slice != 0; K % 128 == 0 always): the steps past K load from an out-of-range buffer offset --
  // zeros into the registers and into LDS alike (the range check is on the VGPR offset) -- and their MFMAs add 0 x 0.
  const int valid_steps = min(STEPS, ((int)p.k - k_begin) / kStepBytes);  // workgroup-uniform
  Frag fb[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int kill = s < valid_steps ? 0 : (int)0x80000000;
    fb[s].lo = ld16(rb, voff_b | kill, k_begin + s * kStepBytes);
    fb[s].hi = ld16(rb, voff_b | kill, k_begin + s * kStepBytes + 64);
    if constexpr (A_SRC == 0) {
      char* dst = lds + s * kUnit + wave * (ROWS / 4) * kStepBytes;
#pragma unroll
      for (int j = 0; j < kPieces; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(dst + j * 1024), 16, voff_a[j] | kill,
                                                 k_begin + s * kStepBytes, 0, 0);
    }
    // the counted waits below count VMEM operations in THIS order, step by step: no reordering across steps
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (A_SRC != 0) {
    constexpr int kKind = MMA == kMmaInt8 ? quant::kInt8 : quant::kFp8Fn;
    const float inv = 1.0f / p.scale_a[0];
    constexpr int kItems = STEPS * ROWS * 8;  // (step, unit row, 16-byte position)
#pragma unroll 4
    for (int it = threadIdx.x; it < kItems; it += kSkThreads) {
      const int s = it / (ROWS * 8), rem = it - s * (ROWS * 8);
      const int rho = rem >> 3, pos = rem & 7;
      const int chunk = pos ^ ((rho >> 1) & 7);
      const int kill = s < valid_steps ? 0 : (int)0x80000000;
      const int voff = (min(m0 + rho, (int)p.m - 1) * (int)p.a_stride_m + chunk * 16) * 2;
      // the step's byte offset depends on the thread (s = it / ...): it belongs in the VGPR offset.  As an soffset hipcc wrapped
      // both loads in a waterfall loop over its distinct values (up to eight trips per load; round 3, tools/isa_waterfalls.py)
      const int koff = voff + (k_begin + s * kStepBytes) * 2;
      const i32x4 lo = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, koff | kill, 0, 0));
      const i32x4 hi = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (koff + 16) | kill, 0, 0));
      float f[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[2 * j] = bits16_to_float<A_SRC>((uint16_t)((uint32_t)lo[j] & 0xffffu));
        f[2 * j + 1] = bits16_to_float<A_SRC>((uint16_t)((uint32_t)lo[j] >> 16));
        f[8 + 2 * j] = bits16_to_float<A_SRC>((uint16_t)((uint32_t)hi[j] & 0xffffu));
        f[8 + 2 * j + 1] = bits16_to_float<A_SRC>((uint16_t)((uint32_t)hi[j] >> 16));
      }
      i32x4 q;
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = (int)quant::quant_four<kKind>(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3], inv);
      *(i32x4*)(lds + s * kUnit + rho * kStepBytes + pos * 16) = q;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // B^T fragments and this thread's ds_writes are complete
  }

  typename AccT<MMA>::type acc[ROWS / 16];
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) acc[i] = typename AccT<MMA>::type{0, 0, 0, 0};

  sp_consume<MMA, ROWS, STEPS, 0>(acc, fb, lds, lane_off);

  // partial sums -> slab [slice][M][N] (4-byte elements); lane: m = m0 + 16 i + r, n = n0 + 4 g + e
  const int n = n0 + 4 * g;
  const int64_t slab_elems = p.m * p.n;
  if constexpr (!FUSED) {
    int* slab = slabs + (int64_t)slice * slab_elems;
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
      const int m = m0 + i * 16 + r;
      if (m < p.m && n + 4 <= p.n) *(i32x4*)(slab + (int64_t)m * p.n + n) = __builtin_bit_cast(i32x4, acc[i]);
    }
  } else {
    const int slices = (int)gridDim.y;
    const uint32_t slab_bytes = (uint32_t)(slab_elems * 4);
    // one descriptor over all slabs (the launcher keeps slices x M x N x 4 below 2 GiB); slice s at scalar offset s x slab_bytes
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, 0, slab_bytes * (uint32_t)slices, 0x00020000);
    int voff_s[ROWS / 16];
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) voff_s[i] = (min(m0 + i * 16 + r, (int)p.m - 1) * (int)p.n + min(n, (int)p.n - 4)) * 4;
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
      const int m = m0 + i * 16 + r;
      if (m < p.m && n + 4 <= p.n)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i]), rs, voff_s[i] + slice * (int)slab_bytes, 0, 16);  // sc1; constant soffset: see gemm_mfma.hip epilogue_rows_body
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its own write-through stores
    __syncthreads();
    unsigned* flag = (unsigned*)lds;  // the operand ring is dead: every wave's last ds_read fed an MFMA before the barrier
    unsigned* cnt = counters + (blockIdx.z * gridDim.x + blockIdx.x);
    if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != (unsigned)(slices - 1)) return;  // workgroup-uniform
    // Last arriver: add the slices IN SLICE ORDER (exact for int32, deterministic for fp32), four slices of loads in
    // flight at a time.  Every slice is loaded, this workgroup's own included (its stores are at L2 like the others'):
    // a per-slice "registers or load" choice would make hipcc branch around, and drain, every load.
    typename AccT<MMA>::type sum[ROWS / 16];
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
          const typename AccT<MMA>::type v = __builtin_bit_cast(typename AccT<MMA>::type, part[j][i]);
          sum[i] = first ? v : live ? sum[i] + v : sum[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
      const int m = m0 + i * 16 + r;
      if (m < p.m && n + 4 <= p.n) {
        const i32x2 pk = sp_epilogue4<OUT_DT>(p, sum[i], m, n);
        uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n;
        if ((((uintptr_t)dst) & 7) == 0) {
          *(i32x2*)dst = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e] = (uint16_t)((uint32_t)pk[e >> 1] >> (16 * (e & 1)));
        }
      }
    }
    if (threadIdx.x == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }
}

// out[m][n..n+3] = epilogue( sum over slices, in slice order )
template <int MMA, int OUT_DT>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(ScaledGemmArgs p, const int* __restrict__ slabs, int slices) {
  // blockIdx.y = row, blockIdx.x x 256 + thread = the 4-column quad of the row (round 3: a flat index split by a 64-bit division
  // per thread had put ~200 instructions ahead of the first load of a kernel that lives 3-4 us)
  const int q = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (q >= (int)(p.n / 4)) return;
  const int m = blockIdx.y, n = q * 4;
  // four slices of loads in flight at a time (a plain `for s` loop is one dependent L2 round trip per slice), added in
  // slice order: exact for int32, deterministic for fp32
  typename AccT<MMA>::type sum;
  for (int sb = 0; sb < slices; sb += 4) {
    i32x4 part[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) part[j] = *(const i32x4*)(slabs + ((int64_t)min(sb + j, slices - 1) * p.m + m) * p.n + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const typename AccT<MMA>::type v = __builtin_bit_cast(typename AccT<MMA>::type, part[j]);
      sum = (sb + j == 0) ? v : (sb + j < slices) ? sum + v : sum;
    }
  }
  const float sa = p.scale_a[p.scale_a_numel != 1 ? m : 0];
  uint16_t o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float sb = p.scale_b[p.scale_b_numel != 1 ? n + e : 0];
    float v = sa * (float)sum[e];   // scaled_gemm.py:21
    v = pin_f32(sb * v);            // :22
    uint16_t h = float_to_bits16<OUT_DT>(v);  // :23
    if (p.bias)                     // :24-25
      h = float_to_bits16<OUT_DT>(pin_f32(bits16_to_float<OUT_DT>(h) + bits16_to_float<OUT_DT>(((const uint16_t*)p.bias)[n + e])));
    o[e] = h;
  }
  uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n;
  if ((((uintptr_t)dst) & 7) == 0) {
    i32x2 pk;
    pk[0] = (int)((uint32_t)o[0] | ((uint32_t)o[1] << 16));
    pk[1] = (int)((uint32_t)o[2] | ((uint32_t)o[3] << 16));
    *(i32x2*)dst = pk;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[e] = o[e];
  }
}

// The same with the gate/up FFN fusion (conch_scaled_gemm_silu_and_mul at decode batch sizes): the slabs hold the plain
// product for n = 2d columns [gate | up]; thread = 4 consecutive OUTPUT columns j: g and u get the scaled_gemm epilogue
// (scales, cast, bias), then the reference's silu_and_mul roundings -- exactly what silu_and_mul_kernel does to a stored G.
template <int MMA, int OUT_DT>
__global__ __launch_bounds__(256) void skinny_reduce_silu_kernel(ScaledGemmArgs p, const int* __restrict__ slabs, int slices) {
  const int64_t d = p.n / 2;
  const int q = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (q >= (int)(d / 4)) return;
  const int m = blockIdx.y, j0 = q * 4;
  const float sa = p.scale_a[p.scale_a_numel != 1 ? m : 0];
  uint16_t gu[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = j0 + h * (int)d;
    typename AccT<MMA>::type sum = __builtin_bit_cast(typename AccT<MMA>::type, *(const i32x4*)(slabs + (int64_t)m * p.n + n));
    for (int s = 1; s < slices; ++s)
      sum += __builtin_bit_cast(typename AccT<MMA>::type, *(const i32x4*)(slabs + ((int64_t)s * p.m + m) * p.n + n));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float sb = p.scale_b[p.scale_b_numel != 1 ? n + e : 0];
      float v = sa * (float)sum[e];
      v = pin_f32(sb * v);
      uint16_t hb = float_to_bits16<OUT_DT>(v);
      if (p.bias)
        hb = float_to_bits16<OUT_DT>(pin_f32(bits16_to_float<OUT_DT>(hb) + bits16_to_float<OUT_DT>(((const uint16_t*)p.bias)[n + e])));
      gu[h][e] = hb;
    }
  }
  uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + j0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float s = bits16_to_float<OUT_DT>(float_to_bits16<OUT_DT>(pin_f32(act_f32(bits16_to_float<OUT_DT>(gu[0][e]), p.fuse_silu))));
    dst[e] = float_to_bits16<OUT_DT>(pin_f32(s * bits16_to_float<OUT_DT>(gu[1][e])));
  }
}

// launches the split-K kernel for a runtime (row count, steps per slice) pair
template <int MMA, int OUT_DT, bool FUSED>
void launch_splitk_kernel(int rows, int steps, dim3 grid, const ScaledGemmArgs& p, int* ws, unsigned* counters, hipStream_t stream) {
  const dim3 block(kSkThreads);
  if constexpr (FUSED) {
    if (steps == 16) {  // 2048-byte slices: half the slabs, rows split instead (rows <= 64: 16 units of 64 rows fill the LDS)
      if (rows == 32) hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, 32, 16, true>), grid, block, 0, stream, p, ws, counters);
      else hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, 64, 16, true>), grid, block, 0, stream, p, ws, counters);
      return;
    }
  }
  if constexpr (!FUSED) {
    if (p.a_src_dtype) {  // quantise A on the way in (conch_static_quant_scaled_gemm)
#define CONCH_QA(R)                                                                                                              \
  do {                                                                                                                           \
    if (p.a_src_dtype == CONCH_DT_FP16)                                                                                          \
      hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, R, kSpSteps, false, CONCH_DT_FP16>), grid, block, 0, stream, p, ws, counters); \
    else                                                                                                                         \
      hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, R, kSpSteps, false, CONCH_DT_BF16>), grid, block, 0, stream, p, ws, counters); \
  } while (0)
      if (rows == 32) CONCH_QA(32);
      else if (rows == 64) CONCH_QA(64);
      else CONCH_QA(128);
#undef CONCH_QA
      return;
    }
  }
  if (rows == 32) hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, 32, kSpSteps, FUSED>), grid, block, 0, stream, p, ws, counters);
  else if (rows == 64) hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, 64, kSpSteps, FUSED>), grid, block, 0, stream, p, ws, counters);
  else hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, 128, kSpSteps, FUSED>), grid, block, 0, stream, p, ws, counters);
}

constexpr int kSpMaxTiles = 16384;  // arrival counters per (device, stream): 64 KiB

int splitk_slices(const ScaledGemmArgs& p) {
  // slices of 1024 K-bytes, the last one possibly shorter (K % 128 == 0 is part of the MFMA contract): K = 11008, the Llama-7B
  // down projection, is 10 slices + 6 steps
  if (p.n % 4 || p.m > 2 * kSkM) return 0;
  return (int)((p.k + kSpSliceK - 1) / kSpSliceK);
}

}  // namespace

// `wide` = the plain GEMM on [gate | up] (n = 2d): can the split-K form run it with the silu fused into its reduce kernel?
bool scaled_gemm_skinny_fused_supported(const ScaledGemmArgs& wide) {
  return scaled_gemm_skinny_supported(wide) && wide.n % 8 == 0 && splitk_slices(wide) >= 1;
}

bool scaled_gemm_skinny_supported(const ScaledGemmArgs& p) {
  if (!scaled_gemm_mfma_supported(p)) return false;
  if (p.m > 2 * kSkM) return false;                      // beyond two row blocks the tiled kernels win
  // the in-workgroup form needs K/4 per wave, an even number of 128-byte steps; the split-K form any K % 128 == 0 but N % 4 == 0
  if (p.k % (2 * kStepBytes * kSkWaves) && p.n % 4) return false;
  return true;
}

// CONCH_TUNE_SKINNY_MODE: 0 = auto, 1 = two launches (partials, then the reduce kernel), 2 = one launch with
// 1024-byte slices, 3 = one launch with 2048-byte slices and <= 64-row blocks
template <int MMA, int OUT_DT>
int launch_splitk(const ScaledGemmArgs& p, int slices, hipStream_t stream) {
  int mode = tuning(CONCH_TUNE_SKINNY_MODE);
  int rows = p.m <= 32 ? 32 : p.m <= 64 ? 64 : 128;
  if (mode == 0) {
    // measured (profiles/r02/splitk_modes.txt, 14 decode shapes x {1, 3}): the one-launch form wins only where its workgroups --
    // 2048-byte slices, two 64-row blocks -- fill exactly one round of the chip (C2: 13.0 against 13.5 us); for <= 64 rows its
    // sixteen steps in flight per workgroup lose to the 8-step slices, and with 1024-byte slices (mode 2) the last arriver's
    // serial pass over four or more slabs costs more than the second launch it saves (16.0 against 13.5 us on C2)
    const int64_t wgs = ((p.n + kSpN - 1) / kSpN) * (p.k / (2 * kSpSliceK)) * ((p.m + 63) / 64);
    if (p.m <= 32) mode = (p.k >= 8192 && p.k % (2 * kSpSliceK) == 0) ? 3 : 2;  // GEMV sizes: the slab tile is 2-8 KiB, one launch wins 4-10 %
    else mode = (p.m > 96 && p.m <= 128 && p.k % (2 * kSpSliceK) == 0 && wgs <= device_cu_count()) ? 3 : 1;  // 96 rows: two launches 4 % ahead
  }
  if (p.a_src_dtype) mode = 1;  // the quantising prologue lives in the two-launch form
  int steps = kSpSteps;
  if (mode == 3 && p.k % (2 * kSpSliceK) == 0) {
    steps = 2 * kSpSteps;
    slices = (slices + 1) / 2;
    rows = p.m <= 32 ? 32 : 64;
  } else if (mode == 3) {
    mode = 2;
  }
  const dim3 grid((unsigned)((p.n + kSpN - 1) / kSpN), (unsigned)slices, (unsigned)((p.m + rows - 1) / rows));
  const size_t bytes = (size_t)slices * p.m * p.n * 4;
  // the in-launch reduce addresses all slabs through one 32-bit buffer descriptor and needs one counter per tile; the
  // gate/up fusion combines two tiles per output and keeps its reduce kernel
  if (p.fuse_silu || p.a_src_dtype || bytes >= ((size_t)1 << 31) || (size_t)grid.x * grid.z > (size_t)kSpMaxTiles) mode = 1;
  void* ws = nullptr;
  if (int rc = get_scratch(stream, kScratchSplitK, bytes, &ws)) return rc;
  if (mode != 1) {
    void* counters = nullptr;
    if (int rc = get_scratch(stream, kScratchCounters, (size_t)kSpMaxTiles * 4, &counters, /*zero_on_alloc=*/true)) return rc;
    launch_splitk_kernel<MMA, OUT_DT, true>(rows, steps, grid, p, (int*)ws, (unsigned*)counters, stream);
    return check_launch("scaled_gemm_skinny_splitk_fused");
  }
  launch_splitk_kernel<MMA, CONCH_DT_BF16, false>(rows, steps, grid, p, (int*)ws, nullptr, stream);
  if (p.fuse_silu) {  // p.n = 2d columns [gate | up], C has d
    hipLaunchKernelGGL((skinny_reduce_silu_kernel<MMA, OUT_DT>), dim3((unsigned)((p.n / 8 + 255) / 256), (unsigned)p.m), dim3(256), 0, stream, p,
                       (const int*)ws, slices);
    return check_launch("scaled_gemm_skinny_splitk_silu");
  }
  hipLaunchKernelGGL((skinny_reduce_kernel<MMA, OUT_DT>), dim3((unsigned)((p.n / 4 + 255) / 256), (unsigned)p.m), dim3(256), 0, stream, p,
                     (const int*)ws, slices);
  return check_launch("scaled_gemm_skinny_splitk");
}

int launch_scaled_gemm_skinny(const ScaledGemmArgs& p, hipStream_t stream) {
  // fuse_silu: `p` is the WIDE problem (n = 2d) whose reduce kernel writes the d-column FFN result (the caller has checked
  // scaled_gemm_skinny_fused_supported); only the split-K form has a reduce kernel
  // CONCH_TUNE_SKINNY_NO_SPLITK = 1 disables split-K (where the in-workgroup form can take the shape)
  const bool in_wg_ok = p.k % (2 * kStepBytes * kSkWaves) == 0;
  const int slices = (tuning(CONCH_TUNE_SKINNY_NO_SPLITK) == 1 && !p.fuse_silu && !p.a_src_dtype && in_wg_ok) ? 0 : splitk_slices(p);
  if (slices >= 1) {
    int rc;
    if (p.in_dtype == CONCH_DT_FP8_E4M3FN)
      rc = p.out_dtype == CONCH_DT_BF16 ? launch_splitk<kMmaFp8, CONCH_DT_BF16>(p, slices, stream)
                                        : launch_splitk<kMmaFp8, CONCH_DT_FP16>(p, slices, stream);
    else
      rc = p.out_dtype == CONCH_DT_BF16 ? launch_splitk<kMmaInt8, CONCH_DT_BF16>(p, slices, stream)
                                        : launch_splitk<kMmaInt8, CONCH_DT_FP16>(p, slices, stream);
    return rc;  // a scratch failure is an error, not a reason to fall through: the in-workgroup kernel has no fused reduce
  }
  const dim3 grid((unsigned)((p.n + kSkN - 1) / kSkN), (unsigned)((p.m + kSkM - 1) / kSkM));
#define CONCH_LAUNCH(MMA, OUT)                                                                          \
  do {                                                                                                  \
    hipLaunchKernelGGL((scaled_gemm_skinny_kernel<MMA, OUT>), grid, dim3(kSkThreads), 0, stream, p);    \
    return check_launch("scaled_gemm_skinny");                                                          \
  } while (0)
  if (p.in_dtype == CONCH_DT_FP8_E4M3FN) {
    if (p.out_dtype == CONCH_DT_BF16) CONCH_LAUNCH(kMmaFp8, CONCH_DT_BF16);
    CONCH_LAUNCH(kMmaFp8, CONCH_DT_FP16);
  } else {
    if (p.out_dtype == CONCH_DT_BF16) CONCH_LAUNCH(kMmaInt8, CONCH_DT_BF16);
    CONCH_LAUNCH(kMmaInt8, CONCH_DT_FP16);
  }
#undef CONCH_LAUNCH
}

}  // namespace conch
