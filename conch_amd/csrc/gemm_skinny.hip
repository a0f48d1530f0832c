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
constexpr int kSpCountersPerTile = 4;  // one-launch form: one arrival counter per (tile, wave)

// ---- e4m3fnuz operands on the OCP fp8 MFMA, every code handled in registers (round 6; split-K kernel only) --------------------
// Read as OCP e4m3fn a fnuz byte is exactly TWICE its value -- except 0x7F / 0xFF (+-240: 480 does not exist in OCP, the byte is
// its NaN) and 0x80 (the fnuz NaN: OCP's -0).  The tile kernels stage both operands by LDS-DMA and can only run clean data that
// way (capi.hip: scan, gate, else the bf16 expansion).  HERE both operands pass through registers on their way into the MFMA, and
// at decode sizes the MFMA pipe is nearly idle, so the special codes are split off on the spot:
//     x = clean + rem      clean: 0x7F -> 0x7E, 0xFF -> 0xFE (+-448 = 2 x 224), 0x80 -> 0x00
//                          rem:   +-32 (0x60 / 0xE0 = 2 x 16) where the byte was +-240, OCP NaN (0x7F) where it was 0x80, else 0
//     acc += b_clean a_clean (+ b_rem a_clean) (+ b_clean a_rem + b_rem a_rem)     -- exact products, the extra MFMAs only for a
//                          fragment pair that holds a special code at all (wave-uniform)
// and the accumulator takes the exact 1/4 once, behind the K loop.  A fnuz NaN reaches the MFMA as an OCP NaN in `rem` and poisons
// exactly its row (A) or column (B), as the reference's fp32 matmul does.  No scan of the operands, no flag, no expansion pass:
// a weight matrix that was clamped by its quantiser costs what a clean one costs (16 x 4096 x 11008: 96 us through the expansion).
constexpr int kMmaFp8Fnuz = 4;  // (mfma_tile.hpp's kinds are 0..3; AccT<4> is the fp32 accumulator of the primary template)

// wave-uniform: does ANY lane's part of this fragment hold 0x7F / 0xFF / 0x80?  Five VALU operations per dword (and, add, add,
// bfi, or3) -- the cost of the fnuz path on clean data: a fragment without special codes goes to the MFMA as it is
__device__ __forceinline__ bool fnuz_special(const Frag& x) {
  uint32_t any = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t u = (uint32_t)(h ? x.hi[e] : x.lo[e]);
      const uint32_t t = u & 0x7F7F7F7Fu;
      any |= (t + 0x01010101u) | (~(t + 0x7F7F7F7Fu) & u);  // bit 7 of a byte: low seven bits all ones | all zeros under a set sign
    }
  }
  return __builtin_amdgcn_ballot_w64((any & 0x80808080u) != 0) != 0;
}

__device__ __forceinline__ void fnuz_split(const Frag& x, Frag& clean, Frag& rem) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t u = (uint32_t)(h ? x.hi[e] : x.lo[e]);
      const uint32_t t = u & 0x7F7F7F7Fu;
      const uint32_t sat = (t + 0x01010101u) & 0x80808080u;        // 0x80 in every byte that is 0x7F / 0xFF
      const uint32_t nan = ~(t + 0x7F7F7F7Fu) & u & 0x80808080u;   // 0x80 in every byte that is 0x80
      const uint32_t c = (u - (sat >> 7)) & ~nan;
      const uint32_t r = (sat >> 1) | (sat >> 2) | (u & sat) | (nan - (nan >> 7));
      if (h) {
        clean.hi[e] = (int)c;
        rem.hi[e] = (int)r;
      } else {
        clean.lo[e] = (int)c;
        rem.lo[e] = (int)r;
      }
    }
  }
}

// The whole slice's B^T fragments and A units are consumed step by step; step S: wait for its operands, (gather form) fetch this lane's
// chunk from lane 4 r + g (`gather` = that lane's ds_bpermute address)
template <int MMA, int ROWS, int STEPS, bool GATHER, int S>
__device__ __forceinline__ void sp_consume(typename AccT<MMA>::type (&acc)[ROWS / 16], const Frag (&fb)[STEPS], const char* lds,
                                           int lane_off, int gather, int live_blocks) {
  constexpr int kOps = 2 + ROWS / 32;
  if constexpr (S < STEPS) {
    wait_vmcnt_n<kOps * (STEPS - 1 - S)>();  // step S of this wave has landed ...
    Frag b = fb[S];
    if constexpr (GATHER) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        b.lo[j] = __builtin_amdgcn_ds_bpermute(gather, fb[S].lo[j]);
        b.hi[j] = __builtin_amdgcn_ds_bpermute(gather, fb[S].hi[j]);
      }
    }
    if constexpr (MMA == kMmaFp8Fnuz) {
      Frag bc = b, br;
      const bool b_special = fnuz_special(b);
      if (b_special) fnuz_split(b, bc, br);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < ROWS / 16; ++i) {
        // (<= 16 rows of a 32-row workgroup: the second block holds copies of row M - 1 and is never stored -- no test, no MFMA: a
        // third of the step's VALU work.  Workgroup-uniform.  Only there: the same test in front of every block of the taller
        // workgroups cost them 2 us, profiles/r06/fnuz_decode_ab.txt.)
        if constexpr (ROWS == 32) {
          if (i >= live_blocks) break;
        }
        const Frag fa = read_frag(lds, S * (ROWS * kStepBytes) + lane_off + i * 2048);
        if (!fnuz_special(fa)) {
          mma_step<kMmaFp8>(acc[i], bc, fa);
          if (b_special) mma_step<kMmaFp8>(acc[i], br, fa);
        } else {
          Frag ac, ar;
          fnuz_split(fa, ac, ar);
          mma_step<kMmaFp8>(acc[i], bc, ac);
          mma_step<kMmaFp8>(acc[i], bc, ar);
          if (b_special) {
            mma_step<kMmaFp8>(acc[i], br, ac);
            mma_step<kMmaFp8>(acc[i], br, ar);
          }
        }
      }
    } else {
      __builtin_amdgcn_s_barrier();            // ... and so has every other wave's quarter of unit S
#pragma unroll
      for (int i = 0; i < ROWS / 16; ++i) {
        const Frag fa = read_frag(lds, S * (ROWS * kStepBytes) + lane_off + i * 2048);
        mma_step<MMA>(acc[i], b, fa);  // D rows = n, D cols = m
      }
    }
    sp_consume<MMA, ROWS, STEPS, GATHER, S + 1>(acc, fb, lds, lane_off, gather, live_blocks);
  }
}

// scale / cast / bias of four consecutive columns of one row (scaled_gemm.py:21-25), packed as 2 dwords; the scales and the
// bias bits are given (sp_epilogue4 loads them; the one-launch split-K kernel fetched them before its K slice)
template <int OUT_DT, class ACC>
__device__ __forceinline__ i32x2 sp_epilogue4_vals(const ACC& sum, float sa, const float (&sb)[4], const uint32_t (&bias)[4], bool has_bias) {
  uint16_t o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = sa * (float)sum[e];   // scaled_gemm.py:21
    v = pin_f32(sb[e] * v);         // :22
    uint16_t h = float_to_bits16<OUT_DT>(v);  // :23
    if (has_bias)                   // :24-25
      h = float_to_bits16<OUT_DT>(pin_f32(bits16_to_float<OUT_DT>(h) + bits16_to_float<OUT_DT>((uint16_t)bias[e])));
    o[e] = h;
  }
  i32x2 pk;
  pk[0] = (int)((uint32_t)o[0] | ((uint32_t)o[1] << 16));
  pk[1] = (int)((uint32_t)o[2] | ((uint32_t)o[3] << 16));
  return pk;
}

template <int OUT_DT, class ACC>
__device__ __forceinline__ i32x2 sp_epilogue4(const ScaledGemmArgs& p, const ACC& sum, int m, int n) {
  const float sa = p.scale_a[p.scale_a_numel != 1 ? m : 0];
  float sb[4];
  uint32_t bias[4] = {0, 0, 0, 0};  // one register each: packing two 16-bit loads into one makes hipcc wait for each in turn
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sb[e] = p.scale_b[p.scale_b_numel != 1 ? n + e : 0];
    if (p.bias) bias[e] = ((const uint16_t*)p.bias)[n + e];
  }
  return sp_epilogue4_vals<OUT_DT>(sum, sa, sb, bias, p.bias != nullptr);
}

// the last arriver's C stores: one branch on the bias for the whole sub-tile (a taken branch per output costs more than the
// arithmetic it skips), one 8-byte store per row block at whatever alignment C has
template <int OUT_DT, int ROWS, bool BIAS, class ACC>
__device__ __forceinline__ void sp_finish(const ScaledGemmArgs& p, const ACC (&sum)[ROWS / 16], const float (&sa)[ROWS / 16],
                                          const float (&sb)[4], const uint32_t (&bias)[4], const int (&coff)[ROWS / 16], int m_first,
                                          bool live_n) {
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) {
    if (m_first + i * 16 < p.m && live_n) {
      const i32x2 pk = sp_epilogue4_vals<OUT_DT>(sum[i], sa[i], sb, bias, BIAS);
      __builtin_memcpy((uint16_t*)p.c + coff[i], &pk, 8);
    }
  }
}

// The whole slice is put in flight at once -- STEPS A units (ROWS x 128 bytes each, LDS-DMA) and STEPS B^T fragments
// (8 VGPRs each) per wave -- so the slice costs one memory latency plus its transfer time instead of a
// latency per K step; the steps are then consumed in issue order with counted vmcnt waits.
//
// FUSED = the reduce of the slices runs in the same launch (BASELINE config C2: one launch instead of two, no second
// pass over the slabs by a second grid).  Protocol (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the
// table of measured hand-offs; cdna_hip_programming.md section 5, in-launch split-K reduction), per WAVE -- wave w of every
// slice's workgroup owns the same 16 output columns, so no workgroup barrier is involved: the wave stores its partial
// sub-tile WRITE-THROUGH (sc1) and drains its own stores (s_waitcnt vmcnt(0)); one lane draws a ticket from the (tile, wave)
// agent-scope counter; the wave that draws the last ticket re-reads the slices' slabs with sc1 loads (they bypass this CU's
// L1, so no acquire is needed) and adds them IN SLICE ORDER (two slices: the other one's slab only, its own partial sums from
// registers, bit-identical to what it stored), then runs the fused epilogue -- scales and bias were fetched before the K
// slice -- and puts the counter back to zero.  Placement-independent: nothing here depends on which XCD or CU a slice runs on.
// Round 3 (profiles/r03/probe_skinny.txt, skinny_tail_ab.txt): scales ahead of the slice, the pair form and the per-wave
// ticket took C2 from 12.5 to 10.4 us; the GATHER fetch order (sp_consume) to 9.3 us.
// A_SRC != 0 (CONCH_DT_FP16 / BF16; two-launch form only): A arrives as 16-bit activations and is quantised on the way into
// LDS with the static per-tensor scale scale_a[0] -- the arithmetic of quant.hip (x * (1 / scale), clamp, convert), so the bytes
// in LDS are the ones static_scaled_{int8,fp8}_quant would have written to HBM and the product is bit-identical to the unfused
// pair.  The B^T fragments are put in flight first; the A slice (ROWS x STEPS x 128 elements) then passes through registers:
// 16 elements per thread and item, laid down where the LDS-DMA would have put them (unit row rho, 16-byte position pos holds
// source chunk pos ^ ((rho >> 1) & 7)).  At decode sizes the separate quantiser is a launch of the GEMM's own order.
#ifdef CONCH_CLOCK_PROBE
__device__ unsigned long long g_probe_skinny[kProbeBlocks * 8];
#define SK_STAMP(slot) CONCH_STAMP(g_probe_skinny, slot, (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)))
#define SK_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// the shader clock (s_memtime) at entry (0) and when the slice is consumed (1): workgroup i < 2048 uses row i + 2048 of the buffer
#define SK_CLOCK(which)                                                                                           \
  do {                                                                                                            \
    const int sk_idx = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));                     \
    if (threadIdx.x == 0 && sk_idx < 2048) g_probe_skinny[(sk_idx + 2048) * 8 + (which)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define SK_CLOCK(which) do { } while (0)
#define SK_STAMP(slot) do { } while (0)
#define SK_DRAIN() do { } while (0)
#endif

template <int MMA, int OUT_DT, int ROWS, int STEPS, bool FUSED, int A_SRC = 0, bool GATHER = false, bool LOOP = false>
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
  SK_STAMP(0);
  SK_CLOCK(0);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * kSpN + wave * 16, m0 = blockIdx.z * ROWS;
  const int slice = blockIdx.y;
  // A slice is `chunks` passes of STEPS K steps (p.split_steps steps in all; 0 = one pass): the accumulators stay in registers
  // from pass to pass, so a long K costs one slab per slice, not one per STEPS steps (round 4: 64 x 28672 x 8192 went through
  // 28 slabs, 117 MB of partial sums beside 235 MB of weights)
  // (LOOP = a separate instantiation: the one-pass kernels -- C2's among them -- keep their code)
  const int chunks = LOOP ? p.split_steps / STEPS : 1;
  const int slice_begin = LOOP ? slice * chunks * (STEPS * kStepBytes) : slice * (STEPS * kStepBytes);

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
  // GATHER: lane L fetches row L >> 2; its chunk is L & 3 for rows 0-7 and (L + 2) & 3 for rows 8-15 (a quad still covers the
  // row's 64 contiguous bytes).  Without the rotation operand lane (r, g) would read lane 4 r + g, and the lanes (r, g) and
  // (r + 8, g) of one half-wave -- sources 32 lanes apart -- hit the same ds_bpermute bank: two extra LDS cycles per
  // instruction, ALL of C2's SQ_LDS_BANK_CONFLICT (262 144 = 131 072 bpermutes x 2: profiles/r03/c2_rocprofv3_summary.txt).
#ifdef CONCH_EXP_GATHER_PLAIN  // A/B variant: round 3's order
  const int rot = 0;
#else
  const int rot = 2;
#endif
  const int voff_b = GATHER ? min(n0 + (lane >> 2), (int)p.n - 1) * (int)p.b_stride_n + 16 * (((lane & 3) + rot * (lane >> 5)) & 3)
                               : min(n0 + r, (int)p.n - 1) * (int)p.b_stride_n + 16 * g;
  const int gather = (4 * r + ((g - rot * (r >> 3)) & 3)) * 4;
  const int lane_off = (r >> 3) * 1024 + (r & 7) * 128 + ((g ^ ((r >> 1) & 7)) * 16);

  // One-launch form: the scales and the bias of this lane's outputs are requested BEFORE the K slice (they are older than
  // every slice load, so the counted waits below are unchanged) -- the last arriver's epilogue then has no load of its own
  // between the sum and the C stores (one L2 round trip off the launch's critical path; profiles/r03/probe_skinny.txt).
  [[maybe_unused]] float pre_sa[ROWS / 16];
  [[maybe_unused]] float pre_sb[4];
  [[maybe_unused]] uint32_t pre_bias[4] = {0, 0, 0, 0};
  [[maybe_unused]] int coff[ROWS / 16];  // element offset of this lane's four outputs in C (the launcher keeps C below 2^31 elements)
  if constexpr (FUSED) {
    const int nq = min(n0 + 4 * g, (int)p.n - 4);
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
      const int mq = min(m0 + i * 16 + r, (int)p.m - 1);
      pre_sa[i] = p.scale_a[p.scale_a_numel != 1 ? mq : 0];
      coff[i] = mq * (int)p.c_stride_m + nq;
    }
    // unconditional loads (without a bias: eight readable bytes of B^T, never used): a load under `if (p.bias)` is waited for
    // at the end of its branch
    const uint16_t* bias4 = p.bias ? (const uint16_t*)p.bias + nq : (const uint16_t*)p.b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pre_sb[e] = p.scale_b[p.scale_b_numel != 1 ? nq + e : 0];
      pre_bias[e] = bias4[e];
    }
    asm volatile("" ::: "memory");
  }

  // A ragged LAST slice (K % slice != 0; K % 128 == 0 always): the steps past K load from an out-of-range buffer offset --
  // zeros into the registers and into LDS alike (the range check is on the VGPR offset) -- and their MFMAs add 0 x 0.
  typename AccT<MMA>::type acc[ROWS / 16];
  if constexpr (LOOP) {
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) acc[i] = typename AccT<MMA>::type{0, 0, 0, 0};
  }
  // The passes are a goto loop that exists only in the LOOP instantiations: with a `for` around the body the ONE-pass kernels
  // came out 20 instructions different and 2-4 % slower (C2 9.03 -> 9.24 us, same box), although the loop folds away.
  [[maybe_unused]] int chunk = 0;
pass_begin: __attribute__((unused));
  {
  const int k_begin = LOOP ? slice_begin + chunk * (STEPS * kStepBytes) : slice_begin;
  if constexpr (LOOP) {
    if (k_begin >= (int)p.k) goto pass_end;  // a ragged last slice ends early (workgroup-uniform)
    if (chunk > 0) {  // every wave is done reading the units of the previous pass before the next pass's LDS-DMA overwrites them
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
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
    constexpr int kPer = kItems / kSkThreads;  // items per thread: ROWS / 4 at eight steps
    constexpr int kBatch = 8;                  // items whose loads are in flight together (64 VGPRs)
    static_assert(kItems % kSkThreads == 0 && kPer % kBatch == 0, "whole batches of items per thread");
    // a batch's sixteen loads are requested before its first item is converted (the loop had been one memory round trip per
    // item: load, wait, convert, store -- tools/isa_issue_waits.py)
#pragma unroll 1
    for (int b0 = 0; b0 < kPer; b0 += kBatch) {
      i32x4 lo[kBatch], hi[kBatch];
      int dst[kBatch];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        const int it = (int)threadIdx.x + (b0 + j) * kSkThreads;
        const int s = it / (ROWS * 8), rem = it - s * (ROWS * 8);
        const int rho = rem >> 3, pos = rem & 7;
        const int chunk = pos ^ ((rho >> 1) & 7);
        const int kill = s < valid_steps ? 0 : (int)0x80000000;
        const int voff = (min(m0 + rho, (int)p.m - 1) * (int)p.a_stride_m + chunk * 16) * 2;
        // the step's byte offset depends on the thread (s = it / ...): it belongs in the VGPR offset.  As an soffset hipcc wrapped
        // both loads in a waterfall loop over its distinct values (up to eight trips per load; round 3, tools/isa_waterfalls.py)
        const int koff = voff + (k_begin + s * kStepBytes) * 2;
        lo[j] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, koff | kill, 0, 0));
        hi[j] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (koff + 16) | kill, 0, 0));
        dst[j] = s * kUnit + rho * kStepBytes + pos * 16;
      }
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        float f[16];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[2 * e] = bits16_to_float<A_SRC>((uint16_t)((uint32_t)lo[j][e] & 0xffffu));
          f[2 * e + 1] = bits16_to_float<A_SRC>((uint16_t)((uint32_t)lo[j][e] >> 16));
          f[8 + 2 * e] = bits16_to_float<A_SRC>((uint16_t)((uint32_t)hi[j][e] & 0xffffu));
          f[8 + 2 * e + 1] = bits16_to_float<A_SRC>((uint16_t)((uint32_t)hi[j][e] >> 16));
        }
        i32x4 q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = (int)quant::quant_four<kKind>(f[4 * e], f[4 * e + 1], f[4 * e + 2], f[4 * e + 3], inv);
        *(i32x4*)(lds + dst[j]) = q;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // B^T fragments and this thread's ds_writes are complete
  }

  SK_STAMP(1);
  if constexpr (!LOOP) {  // one pass: zeroed behind the slice's requests, as in rounds 2-3 (same code as before the loop existed)
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) acc[i] = typename AccT<MMA>::type{0, 0, 0, 0};
  }
  sp_consume<MMA, ROWS, STEPS, GATHER, 0>(acc, fb, lds, lane_off, gather, min(ROWS / 16, ((int)p.m - m0 + 15) / 16));
  }  // pass
  if constexpr (LOOP) {
    if (++chunk < chunks) goto pass_begin;
  }
pass_end: __attribute__((unused));
  if constexpr (MMA == kMmaFp8Fnuz) {  // both operands ran as twice their values: the exact 1/4 (fnuz_split)
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) acc[i] = acc[i] * 0.25f;
  }
  SK_STAMP(2);
  SK_CLOCK(1);

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
    SK_STAMP(3);
    // wave w of every slice's workgroup owns the same 16 output columns, so the hand-off is per WAVE -- one counter per (tile,
    // wave), no workgroup barrier on the way
    unsigned* cnt = counters + kSpCountersPerTile * (blockIdx.z * gridDim.x + blockIdx.x) + wave;
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
    SK_STAMP(4);
    if (ticket != (unsigned)(slices - 1)) return;  // wave-uniform
    typename AccT<MMA>::type sum[ROWS / 16];
    if (slices == 2) {
      // two slices: the other slab only (a quarter of the general loop's loads); this slice's partial sums are still in
      // registers, bit-identical to what was stored.  Operand order = slice order, as in the general loop.
      u32x4 other[ROWS / 16];
      const int soff = (1 - slice) * (int)slab_bytes;
#pragma unroll
      for (int i = 0; i < ROWS / 16; ++i) other[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_s[i], soff, 16);  // sc1
#pragma unroll
      for (int i = 0; i < ROWS / 16; ++i) {
        const typename AccT<MMA>::type v = __builtin_bit_cast(typename AccT<MMA>::type, other[i]);
        sum[i] = (slice == 0 ? acc[i] : v) + (slice == 0 ? v : acc[i]);
      }
    } else {
      // Last arriver: add the slices IN SLICE ORDER (exact for int32, deterministic for fp32), four slices of loads in
      // flight at a time.  Every slice is loaded, this wave's own included (its stores are at L2 like the others'):
      // a per-slice "registers or load" choice would make hipcc branch around, and drain, every load.
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
    }
    SK_DRAIN();
    SK_STAMP(5);
    const bool live_n = n + 4 <= p.n;
    if (p.bias) sp_finish<OUT_DT, ROWS, true>(p, sum, pre_sa, pre_sb, pre_bias, coff, m0 + r, live_n);
    else sp_finish<OUT_DT, ROWS, false>(p, sum, pre_sa, pre_sb, pre_bias, coff, m0 + r, live_n);
    SK_STAMP(6);
    SK_DRAIN();
    SK_STAMP(7);
    if (lane == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
  }
}

// out[m][n..n+3] = epilogue( sum over slices, in slice order )
template <int MMA, int OUT_DT>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(ScaledGemmArgs p, const int* __restrict__ slabs, int slices) {
  // blockIdx.y = row, blockIdx.x x 256 + thread = the 4-column quad of the row (round 3: a flat index split by a 64-bit division
  // per thread had put ~200 instructions ahead of the first load of a kernel that lives 3-4 us)
  const int q = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (q >= (int)(p.n / 4)) return;
  if (gated_off(p.gate, p.gate_run_if)) return;  // the exact leg of the e4m3fnuz flow (gemm.hpp): nothing to reduce, C is not ours
  const int m = blockIdx.y, n = q * 4;
  // four slices of loads in flight at a time (a plain `for s` loop is one dependent L2 round trip per slice), added in
  // slice order: exact for int32, deterministic for fp32
#ifndef CONCH_EXP_REDUCE_LATE_SCALES
  // the scales and the bias are requested ahead of the slabs (as in the one-launch form): no dependent load behind the sum
  const float sa = p.scale_a[p.scale_a_numel != 1 ? m : 0];
  float sb4[4];
  uint32_t bias4[4];
  const uint16_t* bias_src = p.bias ? (const uint16_t*)p.bias + n : (const uint16_t*)p.b;  // unconditional loads (see skinny_splitk_kernel)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sb4[e] = p.scale_b[p.scale_b_numel != 1 ? n + e : 0];
    bias4[e] = bias_src[e];
  }
  asm volatile("" ::: "memory");
#endif
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
#ifndef CONCH_EXP_REDUCE_LATE_SCALES
  const i32x2 pk = sp_epilogue4_vals<OUT_DT>(sum, sa, sb4, bias4, p.bias != nullptr);
#else
  const i32x2 pk = sp_epilogue4<OUT_DT>(p, sum, m, n);
#endif
  __builtin_memcpy((uint16_t*)p.c + (int64_t)m * p.c_stride_m + n, &pk, 8);  // one 8-byte store at whatever alignment C has
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
  // scales and bias of both halves ahead of the slabs; the slabs of a half four slices in flight at a time (the loop had been one
  // dependent L2 round trip per slice and half), added in slice order
  const float sa = p.scale_a[p.scale_a_numel != 1 ? m : 0];
  float sb4[2][4];
  uint32_t bias4[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = j0 + h * (int)d;
    const uint16_t* bias_src = p.bias ? (const uint16_t*)p.bias + n : (const uint16_t*)p.b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sb4[h][e] = p.scale_b[p.scale_b_numel != 1 ? n + e : 0];
      bias4[h][e] = bias_src[e];
    }
  }
  asm volatile("" ::: "memory");
  uint16_t gu[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = j0 + h * (int)d;
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
    const i32x2 pk = sp_epilogue4_vals<OUT_DT>(sum, sa, sb4[h], bias4[h], p.bias != nullptr);
#pragma unroll
    for (int e = 0; e < 4; ++e) gu[h][e] = (uint16_t)((uint32_t)pk[e >> 1] >> (16 * (e & 1)));
  }
  uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + j0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float s = bits16_to_float<OUT_DT>(float_to_bits16<OUT_DT>(pin_f32(act_f32(bits16_to_float<OUT_DT>(gu[0][e]), p.fuse_silu))));
    dst[e] = float_to_bits16<OUT_DT>(pin_f32(s * bits16_to_float<OUT_DT>(gu[1][e])));
  }
}

// launches the split-K kernel for a runtime (row count, steps per slice, B^T fetch order) triple
template <int MMA, int OUT_DT, bool FUSED>
void launch_splitk_kernel(int rows, int steps, bool gather, dim3 grid, const ScaledGemmArgs& p, int* ws, unsigned* counters,
                          hipStream_t stream) {
  const dim3 block(kSkThreads);
  const bool loop = p.split_steps > steps;  // several passes per slice
#define CONCH_SK(R, S)                                                                                                            \
  do {                                                                                                                           \
    if (loop) {                                                                                                                  \
      if (gather)                                                                                                                \
        hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, R, S, FUSED, 0, true, true>), grid, block, 0, stream, p, ws, counters);  \
      else                                                                                                                       \
        hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, R, S, FUSED, 0, false, true>), grid, block, 0, stream, p, ws, counters); \
    } else if (gather)                                                                                                           \
      hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, R, S, FUSED, 0, true>), grid, block, 0, stream, p, ws, counters);     \
    else                                                                                                                         \
      hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, R, S, FUSED, 0, false>), grid, block, 0, stream, p, ws, counters);    \
  } while (0)
  if constexpr (FUSED) {
    if (steps == 16) {  // 2048-byte slices: half the slabs, rows split instead (rows <= 64: 16 units of 64 rows fill the LDS)
      if (rows == 32) CONCH_SK(32, 16);
      else CONCH_SK(64, 16);
      return;
    }
  }
  if constexpr (!FUSED && MMA != kMmaFp8Fnuz) {
    if (p.a_src_dtype) {  // quantise A on the way in (conch_static_quant_scaled_gemm)
#define CONCH_QA(R)                                                                                                              \
  do {                                                                                                                           \
    if (p.a_src_dtype == CONCH_DT_FP16) {                                                                                        \
      if (gather)                                                                                                                \
        hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, R, kSpSteps, false, CONCH_DT_FP16, true>), grid, block, 0, stream, p, ws, counters); \
      else                                                                                                                       \
        hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, R, kSpSteps, false, CONCH_DT_FP16, false>), grid, block, 0, stream, p, ws, counters); \
    } else {                                                                                                                     \
      if (gather)                                                                                                                \
        hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, R, kSpSteps, false, CONCH_DT_BF16, true>), grid, block, 0, stream, p, ws, counters); \
      else                                                                                                                       \
        hipLaunchKernelGGL((skinny_splitk_kernel<MMA, OUT_DT, R, kSpSteps, false, CONCH_DT_BF16, false>), grid, block, 0, stream, p, ws, counters); \
    }                                                                                                                            \
  } while (0)
      if (rows == 32) CONCH_QA(32);
      else if (rows == 64) CONCH_QA(64);
      else CONCH_QA(128);
#undef CONCH_QA
      return;
    }
  }
  if (rows == 32) CONCH_SK(32, kSpSteps);
  else if (rows == 64) CONCH_SK(64, kSpSteps);
  else CONCH_SK(128, kSpSteps);
#undef CONCH_SK
}

constexpr int kSpMaxTiles = 16384;  // arrival counters per (device, stream): 64 KiB; a tile uses kSpCountersPerTile of them

int splitk_slices(const ScaledGemmArgs& p) {
  // slices of 1024 K-bytes, the last one possibly shorter (K % 128 == 0 is part of the MFMA contract): K = 11008, the Llama-7B
  // down projection, is 10 slices + 6 steps
  if (p.n % 4 || p.m > 2 * kSkM) return 0;
  return (int)((p.k + kSpSliceK - 1) / kSpSliceK);
}

// passes of `steps`-step units a slice makes before it leaves its partial sums (1 = a slab per unit, rounds 2-3)
int splitk_chunks(const ScaledGemmArgs& p, int slices, int rows) {
  const int forced = tuning(CONCH_TUNE_SKINNY_CHUNKS);
  if (forced >= 1) return std::min(forced, std::max(slices, 1));
  // auto: keep about two workgroups per CU in the grid
  const int64_t wgs = ((p.n + kSpN - 1) / kSpN) * ((p.m + rows - 1) / rows) * slices;
  const int64_t want = 2 * (int64_t)device_cu_count();
  if (wgs <= want || slices < 2) return 1;
  return (int)std::min<int64_t>(slices, std::max<int64_t>(1, wgs / want));
}

}  // namespace

// `wide` = the plain GEMM on [gate | up] (n = 2d): can the split-K form run it with the silu fused into its reduce kernel?
bool scaled_gemm_skinny_fused_supported(const ScaledGemmArgs& wide) {
  return scaled_gemm_skinny_supported(wide) && wide.n % 8 == 0 && splitk_slices(wide) >= 1;
}

// e4m3fnuz through the split-K kernel with the special codes split off in registers: `f` = the problem with in_dtype already
// rewritten to e4m3fn (the contract checks are dtype-agnostic beyond "one byte per element")
bool scaled_gemm_skinny_fnuz_supported(const ScaledGemmArgs& f) {
  return scaled_gemm_skinny_supported(f) && !f.fuse_silu && !f.a_src_dtype && !f.n_more && splitk_slices(f) >= 1 &&
         tuning(CONCH_TUNE_SKINNY_NO_SPLITK) != 1;
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
    // measured (profiles/r03/sweep_skinny.txt, 22 decode shapes x {1, 2, 3} x both fetch orders; round 2's table:
    // profiles/r02/splitk_modes.txt).  The one-launch forms pay the last arriver's serial pass over the slabs, so they want few
    // slices: 2048-byte slices in two 64-row blocks where those fill at most one round of the chip (C2: 9.3 against 13.2 us; 96 rows
    // 9.3 against 12.1), 1024-byte slices up to 64 rows while there are at most four of them (64 x 4096 x 11008: 15.4 against
    // 16.4 us; at K = 8192 the two-launch form is level or ahead).  GEMV sizes (<= 32 rows): one launch always, the slab tile is 2-8 KiB.
    const int64_t wgs = ((p.n + kSpN - 1) / kSpN) * (p.k / (2 * kSpSliceK)) * ((p.m + 63) / 64);
    const bool two_k = p.k % (2 * kSpSliceK) == 0;
    if (p.m <= 32) mode = (p.k >= 8192 && two_k) ? 3 : 2;
    else if (p.m <= 64) mode = slices <= 4 ? 2 : 1;
    else mode = (p.m <= 128 && two_k && p.k >= 4 * kSpSliceK && wgs <= device_cu_count()) ? 3 : 1;
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
  // Passes per slice (skinny_splitk_kernel, `chunks`): CONCH_TUNE_SKINNY_CHUNKS forces a count; auto = see splitk_chunks
  const int chunks = p.a_src_dtype ? 1 : splitk_chunks(p, slices, rows);
  ScaledGemmArgs q = p;
  if (chunks > 1) {
    q.split_steps = chunks * steps;
    slices = (slices + chunks - 1) / chunks;
  }
  const dim3 grid((unsigned)((p.n + kSpN - 1) / kSpN), (unsigned)slices, (unsigned)((p.m + rows - 1) / rows));
  const size_t bytes = (size_t)slices * p.m * p.n * 4;
  // the in-launch reduce addresses all slabs through one 32-bit buffer descriptor and needs one counter per tile; the
  // gate/up fusion combines two tiles per output and keeps its reduce kernel
  const bool c_fits_int = (p.m - 1) * p.c_stride_m + p.n < ((int64_t)1 << 31);  // the fused epilogue's 32-bit element offsets
  if (p.fuse_silu || p.a_src_dtype || bytes >= ((size_t)1 << 31) || !c_fits_int ||
      (size_t)grid.x * grid.z * kSpCountersPerTile > (size_t)kSpMaxTiles)
    mode = 1;
  // CONCH_TUNE_SKINNY_GATHER.  Measured (profiles/r03/skinny_gather_ab.txt, sweep_skinny.txt): with at most one workgroup per CU
  // the launch is bound by how fast a wave gets its B^T loads ISSUED -- in operand order every quad of lanes touches four lines
  // and the texture path takes them one line at a time -- and the gather form wins 7-11 %; so it does with 32-row blocks and
  // 8-step slices whatever the grid (2-4 %: little A to read back from LDS).  With several rounds of 64- or 128-row workgroups
  // the LDS is the busier unit and the gather's eight ds_bpermutes per step cost 3 %.
  const int gather_mode = tuning(CONCH_TUNE_SKINNY_GATHER);
  const bool gather = gather_mode == 2 || (gather_mode == 0 && ((size_t)grid.x * grid.y * grid.z <= (size_t)device_cu_count() ||
                                                               (rows == 32 && steps == kSpSteps)));
  void* ws = nullptr;
  if (int rc = get_scratch(stream, kScratchSplitK, bytes, &ws)) return rc;
  if (mode != 1) {
    void* counters = nullptr;
    if (int rc = get_scratch(stream, kScratchCounters, (size_t)kSpMaxTiles * 4, &counters, /*zero_on_alloc=*/true)) return rc;
    launch_splitk_kernel<MMA, OUT_DT, true>(rows, steps, gather, grid, q, (int*)ws, (unsigned*)counters, stream);
    return check_launch("scaled_gemm_skinny_splitk_fused");
  }
  launch_splitk_kernel<MMA, CONCH_DT_BF16, false>(rows, steps, gather, grid, q, (int*)ws, nullptr, stream);
  if (p.fuse_silu) {  // p.n = 2d columns [gate | up], C has d
    hipLaunchKernelGGL((skinny_reduce_silu_kernel<MMA, OUT_DT>), dim3((unsigned)((p.n / 8 + 255) / 256), (unsigned)p.m), dim3(256), 0, stream, p,
                       (const int*)ws, slices);
    return check_launch("scaled_gemm_skinny_splitk_silu");
  }
  hipLaunchKernelGGL((skinny_reduce_kernel<MMA, OUT_DT>), dim3((unsigned)((p.n / 4 + 255) / 256), (unsigned)p.m), dim3(256), 0, stream, p,
                     (const int*)ws, slices);
  return check_launch("scaled_gemm_skinny_splitk");
}

int launch_splitk_reduce(const ScaledGemmArgs& p, const void* slabs, int slices, hipStream_t stream) {
  CONCH_CHECK_ARG(p.n % 4 == 0 && slices >= 1, "split-K reduce: N %% 4 != 0 or no slices");
  const dim3 grid((unsigned)((p.n / 4 + 255) / 256), (unsigned)p.m);
  const bool f = p.in_dtype != CONCH_DT_INT8;  // fp32 partial sums (fp8 and the 16-bit operand forms) or int32
#define CONCH_RED(MMA, OUT) hipLaunchKernelGGL((skinny_reduce_kernel<MMA, OUT>), grid, dim3(256), 0, stream, p, (const int*)slabs, slices)
  if (p.out_dtype == CONCH_DT_BF16) {
    if (f) CONCH_RED(kMmaFp8, CONCH_DT_BF16);
    else CONCH_RED(kMmaInt8, CONCH_DT_BF16);
  } else {
    if (f) CONCH_RED(kMmaFp8, CONCH_DT_FP16);
    else CONCH_RED(kMmaInt8, CONCH_DT_FP16);
  }
#undef CONCH_RED
  return check_launch("scaled_gemm_splitk_reduce");
}

int launch_scaled_gemm_skinny(const ScaledGemmArgs& p, hipStream_t stream) {
  // fuse_silu: `p` is the WIDE problem (n = 2d) whose reduce kernel writes the d-column FFN result (the caller has checked
  // scaled_gemm_skinny_fused_supported); only the split-K form has a reduce kernel
  // CONCH_TUNE_SKINNY_NO_SPLITK = 1 disables split-K (where the in-workgroup form can take the shape)
  const bool in_wg_ok = p.k % (2 * kStepBytes * kSkWaves) == 0;
  const int slices = (tuning(CONCH_TUNE_SKINNY_NO_SPLITK) == 1 && !p.fuse_silu && !p.a_src_dtype && in_wg_ok) ? 0 : splitk_slices(p);
  if (slices >= 1) {
    int rc;
    if (p.in_dtype == CONCH_DT_FP8_E4M3FNUZ)  // every fnuz code handled in registers (fnuz_split); capi.hip routes it here
      rc = p.out_dtype == CONCH_DT_BF16 ? launch_splitk<kMmaFp8Fnuz, CONCH_DT_BF16>(p, slices, stream)
                                        : launch_splitk<kMmaFp8Fnuz, CONCH_DT_FP16>(p, slices, stream);
    else if (p.in_dtype == CONCH_DT_FP8_E4M3FN)
      rc = p.out_dtype == CONCH_DT_BF16 ? launch_splitk<kMmaFp8, CONCH_DT_BF16>(p, slices, stream)
                                        : launch_splitk<kMmaFp8, CONCH_DT_FP16>(p, slices, stream);
    else
      rc = p.out_dtype == CONCH_DT_BF16 ? launch_splitk<kMmaInt8, CONCH_DT_BF16>(p, slices, stream)
                                        : launch_splitk<kMmaInt8, CONCH_DT_FP16>(p, slices, stream);
    return rc;  // a scratch failure is an error, not a reason to fall through: the in-workgroup kernel has no fused reduce
  }
  if (p.in_dtype == CONCH_DT_FP8_E4M3FNUZ) {
    set_error("scaled_gemm (skinny): e4m3fnuz operands run on the split-K form only (N %% 4 == 0)");
    return CONCH_ERR_UNSUPPORTED;
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

#ifdef CONCH_CLOCK_PROBE
CONCH_PROBE_READER(conch_debug_probe_skinny, conch::g_probe_skinny)
#endif
