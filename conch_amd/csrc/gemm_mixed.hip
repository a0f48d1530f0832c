// Mixed-precision GEMM for gfx950: fp16/bf16 activations x group-quantized int4/int8 packed weights.
//
// Replaces the mixed-precision instantiation of the reference's Triton kernel
// (conch/kernels/quantization/gemm.py:219-457 via mixed_precision_gemm_launcher :482-545).
//
// Same 256 x (64 NT) / 8-wave geometry as the scaled GEMM (mfma_tile.hpp), K advanced 64 elements (128 bytes
// of fp16/bf16) per step, MFMA = v_mfma_f32_16x16x32_{f16,bf16}, one workgroup barrier per K step:
//   * activations X stream global -> LDS by LDS-DMA (U units), exactly like the scaled GEMM's A;
//   * packed weights are read ONCE per workgroup as 32-bit words straight from the [K/pf][N] tensor (a word
//     holds 8 (int4) / 4 (int8) consecutive k of one column = one 16-byte / 8-byte piece of that column's LDS
//     row) by buffer loads whose per-step advance is a scalar offset, dequantised IN REGISTERS and written to
//     the V units with ds_write_b128, so the MFMA loop reads both operands from LDS with conflict-free
//     ds_read_b128;
//   * dequantisation is bit-identical to the reference's w_ref (quant_utils.py:74, kernels gemm.py:192-210):
//     (q - bias - zp) is formed exactly, then ONE rounding multiply by the fp16/bf16 group scale (ChunkDequant);
//   * every wave dequantises the weights of step t+1 BETWEEN the MFMAs of step t.  A v_mfma_f32_16x16x32
//     occupies the matrix pipe for 16 cycles and the SIMD's vector issue for 8 of them, so two 4-cycle VALU
//     instructions per MFMA are free: the ~100 VALU instructions of a step's dequantisation fit under its
//     32-64 MFMAs.  (Converting right after the barrier, then multiplying, exposes that VALU time -- both waves
//     of a SIMD convert at once: C4 104 us against 85 us.)  The step is written as SLOTS -- one MFMA, then a
//     two-to-four-instruction slice of the dequantisation, a fragment read or a VMEM instruction -- fenced by
//     sched_barrier(0), and the K loop is peeled so that its body is one basic block;
//   * the two waves of a SIMD (w and w + 4) issue the step's VMEM work (4 LDS-DMA pieces, the weight loads of
//     step t+2) at different slots: while one pays the LDS-DMA issue cost the other has the matrix pipe;
//   * work items are (unit row, chunk) pairs dealt so that every thread converts the same number of chunks
//     for every tile width (NT = 3: two chunks of V1 and ONE of V2 -- threads with tq = 1 take chunk cp + 4 of
//     the row their tq = 0 neighbour owns), so there is no divergent branch in the loop;
//   * fp32 accumulation, RNE cast, 16-byte stores (8 consecutive n per lane).
// Tried and dropped (profiles/README.md): a ping-pong K loop with the dequantisation in the load segments, a
// three-stage X ring (LDS-DMA two steps ahead, 160 KiB of LDS), a sched_group_barrier pattern over the whole
// step instead of the hand-made slots.
#include <algorithm>

#include "common.hpp"
#include "gemm.hpp"
#include "dispatch_fit.hpp"
#include "mfma_tile.hpp"
#include "epilogue_rows.hpp"
#include "mixed_dequant.hpp"

namespace conch {
namespace {

using namespace tile;
using mixed::ChunkDequant;

// see ChunkDequant: bf16 x 8-bit weights with a zero point round the difference first
template <int X_DT, int BITS, int ZP, bool PRE = false, bool SILU = false>
#ifdef CONCH_EXP_NO_SHIFTLESS  // A/B variant (tools/ab_lib.py): the int4 upper pairs shifted down as before
using Dequant = ChunkDequant<X_DT, BITS, (X_DT == CONCH_DT_BF16 && BITS == 8 && ZP != CONCH_ZP_NONE), PRE, false>;
#else
using Dequant = ChunkDequant<X_DT, BITS, (X_DT == CONCH_DT_BF16 && BITS == 8 && ZP != CONCH_ZP_NONE), PRE, !SILU>;
#endif

constexpr int kStepK = 64;  // k elements per step (128 bytes of 16-bit)
constexpr int kMixedTall = 5;  // tile code of the 512 x 128 tile (2..4 = 256 x 64 NT)

struct MixedTile {
  f32x4 acc[8][4];
  Frag fm[4];
  Frag fn[2][2];
};

template <int X_DT, int OUT_DT, int NT>
__device__ __forceinline__ void mixed_epilogue(const MixedTile& w, const MixedGemmArgs& p, int bm0, int bn0, int wr,
                                               int wc, int lane) {
  // cast and store; a pair of tiles gives a lane 8 consecutive n (16-byte store), a lone tile 4
  const int g = lane >> 4, jm = lane & 15;
  const bool vec_store = (p.c_stride_m % 8 == 0) && (((uintptr_t)p.c & 15) == 0);
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    if (nh == 1 && NT == 2) continue;
    const bool pair_h = nh == 0 || NT == 4;
    const int width = pair_h ? 8 : 4;
    const int n0 = bn0 + wc * 16 * NT + nh * 32 + width * g;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = bm0 + wr * 128 + mt * 16 + jm;
      if (m >= p.m) continue;
      i32x4 pk;  // one packed convert per output dword
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const int e = 2 * e2;
        const f32x4& a = w.acc[mt][nh * 2 + (pair_h ? (e >> 2) : 0)];
        pk[e2] = (int)pack2_bits16<OUT_DT>(f32x2{a[e & 3], a[(e & 3) + 1]});
      }
      uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n0;
      if (pair_h && vec_store && n0 + 8 <= p.n) {
        *(i32x4*)dst = pk;
      } else if (!pair_h && vec_store && n0 + 4 <= p.n) {
        *(i32x2*)dst = i32x2{pk[0], pk[1]};
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < width && n0 + e < p.n) dst[e] = (uint16_t)((uint32_t)pk[e >> 1] >> (16 * (e & 1)));
      }
    }
  }
}

// Row-major epilogue through LDS for whole tiles (round 3; the scaled kernels' finding, profiles/r03/store_rate.txt: the
// accumulator-layout stores above put 16 different 128-byte lines into every quarter of a store instruction and run at
// 13.6 B/clk/CU whatever else the chip does -- 8-byte stores of a lone third tile at half that -- while whole lines run at
// up to four times the rate).  A wave owns 16 WT columns = 96 bytes of a row at the 192-column tile (C4), not a line, so the
// image is built by the workgroup: every wave parks its cast sub-tile in a [tile rows][tile columns] bf16 / fp16 image in
// the dead operand buffers (16-byte chunk index XOR row & 7 inside each 128-byte line: conflict-free ds_write_b128),
// barrier, and every wave stores its share of the image as whole lines with write-through buffer stores -- 4 rows x 256
// bytes per instruction where a row is a multiple of 256 bytes (a 16-lane group reads one row's 16 chunks: 16 banks
// slots), 8 rows x 128 bytes at the 384-byte rows of the 192-column tile.  Same values as mixed_epilogue, bit for bit.
template <int OUT_DT, int NT, bool TALL>
__device__ __forceinline__ bool mixed_tile_is_whole(const MixedGemmArgs& p, int bm0, int bn0) {
  constexpr int kRows = TALL ? 512 : 256, kCols = TALL ? 128 : 64 * NT;
  return (p.c_stride_m % 8 == 0) && (((uintptr_t)p.c & 15) == 0) && bm0 + kRows <= (int)p.m && bn0 + kCols <= (int)p.n &&
         ((p.m - 1) * p.c_stride_m + p.n) * 2 < ((int64_t)1 << 32);
}

template <int OUT_DT, int NT, bool TALL>
__device__ __forceinline__ void mixed_epilogue_rows(const MixedTile& w, const MixedGemmArgs& p, char* lds, int bm0, int bn0, int wr,
                                                    int wc, int lane, int wave) {
  constexpr int kRows = TALL ? 512 : 256, kCols = TALL ? 128 : 64 * NT;
  constexpr int kPitch = kCols * 2;          // bytes per image row: 256, 384 or 512
  constexpr int WT = TALL ? 4 : NT;          // 16-column MFMA tiles per wave
  constexpr int kWaveChunks = 2 * WT;        // 16-byte chunks per wave and row
  const int g = lane >> 4, jm = lane & 15;
  __syncthreads();  // every wave is past its last operand read: the buffers are dead (a fence too: the raw s_barrier
                    // builtin does not order the stores below for the compiler)
  // ---- park: same lane -> (row, columns) map and the same casts as mixed_epilogue ----
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    if (nh == 1 && WT == 2) continue;
    const bool pair_h = nh == 0 || WT == 4;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int row = wr * 128 + mt * 16 + jm;  // row & 7 == jm & 7
      i32x4 pk;
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const int e = 2 * e2;
        const f32x4& a = w.acc[mt][nh * 2 + (pair_h ? (e >> 2) : 0)];
        pk[e2] = (int)pack2_bits16<OUT_DT>(f32x2{a[e & 3], a[(e & 3) + 1]});
      }
      if (pair_h) {
        const int chunk = wc * kWaveChunks + nh * 4 + g;
        *(i32x4*)(lds + image_chunk_offset<kPitch>(row, chunk)) = pk;
      } else {  // lone third tile: four columns = 8 bytes per lane, two lanes per chunk
        const int chunk = wc * kWaveChunks + 4 + (g >> 1);
        *(i32x2*)(lds + image_chunk_offset<kPitch>(row, chunk) + (g & 1) * 8) = i32x2{pk[0], pk[1]};
      }
    }
  }
  __syncthreads();
  image_store_rows<kRows, kPitch>(lds, p.c, p.m, p.n, p.c_stride_m, bm0, bn0, lane, wave);
}

// Split-K form: the raw fp32 accumulators to this slice's slab [M][N] (N % 4 == 0: 16-byte stores), same lane -> (m, n) map as
// mixed_epilogue; the fp32-slab reduce of gemm_mixed_skinny.hip adds the slices in slice order and casts.
template <int NT>
__device__ __forceinline__ void mixed_epilogue_slab(const MixedTile& w, const MixedGemmArgs& p, int bm0, int bn0, int wr, int wc, int lane,
                                                    float* __restrict__ slab) {
  const int g = lane >> 4, jm = lane & 15;
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    if (nh == 1 && NT == 2) continue;
    const bool pair_h = nh == 0 || NT == 4;
    const int width = pair_h ? 8 : 4;
    const int n0 = bn0 + wc * 16 * NT + nh * 32 + width * g;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = bm0 + wr * 128 + mt * 16 + jm;
      if (m >= p.m) continue;
      float* dst = slab + (int64_t)m * p.n + n0;
      if (n0 + 4 <= p.n) *(f32x4*)dst = w.acc[mt][nh * 2];
      if (pair_h && n0 + 8 <= p.n) *(f32x4*)(dst + 4) = w.acc[mt][nh * 2 + 1];
    }
  }
}

// Fused gate/up epilogue: g = cast(acc_gate), u = cast(acc_up) (what the plain kernel would store), then the reference's
// silu_and_mul with its roundings: s = cast(silu(g)), y = cast(s * u) (reference/activation/silu_and_mul.py:13-16).
template <int OUT_DT>
__device__ __forceinline__ void mixed_epilogue_silu(const MixedTile& w, const MixedGemmArgs& p, int bm0, int bn0, int wr, int wc, int lane) {
  const int g = lane >> 4, jm = lane & 15;
  const bool vec_store = (p.c_stride_m % 8 == 0) && (((uintptr_t)p.c & 15) == 0);
  const int n0 = bn0 + wc * 32 + 8 * g;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = bm0 + wr * 128 + mt * 16 + jm;
    if (m >= p.m) continue;
    i32x4 pk;
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      const int e = 2 * e2;
      const f32x4& ag = w.acc[mt][e >> 2];
      const f32x4& au = w.acc[mt][2 + (e >> 2)];
      const f32x2 gv = unpack2_bits16<OUT_DT>(pack2_bits16<OUT_DT>(f32x2{ag[e & 3], ag[(e & 3) + 1]}));
      const uint32_t ub = pack2_bits16<OUT_DT>(f32x2{au[e & 3], au[(e & 3) + 1]});
      const f32x2 sv = unpack2_bits16<OUT_DT>(pack2_bits16<OUT_DT>(pin_f32x2(f32x2{act_f32(gv[0], p.fuse_silu), act_f32(gv[1], p.fuse_silu)})));
      pk[e2] = (int)pack2_bits16<OUT_DT>(pin_f32x2(sv * unpack2_bits16<OUT_DT>(ub)));
    }
    uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n0;
    if (vec_store && n0 + 8 <= p.n) {
      *(i32x4*)dst = pk;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (n0 + e < p.n) dst[e] = (uint16_t)((uint32_t)pk[e >> 1] >> (16 * (e & 1)));
    }
  }
}

// Weight words, scales and zero points of one thread for one K step.
template <int BITS, int NT>
struct WeightRegs {
  static constexpr int kWpc = BITS == 4 ? 1 : 2;
  uint32_t w[NT][kWpc];  // items 0,1: V1 chunks cp, cp+4; items 2..: V2
  uint32_t scale[2];
  int zp[2];
};

struct WeightLane {  // per-lane constants
  int vq[4];         // byte offset of each item's first word in w_q, less the wave-uniform (step, cp) part
  int vs[2];         // byte offsets of the lane's V1 / V2 column in a row of w_s
  int vz[2];         // ... of w_zp
  int lds[4];        // byte offset, inside a weight buffer, of each item's 16-byte destination
};

struct WeightSrc {   // wave-uniform
  __amdgpu_buffer_rsrc_t q, s, z;
  int q_row;         // bytes per word row of w_q
  int q_step;        // bytes per K step of w_q
  int s_group;       // bytes per group row of w_s
  int z_group;       // bytes per group row of w_zp
  int steps_per_group;
};

struct WeightCursor {  // wave-uniform position of the NEXT weight load
  int q, s, z, left;
};

__device__ __forceinline__ void advance(WeightCursor& c, const WeightSrc& u) {
  const bool wrap = c.left == 1;
  c.q += u.q_step;
  c.left = wrap ? u.steps_per_group : c.left - 1;
  c.s += wrap ? u.s_group : 0;
  c.z += wrap ? u.z_group : 0;
}

// PRE: the words come from the image conch_prepack_mixed_weights laid out -- the NT x kWpc words of a thread and K step are
// contiguous (ONE 8 / 12 / 16-byte load per thread instead of NT 4-byte ones, 1 KiB contiguous per wave-instruction); ln.vq[0]
// is the thread's byte offset inside a (tile column, K step) record, c.q the record's offset.
template <int BITS, int ZP, int NT, bool PRE = false>
__device__ __forceinline__ void load_weights(WeightRegs<BITS, NT>& r, const WeightLane& ln, const WeightSrc& u, const WeightCursor& c) {
  constexpr int kWpc = WeightRegs<BITS, NT>::kWpc;
  if constexpr (PRE) {
#pragma unroll
    for (int i = 0; i < kWpc; ++i) {  // half i of the thread's record: words [it][i] for it = 0 .. NT-1
      const int soff = c.q + i * NT * 4;
      if constexpr (NT == 4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(u.q, ln.vq[0], soff, 0);
#pragma unroll
        for (int it = 0; it < 4; ++it) r.w[it][i] = v[it];
      } else if constexpr (NT == 3) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b96(u.q, ln.vq[0], soff, 0);
#pragma unroll
        for (int it = 0; it < 3; ++it) r.w[it][i] = v[it];
      } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(u.q, ln.vq[0], soff, 0);
#pragma unroll
        for (int it = 0; it < 2; ++it) r.w[it][i] = v[it];
      }
    }
  } else {
#pragma unroll
  for (int it = 0; it < NT; ++it)
#pragma unroll
    for (int i = 0; i < kWpc; ++i) r.w[it][i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(u.q, ln.vq[it], c.q + i * u.q_row, 0);
  }
#pragma unroll
  for (int un = 0; un < (NT > 2 ? 2 : 1); ++un) {
    r.scale[un] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(u.s, ln.vs[un], c.s, 0);
    if constexpr (ZP == CONCH_ZP_TENSOR) r.zp[un] = (int)__builtin_amdgcn_raw_buffer_load_b32(u.z, ln.vz[un], c.z, 0);
    else r.zp[un] = 0;
  }
}

template <int X_DT>
__device__ __forceinline__ void mma1(f32x4& acc, const Frag& fa, const Frag& fb, int h) {
  const i32x4 a = h == 0 ? fa.lo : fa.hi, b = h == 0 ? fb.lo : fb.hi;
  if constexpr (X_DT == CONCH_DT_FP16) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// LDS layout of the interleaved form (160 KiB): a ring of THREE X stages (U1 + U2 = 32 KiB each) and two weight
// buffers (V1 + V2 = 32 KiB each).  X of step t+2 is requested during step t, so an LDS-DMA piece has two full
// steps to land; the weights of step t+2 are loaded to registers during step t, dequantised during step t+1.
// LDS: two stages, each X (U1 + U2 = 32 KiB) followed by the dequantised weights (V1 + V2 = 32 KiB).
constexpr int kXBytes = 2 * kUnitBytes;
constexpr int kStageBytes = 2 * kXBytes;
constexpr int kMixedLdsBytes = 2 * kStageBytes;  // 128 KiB
// TALL form: a tile of 512 rows x 128 columns (wave-rows 0..3 of 128 rows, wave-columns 0..1 of 64 columns -- a wave still owns
// 128 x 64, so the MFMA phases are those of the 256-column tile).  Per K step the same 64 MFMAs per wave but HALF the weights to
// dequantise (the loop's limiter is the vector issue port) against twice the X bytes to stage.  A stage is X (512 rows x 128 B,
// row r of the block at byte 128 r: wave-row wr at 16 KiB wr) followed by ONE weight unit laid out like the 128-column
// tile's (NT = 2: 32 columns per unit-row quarter, pairs of MFMA tiles interleaved); two stages = all 160 KiB of the CU.
constexpr int kTallRows = 512;
constexpr int kTallXBytes = kTallRows * kStepBytes;              // 64 KiB
constexpr int kTallStageBytes = kTallXBytes + kUnitBytes;        // 80 KiB
constexpr int kTallLdsBytes = 2 * kTallStageBytes;               // 160 KiB

template <int UNIT2, int J>
__device__ __forceinline__ void stage_x_piece(char* lds, const Srcs& src, const StageOffsets& so, int wave, int tile) {
  char* dst = lds + (tile & 1) * kStageBytes + UNIT2 * kUnitBytes + wave * 2048 + J * 1024;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(src.a, (lds_void_t*)dst, 16, so.off[UNIT2 ? kU2 : kU1][J], tile * kStepBytes, 0, 0);
}

// TALL: piece J (0..7) of the 8 KiB of X rows [64 wave, 64 wave + 64) this wave stages per step; so.off[J >> 1][J & 1] is the
// lane's source offset
template <int J>
__device__ __forceinline__ void stage_x_piece_tall(char* lds, const Srcs& src, const StageOffsets& so, int wave, int tile) {
  char* dst = lds + (tile & 1) * kTallStageBytes + wave * 8192 + J * 1024;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(src.a, (lds_void_t*)dst, 16, so.off[J >> 1][J & 1], tile * kStepBytes, 0, 0);
}

struct LoopCtx {
#ifdef CONCH_CLOCK_PROBE_STEPS
  // -DCONCH_CLOCK_PROBE_STEPS (on top of the diagnostic twin's -DCONCH_CLOCK_PROBE): time at the step-top wait / barrier.  Off
  // by default: s_memtime is a scalar-memory operation, and with one outstanding the compiler must open every step's fragment
  // waits with lgkmcnt(0) -- the stamps change the loop they measure.
  mutable unsigned long long wait_cycles = 0, barrier_cycles = 0;
#endif
  Srcs src;
  StageOffsets so;
  WeightLane ln;
  WeightSrc ws;
  int wave, m_base, n_base, off_base;
  int m_base_hi, n_base_hi;        // the same fragment offsets with the 16-byte chunk index + 4 (byte offset ^ 64)
  uint32_t and_mask, or_magic, and_mask_hi;  // ChunkDequant::slice's constants, opaque to the compiler
};

// One K step.  MODE 0: steps t+1 and t+2 exist; 1: t+1 exists; 2: last step.  ISSUE = the slot at which this
// wave starts issuing the step's VMEM work.
// DIAG (diagnostic twin only, timing experiments with WRONG results): 1 = no MFMAs (fragments still read), 2 = no fragment reads,
// 3 = no dequantisation (no VALU slices, no ds_write), 4 = no LDS-DMA of X, 5 = no weight loads.
template <int X_DT, int BITS, int ZP, int NT, int MODE, int ISSUE, bool DEFER, bool PRE = false, bool TALL = false, int DIAG = 0, bool SILU = false>
__device__ __forceinline__ void mixed_step(MixedTile& w, WeightRegs<BITS, NT>& regs, char* lds, const LoopCtx& c, WeightCursor& cur, int t) {
  static_assert(!TALL || NT == 2, "the tall tile dequantises two items per thread (the 128-column weight unit)");
  constexpr int N1 = TALL ? 2 : NT - 2;  // MFMA n tiles of a wave beyond the first two
  constexpr int kStage = TALL ? kTallStageBytes : kStageBytes;
  constexpr int kXB = TALL ? kTallXBytes : kXBytes;
  constexpr int kHalfM = TALL ? 4 * 2048 : kUnitBytes;  // from m tile i to m tile 4 + i of the wave
  constexpr int kHalfN = TALL ? 2 * 2048 : kUnitBytes;  // from n tile tt to n tile 2 + tt
  constexpr int kXPieces = TALL ? 8 : 4;
#ifdef CONCH_EXP_HORDER
  constexpr bool kHOrder = true;   // experiment: the two K halves of one accumulator NOT back to back (h outside tt)
#else
  constexpr bool kHOrder = false;
#endif
  constexpr int kWpc = WeightRegs<BITS, NT>::kWpc;
  constexpr int kSlices = Dequant<X_DT, BITS, ZP, PRE, SILU>::kSlices;
  static_assert(kSlices <= 16, "more slices per chunk than MFMA slots per chunk");
  // everything this wave staged for step t (LDS-DMA, ds_write) and loaded for step t+1 is complete ...
#ifdef CONCH_CLOCK_PROBE_STEPS
  const unsigned long long probe_a = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long probe_b = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_barrier();
  const unsigned long long probe_c = __builtin_amdgcn_s_memtime();
  c.wait_cycles += probe_b - probe_a;
  c.barrier_cycles += probe_c - probe_b;
#else
  // (Inline asm, so the compiler's counter model does not know the counters are zero here: it re-waits for the previous step's
  // weight loads after the barrier and opens the first wave's fragment waits with lgkmcnt(0).  The same wait as
  // __builtin_amdgcn_s_waitcnt, which the model does see, gives counted waits throughout -- and measures within +-0.4 % on the
  // 192- and 256-column tiles, 4-6 % SLOWER on the 128-column one: profiles/r02/mixed_loop_ab.txt.)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // ... and every wave is done reading the other stage
#endif
  const int ubuf = (t & 1) * kStage, vbuf = ubuf + kXB;
  const int vnext = ((t + 1) & 1) * kStage + kXB;
  // four per-step base addresses (X / weights, chunk g / g + 4); every fragment read is base + an immediate offset
  const char* ulo = lds + ubuf + c.m_base;
  const char* uhi = lds + ubuf + c.m_base_hi;
  const char* vlo = lds + vbuf + c.n_base;
  const char* vhi = lds + vbuf + c.n_base_hi;
  auto frag = [](const char* lo, const char* hi, int off) {
    Frag f;
    f.lo = *(const i32x4*)(lo + off);
    f.hi = *(const i32x4*)(hi + off);
    return f;
  };
  auto mma = [](f32x4& acc, const Frag& fa, const Frag& fb, int h) {
    if constexpr (DIAG == 1) asm volatile("" ::"v"(fa.lo), "v"(fa.hi), "v"(fb.lo), "v"(fb.hi));  // keep the reads alive
    else mma1<X_DT>(acc, fa, fb, h);
  };
  WeightRegs<BITS, NT> next;

  Dequant<X_DT, BITS, ZP, PRE, SILU> cv[NT];
  int slot = 0;  // a constant in every unrolled copy
  auto tail = [&](int sl) {
    // the step's VMEM work from slot ISSUE on: one per slot the four LDS-DMA pieces of X of step t+1, then the
    // packed words / scales / zero points of step t+2
    if constexpr (MODE <= 1 && !TALL && DIAG != 4) {
      if (sl == ISSUE + 0) stage_x_piece<0, 0>(lds, c.src, c.so, c.wave, t + 1);
      if (sl == ISSUE + 1) stage_x_piece<0, 1>(lds, c.src, c.so, c.wave, t + 1);
      if (sl == ISSUE + 2) stage_x_piece<1, 0>(lds, c.src, c.so, c.wave, t + 1);
      if (sl == ISSUE + 3) stage_x_piece<1, 1>(lds, c.src, c.so, c.wave, t + 1);
    }
    if constexpr (MODE <= 1 && TALL && DIAG != 4) {
      if (sl == ISSUE + 0) stage_x_piece_tall<0>(lds, c.src, c.so, c.wave, t + 1);
      if (sl == ISSUE + 1) stage_x_piece_tall<1>(lds, c.src, c.so, c.wave, t + 1);
      if (sl == ISSUE + 2) stage_x_piece_tall<2>(lds, c.src, c.so, c.wave, t + 1);
      if (sl == ISSUE + 3) stage_x_piece_tall<3>(lds, c.src, c.so, c.wave, t + 1);
      if (sl == ISSUE + 4) stage_x_piece_tall<4>(lds, c.src, c.so, c.wave, t + 1);
      if (sl == ISSUE + 5) stage_x_piece_tall<5>(lds, c.src, c.so, c.wave, t + 1);
      if (sl == ISSUE + 6) stage_x_piece_tall<6>(lds, c.src, c.so, c.wave, t + 1);
      if (sl == ISSUE + 7) stage_x_piece_tall<7>(lds, c.src, c.so, c.wave, t + 1);
    }
    if constexpr (MODE == 0 && DIAG != 5) {
      if (sl == ISSUE + kXPieces) {
        load_weights<BITS, ZP, NT, PRE>(next, c.ln, c.ws, cur);
        advance(cur, c.ws);
      }
    }
    if constexpr (MODE <= 1 && DIAG != 3) {
      const int it = sl / kSlices, sub = sl % kSlices;
      if (it < NT) {
        const int un = it < 2 ? 0 : 1;
        cv[it].slice(sub, regs.w[it][0], regs.w[it][kWpc - 1], c.off_base + regs.zp[un], regs.scale[un], lds + vnext + c.ln.lds[it], c.and_mask, c.or_magic, c.and_mask_hi);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // phase 3 = (m rows 64-127) x (n tiles 0,1): 16 MFMAs on fragments that are in registers when phase 2 ends
  auto phase3 = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int tt = kHOrder ? q % 2 : q / 2, h = kHOrder ? q / 2 : q % 2;
        mma(w.acc[4 + i][tt], w.fn[0][tt], w.fm[i], h);
        tail(slot++);
      }
  };
  // DEFER (waves 4-7, the second wave of every SIMD): the previous step's phase 3 runs here, after the barrier,
  // and needs no LDS access, so the matrix pipe has work while the other wave of the SIMD waits for its first
  // fragments, and the two waves' read bursts, LDS-DMA issue and ds_writes no longer coincide.  (At t = 0 the
  // fragment registers hold zeros: sixteen MFMAs that add nothing.)
  if constexpr (DEFER) phase3();
  // phase-0 fragment reads, ahead of the slots that hide their latency
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) if constexpr (DIAG != 2) w.fn[0][tt] = frag(vlo, vhi, tt * 2048);
#pragma unroll
  for (int i = 0; i < 4; ++i) if constexpr (DIAG != 2) w.fm[i] = frag(ulo, uhi, i * 2048);
  __builtin_amdgcn_sched_barrier(0);
  // phase 0: (m rows 0-63 of the wave) x (n tiles 0,1); the V2 fragments of phase 1 -- or, for the narrowest
  // tile, the U2 fragments -- are fetched underneath
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tt = kHOrder ? q % 2 : q / 2, h = kHOrder ? q / 2 : q % 2;
        mma(w.acc[i][tt], w.fn[0][tt], w.fm[i], h);
        const int idx = i * 4 + q;
        if constexpr (N1 > 0) {
          if (idx >= 2 && idx < 2 + N1) if constexpr (DIAG != 2) w.fn[1][idx - 2] = frag(vlo, vhi, kHalfN + (idx - 2) * 2048);
        } else {
          if (idx % 4 == 3) if constexpr (DIAG != 2) w.fm[i] = frag(ulo, uhi, kHalfM + i * 2048);
        }
        tail(slot++);
      }
  // phase 1: same m rows x n tiles 2..; each U1 fragment is replaced by its U2 counterpart after its last use
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 2 * N1; ++q) {
        const int tt = kHOrder ? q % (N1 > 0 ? N1 : 1) : q / 2, h = kHOrder ? q / (N1 > 0 ? N1 : 1) : q % 2;
        mma(w.acc[i][2 + tt], w.fn[1][tt], w.fm[i], h);
        if (q == 2 * N1 - 1) if constexpr (DIAG != 2) w.fm[i] = frag(ulo, uhi, kHalfM + i * 2048);
        tail(slot++);
      }
  // phase 2: m rows 64-127 x n tiles 2..
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 2 * N1; ++q) {
        const int tt = kHOrder ? q % (N1 > 0 ? N1 : 1) : q / 2, h = kHOrder ? q / (N1 > 0 ? N1 : 1) : q % 2;
        mma(w.acc[4 + i][2 + tt], w.fn[1][tt], w.fm[i], h);
        tail(slot++);
      }
  if constexpr (!DEFER) phase3();
  if constexpr (MODE == 0 && DIAG != 5) regs = next;
}

template <int X_DT, int BITS, int ZP, int NT, int ISSUE, bool DEFER, bool PRE = false, bool TALL = false, int DIAG = 0, bool SILU = false>
__device__ __forceinline__ void mixed_mainloop(MixedTile& w, WeightRegs<BITS, NT>& regs, char* lds, const LoopCtx& c, WeightCursor& cur, int steps) {
  if constexpr (DEFER || DIAG == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) w.fm[i] = Frag{i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}};
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) w.fn[0][tt] = Frag{i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}};
    if constexpr (DIAG == 2) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) w.fn[1][tt] = Frag{i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}};
    }
  }
  int t = 0;
  for (; t + 2 < steps; ++t) mixed_step<X_DT, BITS, ZP, NT, 0, ISSUE, DEFER, PRE, TALL, DIAG, SILU>(w, regs, lds, c, cur, t);
  if (steps > 1) {
    mixed_step<X_DT, BITS, ZP, NT, 1, ISSUE, DEFER, PRE, TALL, DIAG, SILU>(w, regs, lds, c, cur, t);
    ++t;
  }
  mixed_step<X_DT, BITS, ZP, NT, 2, ISSUE, DEFER, PRE, TALL, DIAG, SILU>(w, regs, lds, c, cur, t);
  if constexpr (DEFER) {  // the last step's phase 3
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int h = 0; h < 2; ++h) mma1<X_DT>(w.acc[4 + i][tt], w.fn[0][tt], w.fm[i], h);
  }
}

#ifdef CONCH_CLOCK_PROBE
__device__ unsigned long long g_probe_mixed[kProbeBlocks * 8];
#endif

template <int X_DT, int OUT_DT, int BITS, int ZP, int NT, bool SILU = false, bool PRE = false, bool TALL = false, int DIAG = 0>
__global__ __launch_bounds__(kThreads, 2) void mixed_gemm_kernel(MixedGemmArgs p) {
  static_assert(!(SILU && PRE), "the fused gate/up form reads the plain [K/pf][N] layout");
  static_assert(!SILU || NT == 4, "the fused gate/up form pairs the V1 and V2 units of a full-width tile");
  static_assert(!TALL || (NT == 2 && !SILU), "the tall tile is 512 rows x 128 columns");
  constexpr int kRows = TALL ? kTallRows : kTileM;
  // SILU (conch_mixed_precision_gemm_silu_and_mul): a tile is 256 rows x 128 OUTPUT columns; a wave-column owns 32 of
  // them, V1 holds their gate columns and V2 the up columns p.n further right, so accumulator tiles 0,1 / 2,3 of a lane
  // are gate / up of the same eight outputs (same construction as gemm_mfma.hip's fused epilogue).
  constexpr int kTileW = SILU ? 128 : 64 * NT;
  constexpr int kWpc = WeightRegs<BITS, NT>::kWpc;
  constexpr int kWordRowsPerStep = kStepK * BITS / 32;
  __shared__ __attribute__((aligned(1024))) char lds[TALL ? kTallLdsBytes : kMixedLdsBytes];
  LoopCtx c;
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int wr = TALL ? c.wave >> 1 : c.wave >> 2, wc = TALL ? c.wave & 1 : c.wave & 3;
  const int tiles_m = ((int)p.m + kRows - 1) / kRows;
  const int tiles_n = ((int)p.n + kTileW - 1) / kTileW;
  const TileCoord tc = map_tile(blockIdx.x, tiles_m, tiles_n, p.raster_magic, p.raster_shift);
  const int bm0 = tc.tm * kRows, bn0 = tc.tn * kTileW;

  // split-K form: this workgroup's K range starts at step0 (0 in the plain form); the X buffer is rebased there, the weight
  // cursor starts there, the loop below counts its own steps from 0
  const int total_steps = (int)(p.k / kStepK);
  const int step0 = p.split_steps ? (int)blockIdx.y * p.split_steps : 0;
  // activations: LDS-DMA units (byte strides: 2 bytes per element)
  if constexpr (TALL) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = 64 * c.wave + 8 * j + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      c.so.off[j >> 1][j & 1] = min(bm0 + row, (int)p.m - 1) * ((int)p.x_stride_m * 2) + chunk * 16;
    }
  } else {
    c.so = make_stage_offsets(c.wave, lane, bm0, bn0, (int)p.m - 1, 0, (int)p.x_stride_m * 2, 0);
  }
  c.src.a = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x + (int64_t)step0 * kStepBytes), 0,
                                              (uint32_t)(((p.m - 1) * p.x_stride_m + p.k) * 2 - step0 * kStepBytes), 0x00020000);
  c.src.b = c.src.a;

  // weight work items of this thread: unit row rho of V1, chunks cp and cp + 4; V2: see the file header.
  // Wave-column rho>>5 owns columns [16*NT*wc', 16*NT*(wc'+1)); a PAIR of MFMA tiles is interleaved (rows 4g+e of
  // tile t <-> n = 8g+e+4t: a lane ends up with 8 consecutive n), a LONE third tile (NT == 3) is plain.
  const int rho = threadIdx.x & 127;
  const int cp = c.wave >> 1;  // = threadIdx.x >> 7, wave-uniform
  const int r5 = rho & 31, r = r5 & 15, tq = r5 >> 4;
  const int wcol = (rho >> 5) * (SILU ? 32 : 16 * NT);
  const int pair = 8 * (r >> 2) + (r & 3) + 4 * tq;
  const int row2 = NT == 4 ? rho : (rho & ~16);
  const int n1 = min(bn0 + wcol + pair, (int)p.n - 1);
  const int n2 = SILU ? n1 + (int)p.n : min(bn0 + wcol + 32 + (NT == 4 ? pair : r), (int)p.n - 1);
  const int64_t w_cols = SILU ? 2 * p.n : p.n;  // columns of the weight / scale / zero-point arrays
  {
    const int row[4] = {rho, rho, row2, row2};
    const int dchunk[4] = {0, 4, NT == 4 ? 0 : 4 * tq, 4};  // chunk - cp
    const int ncol[4] = {n1, n1, n2, n2};
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int chunk = cp + dchunk[it];
      c.ln.vq[it] = (dchunk[it] * kWpc * (int)p.wq_stride_k + ncol[it]) * 4;
      c.ln.lds[it] = (it < 2 ? 0 : kUnitBytes) + (row[it] >> 3) * 1024 + (row[it] & 7) * 128 + ((chunk ^ ((row[it] >> 1) & 7)) * 16);
    }
    c.ln.vs[0] = n1 * 2; c.ln.vs[1] = n2 * 2;
    c.ln.vz[0] = n1 * 4; c.ln.vz[1] = n2 * 4;
  }
  const int64_t word_rows = p.k * BITS / 32, groups = p.k / p.group_size;
  c.ws.q = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_q, 0, (uint32_t)(((word_rows - 1) * p.wq_stride_k + w_cols) * 4), 0x00020000);
  c.ws.s = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_s, 0, (uint32_t)(((groups - 1) * p.ws_stride_g + w_cols) * 2), 0x00020000);
  c.ws.z = c.ws.s;
  if constexpr (ZP == CONCH_ZP_TENSOR)
    c.ws.z = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_zp, 0, (uint32_t)(((groups - 1) * p.wzp_stride_g + w_cols) * 4), 0x00020000);
  c.ws.q_row = (int)p.wq_stride_k * 4;
  c.ws.q_step = kWordRowsPerStep * c.ws.q_row;
  if constexpr (PRE) {  // records of 512 threads x NT x kWpc words per (tile column, K step): mixed_prepack.hpp
    constexpr int kRecord = 512 * NT * kWpc * 4;
    const int64_t image_bytes = (int64_t)tiles_n * (p.k / kStepK) * kRecord;
    c.ws.q = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_q, 0, (uint32_t)image_bytes, 0x00020000);
    c.ws.q_step = kRecord;
    c.ln.vq[0] = (int)threadIdx.x * NT * kWpc * 4;
  }
  c.ws.s_group = (int)p.ws_stride_g * 2;
  c.ws.z_group = (int)p.wzp_stride_g * 4;
  c.ws.steps_per_group = p.group_size / kStepK;
  const int group0 = step0 / c.ws.steps_per_group;  // split_steps is a multiple of steps_per_group
  WeightCursor cur = {(PRE ? tc.tn * total_steps * c.ws.q_step : cp * kWpc * c.ws.q_row) + step0 * c.ws.q_step, group0 * c.ws.s_group,
                      group0 * c.ws.z_group, c.ws.steps_per_group};
  c.off_base = p.weight_bias + (ZP == CONCH_ZP_SCALAR ? p.w_zp[0] : 0);

  // fragment read offsets
  const int fr = lane & 15, fg = lane >> 4;
  const int lane_off = (fr >> 3) * 1024 + (fr & 7) * 128 + ((fg ^ ((fr >> 1) & 7)) * 16);
  c.m_base = (TALL ? wr * 16 : wr * 8) * 1024 + lane_off;
  c.n_base = (TALL ? wc * 8 : wc * 4) * 1024 + lane_off;
  c.m_base_hi = c.m_base ^ 64;
  c.n_base_hi = c.n_base ^ 64;
  c.and_mask = BITS == 4 ? 0x000f000fu : 0x00ff00ffu;
  c.or_magic = 0x64006400u;
  c.and_mask_hi = c.and_mask << 4;
  asm volatile("" : "+v"(c.and_mask), "+s"(c.or_magic), "+v"(c.and_mask_hi));

  MixedTile w;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) w.acc[i][j] = f32x4{0, 0, 0, 0};

  const int steps = p.split_steps ? min(p.split_steps, total_steps - step0) : total_steps;
  WeightRegs<BITS, NT> regs;
  // prologue: X and weights of step 0 (converted at once), weights of step 1 to registers
  if constexpr (TALL) {
    stage_x_piece_tall<0>(lds, c.src, c.so, c.wave, 0);
    stage_x_piece_tall<1>(lds, c.src, c.so, c.wave, 0);
    stage_x_piece_tall<2>(lds, c.src, c.so, c.wave, 0);
    stage_x_piece_tall<3>(lds, c.src, c.so, c.wave, 0);
    stage_x_piece_tall<4>(lds, c.src, c.so, c.wave, 0);
    stage_x_piece_tall<5>(lds, c.src, c.so, c.wave, 0);
    stage_x_piece_tall<6>(lds, c.src, c.so, c.wave, 0);
    stage_x_piece_tall<7>(lds, c.src, c.so, c.wave, 0);
  } else {
    stage_x_piece<0, 0>(lds, c.src, c.so, c.wave, 0);
    stage_x_piece<0, 1>(lds, c.src, c.so, c.wave, 0);
    stage_x_piece<1, 0>(lds, c.src, c.so, c.wave, 0);
    stage_x_piece<1, 1>(lds, c.src, c.so, c.wave, 0);
  }
  load_weights<BITS, ZP, NT, PRE>(regs, c.ln, c.ws, cur);
  advance(cur, c.ws);
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    Dequant<X_DT, BITS, ZP, PRE, SILU> cv;
#pragma unroll
    for (int sl = 0; sl < Dequant<X_DT, BITS, ZP, PRE, SILU>::kSlices; ++sl)
      cv.slice(sl, regs.w[it][0], regs.w[it][kWpc - 1], c.off_base + regs.zp[it < 2 ? 0 : 1], regs.scale[it < 2 ? 0 : 1],
               lds + (TALL ? kTallXBytes : kXBytes) + c.ln.lds[it], c.and_mask, c.or_magic, c.and_mask_hi);
  }
  if (steps > 1) {
    load_weights<BITS, ZP, NT, PRE>(regs, c.ln, c.ws, cur);
    advance(cur, c.ws);
  }

  // The two waves of a SIMD (w and w + 4) issue their VMEM work at different places of the step.
  CONCH_PROBE(g_probe_mixed, 0);
#ifdef CONCH_EXP_SETPRIO
  if ((c.wave >> 2) != 0) __builtin_amdgcn_s_setprio(CONCH_EXP_SETPRIO);
#endif
  if ((c.wave >> 2) == 0) mixed_mainloop<X_DT, BITS, ZP, NT, 0, false, PRE, TALL, DIAG, SILU>(w, regs, lds, c, cur, steps);
  else mixed_mainloop<X_DT, BITS, ZP, NT, 16, true, PRE, TALL, DIAG, SILU>(w, regs, lds, c, cur, steps);
  CONCH_PROBE(g_probe_mixed, 1);
#ifdef CONCH_CLOCK_PROBE_STEPS
  if (threadIdx.x == 0 && blockIdx.x < kProbeBlocks) {  // slots 4..7 of the block: cycles at the step-top wait / barrier (wave 0)
    g_probe_mixed[blockIdx.x * 8 + 4] = c.wait_cycles;
    g_probe_mixed[blockIdx.x * 8 + 5] = c.barrier_cycles;
  }
  if (threadIdx.x == 256 && blockIdx.x < kProbeBlocks) {  // slots 6, 7: the same for wave 4 (second wave of SIMD 0, late VMEM issue)
    g_probe_mixed[blockIdx.x * 8 + 6] = c.wait_cycles;
    g_probe_mixed[blockIdx.x * 8 + 7] = c.barrier_cycles;
  }
#endif

  if constexpr (SILU) {
    mixed_epilogue_silu<OUT_DT>(w, p, bm0, bn0, wr, wc, lane);
  } else {
    if (p.slabs) mixed_epilogue_slab<TALL ? 4 : NT>(w, p, bm0, bn0, wr, wc, lane, p.slabs + (int64_t)blockIdx.y * p.m * p.n);
    else if (p.rows_epilogue && mixed_tile_is_whole<OUT_DT, NT, TALL>(p, bm0, bn0))  // workgroup-uniform
      mixed_epilogue_rows<OUT_DT, NT, TALL>(w, p, lds, bm0, bn0, wr, wc, lane, c.wave);
    else mixed_epilogue<X_DT, OUT_DT, TALL ? 4 : NT>(w, p, bm0, bn0, wr, wc, lane);
  }
}

inline int split_slices(const MixedGemmArgs& p) {
  return p.split_steps ? (int)((p.k / kStepK + p.split_steps - 1) / p.split_steps) : 1;
}

template <int X_DT, int OUT_DT, int BITS, int NT>
int launch_zp(const MixedGemmArgs& p, hipStream_t stream) {
  const int tiles_m = (int)((p.m + kTileM - 1) / kTileM);
  if constexpr (NT == 4) {
    if (p.fuse_silu) {
      const dim3 grid((unsigned)(tiles_m * (int)((p.n + 127) / 128)));
      switch (p.zp_mode) {
        case CONCH_ZP_NONE:
          hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_NONE, 4, true>), grid, dim3(kThreads), 0, stream, p);
          break;
        case CONCH_ZP_SCALAR:
          hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_SCALAR, 4, true>), grid, dim3(kThreads), 0, stream, p);
          break;
        default:  // per-group zero points at this width do not fit the register file (mixed_gemm_silu_fused_supported)
          set_error("mixed_precision_gemm_silu_and_mul: no fused kernel for per-group zero points");
          return CONCH_ERR_UNSUPPORTED;
      }
      return check_launch("mixed_gemm_mfma_silu");
    }
  }
  const int tiles_n = (int)((p.n + 64 * NT - 1) / (64 * NT));
  const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)split_slices(p));
  if (p.prepacked) {  // weights in the image of conch_prepack_mixed_weights for THIS tile width; instantiated for OUT_DT == X_DT
    if constexpr (X_DT == OUT_DT) {
      switch (p.zp_mode) {
        case CONCH_ZP_NONE:
          hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_NONE, NT, false, true>), grid, dim3(kThreads), 0, stream, p);
          break;
        case CONCH_ZP_SCALAR:
          hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_SCALAR, NT, false, true>), grid, dim3(kThreads), 0, stream, p);
          break;
        default:
          if constexpr (NT == 4) {
            set_error("mixed_precision_gemm: 256-column tiles are not built for per-group zero points");
            return CONCH_ERR_UNSUPPORTED;
          } else {
            hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_TENSOR, NT, false, true>), grid, dim3(kThreads), 0, stream, p);
          }
          break;
      }
      return check_launch("mixed_gemm_mfma_prepacked");
    } else {
      set_error("mixed_precision_gemm (prepacked): output dtype must equal the activation dtype");
      return CONCH_ERR_UNSUPPORTED;
    }
  }
#ifdef CONCH_CLOCK_PROBE
  if constexpr (X_DT == CONCH_DT_FP16 && OUT_DT == CONCH_DT_FP16 && BITS == 4 && NT == 4) {
    const int diag = tuning(CONCH_TUNE_DIAG);  // diagnostic twin: timing experiments of the K loop (see mixed_step)
    if (diag >= 1 && diag <= 5 && p.zp_mode == CONCH_ZP_NONE) {
#define CONCH_DIAG(D) hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_NONE, NT, false, false, false, D>), grid, dim3(kThreads), 0, stream, p)
      if (diag == 1) CONCH_DIAG(1);
      else if (diag == 2) CONCH_DIAG(2);
      else if (diag == 3) CONCH_DIAG(3);
      else if (diag == 4) CONCH_DIAG(4);
      else CONCH_DIAG(5);
#undef CONCH_DIAG
      return check_launch("mixed_gemm_mfma_diag");
    }
  }
#endif
  switch (p.zp_mode) {
    case CONCH_ZP_NONE:
      hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_NONE, NT>), grid, dim3(kThreads), 0, stream, p);
      break;
    case CONCH_ZP_SCALAR:
      hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_SCALAR, NT>), grid, dim3(kThreads), 0, stream, p);
      break;
    default:
      // per-group zero points: 192 columns at most -- at 256 the zero-point registers on top of the 128 accumulators spilled
      // (4 to 7 dwords of scratch in the K loop, whose traffic also counts against the loop's vmcnt waits)
      if constexpr (NT == 4) {
        set_error("mixed_precision_gemm: 256-column tiles are not built for per-group zero points");
        return CONCH_ERR_UNSUPPORTED;
      } else {
        hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_TENSOR, NT>), grid, dim3(kThreads), 0, stream, p);
      }
      break;
  }
  return check_launch("mixed_gemm_mfma");
}

// The 512 x 128 tile (kMixedTall): plain or pre-packed (the NT = 2 image) weights, every zero-point mode.
template <int X_DT, int OUT_DT, int BITS>
int launch_tall(const MixedGemmArgs& p, hipStream_t stream) {
  const int tiles_m = (int)((p.m + kTallRows - 1) / kTallRows), tiles_n = (int)((p.n + 127) / 128);
  const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)split_slices(p));
#define CONCH_TALL(ZP, PRE) hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, ZP, 2, false, PRE, true>), grid, dim3(kThreads), 0, stream, p)
  if (p.prepacked) {
    if constexpr (X_DT == OUT_DT) {
      if (p.zp_mode == CONCH_ZP_NONE) CONCH_TALL(CONCH_ZP_NONE, true);
      else if (p.zp_mode == CONCH_ZP_SCALAR) CONCH_TALL(CONCH_ZP_SCALAR, true);
      else CONCH_TALL(CONCH_ZP_TENSOR, true);
    } else {
      set_error("mixed_precision_gemm (prepacked): output dtype must equal the activation dtype");
      return CONCH_ERR_UNSUPPORTED;
    }
  } else {
    if (p.zp_mode == CONCH_ZP_NONE) CONCH_TALL(CONCH_ZP_NONE, false);
    else if (p.zp_mode == CONCH_ZP_SCALAR) CONCH_TALL(CONCH_ZP_SCALAR, false);
    else CONCH_TALL(CONCH_ZP_TENSOR, false);
  }
#undef CONCH_TALL
  return check_launch("mixed_gemm_mfma_tall");
}

// Tile shape: 256 rows x 64 NT columns, NT in {4, 3, 2}, or 512 x 128 (kMixedTall) -- the one that needs the least
// (rounds of workgroups) x (time per workgroup).  C4 (1024 x 11008): 256-wide = 172 tiles (67 % of 256 CUs busy), 192-wide = 232
// tiles in ONE round at 3/4 of the work each.  The 512 x 128 tile does the MFMA work of a 256 x 256 tile in 1.03-1.04 of its time
// (profiles/r02/mixed_tall_ab.txt) and exists for the problems the 256-column tile cannot take: per-group zero points (whose
// registers do not fit at 256 columns) at M >= 2048 -- 4096 x 8192 x 4096 with zero points: 352 tiles of 192 columns = two rounds,
// 299 us; 256 tall tiles = one round, 217 us.
int pick_nt(const MixedGemmArgs& p, int num_cus) {
  const int64_t tiles_m = (p.m + kTileM - 1) / kTileM;
  int best = 4;
  double best_cost = 1e30;
  for (int nt = 4; nt >= 2; --nt) {
    if (nt == 4 && p.zp_mode == CONCH_ZP_TENSOR && !p.fuse_silu) continue;  // not built (see launch_zp)
    const int64_t tiles = tiles_m * ((p.n + 64 * nt - 1) / (64 * nt));
    const int64_t rounds = (tiles + num_cus - 1) / num_cus;
    // + fixed per-tile cost (X staging, prologue, epilogue, issue-port share): measured 85 us at 192 columns against
    // 105 us at 256 for K = 4096, i.e. (3 + c) / (4 + c) = 0.81
    const double cost = (double)rounds * (nt + fit::kMixedTileSplit.width_offset);  // (dispatch_fit.hpp)
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = nt;
    }
  }
  if (!p.fuse_silu) {
    const int64_t tiles = ((p.m + kTallRows - 1) / kTallRows) * ((p.n + 127) / 128);
    const double cost = (double)((tiles + num_cus - 1) / num_cus) * fit::kMixedTileSplit.tall_width;
    if (cost < best_cost - 1e-9) best = kMixedTall;
  }
  return best;
}

template <int X_DT, int OUT_DT, int BITS>
int launch_nt(const MixedGemmArgs& p, int nt, hipStream_t stream) {
  if (nt == kMixedTall) return launch_tall<X_DT, OUT_DT, BITS>(p, stream);
  if (nt == 3) return launch_zp<X_DT, OUT_DT, BITS, 3>(p, stream);
  if (nt == 2) return launch_zp<X_DT, OUT_DT, BITS, 2>(p, stream);
  return launch_zp<X_DT, OUT_DT, BITS, 4>(p, stream);
}

template <int X_DT, int OUT_DT>
int launch_bits(const MixedGemmArgs& p, int nt, hipStream_t stream) {
  return p.bits == 4 ? launch_nt<X_DT, OUT_DT, 4>(p, nt, stream) : launch_nt<X_DT, OUT_DT, 8>(p, nt, stream);
}

// ---------------------------------------------------------------------------------------------
// Weight pre-pack (the reference's comparator does the same for its kernel:
// benchmarks/mixed_precision_gemm_benchmark.py:59-75, machete_prepack_B).  The [K/pf][N] tensor is rewritten ONCE, offline,
// into the order the K loop consumes it: for every (tile column of 64 NT output columns, K step of 64) one record of
// 512 threads x (kWpc halves x NT words) -- exactly the registers mixed_gemm_kernel's threads hold for that step -- with the
// nibbles of every word re-ordered so that the in-register dequantisation needs no permute (ChunkDequant, PRE).
// The item <-> (column, 8-k chunk) map below is THE map of the kernel's prologue; pack and unpack share it.
// ---------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void weight_item(int tid, int it, int bn0, int n_max, int& chunk, int& ncol) {
  const int rho = tid & 127, cp = tid >> 7;
  const int r5 = rho & 31, r = r5 & 15, tq = r5 >> 4;
  const int wcol = (rho >> 5) * 16 * NT;
  const int pair = 8 * (r >> 2) + (r & 3) + 4 * tq;
  const int n1 = min(bn0 + wcol + pair, n_max);
  const int n2 = min(bn0 + wcol + 32 + (NT == 4 ? pair : r), n_max);
  const int dchunk = it == 0 ? 0 : it == 1 ? 4 : it == 2 ? (NT == 4 ? 0 : 4 * tq) : 4;
  chunk = cp + dchunk;
  ncol = it < 2 ? n1 : n2;
}

template <int BITS>
__device__ __forceinline__ uint32_t reorder_word(uint32_t w, bool inverse) {
  if constexpr (BITS == 4) {
    // plain: nibble j = k j.  packed: nibble j = k 2j, nibble j + 4 = k 2j + 1
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pj = (j & 1) * 4 + (j >> 1);  // packed position of k j
      out |= inverse ? ((w >> (4 * pj)) & 0xfu) << (4 * j) : ((w >> (4 * j)) & 0xfu) << (4 * pj);
    }
    return out;
  } else {
    // bytes k0 k1 k2 k3 <-> k0 k2 k1 k3 (its own inverse)
    return (w & 0xff0000ffu) | ((w & 0x0000ff00u) << 8) | ((w & 0x00ff0000u) >> 8);
  }
}

template <int BITS, int NT, bool UNPACK>
__global__ __launch_bounds__(kThreads) void prepack_kernel(uint32_t* __restrict__ image, uint32_t* __restrict__ plain, int64_t n, int64_t plain_stride,
                                                           int steps) {
  constexpr int kWpc = BITS == 4 ? 1 : 2;
  constexpr int kWordRowsPerStep = kStepK * BITS / 32;
  const int tn = blockIdx.x / steps, t = blockIdx.x - tn * steps;
  const int tid = threadIdx.x;
  uint32_t* rec = image + ((int64_t)blockIdx.x * kThreads + tid) * (NT * kWpc);
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    int chunk, ncol;
    weight_item<NT>(tid, it, tn * 64 * NT, (int)n - 1, chunk, ncol);
#pragma unroll
    for (int i = 0; i < kWpc; ++i) {
      uint32_t* src = plain + ((int64_t)t * kWordRowsPerStep + chunk * kWpc + i) * plain_stride + ncol;
      if constexpr (UNPACK) *src = reorder_word<BITS>(rec[i * NT + it], true);  // clamped duplicates write equal words
      else rec[i * NT + it] = reorder_word<BITS>(*src, false);
    }
  }
}

template <bool UNPACK>
int launch_prepack(uint32_t* image, uint32_t* plain, int64_t k, int64_t n, int64_t plain_stride, int bits, int nt, hipStream_t stream) {
  const int steps = (int)(k / kStepK);
  const int tiles_n = (int)((n + 64 * nt - 1) / (64 * nt));
  const dim3 grid((unsigned)(tiles_n * steps)), block(kThreads);
#define CONCH_LAUNCH(B, N) hipLaunchKernelGGL((prepack_kernel<B, N, UNPACK>), grid, block, 0, stream, image, plain, n, plain_stride, steps)
  if (bits == 4) {
    if (nt == 4) CONCH_LAUNCH(4, 4);
    else if (nt == 3) CONCH_LAUNCH(4, 3);
    else CONCH_LAUNCH(4, 2);
  } else {
    if (nt == 4) CONCH_LAUNCH(8, 4);
    else if (nt == 3) CONCH_LAUNCH(8, 3);
    else CONCH_LAUNCH(8, 2);
  }
#undef CONCH_LAUNCH
  return check_launch(UNPACK ? "unpack_mixed_weights" : "prepack_mixed_weights");
}

}  // namespace

int64_t prepacked_weight_words(int64_t k, int64_t n, int bits, int nt) {
  const int64_t steps = k / kStepK, tiles_n = (n + 64 * nt - 1) / (64 * nt);
  return tiles_n * steps * kThreads * nt * (bits == 4 ? 1 : 2);
}

int prepack_mixed_weights(uint32_t* image, uint32_t* plain, int64_t k, int64_t n, int64_t plain_stride, int bits, int nt, bool unpack,
                          hipStream_t stream) {
  return unpack ? launch_prepack<true>(image, plain, k, n, plain_stride, bits, nt, stream)
                : launch_prepack<false>(image, plain, k, n, plain_stride, bits, nt, stream);
}

// the tile width the dispatcher would pick for this problem (so that a caller can pre-pack for it)
int mixed_gemm_tile_nt(const MixedGemmArgs& p) {
  const int nt = pick_nt(p, device_cu_count());
  return nt == kMixedTall ? 2 : nt;  // the 512 x 128 tile reads the 128-column image
}

// the fused gate/up epilogue runs on the 256-column tile only, which is not built for per-group zero points
bool mixed_gemm_silu_fused_supported(const MixedGemmArgs& p) { return p.zp_mode != CONCH_ZP_TENSOR && mixed_gemm_mfma_supported(p); }

bool mixed_gemm_mfma_supported(const MixedGemmArgs& p) {
  if (p.bits != 4 && p.bits != 8) return false;
  if (p.x_dtype != CONCH_DT_FP16 && p.x_dtype != CONCH_DT_BF16) return false;
  if (p.out_dtype != CONCH_DT_FP16 && p.out_dtype != CONCH_DT_BF16) return false;
  if (p.m < 1 || p.n < 1 || p.k < kStepK || p.k % kStepK) return false;
  if (p.group_size % kStepK || p.k % p.group_size) return false;
  if ((p.x_stride_m * 2) % 16 || ((uintptr_t)p.x & 15) || ((uintptr_t)p.c & 1)) return false;
  if (p.weight_bias < 0 || p.weight_bias > 512) return false;
  const int64_t lim = (int64_t)1 << 31;
  if (p.m * p.x_stride_m * 2 >= lim) return false;
  if (p.m >= (1 << 24) || p.n >= (1 << 24)) return false;
  // the weight arrays are addressed through 32-bit buffer offsets
  const int64_t lim32 = (int64_t)1 << 32;
  const int64_t w_cols = p.fuse_silu ? 2 * p.n : p.n;
  if (p.prepacked) {
    if (p.prepacked < 2 || p.prepacked > 4 || p.fuse_silu || prepacked_weight_words(p.k, p.n, p.bits, p.prepacked) * 4 >= ((int64_t)1 << 31)) return false;
  } else if (((p.k * p.bits / 32) * p.wq_stride_k + w_cols) * 4 >= lim32) {
    return false;
  }
  if (((p.k / p.group_size) * p.ws_stride_g + w_cols) * 2 >= lim32) return false;
  if (p.zp_mode == CONCH_ZP_TENSOR && ((p.k / p.group_size) * p.wzp_stride_g + w_cols) * 4 >= lim32) return false;
  return true;
}

// Split-K for problems whose tiles leave most of the chip idle (M of a few hundred rows x N of a few thousand columns: 512 x 4096 x
// 4096 is 64 tiles): S slices of the K range per tile, fp32 partial sums to slabs, the fp32-slab reduce.  Worth it while the tile
// kernel's time (~ steps per workgroup) drops by more than the slabs cost (S x M x N x 4 bytes written and read again).
// Round 4: one row of tiles (M <= 256) splits too -- it had been left to the decode-batch kernel, whose time grows with every 64
// rows (256 x 8192 x 8192: 110 us there, 46 here in four slices; 256 x 28672 x 8192: 458 against 122:
// profiles/r04/mixed_splitk_sweep.txt).  `us_out` = the model's time for the pick (the dispatcher compares it with the decode kernel's).
int pick_split(const MixedGemmArgs& p, int nt, int num_cus, double* us_out = nullptr) {
  const int rows = nt == kMixedTall ? kTallRows : kTileM, cols = nt == kMixedTall ? 128 : 64 * nt;
  const int64_t tiles = ((p.m + rows - 1) / rows) * ((p.n + cols - 1) / cols);
  const int steps = (int)(p.k / kStepK), spg = p.group_size / kStepK;
  const double rounds = (double)((tiles + num_cus - 1) / num_cus);
  int best = 1;
  double best_us = 1e30;
  for (int s = 1; s <= 8; s *= 2) {
    if (s > 1 && (p.fuse_silu || p.n % 4 || tiles * s > num_cus || steps / s < 8)) break;  // (a slice of 8 steps: 192 x 4096 x 4096 22.8 us in eight, 25.5 in four)
    const int per = ((steps + s - 1) / s + spg - 1) / spg * spg;  // steps per slice, whole groups
    // ~0.97 us per step on a sparsely filled chip (profiles/r02/dispatch_cold_sweep_after.txt: one row of tiles, 15.5 us per 1024 of
    // K) + the slabs: written once, read once, ~4 TB/s, plus the reduce launch
    // Refitted in round 4 on weights streamed from HBM (profiles/r04/dispatch_cold_sweep.txt): 3 us of prologue + epilogue per
    // workgroup whatever its K range, 5 us for the second launch, the slabs at ~8 bytes per ns (written once, read once, mostly
    // out of L2 / Infinity Cache).  With 3 us and 4 bytes per ns 256 x 4096 x 4096 took four slices (30.6 us) where eight take
    // 25.8, and 96 x 4096 x 4096 left the decode kernel's 17.3 us for 21.1.
    const fit::MixedSplit& fs = fit::kMixedSplit;  // (dispatch_fit.hpp)
    const double us = fs.wg_fixed + fs.step * per * rounds + (s > 1 ? fs.launch2 + (double)s * (double)p.m * (double)p.n * fs.slab_per_elem : 0.0);
    if (us < best_us - 1e-9) {
      best_us = us;
      best = s;
    }
  }
  if (us_out) *us_out = best_us;
  return best;
}

// More than one row of tiles (M > 256): tile shape AND K split together.  pick_nt ignores the split and pick_split takes the
// shape as given, which left 6-15 % at 768-1536 rows (1024 x 28672 x 8192: 445 us on 256 x 128 tiles where 256 x 256 tiles in two K
// slices take 387; 1024 x 13824 x 5120: 167 against 147 on 192-column tiles in two slices).  One model for all (shape, slices)
// candidates, fitted to tools/sweep_mixed_nt_split.py's 120 timings (RMS error 4.5 %, the fastest candidate picked on all ten
// shapes: profiles/r04/mixed_nt_split_sweep.txt): rounds x steps per slice x 0.24 us x (nt + 1.2; 5.4 for the 512 x 128 tile)
// x (1 + 0.2 x the fraction of the chip the launch fills -- a full chip runs each K step ~20 % slower than half a chip: the
// power limit) + 5 us and 1.5 us per MB-slab for a split.  `p` without forced tuning, not fused, not pre-packed.
bool pick_tile_and_split(const MixedGemmArgs& p, int num_cus, int* nt_out, int* split_out, double* us_out) {
  if (p.m <= kTileM || p.fuse_silu || p.prepacked) return false;
  const int steps = (int)(p.k / kStepK), spg = p.group_size / kStepK;
  double best_us = 1e30;
  for (int nt = 2; nt <= kMixedTall; ++nt) {
    if (nt == 4 && p.zp_mode == CONCH_ZP_TENSOR) continue;  // not built (see launch_zp)
    const int rows = nt == kMixedTall ? kTallRows : kTileM, cols = nt == kMixedTall ? 128 : 64 * nt;
    const int64_t tiles = ((p.m + rows - 1) / rows) * ((p.n + cols - 1) / cols);
    const fit::MixedTileSplit& fts = fit::kMixedTileSplit;  // (dispatch_fit.hpp)
    const double w = nt == kMixedTall ? fts.tall_width : nt + fts.width_offset;
    for (int s = 1; s <= 8; s *= 2) {
      if (s > 1 && (p.n % 4 || tiles * s > num_cus || steps / s < 8)) break;
      const int per = ((steps + s - 1) / s + spg - 1) / spg * spg;
      const int64_t wgs = tiles * s, rounds = (wgs + num_cus - 1) / num_cus;
      const double fill = (double)wgs / (double)(rounds * num_cus);
      const double us = (double)rounds * per * fts.step * w * (1.0 + fts.fill_slowdown * fill) +
                        (s > 1 ? fts.launch2 + (double)s * (double)p.m * (double)p.n * fts.slab_per_elem : 0.0);
      if (us < best_us - 1e-9) {
        best_us = us;
        *nt_out = nt;
        *split_out = s;
      }
    }
  }
  if (us_out) *us_out = best_us;
  return best_us < 1e29;
}

// the cost model's time (us) for the LDS-tiled kernel on `p` with the tile shape and the K split it would pick
double mixed_tiles_estimate_us(const MixedGemmArgs& p) {
  const int cus = device_cu_count();
  double us = 0.0;
  int nt = 0, split = 1;
  if (!pick_tile_and_split(p, cus, &nt, &split, &us)) pick_split(p, pick_nt(p, cus), cus, &us);
  return us * (p.bits == 8 ? 1.1 : 1.0);
}

// K slices the LDS-tiled kernel would use on `p` with automatic tuning (the strip kernel, which has no split, defers to a split pick)
int mixed_tiles_auto_split(const MixedGemmArgs& p) {
  const int cus = device_cu_count();
  int nt = 0, split = 1;
  if (pick_tile_and_split(p, cus, &nt, &split, nullptr)) return split;
  return pick_split(p, pick_nt(p, cus), cus);
}

int launch_mixed_gemm_mfma(const MixedGemmArgs& p_in, hipStream_t stream) {
  MixedGemmArgs p = p_in;
  p.rows_epilogue = tuning(CONCH_TUNE_EPILOGUE) != 1;  // auto = row-major (profiles/r03/mixed_epilogue_ab.txt)
  const int num_cus = device_cu_count();
  const int forced = tuning(1);  // CONCH_TUNE_MIXED_TILE_NT: 0 = auto, 2..4 = force 64 NT columns, 5 = force the 512 x 128 tile
  // CONCH_TUNE_MIXED_SPLITK: 0 = auto, 1 = never, 2 / 4 / 8 = force that many K slices (where the shape allows)
  const int forced_split = tuning(CONCH_TUNE_MIXED_SPLITK);
  int joint_nt = 0, joint_split = 1;
  const bool joint = !(forced >= 2 && forced <= kMixedTall) && forced_split == 0 && pick_tile_and_split(p, num_cus, &joint_nt, &joint_split, nullptr);
  int nt = p.fuse_silu ? 4 : (forced >= 2 && forced <= kMixedTall) ? forced : joint ? joint_nt : pick_nt(p, num_cus);
  if (p.zp_mode == CONCH_ZP_TENSOR && nt == 4 && !p.fuse_silu) nt = 3;  // a forced 4
  // a pre-packed image was laid out for one tile width; the 128-column image serves both the 256 x 128 and the 512 x 128 tile
  if (p.prepacked) nt = (p.prepacked == 2 && nt == kMixedTall) ? kMixedTall : p.prepacked;
  int split = forced_split != 0 ? forced_split : joint ? joint_split : pick_split(p, nt, num_cus);
  const int steps = (int)(p.k / kStepK), spg = p.group_size / kStepK;
  if (p.fuse_silu || p.n % 4 || steps / std::max(split, 1) < 2) split = 1;
  if (split > 1) {
    p.split_steps = ((steps + split - 1) / split + spg - 1) / spg * spg;
    const int slices = split_slices(p);
    void* ws = nullptr;
    if (int rc = get_scratch(stream, kScratchMixedSplitK, (size_t)slices * p.m * p.n * 4, &ws)) return rc;
    p.slabs = (float*)ws;
    if (slices == 1) {  // the rounding to whole groups left one slice: the plain form
      p.split_steps = 0;
      p.slabs = nullptr;
    }
  }
  {  // the raster's runtime divisor (GROUP_M x tile columns of the tile shape picked) as a host-made reciprocal
    const int tile_w = p.fuse_silu ? 128 : nt == kMixedTall ? 128 : 64 * nt;
    set_raster_divisor((uint32_t)(kGroupM * ((p.n + tile_w - 1) / tile_w)), &p.raster_magic, &p.raster_shift);
  }
  int rc;
  if (p.x_dtype == CONCH_DT_FP16) {
    rc = p.out_dtype == CONCH_DT_FP16 ? launch_bits<CONCH_DT_FP16, CONCH_DT_FP16>(p, nt, stream)
                                      : launch_bits<CONCH_DT_FP16, CONCH_DT_BF16>(p, nt, stream);
  } else {
    rc = p.out_dtype == CONCH_DT_BF16 ? launch_bits<CONCH_DT_BF16, CONCH_DT_BF16>(p, nt, stream)
                                      : launch_bits<CONCH_DT_BF16, CONCH_DT_FP16>(p, nt, stream);
  }
  if (rc || !p.slabs) return rc;
  return launch_f32_slab_reduce(p.c, p.slabs, split_slices(p), p.m, p.n, p.c_stride_m, p.out_dtype, stream);
}

}  // namespace conch

#ifdef CONCH_CLOCK_PROBE
CONCH_PROBE_READER(conch_debug_probe_mixed, conch::g_probe_mixed)
#endif
