// mixed_precision_gemm: the COLUMN-STRIP kernel (round 4: M > 256; round 5: from 33 rows, on 64- / 128- / 256-row tiles, with K
// split over workgroups where the tiles alone would leave CUs idle).
//
// Replaces, like gemm_mixed.hip, the mixed-precision instantiation of the reference's Triton kernel
// (conch/kernels/quantization/gemm.py:176-216 dequantisation, :219-457 K loop, launcher :482-545).
//
// gemm_mixed.hip passes the dequantised weights through LDS: a thread converts its share of the tile's weights, writes them
// with ds_write_b128 into the image the MFMA loop reads, and ONE workgroup barrier per K step separates the two.  Its loop
// is two streams of ~2000 cycles per step -- the MFMAs and everything else -- that overlap to ~2900.  Here a
// wave owns a COLUMN STRIP of the tile instead of a 128 x 64 block:
//
//   * tile = 256 rows x (64 WA + 64 WB) columns, eight waves: waves 0-3 own 16 WA columns each, waves 4-7 own 16 WB -- the two
//     waves that share a SIMD (w, w + 4) are one wide and one narrow strip, so every SIMD carries the same MFMA work at every
//     tile width: 256 (2, 2), 192 (2, 1), 128 (1, 1) columns;
//   * a lane of the MFMA's weight operand is (column, 8-element k-group) = ONE packed int4 word (two int8 words) of the
//     [K/pf][N] tensor: the wave fetches the words of its own columns, dequantises them in registers (ChunkDequant: the same
//     arithmetic, bit-identical to w_ref) and multiplies -- no ds_write of weights, no weight fragment reads, no duplicated
//     dequantisation (the columns of a strip belong to one wave), and the weights do not wait for the workgroup barrier;
//   * the activations X (256 rows x 64 k per step = 32 KiB) are shared by all eight waves: LDS-DMA into a three-stage ring of
//     the swizzled 8-row x 128-byte image every kernel of this library reads with conflict-free ds_read_b128; each X
//     fragment feeds the strip's one or two MFMA tiles;
//   * EVERYTHING a wave requests from memory is an LDS-DMA piece -- its four X pieces, the 1 KiB of packed words of its
//     32-column group (16-byte chunks XOR-swizzled on the source side: conflict-free ds_read_b32), a 4-byte-per-lane piece
//     with the group's scales (and one with its zero points) -- so the vector-memory queue holds no VGPR destination, hipcc
//     inserts no wait of its own, and one counted `s_waitcnt vmcnt(ops per step)` per step keeps two K steps in flight;
//   * per step and wave: 32 (16) x W MFMAs, 32 X-fragment reads, 2 W chunks dequantised for the NEXT step between the MFMAs
//     (hand-placed slots, as in gemm_mixed.hip), one workgroup barrier.
//
//   * (round 5) the tile height is a template parameter MT = m tiles per wave: 16 (256 rows, the description above), 8 or 4 --
//     the X stage, the wave's X pieces, the slot plan (several dequantisation slices and requests per MFMA slot in the short
//     steps) and the X fragments in flight scale with it; every weight is dequantised once per workgroup whatever MT is.  K
//     slices (blockIdx.y) leave fp32 partial sums in slabs [slice][M][N]; launch_f32_slab_reduce adds them in slice order.
//     strip_plan picks (MT, width, slices) from the cost model of dispatch_fit.hpp (MixedStripSplit).
//
// Contract on top of mixed_gemm_mfma_supported: plain [K/pf][N] weights with 16-byte aligned word rows, output dtype =
// activation dtype, no fused gate/up form, N % 4 == 0 for the K-split forms.
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

constexpr int kSsRows = 256;                       // rows of the full tile: MT = 16 m tiles per wave.  The split-K forms (one row of
                                                   // tiles, M <= 128 / 64) also run MT = 8 / 4: 128- / 64-row tiles, every weight
                                                   // still dequantised once per workgroup
constexpr int kSsStepK = 64;                       // k elements per step = 128 bytes of fp16 / bf16
constexpr int kSsMeta = 256;                       // bytes of one 4-byte-per-lane LDS-DMA piece
constexpr int kStripMaxCounters = 16384;           // arrival counters per (device, stream): the 64 KiB slot the decode kernels use too

#ifdef CONCH_CLOCK_PROBE
__device__ unsigned long long g_probe_mixed_strip[kProbeBlocks * 8];
#endif

template <int BITS, bool ZPT, int MT = 16>
struct StripLds {
  static constexpr int kXStage = MT * 16 * kStepBytes;           // one K step of X: 2 MT subtiles of 8 rows x 128 bytes (32 KiB at MT = 16)
  static constexpr int kXPieces = MT / 4;                        // ... of which every wave requests MT / 4
  static constexpr int kWPieces = BITS == 4 ? 1 : 2;             // 8 (16) word rows x 128 bytes (32 columns) per K step
  static constexpr int kWSlot = kWPieces * 1024;
  // X ring (and every per-wave ring) of S stages: step t read, t + 1 landing, t + 2 .. t + S - 1 requested (the step / tail code
  // below takes any S).  Three at every tile height: the full tile's K step (2 W x 16 MFMAs per wave, >= 1 us) covers the memory
  // latency, and the short tiles' 2x / 4x shorter steps are not waiting for their operands either (profiles/r05/mixed_mid_parts.txt)
  // -- rings of 5 (128 rows) and 7 (64 rows) were built and changed nothing except the LDS they took: with three the 64-row tile fits
  // 54 KiB and 102 registers and the 128-row tile 78 KiB, so two workgroups share a CU where the registers allow: forms with more
  // workgroups than CUs gain 15-20 %, the best form of a shape 1-3 % (profiles/r05/mixed_mid_occupancy_ab.txt)
  static constexpr int kStages = 3;
  static constexpr int kW = kStages * kXStage;                   // + (wave * kStages + slot) * kWSlot
  static constexpr int kS = kW + 8 * kStages * kWSlot;           // + (wave * kStages + slot) * kSsMeta
  static constexpr int kZ = kS + 8 * kStages * kSsMeta;
  static constexpr int kRing = kZ + (ZPT ? 8 * kStages * kSsMeta : 0);
  static_assert(kRing <= 160 * 1024, "LDS budget");
  static constexpr int kOps = kXPieces + kWPieces + 1 + (ZPT ? 1 : 0);  // vector-memory operations per wave and K step
};

// wave-uniform descriptors and strides
struct StripSrc {
  __amdgpu_buffer_rsrc_t x, q, s, z;
  int x_rows8;          // bytes of X between two of a wave's pieces (8 rows)
  int q_step;           // bytes of w_q per K step
  int s_group, z_group; // bytes per group row of w_s / w_zp
  int steps_per_group;
};

// per-lane constants (few: the wide strips run at the 256-register limit)
struct StripLane {
  int vx[2];            // source byte offsets of the wave's X pieces 0 / 1 (pieces 2 / 3: + 16 rows, in the scalar offset; the
                        // swizzle (row >> 1) & 7 of a piece's rows differs between even and odd pieces only)
  int vq;               // ... of its packed-word piece(s), less the wave-uniform column-group part (in StripIssue::q_off)
  int lane4;            // 4 * lane: the 4-byte-per-lane scale / zero-point pieces (column group in the scalar offset)
  int a_lo, a_hi;       // X fragment read offset of this lane inside a stage (chunk g / chunk g + 4), m tile 0
  int w_rd;             // packed-word read offset inside the wave's word slot: n tile 0, k half 0, first word (n tile 1: ^ 64)
  int c2;               // 2 * (this lane's column inside the wave's 32-column group), n tile 0: scale read offset (zero point: 2 x)
};

// wave-uniform cursor of the NEXT K step to request
struct StripIssue {
  int step, slot, q_off, s_off, z_off, left;
};

template <int BITS, bool ZPT, int MT>
__device__ __forceinline__ void issue_op(int op, char* lds, const StripSrc& u, const StripLane& ln, int wave, const StripIssue& is) {
  using L = StripLds<BITS, ZPT, MT>;
  if (op < L::kXPieces) {
#ifdef CONCH_EXP_STRIP_NOXDMA
    if (is.step > 2) return;
#endif
    char* dst = lds + is.slot * L::kXStage + (L::kXPieces * wave + op) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(u.x, (lds_void_t*)dst, 16, ln.vx[op & 1], is.step * kStepBytes + (op >> 1) * 2 * u.x_rows8, 0, 0);
  } else if (op < L::kXPieces + L::kWPieces) {
    const int e = op - L::kXPieces;
    char* dst = lds + L::kW + (wave * L::kStages + is.slot) * L::kWSlot + e * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(u.q, (lds_void_t*)dst, 16, ln.vq, is.q_off + e * (u.q_step / L::kWPieces), 0, 0);
  } else if (op == L::kXPieces + L::kWPieces) {
    char* dst = lds + L::kS + (wave * L::kStages + is.slot) * kSsMeta;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(u.s, (lds_void_t*)dst, 4, ln.lane4, is.s_off, 0, 0);
  } else if (ZPT) {
    char* dst = lds + L::kZ + (wave * L::kStages + is.slot) * kSsMeta;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(u.z, (lds_void_t*)dst, 4, ln.lane4, is.z_off, 0, 0);
  }
}

template <int STAGES>
__device__ __forceinline__ void advance(StripIssue& is, const StripSrc& u) {
  const bool wrap = is.left == 1;
  is.step += 1;
  is.slot = is.slot == STAGES - 1 ? 0 : is.slot + 1;
  is.q_off += u.q_step;
  is.left = wrap ? u.steps_per_group : is.left - 1;
  is.s_off += wrap ? u.s_group : 0;
  is.z_off += wrap ? u.z_group : 0;
}

template <int BITS, bool ZPT, int MT>
__device__ __forceinline__ void issue_all(char* lds, const StripSrc& u, const StripLane& ln, int wave, StripIssue& is) {
#pragma unroll
  for (int op = 0; op < StripLds<BITS, ZPT, MT>::kOps; ++op) issue_op<BITS, ZPT, MT>(op, lds, u, ln, wave, is);
  advance<StripLds<BITS, ZPT, MT>::kStages>(is, u);
}

// the dequantised weights of one K step: [n tile][k half] -> the 8 halfs a lane holds of its column
template <int W>
struct StripB {
  i32x4 v[W][2];
};

// packed words / scale / zero point of one K step as read back from the wave's LDS slots
template <int BITS, int W>
struct StripRaw {
  uint32_t w[W][2][BITS == 4 ? 1 : 2];
  uint32_t scale[W];
  int zp[W];
};

template <int BITS, bool ZPT, int W, int MT>
__device__ __forceinline__ void read_raw(StripRaw<BITS, W>& r, const char* lds, const StripLane& ln, int wave, int slot) {
  using L = StripLds<BITS, ZPT, MT>;
  const char* wq = lds + L::kW + (wave * L::kStages + slot) * L::kWSlot;
  const char* ws = lds + L::kS + (wave * L::kStages + slot) * kSsMeta;
  const char* wz = lds + L::kZ + (wave * L::kStages + slot) * kSsMeta;
#pragma unroll
  for (int nt = 0; nt < W; ++nt) {
    // n tile 1 (wide strips: the group's columns 16-31): chunk index + 4 = ^ 4 (the wave's columns start a group), i.e. ^ 64 bytes
    const int wo = nt ? (ln.w_rd ^ 64) : ln.w_rd;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (BITS == 4) {
        r.w[nt][h][0] = *(const uint32_t*)(wq + wo + h * 512);           // word row 4 h + g
      } else {
        r.w[nt][h][0] = *(const uint32_t*)(wq + wo + h * 1024);          // word rows 8 h + 2 g, + 1
        r.w[nt][h][1] = *(const uint32_t*)(wq + wo + h * 1024 + 128);
      }
    }
    r.scale[nt] = *(const uint16_t*)(ws + ln.c2 + nt * 32);
    if constexpr (ZPT) r.zp[nt] = *(const int*)(wz + 2 * ln.c2 + nt * 64);
    else r.zp[nt] = 0;
  }
}

template <int X_DT>
__device__ __forceinline__ void strip_mma(f32x4& acc, const i32x4& wfrag, const i32x4& xfrag) {
  // operands swapped as everywhere in this library: D rows = n (4 g + e), D columns = m (lane % 16)
#ifdef CONCH_EXP_STRIP_NOMFMA  // timing experiments (wrong results; experiment builds only): see tools/ab_strip_variants.py
  asm volatile("" ::"v"(wfrag), "v"(xfrag));
  return;
#endif
  if constexpr (X_DT == CONCH_DT_FP16)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wfrag), __builtin_bit_cast(f16x8, xfrag), acc, 0, 0, 0);
  else
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, xfrag), acc, 0, 0, 0);
}

template <int X_DT, int BITS>
using StripDequant = ChunkDequant<X_DT, BITS, (X_DT == CONCH_DT_BF16 && BITS == 8)>;  // rounding the difference is a no-op without a zero point

struct StripConst {
  int off_base;
  uint32_t and_mask, or_magic, and_mask_hi;
};

template <int W, int MT = 16>
struct StripSlots {
  static constexpr int kSlots = 2 * MT * W;   // MFMAs of a wave and K step: pair p = (k half p / MT, m tile p % MT) x W n tiles
  static constexpr int kHalf = MT * W;        // first slot of the second k half
  static constexpr int kWait = kHalf - 2;     // the counted wait for this wave's pieces of step t + 1
  static constexpr int kRaw = kHalf - 1;      // ... whose packed words / scales are read back here
  // X fragments in flight.  A fragment feeds W MFMAs (16 W cycles) and returns from LDS after ~130: the full tile rotates eight; with
  // MT / 2 the 64-row tile had two in flight and spent 0.44 us per K step waiting for them (profiles/r05/mixed_mid_sweep.txt)
  static constexpr int kFrags = MT == 16 ? 8 : MT;
  // the workgroup barrier sits in FRONT of this slot's MFMA: behind the last read of this step's X stage (pair 2 MT - kFrags - 1)
  static constexpr int kBar = (2 * MT - kFrags) * W;
  static constexpr int kIssue0 = kBar;        // first of the request slots (step t + 3): the fillers of slot kBar run behind the barrier
  static constexpr int kSl = 9;               // ChunkDequant slices that make a chunk (the tenth is gemm_mixed.hip's ds_write)
  static constexpr int kN = kSl * W;          // slices per k half
  static constexpr int kSpan1 = kHalf - 2;    // k half 1 of THIS step is converted in slots [0, kSpan1)
  static constexpr int kSpan0 = kHalf;        // k half 0 of the NEXT step in slots [kHalf, kSlots)
  // Slice i of a half sits at slot base + i * span / kN: with span >= kN (MT = 16) in distinct slots, the short steps of the
  // 128- / 64-row tiles carry several per slot.  The slices of relative slot r are [first_slice(r), first_slice(r + 1)).
  static constexpr int first_slice(int r, int span) { return r <= 0 ? 0 : r >= span ? kN : (r * kN + span - 1) / span; }
  // requests per slot behind the barrier: one while the slots last (MT = 16), else as many as it takes
  template <int OPS>
  static constexpr int ops_per_slot() { return (OPS + (kSlots - kIssue0) - 1) / (kSlots - kIssue0); }
};

// one slice of the conversion of chunk (n tile NT, k half H)
template <int X_DT, int BITS, int W, int NT, int H, int SUB>
__device__ __forceinline__ void strip_slice(StripB<W>& b, const StripRaw<BITS, W>& raw, StripDequant<X_DT, BITS>& cv, const StripConst& k) {
  cv.slice(SUB, raw.w[NT][H][0], raw.w[NT][H][BITS == 4 ? 0 : 1], k.off_base + raw.zp[NT], raw.scale[NT], nullptr, k.and_mask, k.or_magic, k.and_mask_hi);
  if constexpr (SUB == StripSlots<W>::kSl - 1) b.v[NT][H] = cv.out;
}

// slices I .. END - 1 of k half H, in order
template <int X_DT, int BITS, int W, int H, int I, int END>
__device__ __forceinline__ void strip_slice_run(StripB<W>& b, const StripRaw<BITS, W>& raw, StripDequant<X_DT, BITS>& cv, const StripConst& k) {
  if constexpr (I < END) {
    strip_slice<X_DT, BITS, W, I / StripSlots<W>::kSl, H, I % StripSlots<W>::kSl>(b, raw, cv, k);
    strip_slice_run<X_DT, BITS, W, H, I + 1, END>(b, raw, cv, k);
  }
}

// The work placed behind MFMA number SL of a step.  Slot numbers are TEMPLATE parameters and the step below is a compile-time
// recursion over them: as a `#pragma unroll` loop with the slot in a variable the body was "too large to unroll fully as directed"
// for hipcc, the accumulator array stayed indexed by a run-time value and went to scratch.
// MODE (S = the ring's stages): 0 = steps t + 1 .. t + S exist (t + S is requested here); m = 1 .. S - 1: t + 1 .. t + S - m exist;
// S = last step.
template <int X_DT, int BITS, bool ZPT, int W, int MT, int MODE, int SL>
__device__ __forceinline__ void strip_filler(StripB<W>& b, StripRaw<BITS, W>& raw, StripDequant<X_DT, BITS>& cv, char* lds, const StripSrc& u,
                                             const StripLane& ln, const StripConst& k, int wave, StripIssue& is, int next_slot) {
  using L = StripLds<BITS, ZPT, MT>;
  using S = StripSlots<W, MT>;
#ifndef CONCH_EXP_STRIP_NODQ
  if constexpr (SL < S::kSpan1)  // k half 1 of this step: raw still holds this step's words
    strip_slice_run<X_DT, BITS, W, 1, S::first_slice(SL, S::kSpan1), S::first_slice(SL + 1, S::kSpan1)>(b, raw, cv, k);
#endif
  if constexpr (MODE < L::kStages) {
#ifndef CONCH_EXP_STRIP_NOWAIT
    if constexpr (SL == S::kWait) {  // this wave's pieces of step t + 1 have landed; the requests of the steps behind it may stay in flight
      constexpr int behind = MODE == 0 ? L::kStages - 2 : L::kStages - MODE - 1;  // requested steps younger than t + 1
      wait_vmcnt_n<behind * L::kOps>();
    }
#endif
    if constexpr (SL == S::kRaw) read_raw<BITS, ZPT, W, MT>(raw, lds, ln, wave, next_slot);
#ifndef CONCH_EXP_STRIP_NODQ
    if constexpr (SL >= S::kHalf)  // k half 0 of the next step, into the registers this step's first half has finished with
      strip_slice_run<X_DT, BITS, W, 0, S::first_slice(SL - S::kHalf, S::kSpan0), S::first_slice(SL - S::kHalf + 1, S::kSpan0)>(b, raw, cv, k);
#endif
  }
  if constexpr (MODE == 0 && SL >= S::kIssue0) {
    constexpr int per = S::template ops_per_slot<L::kOps>();
    constexpr int first = (SL - S::kIssue0) * per;
#pragma unroll
    for (int op = first; op < first + per && op < L::kOps; ++op) issue_op<BITS, ZPT, MT>(op, lds, u, ln, wave, is);
    if constexpr (first < L::kOps && first + per >= L::kOps) advance<L::kStages>(is, u);
  }
  __builtin_amdgcn_sched_barrier(0);
}

struct StripPtrs {
  const char* xlo;  // this step's X stage + the lane's fragment offset, chunk g
  const char* xhi;  // ... chunk g + 4
  const char* nlo;  // the next step's stage, chunk g
};

// Slots SL .. 32 W - 1 of one K step of one wave: pair p = SL / W = (k half p / 16, m tile p % 16), n tile SL % W.
template <int X_DT, int BITS, bool ZPT, int W, int MT, int MODE, int SL>
__device__ __forceinline__ void strip_slots(f32x4 (&acc)[MT][W], StripB<W>& b, StripRaw<BITS, W>& raw, StripDequant<X_DT, BITS>& cv, i32x4 (&a)[StripSlots<W, MT>::kFrags],
                                            const StripPtrs& px, char* lds, const StripSrc& u, const StripLane& ln, const StripConst& k, int wave,
                                            StripIssue& is, int next_slot) {
  using S = StripSlots<W, MT>;
  constexpr int kLast = StripLds<BITS, ZPT, MT>::kStages;  // MODE of the last step
  constexpr int P = S::kFrags;  // X fragments in flight
  constexpr int p = SL / W, nt = SL % W;
  if constexpr (MODE < kLast && SL == S::kBar) {
    // Every read of this step's X stage has been ISSUED (the last refill sits behind pair 23); retire them, then meet: behind
    // the barrier every wave's pieces of step t + 1 have landed (each waited for its own at kWait) and this step's stage is
    // free for the requests of step t + 3.
#ifndef CONCH_EXP_STRIP_NOBAR
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  strip_mma<X_DT>(acc[p % MT][nt], b.v[nt][p / MT], a[p % P]);
#ifndef CONCH_EXP_STRIP_NOREAD
  if constexpr (nt == W - 1) {
    constexpr int q = p + P;
    if constexpr (q < 2 * MT) a[p % P] = *(const i32x4*)((q / MT ? px.xhi : px.xlo) + (q % MT) * 2048);
    else if constexpr (MODE < kLast) a[p % P] = *(const i32x4*)(px.nlo + (q - 2 * MT) * 2048);  // the first kFrags pairs of the next step: k half 0
  }
#endif
  strip_filler<X_DT, BITS, ZPT, W, MT, MODE, SL>(b, raw, cv, lds, u, ln, k, wave, is, next_slot);
  if constexpr (SL + 1 < S::kSlots) strip_slots<X_DT, BITS, ZPT, W, MT, MODE, SL + 1>(acc, b, raw, cv, a, px, lds, u, ln, k, wave, is, next_slot);
}

// One K step of one wave.  On entry the first 8 X fragments of the step are in `a` (read behind the previous step's barrier).
// `slot` = ring slot of step t, `next_slot` = of step t + 1.
template <int X_DT, int BITS, bool ZPT, int W, int MT, int MODE>
__device__ __forceinline__ void strip_step(f32x4 (&acc)[MT][W], StripB<W>& b, StripRaw<BITS, W>& raw, i32x4 (&a)[StripSlots<W, MT>::kFrags], char* lds, const StripSrc& u,
                                           const StripLane& ln, const StripConst& k, int wave, StripIssue& is, int slot, int next_slot) {
  using L = StripLds<BITS, ZPT, MT>;
  using S = StripSlots<W, MT>;
  static_assert(S::kWait >= 0 && S::kSpan1 >= 1 && S::kBar >= S::kHalf && S::kBar < S::kSlots, "slot plan");
  static_assert(MT < 16 || (S::kIssue0 + L::kOps <= S::kSlots && S::kSpan1 >= S::kN && S::kSpan0 >= S::kN),
                "the full tile keeps one request and one dequantisation slice per slot at most");
  const StripPtrs px = {lds + slot * L::kXStage + ln.a_lo, lds + slot * L::kXStage + ln.a_hi, lds + next_slot * L::kXStage + ln.a_lo};
  StripDequant<X_DT, BITS> cv;  // ONE conversion in progress: a chunk's slices occupy consecutive slice slots
  strip_slots<X_DT, BITS, ZPT, W, MT, MODE, 0>(acc, b, raw, cv, a, px, lds, u, ln, k, wave, is, next_slot);
}

// the wait in front of the first step: j = min(steps, S) - 1 steps were requested behind step 0
template <int OPS, int J>
__device__ __forceinline__ void strip_wait_first(int steps) {
  if constexpr (J == 0) {
    CONCH_VMCNT(0);
  } else {
    if (steps > J) wait_vmcnt_n<J * OPS>();
    else strip_wait_first<OPS, J - 1>(steps);
  }
}

// the last min(steps, S) steps of a slice: MODE m runs when S - m steps follow it
template <int X_DT, int BITS, bool ZPT, int W, int MT, int MODE>
__device__ __forceinline__ void strip_tail(f32x4 (&acc)[MT][W], StripB<W>& b, StripRaw<BITS, W>& raw, i32x4 (&a)[StripSlots<W, MT>::kFrags], char* lds, const StripSrc& u,
                                           const StripLane& ln, const StripConst& k, int wave, StripIssue& is, int slot, int steps) {
  constexpr int S = StripLds<BITS, ZPT, MT>::kStages;
  if constexpr (MODE == S) {
    strip_step<X_DT, BITS, ZPT, W, MT, S>(acc, b, raw, a, lds, u, ln, k, wave, is, slot, 0);
  } else {
    if (steps > S - MODE) {
      const int nxt = slot == S - 1 ? 0 : slot + 1;
      strip_step<X_DT, BITS, ZPT, W, MT, MODE>(acc, b, raw, a, lds, u, ln, k, wave, is, slot, nxt);
      slot = nxt;
    }
    strip_tail<X_DT, BITS, ZPT, W, MT, MODE + 1>(acc, b, raw, a, lds, u, ln, k, wave, is, slot, steps);
  }
}

template <int X_DT, int BITS, bool ZPT, int W, int MT>
__device__ __forceinline__ void strip_loop(f32x4 (&acc)[MT][W], char* lds, const StripSrc& u, const StripLane& ln, const StripConst& k, int wave,
                                           int step0, int steps, int q0, int s0, int z0) {
  using L = StripLds<BITS, ZPT, MT>;
  StripIssue is = {step0, 0, q0, s0, z0, u.steps_per_group};  // (a K slice starts on a group boundary)
  issue_all<BITS, ZPT, MT>(lds, u, ln, wave, is);
#pragma unroll
  for (int j = 1; j < L::kStages; ++j)
    if (steps > j) issue_all<BITS, ZPT, MT>(lds, u, ln, wave, is);
  strip_wait_first<L::kOps, L::kStages - 1>(steps);  // step 0's pieces have landed; the steps requested behind it stay in flight
  __builtin_amdgcn_sched_barrier(0);
  // k half 0 of step 0, converted at once (k half 1 follows inside the step, like every step's)
  StripB<W> b;
  StripRaw<BITS, W> raw;
  read_raw<BITS, ZPT, W, MT>(raw, lds, ln, wave, 0);
#pragma unroll
  for (int nt = 0; nt < W; ++nt) {
    StripDequant<X_DT, BITS> cv;
#pragma unroll
    for (int s = 0; s < 9; ++s)
      cv.slice(s, raw.w[nt][0][0], raw.w[nt][0][BITS == 4 ? 0 : 1], k.off_base + raw.zp[nt], raw.scale[nt], nullptr, k.and_mask, k.or_magic, k.and_mask_hi);
    b.v[nt][0] = cv.out;
    b.v[nt][1] = i32x4{0, 0, 0, 0};
  }
  __builtin_amdgcn_s_barrier();  // every wave's X pieces of step 0 have landed
  __builtin_amdgcn_sched_barrier(0);
  i32x4 a[StripSlots<W, MT>::kFrags];
#pragma unroll
  for (int j = 0; j < StripSlots<W, MT>::kFrags; ++j) a[j] = *(const i32x4*)(lds + ln.a_lo + j * 2048);
  int slot = 0;
  for (int t = 0; t + L::kStages < steps; ++t) {
    const int nxt = slot == L::kStages - 1 ? 0 : slot + 1;
    strip_step<X_DT, BITS, ZPT, W, MT, 0>(acc, b, raw, a, lds, u, ln, k, wave, is, slot, nxt);
    slot = nxt;
  }
  strip_tail<X_DT, BITS, ZPT, W, MT, 1>(acc, b, raw, a, lds, u, ln, k, wave, is, slot, steps);
}

// Epilogue of one wave: cast to the output dtype (one rounding of the fp32 sum, as the reference's `accumulator.to(out)`), then
// whole tiles leave through the workgroup-built row-major image (epilogue_rows.hpp), others by direct 8-byte stores.
template <int OUT_DT, int W, int TILE_N, int MT>
__device__ __forceinline__ void strip_epilogue(const f32x4 (&acc)[MT][W], const MixedGemmArgs& p, char* lds, int bm0, int bn0, int col0, int lane,
                                               int wave, bool whole) {
  constexpr int kPitch = TILE_N * 2;
  const int g = lane >> 4, r = lane & 15;
  if (whole) {
    // (the caller's closing __syncthreads() made the rings dead)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int nt = 0; nt < W; ++nt) {
        const f32x4& v = acc[i][nt];
        const i32x2 pk = {(int)pack2_bits16<OUT_DT>(f32x2{v[0], v[1]}), (int)pack2_bits16<OUT_DT>(f32x2{v[2], v[3]})};
        const int row = 16 * i + r;
        const int chunk = (col0 >> 3) + 2 * nt + (g >> 1);
        *(i32x2*)(lds + image_chunk_offset<kPitch>(row, chunk) + (g & 1) * 8) = pk;
      }
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = bm0 + 16 * i + r;
      if (m >= (int)p.m) continue;
#pragma unroll
      for (int nt = 0; nt < W; ++nt) {
        const f32x4& v = acc[i][nt];
        const i32x2 pk = {(int)pack2_bits16<OUT_DT>(f32x2{v[0], v[1]}), (int)pack2_bits16<OUT_DT>(f32x2{v[2], v[3]})};
        const int n0 = bn0 + col0 + 16 * nt + 4 * g;
        uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n0;
        if (n0 + 4 <= (int)p.n && (((uintptr_t)dst) & 7) == 0) {
          *(i32x2*)dst = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + e < (int)p.n) dst[e] = (uint16_t)((uint32_t)pk[e >> 1] >> (16 * (e & 1)));
        }
      }
    }
  }
}

// K-split form: the wave's fp32 partial sums go to its slice's slab [M][N] (N % 4 == 0: a lane's four columns are inside N or
// outside together, and 16-byte aligned); launch_f32_slab_reduce adds the slices in slice order and casts once.
template <int W, int MT>
__device__ __forceinline__ void strip_epilogue_slab(const f32x4 (&acc)[MT][W], const MixedGemmArgs& p, float* slab, int bm0, int bn0, int col0, int lane) {
  const int g = lane >> 4, r = lane & 15;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = bm0 + 16 * i + r;
    if (m >= (int)p.m) continue;
#pragma unroll
    for (int nt = 0; nt < W; ++nt) {
      const int n0 = bn0 + col0 + 16 * nt + 4 * g;
      if (n0 < (int)p.n) *(f32x4*)(slab + (int64_t)m * p.n + n0) = acc[i][nt];
    }
  }
}

// K-split form in ONE launch (p.counters): the hand-off of gemm_mixed_skinny.hip's one-launch form (MI355X_MICROARCH.md, table of
// measured hand-offs, first row).  Every wave stores its partial sums write-through (sc1) and drains its own stores; one lane draws
// a ticket from the (tile, wave) agent-scope counter -- per wave, no workgroup barrier: wave w of every slice's workgroup owns the
// same columns --; the wave that draws the last ticket re-reads ALL slices with sc1 loads, adds them in slice order (the order of
// the reduce kernel: the same bits), casts, stores C and puts the counter back to zero.  CH m tiles at a time, 128 registers of
// loads in flight.
template <int OUT_DT, int W, int MT>
__device__ __forceinline__ void strip_epilogue_fused(const f32x4 (&acc)[MT][W], const MixedGemmArgs& p, int tile, int bm0, int bn0, int col0, int lane,
                                                     int wave) {
  const int g = lane >> 4, r = lane & 15;
  const int slices = (int)gridDim.y;
  const uint32_t slab_bytes = (uint32_t)(p.m * p.n * 4);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.slabs, 0, slab_bytes * (uint32_t)slices, 0x00020000);
  // byte offset of this lane's four columns of (m tile i, n tile nt) inside a slab; outside M x N: past the end of the buffer (a store
  // is dropped, a load returns zeros -- the range check is on this offset alone, the slice goes into the scalar offset)
  // (a macro, not a lambda: see gemm_mixed_skinny.hip on lambdas and hipcc's host pass)
#define CONCH_SLAB_OFF(i, nt) \
  ((bm0 + 16 * (i) + r < (int)p.m && bn0 + col0 + 16 * (nt) + 4 * g < (int)p.n) ? ((bm0 + 16 * (i) + r) * (int)p.n + bn0 + col0 + 16 * (nt) + 4 * g) * 4 : (int)0x80000000)
  const int mine = (int)blockIdx.y * (int)slab_bytes;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int nt = 0; nt < W; ++nt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][nt]), rs, CONCH_SLAB_OFF(i, nt), mine, 16);  // sc1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its own write-through stores
  unsigned* cnt = p.counters + 8 * tile + wave;
  unsigned ticket = 0;
  if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
  if (ticket != (unsigned)(slices - 1)) return;  // wave-uniform
  constexpr int CH = 4;                          // m tiles per pass
  constexpr int SF = 32 / (CH * W);              // slices in flight: 128 registers of loads
  static_assert(SF >= 1, "loads in flight");
#pragma unroll
  for (int c = 0; c < MT / CH; ++c) {
    f32x4 sum[CH][W];
    for (int sb = 0; sb < slices; sb += SF) {
      u32x4 part[SF][CH][W];
#pragma unroll
      for (int j = 0; j < SF; ++j) {
        const int soff = min(sb + j, slices - 1) * (int)slab_bytes;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
          for (int nt = 0; nt < W; ++nt) part[j][i][nt] = __builtin_amdgcn_raw_buffer_load_b128(rs, CONCH_SLAB_OFF(c * CH + i, nt), soff, 16);  // sc1
      }
#pragma unroll
      for (int j = 0; j < SF; ++j) {
        const bool first = sb + j == 0, live = sb + j < slices;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
          for (int nt = 0; nt < W; ++nt) {
            const f32x4 v = __builtin_bit_cast(f32x4, part[j][i][nt]);
            sum[i][nt] = first ? v : live ? sum[i][nt] + v : sum[i][nt];
          }
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int m = bm0 + 16 * (c * CH + i) + r;
      if (m >= (int)p.m) continue;
#pragma unroll
      for (int nt = 0; nt < W; ++nt) {
        const f32x4& v = sum[i][nt];
        const i32x2 pk = {(int)pack2_bits16<OUT_DT>(f32x2{v[0], v[1]}), (int)pack2_bits16<OUT_DT>(f32x2{v[2], v[3]})};
        const int n0 = bn0 + col0 + 16 * nt + 4 * g;
        if (n0 >= (int)p.n) continue;  // (N % 4 == 0 in the split forms: the four columns are inside together)
        uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n0;
        if ((((uintptr_t)dst) & 7) == 0) {
          *(i32x2*)dst = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e] = (uint16_t)((uint32_t)pk[e >> 1] >> (16 * (e & 1)));
        }
      }
    }
  }
  if (lane == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
#undef CONCH_SLAB_OFF
}

template <int X_DT, int BITS, bool ZPT, int W, int TILE_N, int MT>
__device__ __forceinline__ void strip_wave(const MixedGemmArgs& p, char* lds, const StripSrc& u, StripLane& ln, const StripConst& k, int wave, int lane,
                                           int bm0, int bn0, int col0, int step0, int steps) {
  const int r = lane & 15, g = lane >> 4;
  const int gb = col0 & ~31, local = col0 & 31;  // the 32-column group whose words / scales this wave fetches, and its place in it
  // packed words: piece row L >> 3 (a word row), 16-byte chunk L & 7 = four columns; chunks XORed with 4 for odd word rows
  // (int8: odd PAIRS of word rows) so that the operand read -- lanes (r, g): word row [2] g, column local + 16 nt + r --
  // spreads its two word rows per half-wave over both halves of the bank row
  {
    const int prow = lane >> 3, pos = lane & 7;
    const int swz = 4 * ((BITS == 4 ? prow : prow >> 1) & 1);
    ln.vq = prow * (u.q_step / (kSsStepK * BITS / 32)) + (pos ^ swz) * 16;
    ln.lane4 = lane * 4;
  }
  {
    const int c = local + r;  // n tile 0
    ln.w_rd = (BITS == 4 ? g : 2 * g) * 128 + ((((c >> 2) ^ (4 * (g & 1)))) * 16) + (c & 3) * 4;
    ln.c2 = c * 2;
  }
  const int group0 = step0 / u.steps_per_group;  // the slice's first group (split_steps is a multiple of steps_per_group)
  const int q0 = (bn0 + gb) * 4 + step0 * u.q_step, s0 = (bn0 + gb) * 2 + group0 * u.s_group, z0 = (bn0 + gb) * 4 + group0 * u.z_group;
  CONCH_PROBE_AT(g_probe_mixed_strip, 0, (int)(blockIdx.x + gridDim.x * blockIdx.y));
  f32x4 acc[MT][W];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int nt = 0; nt < W; ++nt) acc[i][nt] = f32x4{0, 0, 0, 0};
  strip_loop<X_DT, BITS, ZPT, W, MT>(acc, lds, u, ln, k, wave, step0, steps, q0, s0, z0);
  CONCH_PROBE_AT(g_probe_mixed_strip, 1, (int)(blockIdx.x + gridDim.x * blockIdx.y));
  if (p.slabs) {
    // (the 256-row tile keeps the reduce kernel: its loop runs at the 256-register limit and the fused epilogue's loads in flight
    // would be paid for with spills)
#ifdef CONCH_STRIP_ONE_LAUNCH
    if constexpr (MT < 16) {
      if (p.counters) {
        strip_epilogue_fused<X_DT, W, MT>(acc, p, (int)blockIdx.x, bm0, bn0, col0, lane, wave);
        CONCH_PROBE_AT(g_probe_mixed_strip, 3, (int)(blockIdx.x + gridDim.x * blockIdx.y));
        return;
      }
    }
#endif
    strip_epilogue_slab<W, MT>(acc, p, p.slabs + (int64_t)blockIdx.y * p.m * p.n, bm0, bn0, col0, lane);
    CONCH_PROBE_AT(g_probe_mixed_strip, 3, (int)(blockIdx.x + gridDim.x * blockIdx.y));
    return;
  }
  if constexpr (MT == 16) {
    const bool whole = p.rows_epilogue && (p.c_stride_m % 8 == 0) && (((uintptr_t)p.c & 15) == 0) && bm0 + kSsRows <= (int)p.m && bn0 + TILE_N <= (int)p.n &&
                       ((p.m - 1) * p.c_stride_m + p.n) * 2 < ((int64_t)1 << 32);  // workgroup-uniform
    __syncthreads();  // every wave is past its last read of the rings (a fence too: see gemm_mixed.hip, mixed_epilogue_rows)
    strip_epilogue<X_DT, W, TILE_N, MT>(acc, p, lds, bm0, bn0, col0, lane, wave, whole);
    if (whole) {
      __syncthreads();
      image_store_rows<kSsRows, TILE_N * 2>(lds, p.c, p.m, p.n, p.c_stride_m, bm0, bn0, lane, wave);
    }
  } else {
    strip_epilogue<X_DT, W, TILE_N, MT>(acc, p, lds, bm0, bn0, col0, lane, wave, false);
  }
}

template <int X_DT, int BITS, bool ZPT, int WA, int WB, int MT>
__global__ __launch_bounds__(kThreads, MT == 4 ? 4 : 2) void mixed_strip_kernel(MixedGemmArgs p) {
  using L = StripLds<BITS, ZPT, MT>;
  constexpr int kRows = 16 * MT;
  constexpr int kTileN = 64 * WA + 64 * WB;
  constexpr int kImage = MT == 16 ? kSsRows * kTileN * 2 : 0;
  constexpr int kLds = L::kRing > kImage ? L::kRing : kImage;
  static_assert(kLds <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) char lds[kLds];
#if defined(__HIP_DEVICE_COMPILE__)  // the HOST pass of hipcc (ROCm 7.2) fails to substitute issue_op<> inside strip_filler<> ("no
                                     // matching function", no reason given; the device pass takes it): it only needs the stub
  CONCH_PROBE_AT(g_probe_mixed_strip, 2, (int)(blockIdx.x + gridDim.x * blockIdx.y));
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int tiles_m = ((int)p.m + kRows - 1) / kRows;
  const int tiles_n = ((int)p.n + kTileN - 1) / kTileN;
  const TileCoord tc = map_tile(blockIdx.x, tiles_m, tiles_n, p.raster_magic, p.raster_shift);
  const int bm0 = tc.tm * kRows, bn0 = tc.tn * kTileN;
  const int total_steps = (int)(p.k / kSsStepK);
  const int step0 = p.split_steps ? (int)blockIdx.y * p.split_steps : 0;        // K slice (split-K form: blockIdx.y)
  const int steps = p.split_steps ? min(p.split_steps, total_steps - step0) : total_steps;

  StripSrc u;
  const int64_t word_rows = p.k * BITS / 32, groups = p.k / p.group_size;
  u.x = make_uniform_rsrc(p.x, (uint32_t)(((p.m - 1) * p.x_stride_m + p.k) * 2));
  u.q = make_uniform_rsrc(p.w_q, (uint32_t)(((word_rows - 1) * p.wq_stride_k + p.n) * 4));
  u.s = make_uniform_rsrc(p.w_s, (uint32_t)(((groups - 1) * p.ws_stride_g + p.n) * 2));
  u.z = u.s;
  if constexpr (ZPT) u.z = make_uniform_rsrc(p.w_zp, (uint32_t)(((groups - 1) * p.wzp_stride_g + p.n) * 4));
  u.q_step = (kSsStepK * BITS / 32) * (int)p.wq_stride_k * 4;
  u.s_group = (int)p.ws_stride_g * 2;
  u.z_group = (int)p.wzp_stride_g * 4;
  u.steps_per_group = p.group_size / kSsStepK;

  StripLane ln;
  {
    // piece j of the wave = rows 8 (XP wave + j) .. + 7, XP = MT / 4 pieces per wave; rows past M are past the END of the buffer
    // (row M starts at M ldx >= (M - 1) ldx + K): the range check of the buffer load zero-fills them, no clamp
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 8 * L::kXPieces * wave + 8 * j + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      ln.vx[j] = (bm0 + row) * ((int)p.x_stride_m * 2) + chunk * 16;
    }
  }
  u.x_rows8 = 8 * (int)p.x_stride_m * 2;
  {
    const int fr = lane & 15, fg = lane >> 4;
    ln.a_lo = (fr >> 3) * 1024 + (fr & 7) * 128 + ((fg ^ ((fr >> 1) & 7)) * 16);
    ln.a_hi = ln.a_lo ^ 64;
  }
  StripConst k;
  k.off_base = p.weight_bias + (p.zp_mode == CONCH_ZP_SCALAR ? p.w_zp[0] : 0);
  k.and_mask = BITS == 4 ? 0x000f000fu : 0x00ff00ffu;
  k.or_magic = 0x64006400u;
  k.and_mask_hi = k.and_mask << 4;
  asm volatile("" : "+v"(k.and_mask), "+s"(k.or_magic), "+v"(k.and_mask_hi));

  if constexpr (WA == WB) {
    strip_wave<X_DT, BITS, ZPT, WA, kTileN, MT>(p, lds, u, ln, k, wave, lane, bm0, bn0, wave * 16 * WA, step0, steps);
  } else {
    if (wave < 4) strip_wave<X_DT, BITS, ZPT, WA, kTileN, MT>(p, lds, u, ln, k, wave, lane, bm0, bn0, wave * 16 * WA, step0, steps);
    else strip_wave<X_DT, BITS, ZPT, WB, kTileN, MT>(p, lds, u, ln, k, wave, lane, bm0, bn0, 64 * WA + (wave - 4) * 16 * WB, step0, steps);
  }
#endif
}

int strip_slices(const MixedGemmArgs& p) {
  return p.split_steps ? (int)((p.k / kSsStepK + p.split_steps - 1) / p.split_steps) : 1;
}

template <int X_DT, int BITS, bool ZPT, int MT>
int launch_width(const MixedGemmArgs& p, int nt, hipStream_t stream) {
  const int tile_n = 64 * nt, rows = 16 * MT;
  const int tiles_m = (int)((p.m + rows - 1) / rows), tiles_n = (int)((p.n + tile_n - 1) / tile_n);
  const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)strip_slices(p)), block(kThreads);
  if (nt == 4) hipLaunchKernelGGL((mixed_strip_kernel<X_DT, BITS, ZPT, 2, 2, MT>), grid, block, 0, stream, p);
  else if (nt == 3) hipLaunchKernelGGL((mixed_strip_kernel<X_DT, BITS, ZPT, 2, 1, MT>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((mixed_strip_kernel<X_DT, BITS, ZPT, 1, 1, MT>), grid, block, 0, stream, p);
  return check_launch("mixed_gemm_strip");
}

template <int X_DT, int BITS, bool ZPT>
int launch_rows(const MixedGemmArgs& p, int nt, int mt, hipStream_t stream) {
  if (mt == 4) return launch_width<X_DT, BITS, ZPT, 4>(p, nt, stream);
  if (mt == 8) return launch_width<X_DT, BITS, ZPT, 8>(p, nt, stream);
  return launch_width<X_DT, BITS, ZPT, 16>(p, nt, stream);
}

template <int X_DT, int BITS>
int launch_zp(const MixedGemmArgs& p, int nt, int mt, hipStream_t stream) {
  return p.zp_mode == CONCH_ZP_TENSOR ? launch_rows<X_DT, BITS, true>(p, nt, mt, stream) : launch_rows<X_DT, BITS, false>(p, nt, mt, stream);
}

}  // namespace

bool mixed_gemm_strip_supported(const MixedGemmArgs& p) {
  if (!mixed_gemm_mfma_supported(p)) return false;
  if (p.fuse_silu || p.prepacked || p.split_steps || p.slabs) return false;  // (the K split is this launcher's own choice)
  if (p.out_dtype != p.x_dtype) return false;
  // bf16 x 8-bit weights: the widening dequantisation (fp32 pairs, the difference rounded to bf16 first) on top of 128 accumulators
  // and 16 + 16 operand registers spills in the wide strips (288 bytes of scratch per lane inside the K loop); the LDS-tiled
  // kernel keeps that combination
  if (p.x_dtype == CONCH_DT_BF16 && p.bits == 8) return false;
  // LDS-DMA of 16 bytes per lane: aligned word rows; of 4 bytes per lane: aligned scale / zero-point rows
  if ((((uintptr_t)p.w_q) & 15) || p.wq_stride_k % 4) return false;
  if ((((uintptr_t)p.w_s) & 3) || p.ws_stride_g % 2) return false;
  if (p.zp_mode == CONCH_ZP_TENSOR && ((((uintptr_t)p.w_zp) & 3))) return false;
  return true;
}

int pick_strip_nt(const MixedGemmArgs& p, int num_cus);

struct StripPlan {
  int mt = 16, nt = 4, split = 1;  // m tiles per wave (tile rows / 16), tile width / 64, K slices
  double us = 1e30;                // the cost model's time for it (one row of tiles only)
};

// Auto rule, from the interleaved sweep of 48 shapes x the two kernels (profiles/r04/mixed_strip_sweep.txt).  On the benchmark's
// data both kernels run against the chip's POWER limit (in-loop clock 1.75-1.85 GHz; on all-zero activations, where the clock stays
// at 2.4 GHz, the strip kernel is 4-8 % ahead: profiles/r04/mixed_strip_zero_data.txt), so the strip kernel's better schedule buys
// 1-9 % where its tiles fill the chip and N is wide (N >= 8192: 384..768 x 4096 x 11008 -4..-9 %, x 28672 -1..-7 %, M >= 1536 -1..-3 %)
// and nothing at N = 4096; with few tiles the LDS-tiled kernel's split-K form wins by 5-120 % (the strip kernel has none).
bool mixed_strip_beats_tiles(const MixedGemmArgs& p) {
  if (p.n < fit::kMixedStrip.min_n) return false;  // (dispatch_fit.hpp)
  const int cus = device_cu_count();
  const int nt = pick_strip_nt(p, cus);
  const int64_t tiles = ((p.m + kSsRows - 1) / kSsRows) * ((p.n + 64 * nt - 1) / (64 * nt));
  if ((double)tiles < fit::kMixedStrip.min_fill * (double)cus) return false;
  // ... and the LDS-tiled kernel would not split K on this problem (1024 x 28672 x 8192: 256 x 256 tiles in two K slices take 387 us
  // where either kernel's unsplit 256 x 128 tiles take 420-445)
  return mixed_tiles_auto_split(p) == 1;
}

// tile width (64 nt columns) that needs the least (rounds of workgroups) x (work per workgroup): every SIMD carries nt / 4 of
// the 256-column tile's MFMAs at every width
int pick_strip_nt(const MixedGemmArgs& p, int num_cus) {
  const int64_t tiles_m = (p.m + kSsRows - 1) / kSsRows;
  int best = 4;
  double best_cost = 1e30;
  for (int nt = 4; nt >= 2; --nt) {
    const int64_t tiles = tiles_m * ((p.n + 64 * nt - 1) / (64 * nt));
    const double cost = (double)((tiles + num_cus - 1) / num_cus) * (nt + fit::kMixedStrip.width_offset);
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = nt;
    }
  }
  return best;
}

// Tile rows, tile width and K slices of a launch.  Tall problems (M > 256) keep the unsplit 256-row tile at pick_strip_nt's width;
// one row of tiles (M <= 256: the batched-decode sizes) takes the smallest tile that holds M -- 64 / 128 / 256 rows: every weight is
// still dequantised once, the MFMAs of absent rows are not issued -- and as many K slices as fill the chip once, by the cost model
// of dispatch_fit.hpp (MixedStripSplit).  CONCH_TUNE_MIXED_STRIP_ROWS / _TILE_NT / _SPLITK force the three choices.
// the cost model's time (us) for tiles of 16 mt rows x 64 nt columns in s K slices; < 0 = not a candidate
double strip_model_us(const MixedGemmArgs& p, int cus, int mt, int nt, int s) {
  const fit::MixedStripSplit& f = fit::kMixedStripSplit;
  const int steps = (int)(p.k / kSsStepK), spg = p.group_size / kSsStepK;
  const int per = ((steps + s - 1) / s + spg - 1) / spg * spg;  // steps per slice, whole groups
  if ((steps + per - 1) / per != s) return -1.0;                 // (the rounding made it another candidate)
  if (s > 1 && per < 4) return -1.0;
  const int64_t tiles = ((p.m + 16 * mt - 1) / (16 * mt)) * ((p.n + 64 * nt - 1) / (64 * nt));
  const double rounds = (double)((tiles * s + cus - 1) / cus), fill = (double)(tiles * s) / (rounds * cus);
  return f.wg_fixed + rounds * per * (1.0 + f.fill_slowdown * fill) * (f.step + f.step_mt * mt + f.step_nt * nt + f.step_mt_nt * mt * nt) * (p.bits == 8 ? 1.1 : 1.0) +
         (s > 1 ? f.launch2 + (double)s * (double)p.m * (double)p.n * f.slab_per_elem : 0.0);
}

StripPlan strip_plan(const MixedGemmArgs& p) {
  const int cus = device_cu_count();
  const int f_rows = tuning(CONCH_TUNE_MIXED_STRIP_ROWS), f_nt = tuning(CONCH_TUNE_MIXED_TILE_NT), f_split = tuning(CONCH_TUNE_MIXED_SPLITK);
  const fit::MixedStripSplit& f = fit::kMixedStripSplit;
  const bool forced_rows = f_rows == 64 || f_rows == 128 || f_rows == 256;
  const bool forced = forced_rows || (f_nt >= 2 && f_nt <= 4) || f_split >= 2;
  const bool may_split = p.n % 4 == 0 && f_split != 1;
  // tile heights in the race: the forced one; else every height up to the smallest that holds M, all three from 129 rows to
  // max_m_search -- 128- / 64-row tiles pad M less and make more workgroups (384 x 4096 x 11008: 53.1 us on two rows of 256-row tiles
  // = 116 workgroups, 42.9 on three rows of 128-row tiles; 1024 x 4096 x 4096: 48.1 -> 39.6; profiles/r05/mixed_rows_tall_sweep.txt);
  // taller problems: the 256-row tile
  int mts[3], n_mt = 0;
  if (forced_rows) {
    mts[n_mt++] = f_rows / 16;
  } else {
    const int tallest = p.m <= 64 ? 4 : p.m <= 128 ? 8 : 16;
    for (int mt = tallest; mt >= (p.m <= f.max_m_search ? 4 : tallest); mt /= 2) mts[n_mt++] = mt;
  }
  StripPlan tall;  // the unsplit 256-row tile at pick_strip_nt's width: what M > 256 ran before round 5
  tall.nt = (f_nt >= 2 && f_nt <= 4) ? f_nt : pick_strip_nt(p, cus);
  tall.us = strip_model_us(p, cus, 16, tall.nt, 1);
  if (p.m > f.max_m_search && !forced) return tall;
  StripPlan best;
  for (int im = 0; im < n_mt; ++im)
    for (int nt = 4; nt >= 2; --nt) {
      if (f_nt >= 2 && f_nt <= 4 && nt != f_nt) continue;
      for (int s = 1; s <= 8; ++s) {
        if (s > 1 && !may_split) break;
        if (f_split >= 2 && s != std::min(f_split, 8)) continue;
        const double us = strip_model_us(p, cus, mts[im], nt, s);
        if (us >= 0.0 && us < best.us - 1e-9) {
          best.us = us;
          best.mt = mts[im];
          best.nt = nt;
          best.split = s;
        }
      }
    }
  if (best.us > 1e29) {  // nothing admissible under the forced keys: the plain form at the (forced) tile height
    best = tall;
    best.mt = mts[0];
    return best;
  }
  // several rows of tiles, nothing forced: leave the 256-row tile only for a modelled gain beyond the model's own error
  if (p.m > kSsRows && !forced && (best.mt < 16 || best.split > 1) && !(best.us * f.tall_margin < tall.us)) return tall;
  if (p.m > kSsRows && !forced && best.mt == 16 && best.split == 1) return tall;  // (the width rule of round 4 stays)
  return best;
}

// the cost model's time (us) for the strip kernel's pick on one row of tiles (the dispatcher compares it with the other kernels')
double mixed_strip_estimate_us(const MixedGemmArgs& p) { return strip_plan(p).us; }

void mixed_strip_plan_query(const MixedGemmArgs& p, int* rows, int* cols, int* slices) {
  const StripPlan plan = strip_plan(p);
  *rows = 16 * plan.mt;
  *cols = 64 * plan.nt;
  *slices = plan.split;
}

double mixed_strip_short_or_split_us(const MixedGemmArgs& p) {
  const StripPlan plan = strip_plan(p);
  return (plan.mt < 16 || plan.split > 1) && plan.us > 0.0 ? plan.us : 1e30;
}

int launch_mixed_gemm_strip(const MixedGemmArgs& p_in, hipStream_t stream) {
  MixedGemmArgs p = p_in;
  p.rows_epilogue = tuning(CONCH_TUNE_EPILOGUE) != 1;
  const StripPlan plan = strip_plan(p);
  const int nt = plan.nt;
  if (plan.split > 1) {
    const int steps = (int)(p.k / kSsStepK), spg = p.group_size / kSsStepK;
    p.split_steps = ((steps + plan.split - 1) / plan.split + spg - 1) / spg * spg;
    void* ws = nullptr;
    const size_t bytes = (size_t)strip_slices(p) * p.m * p.n * 4;
    if (int rc = get_scratch(stream, kScratchMixedSplitK, bytes, &ws)) return rc;
    p.slabs = (float*)ws;
    // ONE launch (the wave that arrives last at a tile strip adds the slices) was built, is bit-identical and SLOWER at these tile
    // sizes -- 64 x 4096 x 11008 22.6 us against 17.5, 128 x 4096 x 11008 34.8 against 23.3 (profiles/r05/mixed_mid_one_launch.txt):
    // draining 48-96 KiB of write-through stores per workgroup and re-reading them past the L2 costs more than the second launch
    // (the decode kernels' one-launch forms win at <= 8 KiB), and its 128 registers of loads in flight cost the 64-row tile its
    // second workgroup per CU.  Compiled only with -DCONCH_STRIP_ONE_LAUNCH (then CONCH_TUNE_SKINNY_MODE = 2 selects it).
#ifdef CONCH_STRIP_ONE_LAUNCH
    const int64_t tiles = ((p.m + 16 * plan.mt - 1) / (16 * plan.mt)) * ((p.n + 64 * nt - 1) / (64 * nt));
    if (tuning(CONCH_TUNE_SKINNY_MODE) == 2 && plan.mt < 16 && strip_slices(p) > 1 && bytes < ((size_t)1 << 31) && tiles * 8 <= kStripMaxCounters) {
      void* cbuf = nullptr;
      if (int rc = get_scratch(stream, kScratchCounters, (size_t)kStripMaxCounters * 4, &cbuf, /*zero_on_alloc=*/true)) return rc;
      p.counters = (unsigned*)cbuf;
    }
#endif
  }
  set_raster_divisor((uint32_t)(kGroupM * ((p.n + 64 * nt - 1) / (64 * nt))), &p.raster_magic, &p.raster_shift);
  int rc;
  if (p.x_dtype == CONCH_DT_FP16) rc = p.bits == 4 ? launch_zp<CONCH_DT_FP16, 4>(p, nt, plan.mt, stream) : launch_zp<CONCH_DT_FP16, 8>(p, nt, plan.mt, stream);
  else rc = launch_zp<CONCH_DT_BF16, 4>(p, nt, plan.mt, stream);  // (bf16 x 8-bit: not built, see mixed_gemm_strip_supported)
  if (rc || !p.slabs || p.counters) return rc;
  return launch_f32_slab_reduce(p.c, p.slabs, strip_slices(p), p.m, p.n, p.c_stride_m, p.out_dtype, stream);
}

}  // namespace conch

#ifdef CONCH_CLOCK_PROBE
CONCH_PROBE_READER(conch_debug_probe_mixed_strip, conch::g_probe_mixed_strip)
#endif
