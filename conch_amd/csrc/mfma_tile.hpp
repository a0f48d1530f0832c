// Shared building blocks of the 256x256 LDS-tiled MFMA GEMM kernels (gemm_mfma.hip, gemm_mixed.hip):
// tile geometry, XCD-aware rasterisation, LDS-DMA staging of 128-row x 128-byte operand units with
// the source-side XOR swizzle, and the conflict-free ds_read_b128 fragment reads.
#pragma once

#include "common.hpp"

namespace conch {
namespace tile {

constexpr int kThreads = 512;
constexpr int kTileM = 256;
constexpr int kTileN = 256;
constexpr int kStepBytes = 128;                     // K bytes per LDS row / per K step
constexpr int kUnitBytes = 128 * kStepBytes;        // 16 KiB: 128 rows
constexpr int kBufBytes = 4 * kUnitBytes;           // U1, V1, V2, U2
constexpr int kLdsBytes = 2 * kBufBytes;            // double buffered: 128 KiB
#ifdef CONCH_EXP_GROUP_M  // experiment builds: another raster group height
constexpr int kGroupM = CONCH_EXP_GROUP_M;
#else
constexpr int kGroupM = 4;
#endif

// Stream order of the units of one K step.  U = rows of A (m), V = rows of B^T (n).
// U1/U2 hold the first/second 64 rows of BOTH wave-rows' 128-row m ranges, V1/V2 the
// first/second 32 rows of all four wave-columns' 64-row n ranges, so that a unit is needed by
// every wave in the same phase.
enum { kU1 = 0, kV1 = 1, kV2 = 2, kU2 = 3 };

typedef __attribute__((address_space(3))) void lds_void_t;

struct Frag {
  i32x4 lo;  // 16-byte chunk g      of the 128-byte K row
  i32x4 hi;  // 16-byte chunk g + 4
};

struct TileCoord {
  int tm, tn;
};

// XCD-aware + GROUP_M rasterisation.  Workgroups are dealt round-robin over the 8 XCDs, so ids
// b and b+8 share an L2: give each XCD a contiguous run of the GROUP_M-ordered tile list
// (bijective for any grid size).  Pure speed; correctness does not depend on placement.
// `magic` / `shift`: host-made reciprocal of per_group = GROUP_M x tiles_n (gemm.hpp, set_raster_divisor; shift < 0 = not
// given).  The group size is 1..4 (GROUP_M = 4), so the other two divisions are by a small constant each.
__device__ __forceinline__ TileCoord raster_tile(int lin, int tiles_m, int tiles_n, uint32_t magic = 0, int shift = -1) {
  const int per_group = kGroupM * tiles_n;
  const int group = shift < 0 ? lin / per_group : magic ? (int)(__umulhi((uint32_t)lin, magic) >> shift) : lin >> shift;
  const int first_m = group * kGroupM;
  const int gsz = min(tiles_m - first_m, kGroupM);
  const int in_group = lin - group * per_group;
  int q;
  if (gsz == 4) q = in_group >> 2;
  else if (gsz == 3) q = (int)(__umulhi((uint32_t)in_group, 0xAAAAAAABu) >> 1);
  else if (gsz == 2) q = in_group >> 1;
  else if (gsz == 1) q = in_group;
  else q = in_group / gsz;  // only with an experimental GROUP_M above 4
  TileCoord t;
  t.tm = first_m + (in_group - q * gsz);
  t.tn = q;
  return t;
}

__device__ __forceinline__ TileCoord map_tile(int bid, int tiles_m, int tiles_n, uint32_t magic = 0, int shift = -1) {
  const int nwg = tiles_m * tiles_n;
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  return raster_tile(lin, tiles_m, tiles_n, magic, shift);
}

// Two-width tile schedule of the 256-row scaled GEMM (gemm_mfma.hip): `big_cols` tile columns of 256 output columns,
// then `narrow_cols` tile columns of 192 (the last one possibly masked).  With uniform 256-wide tiles a problem whose
// tile count is not a multiple of the CU count idles most of the chip in its last round (C3: 688 tiles on 256 CUs = 2.69
// rounds run as 3); a narrow tile costs 3/4 of a wide one, so [512 wide + 240 narrow] runs as 2 + 0.75 rounds.  Every
// XCD walks ITS share of the wide tiles first and of the narrow tiles last (longest first: the greedy dispatcher then ends
// all CUs together).  narrow_cols == 0 is the plain uniform schedule.
struct TileSchedule {
  int tiles_m;
  int big_cols;
  int narrow_cols;
  int alternate;  // 1 = XCDs with an odd label walk their narrow tiles FIRST: the two halves of the chip then reach their
                  // tile boundaries (C-store bursts) a quarter of a tile apart instead of all at once;
                  // 2 = the same offset between the two halves of EVERY XCD's CUs (see place_tile)
};

struct TilePlace {
  int tm;
  int n0;      // first output column
  int narrow;  // 1 = 192-column tile
};

__device__ __forceinline__ TilePlace place_tile(int bid, const TileSchedule& ts, uint32_t magic = 0, int shift = -1) {
  TilePlace t;
  if (ts.narrow_cols == 0) {
    const TileCoord c = map_tile(bid, ts.tiles_m, ts.big_cols, magic, shift);
    t.tm = c.tm;
    t.n0 = c.tn * kTileN;
    t.narrow = 0;
    return t;
  }
  const int t4 = ts.tiles_m * ts.big_cols, t3 = ts.tiles_m * ts.narrow_cols;
  const int nwg = t4 + t3;
  const int xcd = bid & 7, idx = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int w0 = xcd * q + min(xcd, r);            // tiles of any width given to XCDs below this one
  const int b0 = (xcd * t4) >> 3;                  // wide tiles given to XCDs below this one
  const int b1 = ((xcd + 1) * t4) >> 3;
  const int nbig = b1 - b0;
  const int nnarrow = (q + (xcd < r ? 1 : 0)) - nbig;
  bool is_big;
  int big_idx, narrow_idx;
  if (ts.alternate == 2) {
    // within every XCD the first round deals wide and narrow tiles to alternate workgroup slots (16 + 16 on 32 CUs): the two
    // halves of the XCD's CUs then reach every later tile boundary -- the C-store burst into the XCD's L2 -- about a sixth
    // of a tile apart; then the remaining wide tiles, then the remaining narrow ones
    const int pairs = min(16, min(nbig, nnarrow));
    if (idx < 2 * pairs) {
      is_big = !(idx & 1);
      big_idx = narrow_idx = idx >> 1;
    } else {
      const int rest = idx - 2 * pairs;
      is_big = rest < nbig - pairs;
      big_idx = pairs + rest;
      narrow_idx = pairs + rest - (nbig - pairs);
    }
  } else {
    const bool narrow_first = ts.alternate && (xcd & 1);
    is_big = narrow_first ? idx >= nnarrow : idx < nbig;
    big_idx = narrow_first ? idx - nnarrow : idx;
    narrow_idx = narrow_first ? idx : idx - nbig;
  }
  if (is_big) {
    const TileCoord c = raster_tile(b0 + big_idx, ts.tiles_m, ts.big_cols);
    t.tm = c.tm;
    t.n0 = c.tn * kTileN;
    t.narrow = 0;
  } else {
    const TileCoord c = raster_tile((w0 - b0) + narrow_idx, ts.tiles_m, ts.narrow_cols);
    t.tm = c.tm;
    t.n0 = ts.big_cols * kTileN + c.tn * (kTileN * 3 / 4);
    t.narrow = 1;
  }
  return t;
}

// Per-lane state of the LDS-DMA staging: byte offset of this lane's 16-byte source chunk for each
// (unit kind, j) wave-instruction, relative to the A / B^T base.
struct StageOffsets {
  int off[4][2];
};

// `wave_col_cols` = columns a wave-column owns in V1 (and again in V2), `v2_delta` = column distance from a V1 row to
// the V2 row of the same unit row.  Plain GEMM: 64-column wave-columns whose second 32 columns sit in V2 (32, 32 -- the
// defaults).  Fused gate/up GEMM (scaled_gemm_silu_and_mul): a wave-column owns 32 OUTPUT columns, V1 holds their gate
// columns and V2 the matching up columns n_out further right, so that a lane's accumulators hold gate and up of the same
// eight outputs.  V1 rows are clamped to n_max, V2 rows follow their V1 row.
__device__ __forceinline__ StageOffsets make_stage_offsets(int wave, int lane, int bm0, int bn0,
                                                           int m_max, int n_max, int lda, int ldb,
                                                           int wave_col_cols = 64, int v2_delta = 32) {
  StageOffsets s;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rho = 16 * wave + 8 * j + (lane >> 3);    // row of the unit this lane feeds
    const int chunk = (lane & 7) ^ ((rho >> 1) & 7);    // source chunk that lands at lane&7
    // U kinds: unit row -> block m
    const int mrow = (rho >> 6) * 128 + (rho & 63);
    // V kinds: unit row -> block n, permuted so that MFMA D rows 4g+e of tile t are n = 8g+e+4t
    const int r5 = rho & 31, r = r5 & 15, t = r5 >> 4;
    const int nrow = (rho >> 5) * wave_col_cols + 8 * (r >> 2) + (r & 3) + 4 * t;
    s.off[kU1][j] = min(bm0 + mrow, m_max) * lda + chunk * 16;
    s.off[kU2][j] = min(bm0 + mrow + 64, m_max) * lda + chunk * 16;
    if (wave_col_cols == 48) {
      // 192-column tile: a wave-column owns 48 columns -- 32 in V1 (both 16-row MFMA tiles, permuted as above) and 16 in V2
      // (ONE MFMA tile, rows in plain order: D rows 4g+e = n 4g+e, four consecutive columns per lane); the second 16 rows
      // of a wave-column's V2 share are never multiplied and re-stage the first sixteen
      s.off[kV1][j] = min(bn0 + nrow, n_max) * ldb + chunk * 16;
      s.off[kV2][j] = min(bn0 + (rho >> 5) * 48 + 32 + r, n_max) * ldb + chunk * 16;
    } else if (v2_delta == 32) {
      s.off[kV1][j] = min(bn0 + nrow, n_max) * ldb + chunk * 16;
      s.off[kV2][j] = min(bn0 + nrow + 32, n_max) * ldb + chunk * 16;
    } else {
      const int n1 = min(bn0 + nrow, n_max);
      s.off[kV1][j] = n1 * ldb + chunk * 16;
      s.off[kV2][j] = (n1 + v2_delta) * ldb + chunk * 16;
    }
  }
  return s;
}

struct Srcs {
  __amdgpu_buffer_rsrc_t a;
  __amdgpu_buffer_rsrc_t b;
};

// Issue the two LDS-DMA wave-instructions this wave contributes to unit `KIND` of K step `tile`.
#ifdef CONCH_EXP_DMA_AUX_A  // experiment builds: cache-policy bits of the operand LDS-DMA loads (1 = sc0, 2 = nt, 16 = sc1)
constexpr int kDmaAuxA = CONCH_EXP_DMA_AUX_A;
#else
constexpr int kDmaAuxA = 0;
#endif
#ifdef CONCH_EXP_DMA_AUX_B
constexpr int kDmaAuxB = CONCH_EXP_DMA_AUX_B;
#else
constexpr int kDmaAuxB = 0;
#endif

template <int KIND>
__device__ __forceinline__ void stage_unit(char* lds, const Srcs& src, const StageOffsets& so, int wave,
                                           int tile) {
  const int buf = tile & 1;
  const int koff = tile * kStepBytes;
  char* dst = lds + buf * kBufBytes + KIND * kUnitBytes + wave * 2048;
  if constexpr (KIND == kU1 || KIND == kU2) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src.a, (lds_void_t*)dst, 16, so.off[KIND][0], koff, 0, kDmaAuxA);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src.a, (lds_void_t*)(dst + 1024), 16, so.off[KIND][1], koff, 0, kDmaAuxA);
  } else {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src.b, (lds_void_t*)dst, 16, so.off[KIND][0], koff, 0, kDmaAuxB);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src.b, (lds_void_t*)(dst + 1024), 16, so.off[KIND][1], koff, 0, kDmaAuxB);
  }
}

// One of the two wave-instructions of stage_unit (J = 0 / 1), for loops that spread the issue over several slots.
template <int KIND, int J>
__device__ __forceinline__ void stage_piece(char* lds, const Srcs& src, const StageOffsets& so, int wave, int tile) {
  char* dst = lds + (tile & 1) * kBufBytes + KIND * kUnitBytes + wave * 2048 + J * 1024;
  if constexpr (KIND == kU1 || KIND == kU2)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src.a, (lds_void_t*)dst, 16, so.off[KIND][J], tile * kStepBytes, 0, 0);
  else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src.b, (lds_void_t*)dst, 16, so.off[KIND][J], tile * kStepBytes, 0, 0);
}

// `off` is a byte offset from the (1024-aligned) LDS base; every term of it except the chunk is a
// multiple of 128, so chunk g+4 = (g ^ swizzle) ^ 4 is simply off ^ 64.
__device__ __forceinline__ Frag read_frag(const char* lds, int off) {
  Frag f;
  f.lo = *(const i32x4*)(lds + off);
  f.hi = *(const i32x4*)(lds + (off ^ 64));
  return f;
}

#define CONCH_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) CONCH_VMCNT(0);
  else if constexpr (N == 2) CONCH_VMCNT(2);
  else if constexpr (N == 4) CONCH_VMCNT(4);
  else if constexpr (N == 6) CONCH_VMCNT(6);
  else if constexpr (N == 8) CONCH_VMCNT(8);
  else if constexpr (N == 16) CONCH_VMCNT(16);
  else if constexpr (N == 24) CONCH_VMCNT(24);
  // N < 0: no wait
}


// counted wait with an arbitrary compile-time count
template <int N>
__device__ __forceinline__ void wait_vmcnt_n() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the same, N < 0 = no wait
template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N >= 0) wait_vmcnt_n<N>();
}

enum { kMmaFp8 = 0, kMmaInt8 = 1, kMmaBf16 = 2, kMmaF16 = 3 };

template <int MMA> struct AccT { typedef f32x4 type; };
template <> struct AccT<kMmaInt8> { typedef i32x4 type; };

// D (16x16) += rows(fa) x cols(fb) over the 128 K-bytes both fragments hold.  The MFMA pairs
// element j of k-group g of A with element j of k-group g of B, so any K permutation that is the
// same on both sides is legal: k-group g holds bytes [16g,16g+16) and [64+16g, 64+16g+16).
template <int MMA>
__device__ __forceinline__ void mma_step(typename AccT<MMA>::type& acc, const Frag& fa, const Frag& fb) {
  if constexpr (MMA == kMmaFp8) {
    const i32x8 a = {fa.lo[0], fa.lo[1], fa.lo[2], fa.lo[3], fa.hi[0], fa.hi[1], fa.hi[2], fa.hi[3]};
    const i32x8 b = {fb.lo[0], fb.lo[1], fb.lo[2], fb.lo[3], fb.hi[0], fb.hi[1], fb.hi[2], fb.hi[3]};
    // cbsz = blgp = 0: both operands fp8 e4m3.  With both scale operands the constant 0 hipcc selects
    // the UNSCALED v_mfma_f32_16x16x128_f8f6f4 (one 8-byte instruction, products taken as they are)
    // instead of the v_mfma_ld_scale_b32 + v_mfma_scale pair with E8M0 scales of 2^0.
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 0, 0, 0);
  } else if constexpr (MMA == kMmaBf16) {
    // 128-byte K step = 64 bf16: chunk g holds k = 8g..8g+7, chunk g+4 holds k = 32+8g.. (e4m3fnuz compat path)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa.lo), __builtin_bit_cast(bf16x8, fb.lo), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa.hi), __builtin_bit_cast(bf16x8, fb.hi), acc, 0, 0, 0);
  } else if constexpr (MMA == kMmaF16) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa.lo), __builtin_bit_cast(f16x8, fb.lo), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa.hi), __builtin_bit_cast(f16x8, fb.hi), acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa.lo, fb.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa.hi, fb.hi, acc, 0, 0, 0);
  }
}


// One HALF of mma_step for the operand types that take two MFMAs per 128-byte K step (PART 0: the 16-byte chunks g, PART 1: the
// chunks g + 4); the fp8 form is one instruction (PART 0; PART 1 is empty).  Lets a cluster issue all its PART-0 MFMAs, then all
// its PART-1 MFMAs, each pass in an order in which consecutive instructions share an operand (cluster_mma in gemm_mfma.hip).
template <int MMA, int PART>
__device__ __forceinline__ void mma_part(typename AccT<MMA>::type& acc, const Frag& fa, const Frag& fb) {
  if constexpr (MMA == kMmaFp8) {
    if constexpr (PART == 0) mma_step<MMA>(acc, fa, fb);
  } else if constexpr (MMA == kMmaBf16) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, PART ? fa.hi : fa.lo), __builtin_bit_cast(bf16x8, PART ? fb.hi : fb.lo), acc, 0, 0, 0);
  } else if constexpr (MMA == kMmaF16) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, PART ? fa.hi : fa.lo), __builtin_bit_cast(f16x8, PART ? fb.hi : fb.lo), acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(PART ? fa.hi : fa.lo, PART ? fb.hi : fb.lo, acc, 0, 0, 0);
  }
}


}  // namespace tile
}  // namespace conch
