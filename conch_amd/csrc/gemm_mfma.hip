// LDS-tiled MFMA GEMM kernels for gfx950 (MI355X): scaled_gemm (int8 / fp8-e4m3fn inputs).
//
// Replaces the GemLite-derived Triton kernel of the reference
// (conch/kernels/quantization/gemm.py:219-457, tiles 128x64x64, 8 warps) with a CDNA4-first design:
//
//   * "NT" layout contract: A is [M][K], B is consumed as B^T = [N][K]; both K-contiguous with
//     1-byte elements (exactly what the reference's tests/benchmarks pass: b = (N,K).T).
//   * 256x256 output tile per 512-thread workgroup (8 wave64s as 2(M) x 4(N), 128x64 per wave),
//     K advanced 128 BYTES per step, so fp8 / int8 / (16-bit, see gemm_mixed) share one geometry.
//   * operands stream global -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA, no VGPR staging):
//     a 16 KiB "unit" = 128 rows x 128 B, built of 8-row x 128-B subtiles (one wave-instruction
//     each: full 128-B lines from HBM/L2).  The 16-byte chunks of a row are XOR-swizzled
//     (chunk ^= (row>>1)&7) on the SOURCE side so that every ds_read_b128 fragment read is
//     bank-conflict free (tools/lds_bank_check.py).
//   * fp8 uses v_mfma_scale_f32_16x16x128_f8f6f4 with unit E8M0 scales (the only fp8 form that
//     reaches the 2x-bf16 rate on gfx950); int8 uses v_mfma_i32_16x16x64_i8 (exact int32).
//   * MFMA operands are swapped (D rows = n, D cols = m) and the n rows of a unit are permuted at
//     staging time so that a lane ends up with 8 consecutive n of one output row: 16-byte stores.
//   * epilogue fused: sb * (sa * float(acc)) in that order (bit-parity with
//     conch/reference/quantization/scaled_gemm.py:20-25), RNE cast, bias add in the output dtype.
//   * the K loop is a ping-pong: the two waves that share a SIMD alternate between an MFMA cluster and a
//     {ds_read, LDS-DMA issue} segment, with counted vmcnt waits so that four units stay in flight
//     across the raw s_barriers.  Variant 5 (default) uses two phases per K step (16 MFMAs per cluster),
//     variant 3 four (8 MFMAs per cluster, the form described phase by phase below).
//   * workgroup ids are remapped XCD-aware (8 XCDs, private L2s) on top of a GROUP_M raster.
#include <algorithm>

#include "common.hpp"
#include "gemm.hpp"
#include "mfma_tile.hpp"
#include "epilogue_rows.hpp"

namespace conch {
namespace {

using namespace tile;

// The register state of one wave: 8 (m) x 4 (n) accumulator tiles of 16x16.
template <int MMA>
struct WaveTile {
  typename AccT<MMA>::type acc[8][4];
  Frag fm[4];      // current 64-row m sub-half
  Frag fn[2][2];   // both 32-row n sub-halves
};

// One phase = {optional fragment reads, optional LDS-DMA issue, counted vmcnt} barrier {MFMAs} barrier.
// PHASE 0: read n0,m0 -> Q(m0,n0); 1: read n1 -> Q(m0,n1); 2: read m1 -> Q(m1,n1); 3: Q(m1,n0).
// NT = 16-column MFMA tiles per wave and 16-row m tile: 4 (256-column workgroup tile) or 3 (192-column tile: the V2 unit
// holds ONE 16-row n tile per wave-column instead of two)
template <int MMA, int PHASE, bool PINGPONG, int NT = 4>
__device__ __forceinline__ void phase_reads(WaveTile<MMA>& w, const char* lds, int buf, int m_base, int n_base) {
  if constexpr (PHASE == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) w.fn[0][t] = read_frag(lds, buf + kV1 * kUnitBytes + n_base + t * 2048);
    if constexpr (PINGPONG) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w.fm[i] = read_frag(lds, buf + kU1 * kUnitBytes + m_base + i * 2048);
  } else if constexpr (PHASE == 1) {
#pragma unroll
    for (int t = 0; t < NT - 2; ++t) w.fn[1][t] = read_frag(lds, buf + kV2 * kUnitBytes + n_base + t * 2048);
  } else if constexpr (PHASE == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) w.fm[i] = read_frag(lds, buf + kU2 * kUnitBytes + m_base + i * 2048);
  }
}

template <int MMA, int PHASE, int NT = 4>
__device__ __forceinline__ void phase_mma(WaveTile<MMA>& w) {
  constexpr int MH = (PHASE >= 2) ? 1 : 0;                 // m sub-half
  constexpr int NH = (PHASE == 1 || PHASE == 2) ? 1 : 0;   // n sub-half
  constexpr int TN = NH == 1 ? NT - 2 : 2;                 // 16-row n tiles of this sub-half
#ifdef CONCH_EXP_ROWWISE_MFMA
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < TN; ++t) mma_step<MMA>(w.acc[MH * 4 + i][NH * 2 + t], w.fn[NH][t], w.fm[i]);
#else
  // boustrophedon over (m tile, n tile), the two-instruction operand types in two passes: see cluster_mma
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int tt = 0; tt < TN; ++tt) {
      const int t = (i & 1) ? TN - 1 - tt : tt;
      mma_part<MMA, 0>(w.acc[MH * 4 + i][NH * 2 + t], w.fn[NH][t], w.fm[i]);
    }
  if constexpr (MMA != kMmaFp8) {
#pragma unroll
    for (int i = 3; i >= 0; --i)
#pragma unroll
      for (int tt = 0; tt < TN; ++tt) {
        const int t = (i & 1) ? tt : TN - 1 - tt;
        mma_part<MMA, 1>(w.acc[MH * 4 + i][NH * 2 + t], w.fn[NH][t], w.fm[i]);
      }
  }
#endif
}

// The MFMA half of a phase: the cluster at raised priority, then the closing barrier.
// (Measured and rejected: arriving at the barrier 2-4 MFMAs early, a single barrier per phase with
// group-dependent placement, a 4-wave 512-VGPR in-wave-pipelined tile -- all slower, profiles/README.md.)
template <int MMA, int PHASE>
__device__ __forceinline__ void phase_cluster(WaveTile<MMA>& w) {
  __builtin_amdgcn_s_setprio(1);
  phase_mma<MMA, PHASE>(w);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------
// epilogue
// ---------------------------------------------------------------------------------------------
// The per-row / per-column scales and the bias of the block's 256 rows / 256 columns are parked in
// the 3 KiB of LDS behind the operand buffers at kernel start (one coalesced load per thread), so
// the epilogue is LDS reads + arithmetic + 16-byte stores with no dependent global round trips.
constexpr int kEpiOff = kLdsBytes;            // float sa[256] | float sb[256] | float bias[256]
constexpr int kLdsTotal = kLdsBytes + 3 * 1024;

struct EpiPrefetch {
  float v0;
  uint32_t bias_bits;  // converted when parked, so the load is not waited for here
};

__device__ __forceinline__ EpiPrefetch epilogue_prefetch(const ScaledGemmArgs& p, int bm0, int bn0) {
  // branch-free: threads 0-255 fetch scale_a[row], threads 256-511 scale_b[col] (and the bias).  Fused gate/up form:
  // the tile has 128 output columns; slots 0-127 take their gate columns, slots 128-255 the up columns n further right
  EpiPrefetch e;
  const int t = threadIdx.x;
  const bool is_b = t >= 256;
  const int tt = t & 255;
  int col = min(bn0 + tt, (int)p.n - 1);
  if (p.fuse_silu) col = tt < 128 ? min(bn0 + tt, (int)p.n - 1) : (int)p.n + min(bn0 + tt - 128, (int)p.n - 1);
  const int idx = is_b ? col : min(bm0 + tt, (int)p.m - 1);
  const float* base = is_b ? p.scale_b : p.scale_a;
  const bool vec = (is_b ? p.scale_b_numel : p.scale_a_numel) != 1;
  e.v0 = base ? base[vec ? idx : 0] : 1.0f;  // NULL scale pointer = 1 (the 16-bit operand path of gemm_modes.hip)
  if (!is_b) e.v0 *= p.acc_scale;  // 1, or the exact 1/4 of e4m3fnuz operands on the OCP fp8 MFMA (gemm.hpp)
  uint32_t bits = 0;
  if (p.bias) bits = ((const uint16_t*)p.bias)[col];
  e.bias_bits = bits;
  return e;
}

template <int OUT_DT>
__device__ __forceinline__ void epilogue_park(char* lds, const EpiPrefetch& e) {
  float* f = (float*)(lds + kEpiOff);
  const int t = threadIdx.x;
  uint32_t bits = e.bias_bits;
  asm volatile("" : "+v"(bits));  // keep the conversion (and so the wait for the load) down here
  f[t] = e.v0;  // t < 256: sa[t]; else sb[t-256] at f[256 + (t-256)]
  if (t >= 256) f[t + 256] = bits16_to_float<OUT_DT>((uint16_t)bits);
}

// 16-byte C store.  Experiment builds (python -m conch_amd._build --variant X -DCONCH_EXP_C_STORE=n) pick the cache policy:
// 1 = sc1 (write-through, the line is dropped from the XCD's L2), 2 = nt, 3 = sc0 sc1; the product build stores plainly.
__device__ __forceinline__ void store_c16(uint16_t* dst, const i32x4& pk) {
#if defined(CONCH_EXP_C_STORE) && CONCH_EXP_C_STORE == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(pk) : "memory");
#elif defined(CONCH_EXP_C_STORE) && CONCH_EXP_C_STORE == 2
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(pk) : "memory");
#elif defined(CONCH_EXP_C_STORE) && CONCH_EXP_C_STORE == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(pk) : "memory");
#else
  *(i32x4*)dst = pk;
#endif
}

template <int MMA, int OUT_DT, int NT = 4>
__device__ __forceinline__ void epilogue(const typename AccT<MMA>::type (&acc)[8][4], const ScaledGemmArgs& p, const char* lds,
                                         int bm0, int bn0, int wr, int wc, int lane, int epi_off = kEpiOff) {
  const int g = lane >> 4, jm = lane & 15;
  const bool vec_store = (p.c_stride_m % 8 == 0) && (((uintptr_t)p.c & 15) == 0);
  const bool has_bias = p.bias != nullptr;
  const float* lsa = (const float*)(lds + epi_off);
  const float* lsb = lsa + 256;
  const float* lbias = lsa + 512;
  if constexpr (NT == 3) {
    // 192-column tile, second sub-half: ONE MFMA tile per m tile, D rows 4g+e = columns 4g+e of the wave-column's last 16
    const int nl = wc * 48 + 32 + 4 * g;
    const int n0 = bn0 + nl;
    const f32x4 sb4 = *(const f32x4*)(lsb + nl), bs4 = *(const f32x4*)(lbias + nl);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int ml = wr * 128 + mt * 16 + jm;
      const int m = bm0 + ml;
      const float sa = lsa[ml];
      i32x2 pk;
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int e = 2 * e2;
        const f32x2 a2 = {(float)acc[mt][2][e], (float)acc[mt][2][e + 1]};
        f32x2 v = f32x2{sa, sa} * a2;
        v = pin_f32x2(f32x2{sb4[e], sb4[e + 1]} * v);
        uint32_t h = pack2_bits16<OUT_DT>(v);
        if (has_bias) h = pack2_bits16<OUT_DT>(pin_f32x2(unpack2_bits16<OUT_DT>(h) + f32x2{bs4[e], bs4[e + 1]}));
        pk[e2] = (int)h;
      }
      if (m < p.m) {
        uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n0;
        if (vec_store && n0 + 4 <= p.n) {
          *(i32x2*)dst = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + e < p.n) dst[e] = (uint16_t)((uint32_t)pk[e >> 1] >> (16 * (e & 1)));
        }
      }
    }
  }
#pragma unroll
  for (int nh = 0; nh < (NT == 3 ? 1 : 2); ++nh) {
    const int nl = wc * (NT == 3 ? 48 : 64) + nh * 32 + 8 * g;  // block-local column of this lane's 8 outputs
    const int n0 = bn0 + nl;
    const f32x4 sb_lo = *(const f32x4*)(lsb + nl), sb_hi = *(const f32x4*)(lsb + nl + 4);
    const f32x4 bs_lo = *(const f32x4*)(lbias + nl), bs_hi = *(const f32x4*)(lbias + nl + 4);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int ml = wr * 128 + mt * 16 + jm;
      const int m = bm0 + ml;
      const float sa = lsa[ml];
      // two adjacent columns at a time: v_pk_mul_f32 x 2 and ONE packed convert per output dword, the same
      // IEEE operations per element as the scalar form (fp32 product, fp32 product, RNE cast, bias in the output dtype)
      i32x4 pk;
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const int e = 2 * e2;
        const f32x2 a2 = {(float)acc[mt][nh * 2 + (e >> 2)][e & 3], (float)acc[mt][nh * 2 + (e >> 2)][(e & 3) + 1]};
        const f32x2 sb2 = e < 4 ? f32x2{sb_lo[e & 3], sb_lo[(e & 3) + 1]} : f32x2{sb_hi[e & 3], sb_hi[(e & 3) + 1]};
        f32x2 v = f32x2{sa, sa} * a2;  // scaled_gemm.py:21
        v = pin_f32x2(sb2 * v);        // :22 (rounded to fp32 before the cast, like torch)
        uint32_t h = pack2_bits16<OUT_DT>(v);  // :23
        if (has_bias) {                // :24-25, added in the output dtype
          const f32x2 b2 = e < 4 ? f32x2{bs_lo[e & 3], bs_lo[(e & 3) + 1]} : f32x2{bs_hi[e & 3], bs_hi[(e & 3) + 1]};
          h = pack2_bits16<OUT_DT>(pin_f32x2(unpack2_bits16<OUT_DT>(h) + b2));
        }
        pk[e2] = (int)h;
      }
      if (m < p.m) {
        uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n0;
        if (vec_store && n0 + 8 <= p.n) {
          store_c16(dst, pk);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (n0 + e < p.n) dst[e] = (uint16_t)((uint32_t)pk[e >> 1] >> (16 * (e & 1)));
        }
      }
    }
  }
}

template <int MMA, int OUT_DT, bool MULTI = false>
__device__ __forceinline__ void epilogue_rows(const typename AccT<MMA>::type (&acc)[8][4], const ScaledGemmArgs& p, char* lds,
                                              int stage_off, int bm0, int bn0, int wr, int wc, int lane, int wave,
                                              int epi_off = kEpiOff) {
  if (!tile_stores_whole_lines(p, bm0, bn0)) {  // workgroup-uniform
    epilogue<MMA, OUT_DT, 4>(acc, p, lds, bm0, bn0, wr, wc, lane, epi_off);
    if constexpr (MULTI) {  // ragged tiles: the direct epilogue once more per further destination (rare: edge tiles only)
      for (int d = 0; d < p.n_more; ++d) {
        ScaledGemmArgs q = p;
        q.c = p.c_more[d];
        epilogue<MMA, OUT_DT, 4>(acc, q, lds, bm0, bn0, wr, wc, lane, epi_off);
      }
    }
  } else if (p.bias) {
    epilogue_rows_body<MMA, OUT_DT, true, 8, 256, MULTI>(acc, p, lds, stage_off, bm0, bn0, wr, wc, lane, wave, epi_off);
  } else {
    epilogue_rows_body<MMA, OUT_DT, false, 8, 256, MULTI>(acc, p, lds, stage_off, bm0, bn0, wr, wc, lane, wave, epi_off);
  }
}

// Fused gate/up epilogue (scaled_gemm_silu_and_mul): the lane's accumulator tiles 0,1 are the gate values and tiles 2,3
// the up values of the SAME eight output columns (make_stage_offsets with v2_delta = n).  Per pair of columns:
//   g = cast(sb_g * (sa * acc_g)) [+ bias_g],  u = cast(sb_u * (sa * acc_u)) [+ bias_u]      -- the reference's scaled_gemm
//   s = cast(g / (1 + exp(-g))),  y = cast(s * u)                                            -- its silu_and_mul
// with every cast an RNE rounding to the output dtype, as torch rounds after each op of the unfused pair.  exp and the
// reciprocal are the hardware's v_exp_f32 / v_rcp_f32 (1 ulp each, in fp32): the result can differ from the CPU oracle by
// one output ulp on a few elements in 10^4 (bound asserted in tests/test_gpu_gemm.py).
template <int OUT_DT>
__device__ __forceinline__ uint32_t silu_mul2(uint32_t g_bits, uint32_t u_bits, int act) {
  const f32x2 g = unpack2_bits16<OUT_DT>(g_bits);
  f32x2 s;
#pragma unroll
  for (int i = 0; i < 2; ++i) s[i] = act_f32(g[i], act);
  const f32x2 sr = unpack2_bits16<OUT_DT>(pack2_bits16<OUT_DT>(pin_f32x2(s)));
  return pack2_bits16<OUT_DT>(pin_f32x2(sr * unpack2_bits16<OUT_DT>(u_bits)));
}

template <int MMA, int OUT_DT>
__device__ __forceinline__ void epilogue_silu(const typename AccT<MMA>::type (&acc)[8][4], const ScaledGemmArgs& p, const char* lds,
                                              int bm0, int bn0, int wr, int wc, int lane) {
  const int g = lane >> 4, jm = lane & 15;
  const bool vec_store = (p.c_stride_m % 8 == 0) && (((uintptr_t)p.c & 15) == 0);
  const bool has_bias = p.bias != nullptr;
  const float* lsa = (const float*)(lds + kEpiOff);
  const float* lsb = lsa + 256;
  const float* lbias = lsa + 512;
  const int nl = wc * 32 + 8 * g;  // tile-local output column of this lane's 8 outputs; gate constants at nl, up at 128 + nl
  const int n0 = bn0 + nl;
  f32x4 sbv[2][2], bsv[2][2];      // [gate / up][low / high four columns]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      sbv[h][q] = *(const f32x4*)(lsb + 128 * h + nl + 4 * q);
      bsv[h][q] = *(const f32x4*)(lbias + 128 * h + nl + 4 * q);
    }
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int ml = wr * 128 + mt * 16 + jm;
    const int m = bm0 + ml;
    const float sa = lsa[ml];
    i32x4 pk;
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      const int e = 2 * e2;
      uint32_t gu[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 a2 = {(float)acc[mt][2 * h + (e >> 2)][e & 3], (float)acc[mt][2 * h + (e >> 2)][(e & 3) + 1]};
        const f32x2 sb2 = {sbv[h][e >> 2][e & 3], sbv[h][e >> 2][(e & 3) + 1]};
        f32x2 v = f32x2{sa, sa} * a2;
        v = pin_f32x2(sb2 * v);
        uint32_t hb = pack2_bits16<OUT_DT>(v);
        if (has_bias) {
          const f32x2 b2 = {bsv[h][e >> 2][e & 3], bsv[h][e >> 2][(e & 3) + 1]};
          hb = pack2_bits16<OUT_DT>(pin_f32x2(unpack2_bits16<OUT_DT>(hb) + b2));
        }
        gu[h] = hb;
      }
      pk[e2] = (int)silu_mul2<OUT_DT>(gu[0], gu[1], p.fuse_silu);
    }
    if (m < p.m) {
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
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
struct BlockSetup {
  int wave, lane, wr, wc, bm0, bn0, m_base, n_base, narrow;
  StageOffsets so;
  Srcs src;
};

// `tid`: the thread id to derive lane constants from.  The persistent kernel passes a value made opaque to the optimiser
// at every tile boundary: hipcc otherwise hoists the lane-constant parts of this function out of the tile loop and spills
// them across the K loop, and the reloads' compiler-inserted vmcnt(0) would drain the hand-counted LDS-DMA / store queue.
// `given_bm0` >= 0: the tile's first row / column are known (the persistent walk looks the next tile up once, before the epilogue,
// and keeps the two scalars): the raster arithmetic -- most of this function's time -- is skipped.
__device__ __forceinline__ BlockSetup setup_block(const ScaledGemmArgs& p, const TileSchedule* ts = nullptr, int bid = -1, int tid = -1,
                                                  int given_bm0 = -1, int given_bn0 = 0) {
  BlockSetup s;
  s.narrow = 0;
  if (bid < 0) bid = blockIdx.x;
  if (tid < 0) tid = threadIdx.x;
  s.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  s.lane = tid & 63;
  s.wr = s.wave >> 2;
  s.wc = s.wave & 3;
  // fused gate/up form: a tile is 256 rows x 128 OUTPUT columns (128 gate + 128 up columns of B)
  const int tile_n = p.fuse_silu ? kTileN / 2 : kTileN;
  const int tiles_m = ((int)p.m + kTileM - 1) / kTileM;
  const int tiles_n = ((int)p.n + tile_n - 1) / tile_n;
  if (given_bm0 >= 0) {
    s.bm0 = given_bm0;
    s.bn0 = given_bn0;
  } else if (ts) {
    const TilePlace tp = place_tile(bid, *ts, p.raster_magic, p.raster_shift);
    s.bm0 = tp.tm * kTileM;
    s.bn0 = tp.n0;
    s.narrow = tp.narrow;
  } else {
    const TileCoord tc = map_tile(bid, tiles_m, tiles_n, p.raster_magic, p.raster_shift);
    s.bm0 = tc.tm * kTileM;
    s.bn0 = tc.tn * tile_n;
  }
  const int lda = (int)p.a_stride_m, ldb = (int)p.b_stride_n;
  // timing experiments (wrong results by construction; never in the product build): every tile stages the operand panels
  // of tile (0, 0) [1], of its own row and column 0 [2], or of the first tile of its 4 x 8 block [3] -- an upper bound on
  // what serving the operand stream from L2 instead of the Infinity Cache is worth
#if defined(CONCH_EXP_SAME_PANELS) && CONCH_EXP_SAME_PANELS == 1
  const int sbm0 = 0, sbn0 = 0;
#elif defined(CONCH_EXP_SAME_PANELS) && CONCH_EXP_SAME_PANELS == 2
  const int sbm0 = s.bm0, sbn0 = 0;
#elif defined(CONCH_EXP_SAME_PANELS) && CONCH_EXP_SAME_PANELS == 3
  const int sbm0 = s.bm0 & ~1023, sbn0 = s.bn0 & ~2047;
#else
  const int sbm0 = s.bm0, sbn0 = s.bn0;
#endif
  if (s.narrow) s.so = make_stage_offsets(s.wave, s.lane, sbm0, sbn0, (int)p.m - 1, (int)p.n - 1, lda, ldb, 48, 32);
  else if (p.fuse_silu) s.so = make_stage_offsets(s.wave, s.lane, sbm0, sbn0, (int)p.m - 1, (int)p.n - 1, lda, ldb, 32, (int)p.n);
  else s.so = make_stage_offsets(s.wave, s.lane, sbm0, sbn0, (int)p.m - 1, (int)p.n - 1, lda, ldb);
  const int64_t b_cols = p.fuse_silu ? 2 * p.n : p.n;
  const uint32_t a_bytes = (uint32_t)((p.m - 1) * p.a_stride_m + p.k);
  const uint32_t b_bytes = (uint32_t)((b_cols - 1) * p.b_stride_n + p.k);
  s.src.a = make_uniform_rsrc(p.a, a_bytes);
  s.src.b = make_uniform_rsrc(p.b, b_bytes);
  // fragment read offsets inside a unit (bytes): row r of a 16-row tile, k-group g
  const int r = s.lane & 15, g = s.lane >> 4;
  const int lane_off = (r >> 3) * 1024 + (r & 7) * 128 + ((g ^ ((r >> 1) & 7)) * 16);
  s.m_base = (s.wr * 8) * 1024 + lane_off;   // wave-row's 64 rows inside a U unit
  s.n_base = (s.wc * 4) * 1024 + lane_off;   // wave-column's 32 rows inside a V unit
  return s;
}

template <int MMA>
__device__ __forceinline__ void zero_acc(WaveTile<MMA>& w) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) w.acc[i][j] = typename AccT<MMA>::type{0, 0, 0, 0};
}

// Variant 2: plain double-buffered loop -- one barrier per K step, next step's four units issued
// right after it.  Bring-up / cross-check kernel.
template <int MMA, int OUT_DT>
__global__ __launch_bounds__(kThreads, 2) void scaled_gemm_simple_kernel(ScaledGemmArgs p) {
  __shared__ __attribute__((aligned(1024))) char lds[kLdsTotal];
  if (gated_off(p.gate, p.gate_run_if)) return;
  const BlockSetup s = setup_block(p);
  WaveTile<MMA> w;
  zero_acc<MMA>(w);
  const int steps = (int)(p.k / kStepBytes);

  epilogue_park<OUT_DT>(lds, epilogue_prefetch(p, s.bm0, s.bn0));
  stage_unit<kU1>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kV1>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kV2>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kU2>(lds, s.src, s.so, s.wave, 0);
  for (int t = 0; t < steps; ++t) {
    CONCH_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    if (t + 1 < steps) {
      stage_unit<kU1>(lds, s.src, s.so, s.wave, t + 1);
      stage_unit<kV1>(lds, s.src, s.so, s.wave, t + 1);
      stage_unit<kV2>(lds, s.src, s.so, s.wave, t + 1);
      stage_unit<kU2>(lds, s.src, s.so, s.wave, t + 1);
    }
    const int buf = (t & 1) * kBufBytes;
    phase_reads<MMA, 0, false>(w, lds, buf, s.m_base, s.n_base);
    phase_mma<MMA, 0>(w);
    phase_reads<MMA, 1, false>(w, lds, buf, s.m_base, s.n_base);
    phase_mma<MMA, 1>(w);
    phase_reads<MMA, 2, false>(w, lds, buf, s.m_base, s.n_base);
    phase_mma<MMA, 2>(w);
    phase_mma<MMA, 3>(w);
  }
  epilogue<MMA, OUT_DT>(w.acc, p, lds, s.bm0, s.bn0, s.wr, s.wc, s.lane);
}

// Variant 3: 4 phases per K step, two wave groups (waves 0-3 / 4-7 = the two waves of each SIMD)
// staggered by one barrier so that one group's MFMA cluster overlaps the other's
// {ds_read, LDS-DMA issue} segment.  Unit s of the stream (4 per K step, order U1,V1,V2,U2) is
// issued in phase s-6 and first read in phase s-1 (U1: phase s); after issuing, every phase waits
// vmcnt(8) = "everything but the 4 youngest units has landed", one phase (= one barrier pair)
// before the unit is read.  A slot is rewritten 8 units later, >= 2 phases after its last read.
template <int MMA, int ISSUE0, int ISSUE1, int ISSUE2, int ISSUE3, int VM0, int VM1, int VM2, int VM3>
__device__ __forceinline__ void pingpong_step(WaveTile<MMA>& w, char* lds, const BlockSetup& s, int t) {
  const int buf = (t & 1) * kBufBytes;
  // ---- phase 0 ----
  phase_reads<MMA, 0, true>(w, lds, buf, s.m_base, s.n_base);
  if constexpr (ISSUE0) stage_unit<kV2>(lds, s.src, s.so, s.wave, t + 1);
  wait_vmcnt<VM0>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  phase_cluster<MMA, 0>(w);
  // ---- phase 1 ----
  phase_reads<MMA, 1, true>(w, lds, buf, s.m_base, s.n_base);
  if constexpr (ISSUE1) stage_unit<kU2>(lds, s.src, s.so, s.wave, t + 1);
  wait_vmcnt<VM1>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  phase_cluster<MMA, 1>(w);
  // ---- phase 2 ----
  phase_reads<MMA, 2, true>(w, lds, buf, s.m_base, s.n_base);
  if constexpr (ISSUE2) stage_unit<kU1>(lds, s.src, s.so, s.wave, t + 2);
  wait_vmcnt<VM2>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  phase_cluster<MMA, 2>(w);
  // ---- phase 3 ----
  if constexpr (ISSUE3) stage_unit<kV1>(lds, s.src, s.so, s.wave, t + 2);
  wait_vmcnt<VM3>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  phase_cluster<MMA, 3>(w);
}

template <int MMA, int OUT_DT>
__global__ __launch_bounds__(kThreads, 2) void scaled_gemm_pingpong_kernel(ScaledGemmArgs p) {
  __shared__ __attribute__((aligned(1024))) char lds[kLdsTotal];
  if (gated_off(p.gate, p.gate_run_if)) return;
  const BlockSetup s = setup_block(p);
  WaveTile<MMA> w;
  zero_acc<MMA>(w);
  const int steps = (int)(p.k / kStepBytes);

  // scales / bias of this block: issue the loads now, park them in LDS behind the prologue DMA
  const EpiPrefetch epi = epilogue_prefetch(p, s.bm0, s.bn0);
  // prologue: units 0..5 (all of K step 0, U1/V1 of K step 1)
  stage_unit<kU1>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kV1>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kV2>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kU2>(lds, s.src, s.so, s.wave, 0);
  if (steps > 1) {
    stage_unit<kU1>(lds, s.src, s.so, s.wave, 1);
    stage_unit<kV1>(lds, s.src, s.so, s.wave, 1);
    epilogue_park<OUT_DT>(lds, epi);
    CONCH_VMCNT(8);  // units 0,1 landed
  } else {
    epilogue_park<OUT_DT>(lds, epi);
    CONCH_VMCNT(4);
  }
  __builtin_amdgcn_s_barrier();
  if (s.wr == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of every SIMD

  int t = 0;
  for (; t + 2 < steps; ++t) pingpong_step<MMA, 1, 1, 1, 1, 8, 8, 8, 8>(w, lds, s, t);
  if (steps >= 2) {
    pingpong_step<MMA, 1, 1, 0, 0, 8, 8, 6, 4>(w, lds, s, t);
    ++t;
  }
  pingpong_step<MMA, 0, 0, 0, 0, 2, 0, -1, -1>(w, lds, s, t);
  if (s.wr == 0) __builtin_amdgcn_s_barrier();  // re-balance the barrier count: every wave is past its last LDS read

  epilogue_rows<MMA, OUT_DT>(w.acc, p, lds, 0, s.bm0, s.bn0, s.wr, s.wc, s.lane, s.wave);
}

// ---------------------------------------------------------------------------------------------
// Variant 5 (the default for K >= 256): the same ping-pong with TWO phases per K step instead of four.
// Phase A = quadrants (m0,n0),(m0,n1): reads V1, V2, U1 (16 ds_read_b128), 16 MFMAs (fp8).
// Phase B = quadrants (m1,n1),(m1,n0): reads U2 (8 ds_read_b128), 16 MFMAs.
// Same registers (one m sub-half + both n sub-halves), half the barriers per MFMA; measured 0.3-2 %
// faster than the four-phase form on every shape (the loop is bound by the L2 -> LDS path, not by
// barriers: a five-units-in-flight form was 1-2 % slower and removed).  Two units are
// issued per phase: A(t) issues V2,U2 of step t+1, B(t) issues U1,V1 of step t+2; after issuing, A
// waits vmcnt(8) (units <= U2(t) landed), B waits vmcnt(6) (units <= V2(t+1)).  A slot is re-targeted
// one phase after its last read, so the reads are retired (lgkmcnt(0)) before the phase's first barrier.
// ---------------------------------------------------------------------------------------------
#ifdef CONCH_CLOCK_PROBE
__device__ unsigned long long g_probe_scaled[kProbeBlocks * 8];
#endif

// The 16 (12 at the 192-column tile) MFMA pairs of a two-phase cluster: m tiles 4 * which .. + 3 against all n tiles.  Issued in
// boustrophedon order over (m tile, n tile), so that consecutive MFMAs share an operand (round 3: on a part whose launches are
// bound by energy the operand fetches count -- C3, 8192^3 and the C5 shard -1.1 ... -1.3 %, profiles/r03/mfma_order_ab.txt), and
// for the two-instruction operand types as two passes (all chunk-g halves, then all chunk-g+4 halves).  Every accumulator still
// sees its products in the same order: bit-identical.
template <int MMA, int NT>
__device__ __forceinline__ void cluster_mma(WaveTile<MMA>& w, int which) {
#ifdef CONCH_EXP_ROWWISE_MFMA  // A/B variant: rounds 1-2's order (n sub-half by n sub-half, m tile by m tile)
  if (which == 0) {
    phase_mma<MMA, 0, NT>(w);
    phase_mma<MMA, 1, NT>(w);
  } else {
    phase_mma<MMA, 2, NT>(w);
    phase_mma<MMA, 3, NT>(w);
  }
  return;
#endif
  const int mh = which == 0 ? 0 : 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int bb = 0; bb < NT; ++bb) {
      const int b = (i & 1) ? NT - 1 - bb : bb;
      mma_part<MMA, 0>(w.acc[mh + i][b], w.fn[b >> 1][b & 1], w.fm[i]);
    }
  if constexpr (MMA != kMmaFp8) {
#pragma unroll
    for (int i = 3; i >= 0; --i)
#pragma unroll
      for (int bb = 0; bb < NT; ++bb) {
        const int b = (i & 1) ? bb : NT - 1 - bb;
        mma_part<MMA, 1>(w.acc[mh + i][b], w.fn[b >> 1][b & 1], w.fm[i]);
      }
  }
}

template <int MMA, int NT>
__device__ __forceinline__ void pp2_cluster(WaveTile<MMA>& w, int which) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this phase's fragment reads have left LDS
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(1);
  cluster_mma<MMA, NT>(w, which);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

template <int MMA, int NT, int ISSUE_A, int ISSUE_B, int VM_A, int VM_B>
__device__ __forceinline__ void pp2_step(WaveTile<MMA>& w, char* lds, const BlockSetup& s, int t) {
  const int buf = (t & 1) * kBufBytes;
  // Timing experiments (wrong results; experiment builds only): CONCH_EXP_SKIP_READS drops the phase-B fragment reads (a third
  // of the LDS read bytes; the MFMAs reuse phase A's registers), CONCH_EXP_SKIP_DMA issues one unit per phase instead of two
  // (half the L2 -> LDS stream; the counted waits then wait for less, never for more).
  // ---- phase A ----
  phase_reads<MMA, 0, true>(w, lds, buf, s.m_base, s.n_base);
  phase_reads<MMA, 1, true, NT>(w, lds, buf, s.m_base, s.n_base);
  if constexpr (ISSUE_A) {
    stage_unit<kV2>(lds, s.src, s.so, s.wave, t + 1);
#ifndef CONCH_EXP_SKIP_DMA
    stage_unit<kU2>(lds, s.src, s.so, s.wave, t + 1);
#endif
  }
  wait_vm<VM_A>();
  pp2_cluster<MMA, NT>(w, 0);
  // ---- phase B ----
#ifndef CONCH_EXP_SKIP_READS
  phase_reads<MMA, 2, true>(w, lds, buf, s.m_base, s.n_base);
#endif
  if constexpr (ISSUE_B) {
    stage_unit<kU1>(lds, s.src, s.so, s.wave, t + 2);
#ifndef CONCH_EXP_SKIP_DMA
    stage_unit<kV1>(lds, s.src, s.so, s.wave, t + 2);
#endif
  }
  wait_vm<VM_B>();
  pp2_cluster<MMA, NT>(w, 1);
}

// K loop + epilogue of one tile; NT = 4 (256 columns) or 3 (192 columns: 24 instead of 32 MFMAs per wave and K step, the same
// staging stream and waits)
template <int MMA, int OUT_DT, bool SILU, int NT, bool ROWS = false, bool MULTI = false>
__device__ __forceinline__ void pp2_tile(const ScaledGemmArgs& p, char* lds, const BlockSetup& s, WaveTile<MMA>& w, int steps) {
  CONCH_PROBE(g_probe_scaled, 0);
  int t = 0;
  for (; t + 2 < steps; ++t) pp2_step<MMA, NT, 1, 1, 8, 6>(w, lds, s, t);
  pp2_step<MMA, NT, 1, 0, 8, 2>(w, lds, s, t);
  pp2_step<MMA, NT, 0, 0, 0, -1>(w, lds, s, t + 1);
  CONCH_PROBE(g_probe_scaled, 1);
  if (s.wr == 0) __builtin_amdgcn_s_barrier();  // re-balance the barrier count

  if constexpr (SILU) epilogue_silu<MMA, OUT_DT>(w.acc, p, lds, s.bm0, s.bn0, s.wr, s.wc, s.lane);
  else if constexpr (NT == 4 && ROWS) epilogue_rows<MMA, OUT_DT, MULTI>(w.acc, p, lds, 0, s.bm0, s.bn0, s.wr, s.wc, s.lane, s.wave);
  else epilogue<MMA, OUT_DT, NT>(w.acc, p, lds, s.bm0, s.bn0, s.wr, s.wc, s.lane);
}

// ROWS: the row-major epilogue through LDS (epilogue_rows) instead of the direct accumulator-layout stores
template <int MMA, int OUT_DT, bool SILU, bool ROWS = false, bool MULTI = false>
__global__ __launch_bounds__(kThreads, 2) void scaled_gemm_pp2_kernel(ScaledGemmArgs p, TileSchedule ts) {
  __shared__ __attribute__((aligned(1024))) char lds[kLdsTotal];
  if (gated_off(p.gate, p.gate_run_if)) return;
#ifndef CONCH_EXP_PROBE_AFTER_SETUP
  CONCH_PROBE(g_probe_scaled, 2);
#endif
  const BlockSetup s = SILU ? setup_block(p) : setup_block(p, &ts);
#ifdef CONCH_EXP_PROBE_AFTER_SETUP  // diagnostic: the "entry" stamp taken behind the tile / address arithmetic instead of before it
  asm volatile("" ::"v"(s.so.off[0][0]), "v"(s.so.off[3][1]), "v"(s.m_base), "v"(s.n_base));
  CONCH_PROBE(g_probe_scaled, 2);
#endif
  WaveTile<MMA> w;
  zero_acc<MMA>(w);
  const int steps = (int)(p.k / kStepBytes);  // >= 2 (dispatcher)

  const EpiPrefetch epi = epilogue_prefetch(p, s.bm0, s.bn0);
  stage_unit<kU1>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kV1>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kV2>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kU2>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kU1>(lds, s.src, s.so, s.wave, 1);
  stage_unit<kV1>(lds, s.src, s.so, s.wave, 1);
  epilogue_park<OUT_DT>(lds, epi);
  CONCH_VMCNT(6);  // units 0,1,2 landed (phase A reads all three)
  __builtin_amdgcn_s_barrier();
  if (s.wr == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of every SIMD

  if constexpr (!SILU) {
    if (s.narrow) {  // workgroup-uniform
      pp2_tile<MMA, OUT_DT, false, 3>(p, lds, s, w, steps);
      CONCH_PROBE(g_probe_scaled, 3);
      return;
    }
  }
  pp2_tile<MMA, OUT_DT, SILU, 4, ROWS, MULTI>(p, lds, s, w, steps);
  CONCH_PROBE(g_probe_scaled, 3);
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the two-phase ping-pong kernel: one workgroup per CU walks its tiles (virtual block ids b, b + grid,
// b + 2 grid ...: the same tile -> XCD map as the plain launch).  What it buys is the tile BOUNDARY (in-kernel stamps,
// profiles/r02/probe_sched.txt: 4.2 us from the last MFMA to the last store issued -- the CU's store path moves ~16 B/clk --
// + 0.7 us until the next workgroup enters + 1.9 us from entry to its first MFMA, against a 36.5 us K loop):
//   * after a tile's last barrier, BEFORE its epilogue, the next tile's first TWO K steps (8 units = the whole LDS ring) and
//     its scale vectors are put in flight by LDS-DMA; the epilogue's stores are issued behind them.  vmcnt retires in issue
//     order, so the loads are never queued behind the stores: `vmcnt(16)` after the epilogue = "everything but my 16 stores";
//   * the next K loop starts at once and runs its first two steps from LDS without waiting for any new load, so the stores
//     drain under ~2.2 us of MFMAs; from phase B of step 1 on the steady-state counted waits apply unchanged (whatever they
//     wait for is younger than the stores);
//   * no launch gap, no first-load latency between tiles.
// An earlier persistent attempt (round 1) issued the stores first and lost: the first counted wait of the next tile then
// waited for them.  Contract on top of the tile kernel's: K >= 512 bytes, no bias (its load would need a VGPR destination:
// hipcc drains the whole queue for those while LDS-DMA is in flight), uniform 256-column tiles.  Edge tiles (ragged M / N or
// unaligned C rows: a store count that is not 16 per wave) drain fully instead of counting.
// ---------------------------------------------------------------------------------------------
// The counted waits of the persistent walk, from what a wave puts into its vector-memory queue around a tile boundary (vmcnt
// retires loads, LDS-DMA and stores together, IN ISSUE ORDER -- the property all of this rests on):
//   [stage_scales: kScaleOps] [stage_two_steps: 2 K steps x 4 units x kPiecesPerUnit] [epilogue_rows_body: kStoreOps stores]
constexpr int kPiecesPerUnit = 2;                               // stage_unit issues two LDS-DMA wave-instructions
constexpr int kUnitsPerStep = 4;                                // U1, V1, V2, U2
constexpr int kScaleOps = 1;                                    // stage_scales: one 4-byte LDS-DMA per wave (the OLDEST of the group)
constexpr int kStoreOps = epilogue_rows_vm_ops(8);              // 16 whole-line stores per wave and whole tile
constexpr int kRingPieces = 2 * kUnitsPerStep * kPiecesPerUnit; // the whole ring in flight: 16 pieces
// boundary: units 0-2 of the next tile landed (the plain kernel's prologue wait) = all but the youngest 5 units + the stores
constexpr int kVmBoundary = (2 * kUnitsPerStep - 3) * kPiecesPerUnit + kStoreOps;               // 26
// A(0): units <= U2(0) landed = step 1's four units and the stores may stay in flight
constexpr int kVmA0 = kUnitsPerStep * kPiecesPerUnit + kStoreOps;                                 // 24
// B(0): units <= V2(1) landed; B(0) has issued U1(2), V1(2) itself: U2(1) + those two + the stores
constexpr int kVmB0 = 3 * kPiecesPerUnit + kStoreOps;                                             // 22
// A(1): units <= U2(1) landed; issued since: U1(2), V1(2), V2(2), U2(2) = the steady-state 8, plus the stores
constexpr int kVmA1 = kUnitsPerStep * kPiecesPerUnit + kStoreOps;                                 // 24
static_assert(kVmBoundary == 26 && kVmA0 == 24 && kVmB0 == 22 && kVmA1 == 24, "persistent walk: counted waits re-derived");
static_assert(kVmBoundary < 64, "vmcnt is a 6-bit field");
constexpr int kEpiBytes = 3 * 1024;
constexpr int kStageOffPersistent = kLdsBytes + 2 * kEpiBytes;          // 2 KiB per wave for the row-major epilogue
constexpr int kLdsTotalPersistent = kStageOffPersistent + 8 * 2048;     // operand ring + two sets of parked scales + staging

// scale_a / scale_b of a tile -> LDS by LDS-DMA (4 bytes per lane; waves 0-3: the 256 row scales, waves 4-7: the column scales)
__device__ __forceinline__ void stage_scales(char* lds, int epi_off, const ScaledGemmArgs& p, int wave, int tid, int bm0, int bn0) {
  const int tt = tid & 255;
  char* dst = lds + epi_off + wave * 256;
  if (wave < 4) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p.scale_a, 0, (uint32_t)(p.scale_a_numel * 4), 0x00020000);
    const int idx = p.scale_a_numel != 1 ? min(bm0 + tt, (int)p.m - 1) : 0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)dst, 4, idx * 4, 0, 0, 0);
  } else {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p.scale_b, 0, (uint32_t)(p.scale_b_numel * 4), 0x00020000);
    const int idx = p.scale_b_numel != 1 ? min(bn0 + tt, (int)p.n - 1) : 0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)dst, 4, idx * 4, 0, 0, 0);
  }
}

__device__ __forceinline__ void stage_two_steps(char* lds, const BlockSetup& s) {
  stage_unit<kU1>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kV1>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kV2>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kU2>(lds, s.src, s.so, s.wave, 0);
  stage_unit<kU1>(lds, s.src, s.so, s.wave, 1);
  stage_unit<kV1>(lds, s.src, s.so, s.wave, 1);
  stage_unit<kV2>(lds, s.src, s.so, s.wave, 1);
  stage_unit<kU2>(lds, s.src, s.so, s.wave, 1);
}

template <int MMA, int OUT_DT>
__global__ __launch_bounds__(kThreads, 2) void scaled_gemm_pp2_persistent_kernel(ScaledGemmArgs p, TileSchedule ts, int total_tiles, int direct_epilogue) {
  __shared__ __attribute__((aligned(1024))) char lds[kLdsTotalPersistent];
  const int steps = (int)(p.k / kStepBytes);  // >= 4 (dispatcher)
  int vb = blockIdx.x;
  int eb = 0;  // which set of parked scales the CURRENT tile uses
  BlockSetup s = setup_block(p, &ts, vb);
  stage_scales(lds, kLdsBytes, p, s.wave, threadIdx.x, s.bm0, s.bn0);
  stage_two_steps(lds, s);
  CONCH_VMCNT(0);
  __builtin_amdgcn_s_barrier();
  WaveTile<MMA> w;
  for (;;) {
    zero_acc<MMA>(w);
    if (s.wr == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of every SIMD
    CONCH_PROBE_AT(g_probe_scaled, 0, vb);
    // The whole ring (units 0-7 = K steps 0 and 1) was put in flight before the previous tile's 16 stores; the boundary
    // waited for units 0-2 only.  vmcnt retires in issue order, so a count of [loads allowed in flight] + 16 stores waits
    // for exactly the loads the plain kernel's counts wait for: A(0) units <= U2(0), B(0) <= V2(1), A(1) <= U2(1); the
    // stores may stay in flight until B(1), whose wait (6) is for a load issued behind them.  (First tile, edge tiles:
    // everything has drained and the counts are trivially met.)
    pp2_step<MMA, 4, 0, 1, kVmA0, kVmB0>(w, lds, s, 0);  // V2 / U2 of step 1 are already in flight or landed
    pp2_step<MMA, 4, 1, 1, kVmA1, 6>(w, lds, s, 1);
    int t = 2;
    for (; t + 2 < steps; ++t) pp2_step<MMA, 4, 1, 1, 8, 6>(w, lds, s, t);
    pp2_step<MMA, 4, 1, 0, 8, 2>(w, lds, s, t);
    pp2_step<MMA, 4, 0, 0, 0, -1>(w, lds, s, t + 1);
    CONCH_PROBE_AT(g_probe_scaled, 1, vb);
    if (s.wr == 0) __builtin_amdgcn_s_barrier();  // re-balance: every wave is past its last LDS read of this tile
    CONCH_PROBE_AT(g_probe_scaled, 2, vb);  // (slot 2 = boundary start here, not kernel entry)

    const int bm0 = s.bm0, bn0 = s.bn0;
    const int next = vb + (int)gridDim.x;
    const bool more = next < total_tiles;  // workgroup-uniform
    // The tile change looks the next tile up ONCE (the raster arithmetic is most of setup_block's 0.56 us,
    // profiles/r03/probe_setup.txt) and keeps its two scalars; the per-lane parts are evaluated before the epilogue (to issue
    // the loads) and again behind it, not carried across it: eight more live registers there would spill, and scratch traffic
    // would break the counted waits.
    int nbm0 = 0, nbn0 = 0;
    if (more) {
      const TilePlace tp = place_tile(next, ts, p.raster_magic, p.raster_shift);
      nbm0 = tp.tm * kTileM;
      nbn0 = tp.n0;
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));  // see setup_block
      const BlockSetup sn = setup_block(p, &ts, next, tid, nbm0, nbn0);
      stage_scales(lds, kLdsBytes + (eb ^ 1) * kEpiBytes, p, sn.wave, tid, nbm0, nbn0);
      stage_two_steps(lds, sn);
      __builtin_amdgcn_sched_barrier(0);  // the loads go out before the first store
    }
    // whole tiles: row-major through the spare LDS (the operand ring is being refilled), kStoreOps whole-line stores per wave;
    // direct_epilogue (CONCH_TUNE_EPILOGUE = 1) and tiles that are not whole store from the accumulator layout and drain
    const bool counted = !direct_epilogue && tile_stores_whole_lines(p, bm0, bn0);  // workgroup-uniform
    if (direct_epilogue) epilogue<MMA, OUT_DT, 4>(w.acc, p, lds, bm0, bn0, s.wr, s.wc, s.lane, kLdsBytes + eb * kEpiBytes);
    else epilogue_rows<MMA, OUT_DT>(w.acc, p, lds, kStageOffPersistent, bm0, bn0, s.wr, s.wc, s.lane, s.wave, kLdsBytes + eb * kEpiBytes);
    CONCH_PROBE_AT(g_probe_scaled, 3, vb);
    if (!more) return;
    __builtin_amdgcn_sched_barrier(0);
    // a whole tile issues exactly kStoreOps stores per wave: they and the 10 youngest LDS-DMA pieces may stay in flight = units
    // 0-2 of the next tile have landed (the plain kernel's prologue wait); anything else (masked / direct stores) drains
    if (counted) wait_vmcnt_n<kVmBoundary>();
    else CONCH_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    int tid2 = threadIdx.x;
    asm volatile("" : "+s"(nbm0), "+s"(nbn0), "+v"(tid2));  // keep hipcc from carrying the first evaluation across the epilogue
    s = setup_block(p, &ts, next, tid2, nbm0, nbn0);
    vb = next;
    eb ^= 1;
  }
}

}  // namespace

bool scaled_gemm_mfma_supported(const ScaledGemmArgs& p) {
  if (p.in_dtype != CONCH_DT_INT8 && p.in_dtype != CONCH_DT_FP8_E4M3FN) return false;
  if (p.out_dtype != CONCH_DT_FP16 && p.out_dtype != CONCH_DT_BF16) return false;
  if (p.m < 1 || p.n < 1 || p.k < kStepBytes || p.k % kStepBytes) return false;
  if (p.a_stride_k != 1 || p.b_stride_k != 1 || p.c_stride_n != 1) return false;
  if (p.a_stride_m % 16 || p.b_stride_n % 16) return false;
  if (((uintptr_t)p.a & 15) || ((uintptr_t)p.b & 15) || ((uintptr_t)p.c & 1)) return false;
  // 32-bit buffer offsets
  const int64_t lim = (int64_t)1 << 31;
  const int64_t b_cols = p.fuse_silu ? 2 * p.n : p.n;
  if (p.m * p.a_stride_m >= lim || b_cols * p.b_stride_n >= lim) return false;
  if (p.m >= (1 << 24) || b_cols >= (1 << 24)) return false;
  if (p.fuse_silu && p.k < 2 * kStepBytes) return false;  // the fused epilogue lives in the two-phase kernel only
  return true;
}

// Greedy (dispatch-order) makespan, in units of one 256-column tile's K loop, of t4 wide tiles followed by t3 narrow ones
// on `cus` workgroup slots; `e` = the fixed per-tile cost (epilogue store burst, re-entry, prologue) in the same unit.
static double schedule_makespan(int64_t t4, int64_t t3, int cus, double e) {
  const double c4 = 1.0 + e, c3 = 0.75 + e;
  const int64_t r4 = t4 / cus, rem4 = t4 % cus;
  // group A: slots free after r4 wide rounds; group B: the rem4 slots that run one more wide tile
  double ta = (double)r4 * c4, tb = ta + c4;
  int64_t na = cus - rem4, nb = rem4;
  double end = rem4 ? tb : ta;
  while (t3 > 0) {
    const bool use_a = nb == 0 || ta <= tb;
    double& t = use_a ? ta : tb;
    const int64_t slots = use_a ? na : nb;
    const int64_t take = t3 < slots ? t3 : slots;
    t += c3;
    if (t > end) end = t;
    t3 -= take;
  }
  return end;
}

// CONCH_TUNE_TILE_SCHEDULE: 0 = auto, 1 = uniform 256-column tiles, 2 = the best two-width schedule, wide tiles first on
// every XCD, 3 = the same with odd XCDs walking their narrow tiles first, 4 = wide and narrow tiles dealt to alternate slots of
// every XCD's first round.
// auto = uniform: measured on MI355X (profiles/r02/tile_schedule_ab.txt) the two-width schedule does not shorten C3 although
// it removes the idle last round -- the chip is power-limited, the idle CUs of a last round hand their power to the busy ones.
static TileSchedule choose_tile_schedule(const ScaledGemmArgs& p) {
  const int tiles_m = (int)((p.m + kTileM - 1) / kTileM);
  const int tiles_n = (int)((p.n + kTileN - 1) / kTileN);
  TileSchedule uni{tiles_m, tiles_n, 0, 0};
  const int mode = tuning(CONCH_TUNE_TILE_SCHEDULE);
  if (mode <= 1 || p.fuse_silu) return uni;
  const int cus = device_cu_count();
  // per-tile fixed cost: ~6 us (C-store burst, re-entry, first LDS-DMA landing) against 36.8 us of K loop at K = 4096
  const double e = 0.163 * 4096.0 / (double)p.k;
  const double t_uni = (double)(((int64_t)tiles_m * tiles_n + cus - 1) / cus) * (1.0 + e);
  double best = t_uni * 10.0;  // forced: the best two-width schedule whatever the model says
  TileSchedule pick = uni;
  const int a_max = (int)(p.n / kTileN);
  for (int a = a_max; a >= 0; --a) {
    const int64_t rest = p.n - (int64_t)a * kTileN;
    const int b = (int)((rest + 191) / 192);
    if ((int64_t)tiles_m * b < 8) continue;  // place_tile wants at least one narrow tile per XCD
    const double t = schedule_makespan((int64_t)tiles_m * a, (int64_t)tiles_m * b, cus, e);
    if (t < best - 1e-9) {
      best = t;
      pick = TileSchedule{tiles_m, a, b, mode == 3 ? 1 : mode == 4 ? 2 : 0};
    }
  }
  return pick;
}

int launch_scaled_gemm_mfma(const ScaledGemmArgs& p_in, int variant, hipStream_t stream) {
  ScaledGemmArgs p = p_in;
  const int tile_n = p.fuse_silu ? kTileN / 2 : kTileN;
  const int tiles_m = (int)((p.m + kTileM - 1) / kTileM);
  const int tiles_n = (int)((p.n + tile_n - 1) / tile_n);
  if (p.fuse_silu) variant = 5;
  dim3 grid((unsigned)(tiles_m * tiles_n));
  TileSchedule ts{tiles_m, tiles_n, 0, 0};
  if (!p.fuse_silu && variant != 2 && variant != 3 && p.k >= 2 * kStepBytes && p.n_more == 0) {  // (further destinations: uniform tiles)
    ts = choose_tile_schedule(p);
    grid = dim3((unsigned)(ts.tiles_m * (ts.big_cols + ts.narrow_cols)));
  }
  // CONCH_TUNE_PERSISTENT: 0 = auto, 1 = plain launch, 2 = persistent workgroups (one per CU) whenever the contract holds,
  // n > 2 = persistent with n workgroups (test hook: many tiles per workgroup on small problems)
  // CONCH_TUNE_EPILOGUE: 0 = auto, 1 = direct stores from the accumulator layout, 2 = row-major through LDS
  const int epi_mode = tuning(CONCH_TUNE_EPILOGUE);
  // auto = row-major (round 3: the pipelined single-buffer form with write-through whole-line stores, C3 -3.5 %, 8192^3 / the
  // C5 shard -1 %, int8 -3 %: profiles/r03/epilogue_ab.txt; round 2's unpipelined form had measured 0-2 % slower)
  const bool rows_epilogue = epi_mode != 1;
  // the raster's runtime divisor as a host-made reciprocal (uniform schedules; the two-width ones divide in the kernel)
  if (ts.narrow_cols == 0) set_raster_divisor((uint32_t)(kGroupM * ts.big_cols), &p.raster_magic, &p.raster_shift);
  const int persist_mode = tuning(CONCH_TUNE_PERSISTENT);
  const int cus = device_cu_count();
  const int total_tiles = (int)grid.x;
  const bool contract = !p.fuse_silu && (variant == 0 || variant == 5) && ts.narrow_cols == 0 && !p.bias && p.k >= 4 * kStepBytes &&
                        p.acc_scale == 1.0f && !p.gate;  // (its scales reach LDS by LDS-DMA: nothing multiplies them on the way)
  // further destinations (conch_scaled_gemm_multi): the plain two-phase kernel with the row-major epilogue, uniform tiles
  const bool multi = p.n_more > 0;
  const bool persistent = contract && persist_mode >= 2 && !multi;  // auto = plain launch: profiles/r02/probe_boundary.txt
  if (persistent) grid = dim3((unsigned)std::min(total_tiles, persist_mode > 2 ? persist_mode : cus));
#define CONCH_LAUNCH(MMA, OUT)                                                                           \
  do {                                                                                                   \
    if (persistent)                                                                                      \
      hipLaunchKernelGGL((scaled_gemm_pp2_persistent_kernel<MMA, OUT>), grid, dim3(kThreads), 0, stream, p, ts, total_tiles, rows_epilogue ? 0 : 1); \
    else if (variant == 2)                                                                               \
      hipLaunchKernelGGL((scaled_gemm_simple_kernel<MMA, OUT>), grid, dim3(kThreads), 0, stream, p);     \
    else if (multi)                                                                                      \
      hipLaunchKernelGGL((scaled_gemm_pp2_kernel<MMA, OUT, false, true, true>), grid, dim3(kThreads), 0, stream, p, ts); \
    else if (p.fuse_silu)                                                                                \
      hipLaunchKernelGGL((scaled_gemm_pp2_kernel<MMA, OUT, true>), grid, dim3(kThreads), 0, stream, p, ts);  \
    else if (variant != 3 && p.k >= 2 * kStepBytes && rows_epilogue)                                     \
      hipLaunchKernelGGL((scaled_gemm_pp2_kernel<MMA, OUT, false, true>), grid, dim3(kThreads), 0, stream, p, ts); \
    else if (variant != 3 && p.k >= 2 * kStepBytes)                                                      \
      hipLaunchKernelGGL((scaled_gemm_pp2_kernel<MMA, OUT, false>), grid, dim3(kThreads), 0, stream, p, ts); \
    else                                                                                                 \
      hipLaunchKernelGGL((scaled_gemm_pingpong_kernel<MMA, OUT>), grid, dim3(kThreads), 0, stream, p); \
    return check_launch("scaled_gemm_mfma");                                                             \
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

// 16-bit operands (bf16: the e4m3fnuz compat path, operands expanded exactly by repack.hip; fp16 / bf16: the dequantise-first
// path of gemm_modes.hip).  `p` is in BYTE units (k = 2 * K elements, strides in bytes) and satisfies the tile contract by
// construction; p.in_dtype says which 16-bit MFMA runs.
int launch_scaled_gemm_mfma_16bit(const ScaledGemmArgs& p_in, hipStream_t stream) {
  ScaledGemmArgs p = p_in;
  // few 256 x 256 tiles (M of a few hundred rows): the 128 x 128-tile kernel, by the cost model the 8-bit dispatcher uses
  // (its unit is the 128-byte K step, the same here) -- round 4: these completeness paths had run the 256 x 256 tiles whatever M was
  if (tuning(CONCH_TUNE_GEMM_VARIANT) == 0 && !p.fuse_silu && scaled_kernel_choice(p) == 1) return launch_scaled_gemm_mid(p, stream);
  const int tiles_m = (int)((p.m + kTileM - 1) / kTileM);
  const int tiles_n = (int)((p.n + kTileN - 1) / kTileN);
  set_raster_divisor((uint32_t)(kGroupM * tiles_n), &p.raster_magic, &p.raster_shift);
  const dim3 grid((unsigned)(tiles_m * tiles_n));
  if (p.in_dtype == CONCH_DT_FP16) {
    if (p.out_dtype == CONCH_DT_BF16)
      hipLaunchKernelGGL((scaled_gemm_pingpong_kernel<kMmaF16, CONCH_DT_BF16>), grid, dim3(kThreads), 0, stream, p);
    else
      hipLaunchKernelGGL((scaled_gemm_pingpong_kernel<kMmaF16, CONCH_DT_FP16>), grid, dim3(kThreads), 0, stream, p);
  } else if (p.out_dtype == CONCH_DT_BF16) {
    hipLaunchKernelGGL((scaled_gemm_pingpong_kernel<kMmaBf16, CONCH_DT_BF16>), grid, dim3(kThreads), 0, stream, p);
  } else {
    hipLaunchKernelGGL((scaled_gemm_pingpong_kernel<kMmaBf16, CONCH_DT_FP16>), grid, dim3(kThreads), 0, stream, p);
  }
  return check_launch("scaled_gemm_mfma_16bit");
}

int launch_scaled_gemm_mfma_bf16(const ScaledGemmArgs& p, hipStream_t stream) {
  ScaledGemmArgs q = p;
  q.in_dtype = CONCH_DT_BF16;
  return launch_scaled_gemm_mfma_16bit(q, stream);
}

// mixed-precision MFMA kernels live in gemm_mixed.hip
}  // namespace conch

#ifdef CONCH_CLOCK_PROBE
CONCH_PROBE_READER(conch_debug_probe_scaled, conch::g_probe_scaled)
#endif
