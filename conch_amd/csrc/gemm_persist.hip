// Persistent, N-balanced form of the scaled GEMM MFMA kernel (variant 4, the default fast path).
//
// Same tile machinery as gemm_mfma.hip (mfma_tile.hpp: 256-row tiles, 128-byte K steps, LDS-DMA
// units with the source-side swizzle, 4-phase ping-pong K loop), but:
//   * ONE workgroup per CU slot.  The output is cut into tiles_m row strips of 256; every strip's
//     columns, counted in 64-column units, are divided evenly over R = #CUs / tiles_m workgroups.
//     A workgroup walks its unit range left to right in tiles of NT*64 columns, NT in {1,2,3,4}
//     (a wave owns 16*NT columns, so ALL four SIMDs shrink together on a narrow tile).  C3
//     (4096x11008): 16 strips x 16 ranges of 10-11 units = 256 workgroups doing 256+256+192(128)
//     columns each, instead of 688 full tiles in three rounds with the last 31 % empty.
//   * the next tile's prologue LDS-DMA (and its scale/bias fetch) is issued BEFORE the current
//     tile's epilogue, so the HBM/L2 latency of a tile start hides behind the stores of the
//     previous tile.  The epilogue stores are buffer stores whose count per wave is fixed (rows or
//     columns outside the matrix are dropped by the buffer range check, not branched around), which
//     keeps the counted `s_waitcnt vmcnt(N)` exact.
//   * (strip, range) pairs are rasterised with the same XCD-aware GROUP_M map as the tiles of
//     variant 3, so the 32 workgroups sharing an L2 work on 4 strips x 8 neighbouring ranges.
#include "common.hpp"
#include "gemm.hpp"
#include "mfma_tile.hpp"

namespace conch {
namespace {

using namespace tile;

constexpr int kEpiRegion = 3 * 1024;                 // float sa[256] | sb[256] | bias[256]
constexpr int kEpiOff = kLdsBytes;                   // two regions, ping-ponged per tile
constexpr int kLdsTotal = kLdsBytes + 2 * kEpiRegion;
constexpr uint32_t kDropOffset = 0xffffff00u;        // beyond any num_records: the store is discarded

template <int MMA>
struct PTile {
  typename AccT<MMA>::type acc[8][4];
  Frag fm[4];   // current 64-row m sub-half
  Frag fn[2];   // current 32-row n sub-half (re-read in phase 3: 16 fewer live VGPRs than keeping both)
};

// Unit row -> source offsets for a tile of `nt` 64-column units starting at column bn0.
// Wave-column wc owns columns [wc*16*nt, (wc+1)*16*nt) of the tile.  Two adjacent 16-column MFMA
// tiles are interleaved (rows 4g+e of tile t <-> n = 8g+e+4t) so that a lane owns 8 consecutive n;
// a lone tile (nt odd) is laid out plainly (a lane owns 4 consecutive n).
__device__ __forceinline__ StageOffsets make_offsets_nt(int wave, int lane, int bm0, int bn0, int m_max, int n_max,
                                                        int lda, int ldb, int nt) {
  StageOffsets s;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rho = 16 * wave + 8 * j + (lane >> 3);
    const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
    const int mrow = (rho >> 6) * 128 + (rho & 63);
    const int wcol = (rho >> 5) * 16 * nt;
    const int r5 = rho & 31, r = r5 & 15, t = r5 >> 4;
    const int pair = 8 * (r >> 2) + (r & 3) + 4 * t;
    // V1: tiles 0,1 (pair) or tile 0 alone (nt == 1); V2: tiles 2,3 (pair, nt == 4) or tile 2 alone (nt == 3)
    const int n_v1 = wcol + (nt >= 2 ? pair : r);
    const int n_v2 = wcol + 32 + (nt == 4 ? pair : r);
    s.off[kU1][j] = min(bm0 + mrow, m_max) * lda + chunk * 16;
    s.off[kU2][j] = min(bm0 + mrow + 64, m_max) * lda + chunk * 16;
    s.off[kV1][j] = min(bn0 + n_v1, n_max) * ldb + chunk * 16;
    s.off[kV2][j] = min(bn0 + n_v2, n_max) * ldb + chunk * 16;
  }
  return s;
}

// PHASE 0: read n0, m0 -> Q(m0,n0); 1: read n1 -> Q(m0,n1); 2: read m1 -> Q(m1,n1); 3: re-read n0 -> Q(m1,n0).
template <int MMA, int PHASE>
__device__ __forceinline__ void p_reads(PTile<MMA>& w, const char* lds, int buf, int m_base, int n_base) {
  if constexpr (PHASE == 0 || PHASE == 3) {
#pragma unroll
    for (int t = 0; t < 2; ++t) w.fn[t] = read_frag(lds, buf + kV1 * kUnitBytes + n_base + t * 2048);
  }
  if constexpr (PHASE == 0) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w.fm[i] = read_frag(lds, buf + kU1 * kUnitBytes + m_base + i * 2048);
  } else if constexpr (PHASE == 1) {
#pragma unroll
    for (int t = 0; t < 2; ++t) w.fn[t] = read_frag(lds, buf + kV2 * kUnitBytes + n_base + t * 2048);
  } else if constexpr (PHASE == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) w.fm[i] = read_frag(lds, buf + kU2 * kUnitBytes + m_base + i * 2048);
  }
}

// `nt` (wave-uniform) = number of 16-column MFMA tiles this wave owns: tiles 0,1 live in the first
// sub-half, 2,3 in the second; absent tiles are skipped by scalar branches around groups of 4 MFMAs.
template <int MMA, int PHASE>
__device__ __forceinline__ void p_mma(PTile<MMA>& w, int nt) {
  constexpr int MH = (PHASE >= 2) ? 1 : 0;
  constexpr int NH = (PHASE == 1 || PHASE == 2) ? 1 : 0;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (NH * 2 + t < nt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) mma_step<MMA>(w.acc[MH * 4 + i][NH * 2 + t], w.fn[t], w.fm[i]);
    }
  }
}

struct PState {
  int wave, lane, wr, wc, m_base, n_base;
  StageOffsets so;
  Srcs src;
};

#define CONCH_PHASE_BARRIER()             \
  __builtin_amdgcn_sched_barrier(0);      \
  __builtin_amdgcn_s_barrier();           \
  __builtin_amdgcn_sched_barrier(0)

template <int MMA, int PHASE, int VM>
__device__ __forceinline__ void p_phase_tail(PTile<MMA>& w, int nt) {
  wait_vmcnt<VM>();
  CONCH_PHASE_BARRIER();
  __builtin_amdgcn_s_setprio(1);
  p_mma<MMA, PHASE>(w, nt);
  __builtin_amdgcn_s_setprio(0);
  CONCH_PHASE_BARRIER();
}

// One K step (4 phases).  Ix: whether phase x issues its unit (V2,U2 of step t+1; U1,V1 of t+2).
template <int MMA, int I0, int I1, int I2, int I3, int VM0, int VM1, int VM2, int VM3>
__device__ __forceinline__ void p_step(PTile<MMA>& w, char* lds, const PState& s, int t, int nt) {
  const int buf = (t & 1) * kBufBytes;
  p_reads<MMA, 0>(w, lds, buf, s.m_base, s.n_base);
  if constexpr (I0) stage_unit<kV2>(lds, s.src, s.so, s.wave, t + 1);
  p_phase_tail<MMA, 0, VM0>(w, nt);
  p_reads<MMA, 1>(w, lds, buf, s.m_base, s.n_base);
  if constexpr (I1) stage_unit<kU2>(lds, s.src, s.so, s.wave, t + 1);
  p_phase_tail<MMA, 1, VM1>(w, nt);
  p_reads<MMA, 2>(w, lds, buf, s.m_base, s.n_base);
  if constexpr (I2) stage_unit<kU1>(lds, s.src, s.so, s.wave, t + 2);
  p_phase_tail<MMA, 2, VM2>(w, nt);
  p_reads<MMA, 3>(w, lds, buf, s.m_base, s.n_base);
  if constexpr (I3) stage_unit<kV1>(lds, s.src, s.so, s.wave, t + 2);
  p_phase_tail<MMA, 3, VM3>(w, nt);
}

// Whole K loop of one tile; on entry units 0..5 have been issued and units 0,1 have landed.
template <int MMA>
__device__ __forceinline__ void p_kloop(PTile<MMA>& w, char* lds, const PState& s, int steps, int nt) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) w.acc[i][j] = typename AccT<MMA>::type{0, 0, 0, 0};
  if (s.wr == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of every SIMD
  int t = 0;
  for (; t + 2 < steps; ++t) p_step<MMA, 1, 1, 1, 1, 8, 8, 8, 8>(w, lds, s, t, nt);
  p_step<MMA, 1, 1, 0, 0, 8, 8, 6, 4>(w, lds, s, t, nt);
  p_step<MMA, 0, 0, 0, 0, 2, 0, -1, -1>(w, lds, s, t + 1, nt);
  if (s.wr == 0) __builtin_amdgcn_s_barrier();  // re-balance; also: every wave is done with the LDS buffers
}

struct EpiRegs {
  float scale;
  uint32_t bias_bits;
};

// threads 0-255 fetch scale_a of the tile's rows, 256-511 scale_b / bias of its columns
__device__ __forceinline__ EpiRegs epi_fetch(const ScaledGemmArgs& p, int bm0, int bn0) {
  EpiRegs e;
  const int t = threadIdx.x;
  const bool is_b = t >= 256;
  const int idx = min((is_b ? bn0 : bm0) + (t & 255), (int)(is_b ? p.n : p.m) - 1);
  const float* base = is_b ? p.scale_b : p.scale_a;
  const bool vec = (is_b ? p.scale_b_numel : p.scale_a_numel) != 1;
  e.scale = base[vec ? idx : 0];
  uint32_t bits = 0;
  if (p.bias) bits = ((const uint16_t*)p.bias)[min(bn0 + (t & 255), (int)p.n - 1)];
  e.bias_bits = bits;
  return e;
}

template <int OUT_DT>
__device__ __forceinline__ void epi_park(char* lds, int region, const EpiRegs& e) {
  float* f = (float*)(lds + kEpiOff + region * kEpiRegion);
  const int t = threadIdx.x;
  uint32_t bits = e.bias_bits;
  asm volatile("" : "+v"(bits));  // keep the conversion (and the wait for the load) down here
  f[t] = e.scale;                 // t < 256: sa[t]; else sb[t - 256]
  if (t >= 256) f[t + 256] = bits16_to_float<OUT_DT>((uint16_t)bits);
}

template <int MMA, int OUT_DT>
__device__ __forceinline__ uint16_t finish(typename AccT<MMA>::type const& acc, int e, float sa, float sb, float bias,
                                           bool has_bias) {
  float v = sa * (float)acc[e];   // scaled_gemm.py:21
  v = pin_f32(sb * v);            // :22, rounded to fp32 before the cast like torch
  uint16_t h = float_to_bits16<OUT_DT>(v);  // :23
  if (has_bias) h = float_to_bits16<OUT_DT>(pin_f32(bits16_to_float<OUT_DT>(h) + bias));  // :24-25
  return h;
}

// Stores of one half (nh) of the wave's columns: a PAIR of MFMA tiles (8 consecutive n per lane, 16-byte
// stores) or a LONE tile (4 consecutive n, 8-byte stores).  Always 8 buffer-store instructions per
// wave: rows / columns outside the matrix are discarded by the buffer range check, not branched
// around, so the number of VMEM operations in flight is known exactly.
template <int MMA, int OUT_DT, bool PAIR>
__device__ __forceinline__ void p_store_half(const PTile<MMA>& w, const ScaledGemmArgs& p, const float* lsa,
                                             const float* lsb, const float* lbias, __amdgpu_buffer_rsrc_t crsrc,
                                             int bm0, int bn0, int nl0, int nh, const PState& s) {
  constexpr int W = PAIR ? 8 : 4;
  const int g = s.lane >> 4, jm = s.lane & 15;
  const bool has_bias = p.bias != nullptr;
  const int nl = nl0 + W * g;  // tile-local column of this lane's first output
  const int n0 = bn0 + nl;
  float sb[W], bs[W];
#pragma unroll
  for (int e = 0; e < W; ++e) {
    sb[e] = lsb[nl + e];
    bs[e] = lbias[nl + e];
  }
  const bool col_ok = n0 + W <= p.n;  // N % 8 == 0 is part of the contract: all or nothing
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int ml = s.wr * 128 + mt * 16 + jm;
    const int m = bm0 + ml;
    const float sa = lsa[ml];
    const bool ok = col_ok && m < p.m;
    const uint32_t voff = ok ? (uint32_t)((m * (int)p.c_stride_m + n0) * 2) : kDropOffset;
    uint16_t o[W];
#pragma unroll
    for (int e = 0; e < W; ++e)
      o[e] = finish<MMA, OUT_DT>(w.acc[mt][nh * 2 + (e >> 2)], e & 3, sa, sb[e], bs[e], has_bias);
    if constexpr (PAIR) {
      i32x4 pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) pk[e] = (int)((uint32_t)o[2 * e] | ((uint32_t)o[2 * e + 1] << 16));
      __builtin_amdgcn_raw_buffer_store_b128(pk, crsrc, (int)voff, 0, 0);
    } else {
      i32x2 pk;
      pk[0] = (int)((uint32_t)o[0] | ((uint32_t)o[1] << 16));
      pk[1] = (int)((uint32_t)o[2] | ((uint32_t)o[3] << 16));
      __builtin_amdgcn_raw_buffer_store_b64(pk, crsrc, (int)voff, 0, 0);
    }
  }
}

// 8 stores per wave for nt <= 2, 16 for nt >= 3.
template <int MMA, int OUT_DT>
__device__ __forceinline__ void p_epilogue(const PTile<MMA>& w, const ScaledGemmArgs& p, const char* lds, int region,
                                           __amdgpu_buffer_rsrc_t crsrc, int bm0, int bn0, int nt, const PState& s) {
  const float* lsa = (const float*)(lds + kEpiOff + region * kEpiRegion);
  const float* lsb = lsa + 256;
  const float* lbias = lsa + 512;
  const int wcol = s.wc * 16 * nt;
  if (nt >= 2) p_store_half<MMA, OUT_DT, true>(w, p, lsa, lsb, lbias, crsrc, bm0, bn0, wcol, 0, s);
  else p_store_half<MMA, OUT_DT, false>(w, p, lsa, lsb, lbias, crsrc, bm0, bn0, wcol, 0, s);
  if (nt == 4) p_store_half<MMA, OUT_DT, true>(w, p, lsa, lsb, lbias, crsrc, bm0, bn0, wcol + 32, 1, s);
  else if (nt == 3) p_store_half<MMA, OUT_DT, false>(w, p, lsa, lsb, lbias, crsrc, bm0, bn0, wcol + 32, 1, s);
}

struct Sched {
  int tiles_m;   // 256-row strips
  int ranges;    // workgroups per strip
  int n_units;   // 64-column units per strip
  int debug;     // timing experiments only (CONCH_TUNE_DEBUG_FLAGS): 1 = drop A loads, 2 = drop B loads
};

template <int MMA, int OUT_DT>
__global__ __launch_bounds__(kThreads, 2) void scaled_gemm_persistent_kernel(ScaledGemmArgs p, Sched sc) {
  __shared__ __attribute__((aligned(1024))) char lds[kLdsTotal];
  PState s;
  s.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  s.lane = threadIdx.x & 63;
  s.wr = s.wave >> 2;
  s.wc = s.wave & 3;
  const int r = s.lane & 15, g = s.lane >> 4;
  const int lane_off = (r >> 3) * 1024 + (r & 7) * 128 + ((g ^ ((r >> 1) & 7)) * 16);
  s.m_base = (s.wr * 8) * 1024 + lane_off;
  s.n_base = (s.wc * 4) * 1024 + lane_off;
  const uint32_t a_bytes = (uint32_t)((p.m - 1) * p.a_stride_m + p.k);
  const uint32_t b_bytes = (uint32_t)((p.n - 1) * p.b_stride_n + p.k);
  const uint32_t c_bytes = (uint32_t)(((p.m - 1) * p.c_stride_m + p.n) * 2);
  // a zero-record descriptor makes the range check drop every load through it (traffic-pricing builds)
  s.src.a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (sc.debug & 1) ? 0u : a_bytes, 0x00020000);
  s.src.b = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, (sc.debug & 2) ? 0u : b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, c_bytes, 0x00020000);

  const TileCoord tc = map_tile(blockIdx.x, sc.tiles_m, sc.ranges);  // (strip, range)
  const int bm0 = tc.tm * kTileM;
  int pos = __builtin_amdgcn_readfirstlane((int)((int64_t)tc.tn * sc.n_units / sc.ranges));
  const int end = __builtin_amdgcn_readfirstlane((int)((int64_t)(tc.tn + 1) * sc.n_units / sc.ranges));
  if (pos >= end) return;
  const int steps = (int)(p.k / kStepBytes);
  const int m_max = (int)p.m - 1, n_max = (int)p.n - 1, lda = (int)p.a_stride_m, ldb = (int)p.b_stride_n;

  // first tile: prologue with nothing to hide behind
  int nt = __builtin_amdgcn_readfirstlane(min(4, end - pos));
  int bn0 = pos * 64;
  s.so = make_offsets_nt(s.wave, s.lane, bm0, bn0, m_max, n_max, lda, ldb, nt);
  {
    const EpiRegs epi = epi_fetch(p, bm0, bn0);
    stage_unit<kU1>(lds, s.src, s.so, s.wave, 0);
    stage_unit<kV1>(lds, s.src, s.so, s.wave, 0);
    stage_unit<kV2>(lds, s.src, s.so, s.wave, 0);
    stage_unit<kU2>(lds, s.src, s.so, s.wave, 0);
    stage_unit<kU1>(lds, s.src, s.so, s.wave, 1);
    stage_unit<kV1>(lds, s.src, s.so, s.wave, 1);
    epi_park<OUT_DT>(lds, 0, epi);
    CONCH_VMCNT(8);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  PTile<MMA> w;
  int region = 0;
  while (true) {
    p_kloop<MMA>(w, lds, s, steps, nt);
    const int next_pos = pos + nt;
    const bool has_next = next_pos < end;
    const int next_nt = __builtin_amdgcn_readfirstlane(min(4, end - next_pos));
    const int next_bn0 = next_pos * 64;
    EpiRegs next_epi;
    if (has_next) {
      // every wave is past its last LDS read (re-balance barrier): restage for the next tile now, so
      // that the DMA latency overlaps this tile's epilogue
      s.so = make_offsets_nt(s.wave, s.lane, bm0, next_bn0, m_max, n_max, lda, ldb, next_nt);
      next_epi = epi_fetch(p, bm0, next_bn0);
      stage_unit<kU1>(lds, s.src, s.so, s.wave, 0);
      stage_unit<kV1>(lds, s.src, s.so, s.wave, 0);
      stage_unit<kV2>(lds, s.src, s.so, s.wave, 0);
      stage_unit<kU2>(lds, s.src, s.so, s.wave, 0);
      stage_unit<kU1>(lds, s.src, s.so, s.wave, 1);
      stage_unit<kV1>(lds, s.src, s.so, s.wave, 1);
    }
    p_epilogue<MMA, OUT_DT>(w, p, lds, region, crsrc, bm0, bn0, nt, s);
    if (!has_next) break;
    epi_park<OUT_DT>(lds, region ^ 1, next_epi);
    // VMEM ops younger than unit 1's DMA: units 2..5 (8) + this tile's stores (8 or 16)
    if (nt >= 3) wait_vmcnt<24>();
    else wait_vmcnt<16>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    pos = next_pos;
    nt = next_nt;
    bn0 = next_bn0;
    region ^= 1;
  }
}

int g_num_cus = 0;

int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_num_cus = prop.multiProcessorCount;
    if (g_num_cus <= 0) g_num_cus = 256;
  }
  return g_num_cus;
}

}  // namespace

bool scaled_gemm_persistent_supported(const ScaledGemmArgs& p) {
  if (!scaled_gemm_mfma_supported(p)) return false;
  if (p.k < 2 * kStepBytes) return false;                          // prologue stages two K steps
  if (p.n % 8 || p.c_stride_m % 8 || ((uintptr_t)p.c & 15)) return false;  // 16-byte stores, all-or-nothing columns
  if (p.m * p.c_stride_m * 2 >= ((int64_t)1 << 31)) return false;  // 32-bit store offsets
  return true;
}

int launch_scaled_gemm_persistent(const ScaledGemmArgs& p, hipStream_t stream) {
  Sched sc;
  sc.tiles_m = (int)((p.m + kTileM - 1) / kTileM);
  sc.n_units = (int)((p.n + 63) / 64);
  int ranges = num_cus() / sc.tiles_m;
  if (ranges < 1) ranges = 1;
  if (ranges > sc.n_units) ranges = sc.n_units;
  sc.ranges = ranges;
  sc.debug = tuning(1);
  const dim3 grid((unsigned)(sc.tiles_m * sc.ranges));
#define CONCH_LAUNCH(MMA, OUT)                                                                              \
  do {                                                                                                      \
    hipLaunchKernelGGL((scaled_gemm_persistent_kernel<MMA, OUT>), grid, dim3(kThreads), 0, stream, p, sc);  \
    return check_launch("scaled_gemm_persistent");                                                          \
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
