// Mixed-precision GEMM for gfx950: fp16/bf16 activations x group-quantized int4/int8 packed weights.
//
// Replaces the mixed-precision instantiation of the reference's Triton kernel
// (conch/kernels/quantization/gemm.py:219-457 via mixed_precision_gemm_launcher :482-545).
//
// Same 256x256 / 8-wave geometry as the scaled GEMM (mfma_tile.hpp), K advanced 64 elements
// (128 bytes of fp16/bf16) per step, MFMA = v_mfma_f32_16x16x32_{f16,bf16}:
//   * activations X stream global -> LDS by LDS-DMA (U units), exactly like the scaled GEMM's A;
//   * packed weights are read ONCE per workgroup as 32-bit words straight from the [K/pf][N] tensor
//     (a word holds 8 (int4) / 4 (int8) consecutive k of one column = one 16-byte / 8-byte piece of
//     that column's LDS row), dequantised IN REGISTERS and written to the V units with
//     ds_write_b128, so the MFMA loop reads both operands from LDS with conflict-free ds_read_b128;
//   * dequantisation is bit-identical to the reference's w_ref (quant_utils.py:74,
//     kernels gemm.py:192-210): (q - bias - zp) is formed exactly, then ONE rounding multiply by the
//     fp16/bf16 group scale.  fp16 uses the 0x6400 magic-number trick (1024+q as an fp16 bit
//     pattern), v_pk_add_f16 and v_pk_mul_f16; bf16 multiplies in fp32 (exact 8x8-bit product)
//     and rounds once with v_cvt_pk_bf16_f32;
//   * the weight words / scales / zero-points of K step t+2 are prefetched into registers while
//     step t is multiplied; step t+1's are converted right after the step's barrier.
//   * fp32 accumulation, RNE cast, 16-byte stores (8 consecutive n per lane).
#include "common.hpp"
#include "gemm.hpp"
#include "mfma_tile.hpp"

namespace conch {
namespace {

using namespace tile;

constexpr int kStepK = 64;  // k elements per step (128 bytes of 16-bit)

template <int X_DT>
__device__ __forceinline__ void mma16(f32x4& acc, const Frag& fa, const Frag& fb) {
  if constexpr (X_DT == CONCH_DT_FP16) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa.lo), __builtin_bit_cast(f16x8, fb.lo), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa.hi), __builtin_bit_cast(f16x8, fb.hi), acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa.lo), __builtin_bit_cast(bf16x8, fb.lo), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa.hi), __builtin_bit_cast(bf16x8, fb.hi), acc, 0, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------
// in-register dequantisation of one 16-byte LDS chunk (8 consecutive k of one column)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pk_sub_mul_f16(uint32_t v, uint32_t sub, uint32_t scale) {
  f16x2 x = __builtin_bit_cast(f16x2, v);
  x = x - __builtin_bit_cast(f16x2, sub);    // exact: (1024 + q) - (1024 + bias + zp)
  x = x * __builtin_bit_cast(f16x2, scale);  // the single rounding
  return __builtin_bit_cast(uint32_t, x);
}

// 8 nibbles / 2x4 bytes -> 8 halves in k order.  `off` = weight_bias + zero point (an integer),
// `scale_bits` = the group scale's 16-bit pattern.
template <int X_DT, int BITS>
__device__ __forceinline__ i32x4 dequant_chunk(uint32_t w0, uint32_t w1, int off, uint32_t scale_bits) {
  i32x4 out;
  if constexpr (X_DT == CONCH_DT_FP16) {
    constexpr uint32_t kMagic = 0x64006400u;  // fp16 1024.0 in both halves: 1024 + q has q in the mantissa LSBs
    const uint32_t sub1 = (uint32_t)float_to_half_bits((float)(1024 + off));
    const uint32_t sub = sub1 | (sub1 << 16);
    const uint32_t sc = scale_bits | (scale_bits << 16);
    if constexpr (BITS == 4) {
      const uint32_t a = pk_sub_mul_f16((w0 & 0x000f000fu) | kMagic, sub, sc);          // k0, k4
      const uint32_t b = pk_sub_mul_f16(((w0 >> 4) & 0x000f000fu) | kMagic, sub, sc);   // k1, k5
      const uint32_t c = pk_sub_mul_f16(((w0 >> 8) & 0x000f000fu) | kMagic, sub, sc);   // k2, k6
      const uint32_t d = pk_sub_mul_f16(((w0 >> 12) & 0x000f000fu) | kMagic, sub, sc);  // k3, k7
      out[0] = (int)((a & 0xffffu) | (b << 16));
      out[1] = (int)((c & 0xffffu) | (d << 16));
      out[2] = (int)((a >> 16) | (b & 0xffff0000u));
      out[3] = (int)((c >> 16) | (d & 0xffff0000u));
    } else {
      const uint32_t e0 = pk_sub_mul_f16((w0 & 0x00ff00ffu) | kMagic, sub, sc);         // k0, k2
      const uint32_t f0 = pk_sub_mul_f16(((w0 >> 8) & 0x00ff00ffu) | kMagic, sub, sc);  // k1, k3
      const uint32_t e1 = pk_sub_mul_f16((w1 & 0x00ff00ffu) | kMagic, sub, sc);         // k4, k6
      const uint32_t f1 = pk_sub_mul_f16(((w1 >> 8) & 0x00ff00ffu) | kMagic, sub, sc);  // k5, k7
      out[0] = (int)((e0 & 0xffffu) | (f0 << 16));
      out[1] = (int)((e0 >> 16) | (f0 & 0xffff0000u));
      out[2] = (int)((e1 & 0xffffu) | (f1 << 16));
      out[3] = (int)((e1 >> 16) | (f1 & 0xffff0000u));
    }
  } else {
    const float s = bf16_bits_to_float((uint16_t)scale_bits);
    uint16_t h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int q;
      if constexpr (BITS == 4) q = (int)((w0 >> (4 * j)) & 0xfu);
      else q = (int)(((j < 4 ? w0 : w1) >> (8 * (j & 3))) & 0xffu);
      // (q - off) is exact in fp32 and in bf16; the fp32 product is exact, so the cast is the one rounding
      h[j] = float_to_bf16_bits((float)(q - off) * s);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = (int)((uint32_t)h[2 * j] | ((uint32_t)h[2 * j + 1] << 16));
  }
  return out;
}

// Weight words, scale and zero-point of one thread for one K step: the thread owns unit row
// rho = tid & 127 of V1 (column n1) and of V2 (column n2 = n1 + 32) and chunks cp, cp + 4.
template <int BITS>
struct WeightRegs {
  static constexpr int kWordsPerChunk = BITS == 4 ? 1 : 2;
  uint32_t w[2][2][kWordsPerChunk];  // [unit V1/V2][chunk cp / cp+4][word]
  uint32_t scale[2];
  int zp[2];
};

template <int BITS, int ZP>
__device__ __forceinline__ void load_weights(WeightRegs<BITS>& r, const MixedGemmArgs& p, int step, int n1, int n2,
                                             int cp, bool v2_valid) {
  constexpr int kWordRowsPerStep = kStepK * BITS / 32;  // 8 (int4) or 16 (int8)
  constexpr int kWpc = WeightRegs<BITS>::kWordsPerChunk;
  const int k0 = step * kStepK;
  const int64_t g = k0 / p.group_size;
  const int ncol[2] = {n1, n2};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (u == 1 && !v2_valid) continue;  // this thread's V2 row is outside a narrow tile
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int chunk = cp + 4 * c;
#pragma unroll
      for (int i = 0; i < kWpc; ++i) {
        const int64_t row = (int64_t)step * kWordRowsPerStep + chunk * kWpc + i;
        r.w[u][c][i] = (uint32_t)p.w_q[row * p.wq_stride_k + ncol[u]];
      }
    }
    r.scale[u] = ((const uint16_t*)p.w_s)[g * p.ws_stride_g + ncol[u]];
    if constexpr (ZP == CONCH_ZP_TENSOR) r.zp[u] = p.w_zp[g * p.wzp_stride_g + ncol[u]];
    else r.zp[u] = 0;
  }
}

template <int X_DT, int BITS>
__device__ __forceinline__ void convert_weights(const WeightRegs<BITS>& r, char* lds, int buf, int row_off, int swz,
                                                int cp, int off_base, bool v2_valid) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (u == 1 && !v2_valid) continue;
    const int kind = u == 0 ? kV1 : kV2;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int chunk = cp + 4 * c;
      const uint32_t w1 = WeightRegs<BITS>::kWordsPerChunk == 2 ? r.w[u][c][WeightRegs<BITS>::kWordsPerChunk - 1] : 0u;
      const i32x4 v = dequant_chunk<X_DT, BITS>(r.w[u][c][0], w1, off_base + r.zp[u], r.scale[u]);
      *(i32x4*)(lds + buf + kind * kUnitBytes + row_off + ((chunk ^ swz) * 16)) = v;
    }
  }
}

struct MixedTile {
  f32x4 acc[8][4];
  Frag fm[4];
  Frag fn[2][2];
};

// NT = 16-column MFMA tiles per wave (tile width 64*NT): tiles 0,1 form the first n sub-half, tiles
// 2..NT-1 the second.
template <int X_DT, int PHASE, int NT>
__device__ __forceinline__ void mixed_phase(MixedTile& w, const char* lds, int buf, int m_base, int n_base) {
  constexpr int N1 = NT - 2;  // tiles in the second sub-half
  if constexpr (PHASE == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) w.fn[0][t] = read_frag(lds, buf + kV1 * kUnitBytes + n_base + t * 2048);
#pragma unroll
    for (int i = 0; i < 4; ++i) w.fm[i] = read_frag(lds, buf + kU1 * kUnitBytes + m_base + i * 2048);
  } else if constexpr (PHASE == 1) {
#pragma unroll
    for (int t = 0; t < N1; ++t) w.fn[1][t] = read_frag(lds, buf + kV2 * kUnitBytes + n_base + t * 2048);
  } else if constexpr (PHASE == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) w.fm[i] = read_frag(lds, buf + kU2 * kUnitBytes + m_base + i * 2048);
  }
  constexpr int MH = (PHASE >= 2) ? 1 : 0;
  constexpr int NH = (PHASE == 1 || PHASE == 2) ? 1 : 0;
  constexpr int CNT = NH == 0 ? 2 : N1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < CNT; ++t) mma16<X_DT>(w.acc[MH * 4 + i][NH * 2 + t], w.fn[NH][t], w.fm[i]);
}

template <int X_DT, int OUT_DT, int BITS, int ZP, int NT>
__global__ __launch_bounds__(kThreads, 2) void mixed_gemm_kernel(MixedGemmArgs p) {
  constexpr int kTileW = 64 * NT;  // columns per workgroup
  __shared__ __attribute__((aligned(1024))) char lds[kLdsBytes];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_m = ((int)p.m + kTileM - 1) / kTileM;
  const int tiles_n = ((int)p.n + kTileW - 1) / kTileW;
  const TileCoord tc = map_tile(blockIdx.x, tiles_m, tiles_n);
  const int bm0 = tc.tm * kTileM, bn0 = tc.tn * kTileW;

  // activations: LDS-DMA units (byte strides: 2 bytes per element)
  const int ldx = (int)p.x_stride_m * 2;
  const StageOffsets so = make_stage_offsets(wave, lane, bm0, bn0, (int)p.m - 1, 0, ldx, 0);
  Srcs src;
  src.a = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (uint32_t)(((p.m - 1) * p.x_stride_m + p.k) * 2), 0x00020000);
  src.b = src.a;

  // weights: this thread's unit row / columns / chunk pair
  const int rho = threadIdx.x & 127;
  const int cp = threadIdx.x >> 7;
  const int r5 = rho & 31, r = r5 & 15, tq = r5 >> 4;
  // wave-column rho>>5 owns columns [16*NT*wc, 16*NT*(wc+1)); a PAIR of MFMA tiles is interleaved (rows 4g+e
  // of tile t <-> n = 8g+e+4t: a lane ends up with 8 consecutive n), a LONE third tile (NT == 3) is plain
  const int wcol = (rho >> 5) * 16 * NT;
  const int pair = 8 * (r >> 2) + (r & 3) + 4 * tq;
  const int n1 = min(bn0 + wcol + pair, (int)p.n - 1);
  const int n2 = min(bn0 + wcol + 32 + (NT == 4 ? pair : r), (int)p.n - 1);
  const bool v2_valid = NT == 4 || (NT == 3 && tq == 0);
  const int w_row_off = (rho >> 3) * 1024 + (rho & 7) * 128;
  const int w_swz = (rho >> 1) & 7;
  const int off_base = p.weight_bias + (ZP == CONCH_ZP_SCALAR ? p.w_zp[0] : 0);

  // fragment read offsets
  const int fr = lane & 15, fg = lane >> 4;
  const int lane_off = (fr >> 3) * 1024 + (fr & 7) * 128 + ((fg ^ ((fr >> 1) & 7)) * 16);
  const int m_base = (wr * 8) * 1024 + lane_off;
  const int n_base = (wc * 4) * 1024 + lane_off;

  MixedTile w;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) w.acc[i][j] = f32x4{0, 0, 0, 0};

  const int steps = (int)(p.k / kStepK);
  WeightRegs<BITS> regs;

  // prologue: step 0 operands into buffer 0, step 1 weights into registers
  stage_unit<kU1>(lds, src, so, wave, 0);
  stage_unit<kU2>(lds, src, so, wave, 0);
  load_weights<BITS, ZP>(regs, p, 0, n1, n2, cp, v2_valid);
  convert_weights<X_DT, BITS>(regs, lds, 0, w_row_off, w_swz, cp, off_base, v2_valid);
  if (steps > 1) load_weights<BITS, ZP>(regs, p, 1, n1, n2, cp, v2_valid);

  for (int t = 0; t < steps; ++t) {
    // everything this wave staged for step t (LDS-DMA and ds_write) is complete ...
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // ... and every wave is done reading the other buffer
    const int buf = (t & 1) * kBufBytes;
    const int nbuf = ((t + 1) & 1) * kBufBytes;
    if (t + 1 < steps) {
      stage_unit<kU1>(lds, src, so, wave, t + 1);
      stage_unit<kU2>(lds, src, so, wave, t + 1);
      convert_weights<X_DT, BITS>(regs, lds, nbuf, w_row_off, w_swz, cp, off_base, v2_valid);
      if (t + 2 < steps) load_weights<BITS, ZP>(regs, p, t + 2, n1, n2, cp, v2_valid);
    }
    mixed_phase<X_DT, 0, NT>(w, lds, buf, m_base, n_base);
    mixed_phase<X_DT, 1, NT>(w, lds, buf, m_base, n_base);
    mixed_phase<X_DT, 2, NT>(w, lds, buf, m_base, n_base);
    mixed_phase<X_DT, 3, NT>(w, lds, buf, m_base, n_base);
  }

  // epilogue: cast and store; a pair of tiles gives a lane 8 consecutive n (16-byte store), a lone tile 4
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
      uint16_t o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = float_to_bits16<OUT_DT>(pin_f32(w.acc[mt][nh * 2 + (pair_h ? (e >> 2) : 0)][e & 3]));
      uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n0;
      if (pair_h && vec_store && n0 + 8 <= p.n) {
        i32x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = (int)((uint32_t)o[2 * e] | ((uint32_t)o[2 * e + 1] << 16));
        *(i32x4*)dst = pk;
      } else if (!pair_h && vec_store && n0 + 4 <= p.n) {
        i32x2 pk;
        pk[0] = (int)((uint32_t)o[0] | ((uint32_t)o[1] << 16));
        pk[1] = (int)((uint32_t)o[2] | ((uint32_t)o[3] << 16));
        *(i32x2*)dst = pk;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < width && n0 + e < p.n) dst[e] = o[e];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Ping-pong form (variant 3, the default): the K loop of gemm_mfma.hip's variant 3 -- 4 phases per K
// step, {load segment} barrier {MFMA cluster} barrier, waves 4-7 one barrier behind waves 0-3 -- with
// the weight units produced in the load segments: the packed words of V unit s are loaded to
// registers in phase s-6, dequantised and written to LDS in phase s-3, read as fragments in phase
// s-1.  A wave's dequantisation VALU work therefore overlaps the MFMA cluster of the other wave on
// its SIMD instead of both waves converting, then both multiplying.
// Counted waits: of any four consecutive units two are LDS-DMA (2 VMEM ops each per wave) and two
// are register loads (NV1 / NV2 ops), so "all but the 4 youngest units" = vmcnt(4 + NV1 + NV2).
// ---------------------------------------------------------------------------------------------
template <int BITS>
struct UnitRegs {
  static constexpr int kWpc = BITS == 4 ? 1 : 2;
  uint32_t w[2][kWpc];  // [chunk cp / cp+4][word]
  uint32_t scale;
  int zp;
};

template <int BITS, int ZP>
constexpr int unit_vmem_ops() { return 2 * UnitRegs<BITS>::kWpc + 1 + (ZP == CONCH_ZP_TENSOR ? 1 : 0); }

template <int BITS, int ZP>
__device__ __forceinline__ void load_unit(UnitRegs<BITS>& r, const MixedGemmArgs& p, int step, int ncol, int cp,
                                          bool valid) {
  constexpr int kWordRowsPerStep = kStepK * BITS / 32;
  constexpr int kWpc = UnitRegs<BITS>::kWpc;
  if (!valid) return;
  const int64_t g = (step * kStepK) / p.group_size;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < kWpc; ++i) {
      const int64_t row = (int64_t)step * kWordRowsPerStep + (cp + 4 * c) * kWpc + i;
      r.w[c][i] = (uint32_t)p.w_q[row * p.wq_stride_k + ncol];
    }
  r.scale = ((const uint16_t*)p.w_s)[g * p.ws_stride_g + ncol];
  if constexpr (ZP == CONCH_ZP_TENSOR) r.zp = p.w_zp[g * p.wzp_stride_g + ncol];
  else r.zp = 0;
}

template <int X_DT, int BITS, int KIND>
__device__ __forceinline__ void convert_unit(const UnitRegs<BITS>& r, char* lds, int buf, int row_off, int swz, int cp,
                                             int off_base, bool valid) {
  if (!valid) return;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const uint32_t w1 = UnitRegs<BITS>::kWpc == 2 ? r.w[c][UnitRegs<BITS>::kWpc - 1] : 0u;
    const i32x4 v = dequant_chunk<X_DT, BITS>(r.w[c][0], w1, off_base + r.zp, r.scale);
    *(i32x4*)(lds + buf + KIND * kUnitBytes + row_off + (((cp + 4 * c) ^ swz) * 16)) = v;
  }
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  if constexpr (N >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct MixedCtx {
  int wave, wr, m_base, n_base;
  StageOffsets so;
  Srcs src;
  int n1, n2, cp, w_row_off, w_swz, off_base;
  bool v2_valid;
};

template <int X_DT, int PHASE, int NT>
__device__ __forceinline__ void mixed_reads(MixedTile& w, const char* lds, int buf, int m_base, int n_base) {
  if constexpr (PHASE == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) w.fn[0][t] = read_frag(lds, buf + kV1 * kUnitBytes + n_base + t * 2048);
    __builtin_amdgcn_sched_barrier(0);
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

template <int X_DT, int PHASE, int NT>
__device__ __forceinline__ void mixed_cluster(MixedTile& w) {
  constexpr int MH = (PHASE >= 2) ? 1 : 0;
  constexpr int NH = (PHASE == 1 || PHASE == 2) ? 1 : 0;
  constexpr int CNT = NH == 0 ? 2 : NT - 2;
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < CNT; ++t) mma16<X_DT>(w.acc[MH * 4 + i][NH * 2 + t], w.fn[NH][t], w.fm[i]);
  __builtin_amdgcn_s_setprio(0);
  // this wave's ds_writes (dequantised weights) must have landed before the barrier that publishes them
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// MODE 0: steady (steps t+1 and t+2 exist); 1: penultimate (t+1 exists); 2: last.
template <int X_DT, int BITS, int ZP, int NT, int MODE>
__device__ __forceinline__ void mixed_pp_step(MixedTile& w, UnitRegs<BITS>& rv1, UnitRegs<BITS>& rv2, char* lds,
                                              const MixedGemmArgs& p, const MixedCtx& c, int t) {
  constexpr int NV1 = unit_vmem_ops<BITS, ZP>();
  constexpr int NV2 = NT == 2 ? 0 : NV1;
  constexpr int kSteadyVm = 4 + NV1 + NV2;
  constexpr bool NEXT = MODE <= 1;    // K step t+1 exists
  constexpr bool NEXT2 = MODE == 0;   // K step t+2 exists
  const int buf = (t & 1) * kBufBytes, nbuf = ((t + 1) & 1) * kBufBytes;
  // ---- phase 0: reads V1,U1; issue V2(t+1) words
  mixed_reads<X_DT, 0, NT>(w, lds, buf, c.m_base, c.n_base);
  if constexpr (NEXT && NT > 2) load_unit<BITS, ZP>(rv2, p, t + 1, c.n2, c.cp, c.v2_valid);
  wait_vm<MODE <= 1 ? kSteadyVm : 2>();
  mixed_cluster<X_DT, 0, NT>(w);
  // ---- phase 1: reads V2; issue U2(t+1) DMA
  mixed_reads<X_DT, 1, NT>(w, lds, buf, c.m_base, c.n_base);
  if constexpr (NEXT) stage_unit<kU2>(lds, c.src, c.so, c.wave, t + 1);
  wait_vm<MODE <= 1 ? kSteadyVm : 0>();
  mixed_cluster<X_DT, 1, NT>(w);
  // ---- phase 2: reads U2; issue U1(t+2) DMA; dequantise V1(t+1) (loaded three phases ago)
  mixed_reads<X_DT, 2, NT>(w, lds, buf, c.m_base, c.n_base);
  if constexpr (NEXT2) stage_unit<kU1>(lds, c.src, c.so, c.wave, t + 2);
  if constexpr (NEXT) convert_unit<X_DT, BITS, kV1>(rv1, lds, nbuf, c.w_row_off, c.w_swz, c.cp, c.off_base, true);
  wait_vm<MODE == 0 ? kSteadyVm : (MODE == 1 ? NV1 + NV2 + 2 : -1)>();
  mixed_cluster<X_DT, 2, NT>(w);
  // ---- phase 3: issue V1(t+2) words; dequantise V2(t+1)
  if constexpr (NEXT && NT > 2) convert_unit<X_DT, BITS, kV2>(rv2, lds, nbuf, c.w_row_off, c.w_swz, c.cp, c.off_base, c.v2_valid);
  if constexpr (NEXT2) load_unit<BITS, ZP>(rv1, p, t + 2, c.n1, c.cp, true);
  wait_vm<MODE == 0 ? kSteadyVm : (MODE == 1 ? NV2 + 2 : -1)>();
  mixed_cluster<X_DT, 3, NT>(w);
}

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
      uint16_t o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = float_to_bits16<OUT_DT>(pin_f32(w.acc[mt][nh * 2 + (pair_h ? (e >> 2) : 0)][e & 3]));
      uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n0;
      if (pair_h && vec_store && n0 + 8 <= p.n) {
        i32x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = (int)((uint32_t)o[2 * e] | ((uint32_t)o[2 * e + 1] << 16));
        *(i32x4*)dst = pk;
      } else if (!pair_h && vec_store && n0 + 4 <= p.n) {
        i32x2 pk;
        pk[0] = (int)((uint32_t)o[0] | ((uint32_t)o[1] << 16));
        pk[1] = (int)((uint32_t)o[2] | ((uint32_t)o[3] << 16));
        *(i32x2*)dst = pk;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < width && n0 + e < p.n) dst[e] = o[e];
      }
    }
  }
}

template <int X_DT, int OUT_DT, int BITS, int ZP, int NT>
__global__ __launch_bounds__(kThreads, 2) void mixed_gemm_pp_kernel(MixedGemmArgs p) {
  constexpr int kTileW = 64 * NT;
  constexpr int NV1 = unit_vmem_ops<BITS, ZP>();
  __shared__ __attribute__((aligned(1024))) char lds[kLdsBytes];
  MixedCtx c;
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  c.wr = c.wave >> 2;
  const int wc = c.wave & 3;
  const int tiles_m = ((int)p.m + kTileM - 1) / kTileM;
  const int tiles_n = ((int)p.n + kTileW - 1) / kTileW;
  const TileCoord tc = map_tile(blockIdx.x, tiles_m, tiles_n);
  const int bm0 = tc.tm * kTileM, bn0 = tc.tn * kTileW;
  c.so = make_stage_offsets(c.wave, lane, bm0, bn0, (int)p.m - 1, 0, (int)p.x_stride_m * 2, 0);
  c.src.a = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (uint32_t)(((p.m - 1) * p.x_stride_m + p.k) * 2), 0x00020000);
  c.src.b = c.src.a;
  const int rho = threadIdx.x & 127;
  c.cp = threadIdx.x >> 7;
  const int r5 = rho & 31, r = r5 & 15, tq = r5 >> 4;
  const int wcol = (rho >> 5) * 16 * NT;
  const int pair = 8 * (r >> 2) + (r & 3) + 4 * tq;
  c.n1 = min(bn0 + wcol + pair, (int)p.n - 1);
  c.n2 = min(bn0 + wcol + 32 + (NT == 4 ? pair : r), (int)p.n - 1);
  c.v2_valid = NT == 4 || (NT == 3 && tq == 0);
  c.w_row_off = (rho >> 3) * 1024 + (rho & 7) * 128;
  c.w_swz = (rho >> 1) & 7;
  c.off_base = p.weight_bias + (ZP == CONCH_ZP_SCALAR ? p.w_zp[0] : 0);
  const int fr = lane & 15, fg = lane >> 4;
  const int lane_off = (fr >> 3) * 1024 + (fr & 7) * 128 + ((fg ^ ((fr >> 1) & 7)) * 16);
  c.m_base = (c.wr * 8) * 1024 + lane_off;
  c.n_base = (wc * 4) * 1024 + lane_off;

  MixedTile w;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) w.acc[i][j] = f32x4{0, 0, 0, 0};
  const int steps = (int)(p.k / kStepK);  // >= 2 (dispatcher)
  UnitRegs<BITS> rv1, rv2;

  // prologue: units 0..5 = U1,V1,V2,U2 of step 0 (weights dequantised here) and U1, V1-words of step 1
  stage_unit<kU1>(lds, c.src, c.so, c.wave, 0);
  stage_unit<kU2>(lds, c.src, c.so, c.wave, 0);
  load_unit<BITS, ZP>(rv1, p, 0, c.n1, c.cp, true);
  if constexpr (NT > 2) load_unit<BITS, ZP>(rv2, p, 0, c.n2, c.cp, c.v2_valid);
  stage_unit<kU1>(lds, c.src, c.so, c.wave, 1);
  convert_unit<X_DT, BITS, kV1>(rv1, lds, 0, c.w_row_off, c.w_swz, c.cp, c.off_base, true);
  if constexpr (NT > 2) convert_unit<X_DT, BITS, kV2>(rv2, lds, 0, c.w_row_off, c.w_swz, c.cp, c.off_base, c.v2_valid);
  load_unit<BITS, ZP>(rv1, p, 1, c.n1, c.cp, true);
  wait_vm<NV1>();  // everything but the step-1 weight words (the youngest NV1 ops) has landed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (c.wr == 1) __builtin_amdgcn_s_barrier();  // stagger the second wave of every SIMD

  int t = 0;
  for (; t + 2 < steps; ++t) mixed_pp_step<X_DT, BITS, ZP, NT, 0>(w, rv1, rv2, lds, p, c, t);
  mixed_pp_step<X_DT, BITS, ZP, NT, 1>(w, rv1, rv2, lds, p, c, t);
  mixed_pp_step<X_DT, BITS, ZP, NT, 2>(w, rv1, rv2, lds, p, c, t + 1);
  if (c.wr == 0) __builtin_amdgcn_s_barrier();  // re-balance the barrier count

  mixed_epilogue<X_DT, OUT_DT, NT>(w, p, bm0, bn0, c.wr, wc, lane);
}

template <int X_DT, int OUT_DT, int BITS, int NT>
int launch_zp(const MixedGemmArgs& p, bool pingpong, hipStream_t stream) {
  const int tiles_m = (int)((p.m + kTileM - 1) / kTileM);
  const int tiles_n = (int)((p.n + 64 * NT - 1) / (64 * NT));
  const dim3 grid((unsigned)(tiles_m * tiles_n));
  if (pingpong) {
    switch (p.zp_mode) {
      case CONCH_ZP_NONE:
        hipLaunchKernelGGL((mixed_gemm_pp_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_NONE, NT>), grid, dim3(kThreads), 0, stream, p);
        break;
      case CONCH_ZP_SCALAR:
        hipLaunchKernelGGL((mixed_gemm_pp_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_SCALAR, NT>), grid, dim3(kThreads), 0, stream, p);
        break;
      default:
        hipLaunchKernelGGL((mixed_gemm_pp_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_TENSOR, NT>), grid, dim3(kThreads), 0, stream, p);
        break;
    }
    return check_launch("mixed_gemm_pingpong");
  }
  switch (p.zp_mode) {
    case CONCH_ZP_NONE:
      hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_NONE, NT>), grid, dim3(kThreads), 0, stream, p);
      break;
    case CONCH_ZP_SCALAR:
      hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_SCALAR, NT>), grid, dim3(kThreads), 0, stream, p);
      break;
    default:
      hipLaunchKernelGGL((mixed_gemm_kernel<X_DT, OUT_DT, BITS, CONCH_ZP_TENSOR, NT>), grid, dim3(kThreads), 0, stream, p);
      break;
  }
  return check_launch("mixed_gemm_mfma");
}

// Tile width 64*NT, NT in {4, 3, 2}: the width that needs the least (rounds of workgroups) x (work per
// workgroup).  C4 (1024 x 11008): 256-wide = 172 tiles (67 % of 256 CUs busy), 192-wide = 232 tiles in
// ONE round at 3/4 of the work each.
int pick_nt(const MixedGemmArgs& p, int num_cus) {
  const int64_t tiles_m = (p.m + kTileM - 1) / kTileM;
  int best = 4;
  double best_cost = 1e30;
  for (int nt = 4; nt >= 2; --nt) {
    const int64_t tiles = tiles_m * ((p.n + 64 * nt - 1) / (64 * nt));
    const int64_t rounds = (tiles + num_cus - 1) / num_cus;
    const double cost = (double)rounds * (nt + 0.35);  // + fixed per-tile cost: narrower tiles re-stage A more often
    if (cost < best_cost - 1e-9) {
      best_cost = cost;
      best = nt;
    }
  }
  return best;
}

template <int X_DT, int OUT_DT, int BITS>
int launch_nt(const MixedGemmArgs& p, int nt, bool pp, hipStream_t stream) {
  if (nt == 3) return launch_zp<X_DT, OUT_DT, BITS, 3>(p, pp, stream);
  if (nt == 2) return launch_zp<X_DT, OUT_DT, BITS, 2>(p, pp, stream);
  return launch_zp<X_DT, OUT_DT, BITS, 4>(p, pp, stream);
}

template <int X_DT, int OUT_DT>
int launch_bits(const MixedGemmArgs& p, int nt, bool pp, hipStream_t stream) {
  return p.bits == 4 ? launch_nt<X_DT, OUT_DT, 4>(p, nt, pp, stream) : launch_nt<X_DT, OUT_DT, 8>(p, nt, pp, stream);
}

}  // namespace

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
  return true;
}

int launch_mixed_gemm_mfma(const MixedGemmArgs& p, int variant, hipStream_t stream) {
  static int num_cus = 0;
  if (num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    num_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
               prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const int forced = tuning(1);  // CONCH_TUNE_MIXED_TILE_NT: 0 = auto, 2..4 = force
  const int nt = (forced >= 2 && forced <= 4) ? forced : pick_nt(p, num_cus);
  // auto = the plain one-barrier-per-K-step loop: measured equal or faster than the ping-pong form on
  // every shape tried (C4 106 vs 113 us, README shape 252 vs 247 us); variant 3 forces ping-pong
  const bool pp = variant == 3 && p.k >= 2 * kStepK;
  if (p.x_dtype == CONCH_DT_FP16) {
    return p.out_dtype == CONCH_DT_FP16 ? launch_bits<CONCH_DT_FP16, CONCH_DT_FP16>(p, nt, pp, stream)
                                        : launch_bits<CONCH_DT_FP16, CONCH_DT_BF16>(p, nt, pp, stream);
  }
  return p.out_dtype == CONCH_DT_BF16 ? launch_bits<CONCH_DT_BF16, CONCH_DT_BF16>(p, nt, pp, stream)
                                      : launch_bits<CONCH_DT_BF16, CONCH_DT_FP16>(p, nt, pp, stream);
}

}  // namespace conch
