// In-register dequantisation of packed int4 / int8 weights, shared by the mixed-precision GEMM kernels
// (gemm_mixed.hip: one slice after each MFMA; gemm_mixed_skinny.hip: a whole chunk at a time).
//
// One 16-byte chunk = 8 consecutive k of one column, cut into slices of two to four VALU instructions.  The packed pairs
// are a = (k0,k4) b = (k1,k5) c = (k2,k6) d = (k3,k7) for int4 and a = (k0,k2) b = (k1,k3) c = (k4,k6) d = (k5,k7) for
// int8, each as the fp16 numbers 1024 + q (0x6400 | q); (1024 + q) - (1024 + bias + zp) is exact, and the product with
// the group scale is the one rounding: v_pk_mul_f16 for fp16, an exact fp32 product + v_cvt_pk_bf16_f32 for bf16 --
// bit-identical to the reference's w_ref (third_party/vllm/quant_utils.py:74, kernels/quantization/gemm.py:192-210).
#pragma once

#include "common.hpp"

namespace conch {
namespace mixed {

// ROUND_DIFF: round (q - bias - zp) to bf16 before the scale multiply.  The reference forms the difference IN the meta dtype
// (kernels/quantization/gemm.py:205-210; w_ref likewise casts the integer difference first, quant_utils.py:74): with 8-bit
// weights, a bias and a zero point it can reach 383 in magnitude, which bf16 (8 significant bits) rounds -- fp16 never has to.
// PRE: the chunk comes from a tensor laid out by conch_prepack_mixed_weights -- int4: nibble j holds k 2j and nibble j + 4 holds
// k 2j + 1 (int8: bytes k0, k2, k1, k3), so that ((w >> 4j) & mask) | magic IS output dword j = (k 2j, k 2j + 1): the four
// v_perm_b32 of the fp16 path and the cross pairing of the bf16 path disappear (19 -> 15 VALU per eight int4 weights).
template <int X_DT, int BITS, bool ROUND_DIFF = false, bool PRE = false, bool SHIFTLESS = true>
struct ChunkDequant {
  static constexpr bool kShiftless = SHIFTLESS && BITS == 4;  // see slice(): the upper pairs are taken without a shift
  static constexpr bool kHalf = X_DT == CONCH_DT_FP16;
  static constexpr int kSlices = 10;  // the last one is the ds_write_b128
  uint32_t w0, w1, sub, nsub16, sc;
  uint32_t a, b, c, d;
  f32x2 p_lo, p_hi, q_lo, q_hi;  // bf16 path: (a.lo, b.lo), (a.hi, b.hi), (c.lo, d.lo), (c.hi, d.hi) widened to fp32
  float fs;
  i32x4 out;

  static __device__ __forceinline__ uint32_t pk_sub(uint32_t v, uint32_t s) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, v) - __builtin_bit_cast(f16x2, s));
  }
  // v / 16 + c, one rounding (the operands used here make the result exact)
  static __device__ __forceinline__ uint32_t pk_fma16(uint32_t v, uint32_t c) {
    const f16x2 k = {(_Float16)0.0625f, (_Float16)0.0625f};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(f16x2, v), k, __builtin_bit_cast(f16x2, c)));
  }
  static __device__ __forceinline__ uint32_t pk_mul(uint32_t v, uint32_t s) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, v) * __builtin_bit_cast(f16x2, s));
  }
  // the fp32 pairs are formed the way they are packed again: (element of a, element of b) -> one output dword, so the
  // scale multiply is a v_pk_mul_f32 and the rounding ONE v_cvt_pk_bf16_f32 per dword
  static __device__ __forceinline__ void widen(uint32_t x, uint32_t y, f32x2& lo, f32x2& hi) {
    const f16x2 hx = __builtin_bit_cast(f16x2, x), hy = __builtin_bit_cast(f16x2, y);
    lo = f32x2{(float)hx[0], (float)hy[0]};
    hi = f32x2{(float)hx[1], (float)hy[1]};
    if constexpr (ROUND_DIFF) {
      lo = unpack2_bits16<CONCH_DT_BF16>(pack2_bits16<CONCH_DT_BF16>(lo));
      hi = unpack2_bits16<CONCH_DT_BF16>(pack2_bits16<CONCH_DT_BF16>(hi));
    }
  }
  static __device__ __forceinline__ f32x2 widen1(uint32_t x) {
    const f16x2 hx = __builtin_bit_cast(f16x2, x);
    f32x2 v = f32x2{(float)hx[0], (float)hx[1]};
    if constexpr (ROUND_DIFF) v = unpack2_bits16<CONCH_DT_BF16>(pack2_bits16<CONCH_DT_BF16>(v));
    return v;
  }
  // output dword j of the chunk is the pair (k 2j, k 2j+1)
  static constexpr int kOutLoP = 0, kOutHiP = PRE ? 1 : (BITS == 4 ? 2 : 1), kOutLoQ = PRE ? 2 : (BITS == 4 ? 1 : 2), kOutHiQ = 3;

  // int4: the pairs that sit four bits up (b, d) are NOT shifted down: (w & (mask << 4)) | magic is the fp16 number 1024 + 16 q, and
  // (1024 + 16 q) / 16 - (64 + bias + zp) = q - bias - zp in ONE v_pk_fma_f16 (every intermediate exact) -- the same count as the
  // v_pk_add_f16 it replaces, two shifts fewer per eight weights.  (SHIFTLESS = false keeps the shifts: two VGPRs fewer, for the
  // fused gate/up kernel, whose epilogue would otherwise spill.)
  // `mask` (a VGPR) and `magic` (an SGPR) hold 0x000f000f / 0x00ff00ff and 0x64006400 as VALUES the compiler cannot see:
  // with literals it emits v_and_b32 + v_or_b32 (a VOP3 instruction cannot carry a 32-bit literal on gfx9); with
  // registers (x & mask) | magic is ONE v_and_or_b32 -- 16 fewer VALU instructions per K step and thread
  __device__ __forceinline__ void slice(int s, uint32_t word0, uint32_t word1, int off, uint32_t scale_bits, char* dst,
                                        uint32_t mask, uint32_t magic, uint32_t mask_hi = 0) {
    constexpr uint32_t kLowHalves = 0x05040100u, kHighHalves = 0x07060302u;
    constexpr int kShift = BITS == 4 ? 4 : 8;
    if (s == 0) {
      w0 = word0;
      w1 = BITS == 4 ? (word0 >> 8) : word1;
      const uint32_t sub1 = (uint32_t)float_to_half_bits((float)(1024 + off));
      sub = sub1 | (sub1 << 16);
      if constexpr (kShiftless) nsub16 = pk_sub(0x63806380u, sub);  // 960 - (1024 + off) = -(64 + off)
      if constexpr (kHalf) sc = scale_bits | (scale_bits << 16);
      else fs = bf16_bits_to_float((uint16_t)scale_bits);
    } else if (s == 1) {
      a = (w0 & mask) | magic;
      if constexpr (kShiftless) b = (w0 & mask_hi) | magic;
      else b = ((w0 >> kShift) & mask) | magic;
    } else if (s == 2) {
      c = (w1 & mask) | magic;
      if constexpr (kShiftless) d = (w1 & mask_hi) | magic;
      else d = ((w1 >> kShift) & mask) | magic;
    } else if (s == 3) {
      a = pk_sub(a, sub);
      b = kShiftless ? pk_fma16(b, nsub16) : pk_sub(b, sub);
    } else if (s == 4) {
      c = pk_sub(c, sub);
      d = kShiftless ? pk_fma16(d, nsub16) : pk_sub(d, sub);
    } else if constexpr (kHalf) {
      if (s == 5) {
        a = pk_mul(a, sc);
        b = pk_mul(b, sc);
      } else if (s == 6) {
        c = pk_mul(c, sc);
        d = pk_mul(d, sc);
      } else if (s == 7) {
        if constexpr (PRE) {
          out[0] = (int)a;
          out[1] = (int)b;
        } else {
          out[kOutLoP] = (int)__builtin_amdgcn_perm(b, a, kLowHalves);
          out[kOutHiP] = (int)__builtin_amdgcn_perm(b, a, kHighHalves);
        }
      } else if (s == 8) {
        if constexpr (PRE) {
          out[2] = (int)c;
          out[3] = (int)d;
        } else {
          out[kOutLoQ] = (int)__builtin_amdgcn_perm(d, c, kLowHalves);
          out[kOutHiQ] = (int)__builtin_amdgcn_perm(d, c, kHighHalves);
        }
      } else if (s == 9) {
        *(i32x4*)dst = out;
      }
    } else {
      if (s == 5) {
        if constexpr (PRE) {
          p_lo = widen1(a);  // -> output dword 0
          p_hi = widen1(b);  // -> 1
        } else {
          widen(a, b, p_lo, p_hi);
        }
      } else if (s == 6) {
        if constexpr (PRE) {
          q_lo = widen1(c);  // -> 2
          q_hi = widen1(d);  // -> 3
        } else {
          widen(c, d, q_lo, q_hi);
        }
      } else if (s == 7) {
        out[kOutLoP] = (int)pack2_bits16<CONCH_DT_BF16>(p_lo * f32x2{fs, fs});
        out[kOutHiP] = (int)pack2_bits16<CONCH_DT_BF16>(p_hi * f32x2{fs, fs});
      } else if (s == 8) {
        out[kOutLoQ] = (int)pack2_bits16<CONCH_DT_BF16>(q_lo * f32x2{fs, fs});
        out[kOutHiQ] = (int)pack2_bits16<CONCH_DT_BF16>(q_hi * f32x2{fs, fs});
      } else if (s == 9) {
        *(i32x4*)dst = out;
      }
    }
  }
};

}  // namespace mixed
}  // namespace conch
