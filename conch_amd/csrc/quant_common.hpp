// Element-wise quantisation arithmetic shared by the static (quant.hip) and dynamic per-token (quant_dynamic.hip) feeders.
#pragma once

#include "common.hpp"

namespace conch {
namespace quant {

constexpr int kQuantThreads = 256;
constexpr int kVec = 16;  // elements per lane per step

enum QuantKind { kInt8 = 0, kFp8Fn = 1, kFp8Fnuz = 2 };

template <int KIND>
__device__ __forceinline__ uint32_t quant_one(float v, float inv) {
  const float p = v * inv;
  if constexpr (KIND == kInt8) {
    // clamp(min=-128,max=127) then C-style truncation, as torch's .to(int8) does.
    // fminf/fmaxf would swallow NaN; torch.clamp propagates it and the cast of NaN is
    // implementation-defined, so NaN inputs are outside the contract.
    const float c = fminf(fmaxf(p, -128.0f), 127.0f);
    return (uint32_t)(int)c & 0xffu;
  } else {
    constexpr bool FNUZ = KIND == kFp8Fnuz;
    const float mx = fp8_max<FNUZ>();
    // NaN-propagating clamp (torch.clamp keeps NaN; the cast then yields the NaN code)
    const float c = (p != p) ? p : fminf(fmaxf(p, -mx), mx);
    if constexpr (FNUZ) {
      return encode_fp8_sw<true>(c);
    } else {
      // gfx950's v_cvt_pk_fp8_f32 is OCP e4m3fn, RNE; the input is already within +-448.
      return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, c, 0, false) & 0xffu;
    }
  }
}

template <int KIND>
__device__ __forceinline__ uint32_t quant_four(float a, float b, float c, float d, float inv) {
  if constexpr (KIND == kFp8Fn) {
    const float mx = 448.0f;
    float v[4] = {a * inv, b * inv, c * inv, d * inv};
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (v[i] != v[i]) ? v[i] : fminf(fmaxf(v[i], -mx), mx);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
    return (uint32_t)w;
  } else {
    return quant_one<KIND>(a, inv) | (quant_one<KIND>(b, inv) << 8) |
           (quant_one<KIND>(c, inv) << 16) | (quant_one<KIND>(d, inv) << 24);
  }
}

// 16 consecutive elements -> 16 floats
template <int XDT>
__device__ __forceinline__ void load16(const void* x, int64_t base, float (&f)[kVec]) {
  if constexpr (XDT == CONCH_DT_FP32) {
    const f32x4* p = (const f32x4*)((const float*)x + base);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 v = __builtin_nontemporal_load(p + j);
#pragma unroll
      for (int i = 0; i < 4; ++i) f[j * 4 + i] = v[i];
    }
  } else {
    const u16x8* p = (const u16x8*)((const uint16_t*)x + base);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const u16x8 v = __builtin_nontemporal_load(p + j);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[j * 8 + i] = bits16_to_float<XDT>(v[i]);
    }
  }
}

// The same 16 elements per thread, laid out so that every load instruction of a wave covers CONTIGUOUS memory: a group of G threads
// works on a chunk of G x 16 elements starting at element e0, thread t takes piece j = elements [e0 + j G P + t P, + P), P = 8 for
// 16-bit inputs (2 pieces, one 16-byte load each), P = 4 for fp32 (4 pieces).  Elements at or beyond `limit` read as 0 and are not
// stored (limit - e0 is a multiple of 16).  Round 4: against 16 consecutive elements per lane (a 32- / 64-byte lane stride per load
// instruction) the static quantiser gained 9-16 % on 16-bit inputs and 33 % on fp32 (profiles/r04/quant_coalesced_ab.txt).
template <int XDT>
constexpr int kPieceElems = XDT == CONCH_DT_FP32 ? 4 : 8;

template <int XDT, int G>
__device__ __forceinline__ void load_pieces(const void* x, int64_t e0, int t, int64_t limit, float (&f)[kVec]) {
  constexpr int P = kPieceElems<XDT>;
#pragma unroll
  for (int j = 0; j < kVec / P; ++j) {
    const int64_t e = e0 + (int64_t)j * (G * P) + (int64_t)t * P;
    if (e < limit) {
      if constexpr (XDT == CONCH_DT_FP32) {
        const f32x4 v = __builtin_nontemporal_load((const f32x4*)((const float*)x + e));
#pragma unroll
        for (int i = 0; i < 4; ++i) f[j * 4 + i] = v[i];
      } else {
        const u16x8 v = __builtin_nontemporal_load((const u16x8*)((const uint16_t*)x + e));
#pragma unroll
        for (int i = 0; i < 8; ++i) f[j * 8 + i] = bits16_to_float<XDT>(v[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < P; ++i) f[j * P + i] = 0.0f;
    }
  }
}

template <int XDT, int KIND, int G>
__device__ __forceinline__ void store_pieces(uint8_t* out, int64_t e0, int t, int64_t limit, const float (&f)[kVec], float mul) {
  constexpr int P = kPieceElems<XDT>;
#pragma unroll
  for (int j = 0; j < kVec / P; ++j) {
    const int64_t e = e0 + (int64_t)j * (G * P) + (int64_t)t * P;
    if (e < limit) {
      if constexpr (P == 4) {
        __builtin_nontemporal_store((int)quant_four<KIND>(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3], mul), (int*)(out + e));
      } else {
        i32x2 o;
        o[0] = (int)quant_four<KIND>(f[8 * j], f[8 * j + 1], f[8 * j + 2], f[8 * j + 3], mul);
        o[1] = (int)quant_four<KIND>(f[8 * j + 4], f[8 * j + 5], f[8 * j + 6], f[8 * j + 7], mul);
        __builtin_nontemporal_store(o, (i32x2*)(out + e));
      }
    }
  }
}

}  // namespace quant
}  // namespace conch
