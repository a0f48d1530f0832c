// Static per-tensor quantisation feeders: fp16/bf16/fp32 -> int8 / fp8 (e4m3fn | e4m3fnuz).
//
// Replaces conch/kernels/quantization/int8.py:11-97 and fp8.py:13-97 (one Triton program per
// token row).  These ops are pure HBM streaming (2|4 bytes in, 1 byte out per element), so the
// gfx950 design is: flat 1-D view when the tensor is contiguous, 16 elements per lane per step
// (16-byte loads, one 16-byte store), grid capped at 16 blocks per CU with a grid-stride loop.
// Arithmetic follows the reference ORACLE (conch/reference/quantization/int8.py:12-18,
// fp8.py:12-18): fp32 multiply by the reciprocal of the scale, clamp, then truncate (int8) or
// round-to-nearest-even (fp8).
#include "common.hpp"
#include "quant_common.hpp"

namespace conch {
namespace {

using namespace quant;

// Flat, 16-byte aligned, contiguous view of n elements.
// ROUNDP (int8, 16-bit inputs): the product is rounded to the INPUT dtype before the clamp -- what torch's type promotion
// makes of `x * scale.reciprocal()` when `scale` is a 0-dim tensor (conch/reference/quantization/int8.py:12-18: a 0-dim
// operand does not take part in the result dtype; the fp32 reciprocal itself is not rounded).
template <int XDT, int KIND, bool ROUNDP>
__device__ __forceinline__ float round_product(float v) {
  if constexpr (ROUNDP && XDT != CONCH_DT_FP32) return bits16_to_float<XDT>(float_to_bits16<XDT>(v));
  return v;
}

template <int XDT, int KIND, bool ROUNDP = false>
__global__ __launch_bounds__(kQuantThreads) void quant_flat_kernel(uint8_t* __restrict__ out,
                                                                   const void* __restrict__ x,
                                                                   const float* __restrict__ scale,
                                                                   int64_t n) {
  const float inv = 1.0f / scale[0];
  const int64_t nvec = n / kVec;
  const int64_t stride = (int64_t)gridDim.x * kQuantThreads;
  int64_t v_first = (int64_t)blockIdx.x * kQuantThreads + threadIdx.x;
#ifndef CONCH_EXP_QUANT_LANE_CHUNKS  // (the A/B variant: rounds 1-3's form, 16 consecutive elements per lane, below)
  // Every load instruction of a wave covers one contiguous kilobyte (lane l its 16 bytes at 16 l), not 16 bytes per lane at a
  // 32- / 64-byte lane stride: a thread's 16 elements sit in 2 (16-bit input) or 4 (fp32) pieces of a 4096-element chunk and leave
  // as 8- / 4-byte stores.  Round 4, profiles/r04/quant_coalesced_ab.txt: C1 9.32 -> 7.81 us (5.4 -> 6.45 TB/s), the other
  // 16-bit cases -9 %, fp32 inputs -32...-36 % (they had been SLOWER than the scalar row kernel), bit-identical.
  {
    constexpr int kPieces = XDT == CONCH_DT_FP32 ? 4 : 2, kPer = kVec / kPieces;
    const int64_t nchunk = n / (kQuantThreads * kVec);
    for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
      const int64_t base = c * (kQuantThreads * kVec) + threadIdx.x * kPer;
      float f[kVec];
      if constexpr (XDT == CONCH_DT_FP32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 v = __builtin_nontemporal_load((const f32x4*)((const float*)x + base + j * (kQuantThreads * kPer)));
#pragma unroll
          for (int i = 0; i < 4; ++i) f[j * 4 + i] = v[i];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const u16x8 v = __builtin_nontemporal_load((const u16x8*)((const uint16_t*)x + base + j * (kQuantThreads * kPer)));
#pragma unroll
          for (int i = 0; i < 8; ++i) f[j * 8 + i] = bits16_to_float<XDT>(v[i]);
        }
      }
      if constexpr (ROUNDP) {
#pragma unroll
        for (int j = 0; j < kVec; ++j) f[j] = round_product<XDT, KIND, ROUNDP>(pin_f32(f[j] * inv));
      }
      const float mul = ROUNDP ? 1.0f : inv;
#pragma unroll
      for (int j = 0; j < kPieces; ++j) {
        uint8_t* dst = out + base + j * (kQuantThreads * kPer);
        if constexpr (kPer == 4) {
          __builtin_nontemporal_store((int)quant_four<KIND>(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3], mul), (int*)dst);
        } else {
          i32x2 o;
          o[0] = (int)quant_four<KIND>(f[8 * j], f[8 * j + 1], f[8 * j + 2], f[8 * j + 3], mul);
          o[1] = (int)quant_four<KIND>(f[8 * j + 4], f[8 * j + 5], f[8 * j + 6], f[8 * j + 7], mul);
          __builtin_nontemporal_store(o, (i32x2*)dst);
        }
      }
    }
    v_first += nchunk * kQuantThreads;  // the rest (< 4096 elements + the scalar tail) in the per-thread form below
  }
#endif
  for (int64_t v = v_first; v < nvec; v += stride) {
    float f[kVec];
    load16<XDT>(x, v * kVec, f);
    if constexpr (ROUNDP) {  // quant_four multiplies by its last argument: hand it the rounded product and 1
#pragma unroll
      for (int j = 0; j < kVec; ++j) f[j] = round_product<XDT, KIND, ROUNDP>(pin_f32(f[j] * inv));
    }
    const float mul = ROUNDP ? 1.0f : inv;
    i32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = (int)quant_four<KIND>(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3], mul);
    __builtin_nontemporal_store(o, (i32x4*)(out + v * kVec));
  }
  // scalar tail (< 16 elements)
  const int64_t tail0 = nvec * kVec;
  const int64_t t = tail0 + (int64_t)blockIdx.x * kQuantThreads + threadIdx.x;
  if (t < n) {
    const float v = load_as_float<XDT>(x, t);
    out[t] = ROUNDP ? (uint8_t)quant_one<KIND>(round_product<XDT, KIND, ROUNDP>(pin_f32(v * inv)), 1.0f) : (uint8_t)quant_one<KIND>(v, inv);
  }
}

// Row-strided or unaligned tensors: one block walks one row with scalar accesses.
template <int XDT, int KIND, bool ROUNDP = false>
__global__ __launch_bounds__(kQuantThreads) void quant_rows_kernel(uint8_t* __restrict__ out,
                                                                   const void* __restrict__ x,
                                                                   const float* __restrict__ scale,
                                                                   int64_t tokens, int64_t hidden,
                                                                   int64_t x_row_stride,
                                                                   int64_t out_row_stride) {
  const float inv = 1.0f / scale[0];
  for (int64_t row = blockIdx.x; row < tokens; row += gridDim.x) {
    const int64_t xb = row * x_row_stride;
    const int64_t ob = row * out_row_stride;
    for (int64_t h = threadIdx.x; h < hidden; h += kQuantThreads) {
      const float v = load_as_float<XDT>(x, xb + h);
      out[ob + h] = ROUNDP ? (uint8_t)quant_one<KIND>(round_product<XDT, KIND, ROUNDP>(pin_f32(v * inv)), 1.0f) : (uint8_t)quant_one<KIND>(v, inv);
    }
  }
}

// Row-strided but 16-byte aligned rows of a multiple of 16 elements (a column slice of a wider activation): one block walks one
// row in chunks of 4096 elements with the contiguous-per-instruction loads of the flat kernel (quant_common.hpp, load_pieces).
template <int XDT, int KIND, bool ROUNDP = false>
__global__ __launch_bounds__(kQuantThreads) void quant_rows_vec_kernel(uint8_t* __restrict__ out, const void* __restrict__ x,
                                                                       const float* __restrict__ scale, int64_t tokens, int64_t hidden,
                                                                       int64_t x_row_stride, int64_t out_row_stride) {
  const float inv = 1.0f / scale[0];
  const float mul = ROUNDP ? 1.0f : inv;
  for (int64_t row = blockIdx.x; row < tokens; row += gridDim.x) {
    const int64_t xb = row * x_row_stride;
    uint8_t* orow = out + row * out_row_stride;
    for (int64_t e0 = 0; e0 < hidden; e0 += (int64_t)kQuantThreads * kVec) {
      float f[kVec];
      load_pieces<XDT, kQuantThreads>(x, xb + e0, (int)threadIdx.x, xb + hidden, f);
      if constexpr (ROUNDP) {
#pragma unroll
        for (int j = 0; j < kVec; ++j) f[j] = round_product<XDT, KIND, ROUNDP>(pin_f32(f[j] * inv));
      }
      store_pieces<XDT, KIND, kQuantThreads>(orow, e0, (int)threadIdx.x, hidden, f, mul);
    }
  }
}

template <int XDT, int KIND, bool ROUNDP = false>
int launch_quant(uint8_t* out, const void* x, const float* scale, int64_t tokens, int64_t hidden,
                 int64_t x_row_stride, int64_t out_row_stride, hipStream_t stream) {
  const int64_t n = tokens * hidden;
  if (n == 0) return CONCH_OK;
  const bool flat = x_row_stride == hidden && out_row_stride == hidden &&
                    ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (flat) {
    const int64_t nvec = n / kVec;
    int64_t blocks = (nvec + kQuantThreads - 1) / kQuantThreads;
    if (blocks < 1) blocks = 1;
    // 16 blocks per CU, grid-stride beyond that (C1 = 4096 blocks: one vector per thread, no loop trip: 9.74 -> 9.42 us
    // against a cap of 8, graph-replayed; two vectors per trip with all loads up front measured 1-3 % slower)
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((quant_flat_kernel<XDT, KIND, ROUNDP>), dim3((unsigned)blocks), dim3(kQuantThreads),
                       0, stream, out, x, scale, n);
  } else {
    int64_t blocks = tokens < 256 * 8 ? tokens : 256 * 8;
    constexpr int kElem = XDT == CONCH_DT_FP32 ? 4 : 2;
    const bool vec_rows = hidden % kVec == 0 && (x_row_stride * kElem) % 16 == 0 && ((uintptr_t)x % 16 == 0) && out_row_stride % 8 == 0 &&
                          ((uintptr_t)out % 8 == 0);
    if (vec_rows)
      hipLaunchKernelGGL((quant_rows_vec_kernel<XDT, KIND, ROUNDP>), dim3((unsigned)blocks), dim3(kQuantThreads), 0, stream, out, x, scale, tokens,
                         hidden, x_row_stride, out_row_stride);
    else
      hipLaunchKernelGGL((quant_rows_kernel<XDT, KIND, ROUNDP>), dim3((unsigned)blocks), dim3(kQuantThreads),
                         0, stream, out, x, scale, tokens, hidden, x_row_stride, out_row_stride);
  }
  return check_launch("static_scaled_quant");
}

template <int KIND>
int dispatch_xdtype(uint8_t* out, const void* x, const float* scale, int64_t tokens, int64_t hidden,
                    int64_t xs, int64_t os, int x_dtype, hipStream_t stream) {
  switch (x_dtype) {
    case CONCH_DT_FP32:
      return launch_quant<CONCH_DT_FP32, KIND>(out, x, scale, tokens, hidden, xs, os, stream);
    case CONCH_DT_FP16:
      return launch_quant<CONCH_DT_FP16, KIND>(out, x, scale, tokens, hidden, xs, os, stream);
    case CONCH_DT_BF16:
      return launch_quant<CONCH_DT_BF16, KIND>(out, x, scale, tokens, hidden, xs, os, stream);
    default:
      set_error("static quant: unsupported input dtype %d (want FP32/FP16/BF16)", x_dtype);
      return CONCH_ERR_UNSUPPORTED;
  }
}

int check_quant_args(const void* out, const void* x, const float* scale, int64_t tokens,
                     int64_t hidden, int64_t xs, int64_t os) {
  CONCH_CHECK_ARG(tokens >= 0 && hidden >= 0, "static quant: negative shape (%lld, %lld)",
                  (long long)tokens, (long long)hidden);
  if (tokens == 0 || hidden == 0) return CONCH_OK;
  CONCH_CHECK_ARG(out && x && scale, "static quant: NULL pointer (out=%p x=%p scale=%p)", out, x,
                  (const void*)scale);
  CONCH_CHECK_ARG(xs >= hidden && os >= hidden,
                  "static quant: row strides (%lld, %lld) smaller than hidden %lld", (long long)xs,
                  (long long)os, (long long)hidden);
  return CONCH_OK;
}

}  // namespace
}  // namespace conch

extern "C" int conch_static_scaled_int8_quant(int8_t* out, const void* x, const float* scale,
                                              int64_t tokens, int64_t hidden, int64_t x_row_stride,
                                              int64_t out_row_stride, int x_dtype, void* stream) {
  using namespace conch;
  if (int rc = check_quant_args(out, x, scale, tokens, hidden, x_row_stride, out_row_stride)) return rc;
  return dispatch_xdtype<kInt8>((uint8_t*)out, x, scale, tokens, hidden, x_row_stride,
                                out_row_stride, x_dtype, (hipStream_t)stream);
}

extern "C" int conch_static_scaled_int8_quant_typed(int8_t* out, const void* x, const float* scale,
                                                    int64_t tokens, int64_t hidden, int64_t x_row_stride,
                                                    int64_t out_row_stride, int x_dtype, int product_dtype, void* stream) {
  using namespace conch;
  if (product_dtype == CONCH_DT_FP32)
    return conch_static_scaled_int8_quant(out, x, scale, tokens, hidden, x_row_stride, out_row_stride, x_dtype, stream);
  if (int rc = check_quant_args(out, x, scale, tokens, hidden, x_row_stride, out_row_stride)) return rc;
  CONCH_CHECK_ARG(product_dtype == x_dtype && (x_dtype == CONCH_DT_FP16 || x_dtype == CONCH_DT_BF16),
                  "static int8 quant: product dtype %d must be FP32 or the 16-bit input dtype (input dtype %d)", product_dtype, x_dtype);
  if (x_dtype == CONCH_DT_FP16)
    return launch_quant<CONCH_DT_FP16, kInt8, true>((uint8_t*)out, x, scale, tokens, hidden, x_row_stride, out_row_stride, (hipStream_t)stream);
  return launch_quant<CONCH_DT_BF16, kInt8, true>((uint8_t*)out, x, scale, tokens, hidden, x_row_stride, out_row_stride, (hipStream_t)stream);
}

extern "C" int conch_static_scaled_fp8_quant(uint8_t* out, const void* x, const float* scale,
                                             int64_t tokens, int64_t hidden, int64_t x_row_stride,
                                             int64_t out_row_stride, int x_dtype, int fp8_dtype,
                                             void* stream) {
  using namespace conch;
  if (int rc = check_quant_args(out, x, scale, tokens, hidden, x_row_stride, out_row_stride)) return rc;
  if (fp8_dtype == CONCH_DT_FP8_E4M3FN)
    return dispatch_xdtype<kFp8Fn>(out, x, scale, tokens, hidden, x_row_stride, out_row_stride,
                                   x_dtype, (hipStream_t)stream);
  if (fp8_dtype == CONCH_DT_FP8_E4M3FNUZ)
    return dispatch_xdtype<kFp8Fnuz>(out, x, scale, tokens, hidden, x_row_stride, out_row_stride,
                                     x_dtype, (hipStream_t)stream);
  set_error("static fp8 quant: unsupported fp8 dtype %d (want E4M3FN or E4M3FNUZ)", fp8_dtype);
  return CONCH_ERR_UNSUPPORTED;
}
