// Dynamic per-token quantisation feeders: fp16/bf16/fp32 [tokens][hidden] -> int8 / fp8 e4m3 + one fp32 scale per token.
//
// SURVEY.md 8(f) N1.  The reference stops at `scale is None -> NotImplementedError`
// (conch/ops/quantization/int8.py:42-44, fp8.py:46-48); the per-row scale this op returns is exactly the (M, 1) `scale_a`
// that scaled_gemm already takes (conch/ops/quantization/gemm.py:199-206).  Definition (oracle:
// oracle/reference.py dynamic_scaled_{int8,fp8}_quant_ref): the STATIC op's arithmetic with a per-row scale,
//   absmax[t] = max_h |float(x[t][h])|
//   scale[t]  = absmax[t] / QMAX  (fp32 division; QMAX = 127 int8, 448 e4m3fn, 240 e4m3fnuz),  1.0 for an all-zero row
//   q[t][h]   = static_quant(x[t][h], scale[t])   -- fp32 multiply by 1/scale, clamp, truncate (int8) / RNE (fp8)
// One pass over HBM: a row of up to 8192 elements stays in registers between the absmax reduction and the quantisation
// (a wave per row up to 1024 elements, a 256-thread workgroup per row above); longer rows are re-read (from L2).
#include "common.hpp"
#include "quant_common.hpp"

namespace conch {
namespace {

using namespace quant;

template <int KIND>
__device__ __forceinline__ float qmax() {
  return KIND == kInt8 ? 127.0f : KIND == kFp8Fn ? 448.0f : 240.0f;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

template <int KIND>
__device__ __forceinline__ float row_scale(float absmax) {
  return absmax > 0.0f ? absmax / qmax<KIND>() : 1.0f;  // IEEE division (-fno-fast-math), like torch
}

// Vector form: rows of a multiple of 16 elements, 16-byte aligned.  GROUP = threads per row (64: four rows per workgroup,
// one per wave; 256: one row per workgroup).  CACHE granules of 16 elements per thread stay in registers.
template <int XDT, int KIND, int GROUP, int CACHE>
__global__ __launch_bounds__(kQuantThreads) void dyn_quant_vec_kernel(uint8_t* __restrict__ out, float* __restrict__ scale_out,
                                                                      const void* __restrict__ x, int64_t tokens, int64_t hidden,
                                                                      int64_t x_row_stride, int64_t out_row_stride) {
  __shared__ float red[4];
  constexpr int kRowsPerWg = kQuantThreads / GROUP;
  const int tid = threadIdx.x % GROUP;
  const int64_t row = (int64_t)blockIdx.x * kRowsPerWg + threadIdx.x / GROUP;
  const bool live = row < tokens;  // wave-uniform when GROUP == 64, workgroup-uniform when 256
  const int64_t nvec = hidden / kVec;
  const int64_t xb = (live ? row : 0) * x_row_stride;
  float cache[CACHE][kVec];
  float amax = 0.0f;
#ifndef CONCH_EXP_QUANT_LANE_CHUNKS
  constexpr bool kPiecesForm = XDT == CONCH_DT_FP32;
#else
  constexpr bool kPiecesForm = false;
#endif
  // fp32 rows: chunks of GROUP x 16 elements, every load instruction contiguous across the lanes (quant_common.hpp, load_pieces):
  // 1.4-1.6x (profiles/r04/quant_coalesced_ab.txt).  16-bit rows keep 16 consecutive elements per lane (two loads at a 32-byte lane
  // stride, one 16-byte store): with a row in registers between the two passes the pieces form measured 2-5 % SLOWER there.
  constexpr int64_t kChunk = (int64_t)GROUP * kVec;
  const int64_t x_end = xb + hidden;
  if constexpr (kPiecesForm) {
#pragma unroll
  for (int c = 0; c < CACHE; ++c) {
    if ((int64_t)c * kChunk < hidden) {
      load_pieces<XDT, GROUP>(x, xb + c * kChunk, tid, x_end, cache[c]);
#pragma unroll
      for (int j = 0; j < kVec; ++j) amax = fmaxf(amax, fabsf(cache[c][j]));
    }
  }
  for (int64_t e0 = (int64_t)CACHE * kChunk; e0 < hidden; e0 += kChunk) {
    float f[kVec];
    load_pieces<XDT, GROUP>(x, xb + e0, tid, x_end, f);
#pragma unroll
    for (int j = 0; j < kVec; ++j) amax = fmaxf(amax, fabsf(f[j]));
  }
  } else {
#pragma unroll
  for (int c = 0; c < CACHE; ++c) {
    const int64_t v = tid + (int64_t)c * GROUP;
    if (v < nvec) {
      load16<XDT>(x, xb + v * kVec, cache[c]);
#pragma unroll
      for (int j = 0; j < kVec; ++j) amax = fmaxf(amax, fabsf(cache[c][j]));
    }
  }
  for (int64_t v = tid + (int64_t)CACHE * GROUP; v < nvec; v += GROUP) {
    float f[kVec];
    load16<XDT>(x, xb + v * kVec, f);
#pragma unroll
    for (int j = 0; j < kVec; ++j) amax = fmaxf(amax, fabsf(f[j]));
  }
  }
  amax = wave_max(amax);
  if constexpr (GROUP == 256) {
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
  if (!live) return;
  const float scale = row_scale<KIND>(amax);
  const float inv = 1.0f / scale;
  if (tid == 0) scale_out[row] = scale;
  uint8_t* orow = out + row * out_row_stride;
  if constexpr (kPiecesForm) {
#pragma unroll
    for (int c = 0; c < CACHE; ++c)
      if ((int64_t)c * kChunk < hidden) store_pieces<XDT, KIND, GROUP>(orow, c * kChunk, tid, hidden, cache[c], inv);
    for (int64_t e0 = (int64_t)CACHE * kChunk; e0 < hidden; e0 += kChunk) {
      float f[kVec];
      load_pieces<XDT, GROUP>(x, xb + e0, tid, x_end, f);
      store_pieces<XDT, KIND, GROUP>(orow, e0, tid, hidden, f, inv);
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < CACHE; ++c) {
    const int64_t v = tid + (int64_t)c * GROUP;
    if (v < nvec) {
      i32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = (int)quant_four<KIND>(cache[c][4 * j], cache[c][4 * j + 1], cache[c][4 * j + 2], cache[c][4 * j + 3], inv);
      __builtin_nontemporal_store(o, (i32x4*)(orow + v * kVec));
    }
  }
  for (int64_t v = tid + (int64_t)CACHE * GROUP; v < nvec; v += GROUP) {
    float f[kVec];
    load16<XDT>(x, xb + v * kVec, f);
    i32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (int)quant_four<KIND>(f[4 * j], f[4 * j + 1], f[4 * j + 2], f[4 * j + 3], inv);
    __builtin_nontemporal_store(o, (i32x4*)(orow + v * kVec));
  }
}

// Any row length / alignment: one workgroup per row, scalar accesses, two passes.
template <int XDT, int KIND>
__global__ __launch_bounds__(kQuantThreads) void dyn_quant_rows_kernel(uint8_t* __restrict__ out, float* __restrict__ scale_out,
                                                                       const void* __restrict__ x, int64_t tokens, int64_t hidden,
                                                                       int64_t x_row_stride, int64_t out_row_stride) {
  __shared__ float red[4];
  for (int64_t row = blockIdx.x; row < tokens; row += gridDim.x) {
    const int64_t xb = row * x_row_stride;
    float amax = 0.0f;
    for (int64_t h = threadIdx.x; h < hidden; h += kQuantThreads) amax = fmaxf(amax, fabsf(load_as_float<XDT>(x, xb + h)));
    amax = wave_max(amax);
    __syncthreads();  // the previous row's readers are done with `red`
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = row_scale<KIND>(amax);
    const float inv = 1.0f / scale;
    if (threadIdx.x == 0) scale_out[row] = scale;
    for (int64_t h = threadIdx.x; h < hidden; h += kQuantThreads)
      out[row * out_row_stride + h] = (uint8_t)quant_one<KIND>(load_as_float<XDT>(x, xb + h), inv);
  }
}

template <int XDT, int KIND>
int launch_dyn(uint8_t* out, float* scale_out, const void* x, int64_t tokens, int64_t hidden, int64_t xs, int64_t os,
               hipStream_t stream) {
  constexpr int kElt = XDT == CONCH_DT_FP32 ? 4 : 2;
  const bool vec = hidden % kVec == 0 && (xs * kElt) % 16 == 0 && os % 16 == 0 && ((uintptr_t)x % 16 == 0) &&
                   ((uintptr_t)out % 16 == 0);
  if (!vec) {
    const int64_t blocks = tokens < 256 * 8 ? tokens : 256 * 8;
    hipLaunchKernelGGL((dyn_quant_rows_kernel<XDT, KIND>), dim3((unsigned)blocks), dim3(kQuantThreads), 0, stream, out, scale_out,
                       x, tokens, hidden, xs, os);
  } else if (hidden <= 1024) {
    hipLaunchKernelGGL((dyn_quant_vec_kernel<XDT, KIND, 64, 1>), dim3((unsigned)((tokens + 3) / 4)), dim3(kQuantThreads), 0, stream,
                       out, scale_out, x, tokens, hidden, xs, os);
  } else {
    hipLaunchKernelGGL((dyn_quant_vec_kernel<XDT, KIND, 256, 2>), dim3((unsigned)tokens), dim3(kQuantThreads), 0, stream, out,
                       scale_out, x, tokens, hidden, xs, os);
  }
  return check_launch("dynamic_scaled_quant");
}

template <int KIND>
int dispatch_dyn(uint8_t* out, float* scale_out, const void* x, int64_t tokens, int64_t hidden, int64_t xs, int64_t os, int x_dtype,
                 hipStream_t stream) {
  CONCH_CHECK_ARG(tokens >= 0 && hidden >= 0, "dynamic quant: negative shape (%lld, %lld)", (long long)tokens, (long long)hidden);
  if (tokens == 0) return CONCH_OK;
  CONCH_CHECK_ARG(scale_out, "dynamic quant: NULL scale output");
  CONCH_CHECK_ARG(tokens < ((int64_t)1 << 31), "dynamic quant: %lld tokens exceed the grid limit", (long long)tokens);
  CONCH_CHECK_ARG(hidden == 0 || (out && x), "dynamic quant: NULL pointer (out=%p x=%p)", (void*)out, x);
  CONCH_CHECK_ARG(xs >= hidden && os >= hidden, "dynamic quant: row strides (%lld, %lld) smaller than hidden %lld", (long long)xs,
                  (long long)os, (long long)hidden);
  switch (x_dtype) {
    case CONCH_DT_FP32: return launch_dyn<CONCH_DT_FP32, KIND>(out, scale_out, x, tokens, hidden, xs, os, stream);
    case CONCH_DT_FP16: return launch_dyn<CONCH_DT_FP16, KIND>(out, scale_out, x, tokens, hidden, xs, os, stream);
    case CONCH_DT_BF16: return launch_dyn<CONCH_DT_BF16, KIND>(out, scale_out, x, tokens, hidden, xs, os, stream);
    default:
      set_error("dynamic quant: unsupported input dtype %d (want FP32/FP16/BF16)", x_dtype);
      return CONCH_ERR_UNSUPPORTED;
  }
}

}  // namespace
}  // namespace conch

extern "C" int conch_dynamic_scaled_int8_quant(int8_t* out, float* scale_out, const void* x, int64_t tokens, int64_t hidden,
                                               int64_t x_row_stride, int64_t out_row_stride, int x_dtype, void* stream) {
  using namespace conch;
  return dispatch_dyn<quant::kInt8>((uint8_t*)out, scale_out, x, tokens, hidden, x_row_stride, out_row_stride, x_dtype,
                                    (hipStream_t)stream);
}

extern "C" int conch_dynamic_scaled_fp8_quant(uint8_t* out, float* scale_out, const void* x, int64_t tokens, int64_t hidden,
                                              int64_t x_row_stride, int64_t out_row_stride, int x_dtype, int fp8_dtype,
                                              void* stream) {
  using namespace conch;
  if (fp8_dtype == CONCH_DT_FP8_E4M3FN)
    return dispatch_dyn<quant::kFp8Fn>(out, scale_out, x, tokens, hidden, x_row_stride, out_row_stride, x_dtype, (hipStream_t)stream);
  if (fp8_dtype == CONCH_DT_FP8_E4M3FNUZ)
    return dispatch_dyn<quant::kFp8Fnuz>(out, scale_out, x, tokens, hidden, x_row_stride, out_row_stride, x_dtype, (hipStream_t)stream);
  set_error("dynamic fp8 quant: unsupported fp8 dtype %d (want E4M3FN or E4M3FNUZ)", fp8_dtype);
  return CONCH_ERR_UNSUPPORTED;
}
