// Shape- and stride-agnostic GEMM kernels (no MFMA, no alignment requirements).
//
// These are the "always correct" device paths: any M/N/K, any element strides, every dtype
// combination the C ABI accepts.  The dispatcher (capi.hip) uses them when a request does not meet
// the layout contract of the MFMA kernels (gemm_mfma.hip) and tests use them as an on-device
// cross-check.  One thread per output element, 16x16 thread tiles.
#include "common.hpp"
#include "gemm.hpp"

namespace conch {
namespace {

template <int IN_DT>
__device__ __forceinline__ float decode8(uint8_t c) {
  if constexpr (IN_DT == CONCH_DT_FP8_E4M3FN) return decode_fp8_sw<false>(c);
  else if constexpr (IN_DT == CONCH_DT_FP8_E4M3FNUZ) return decode_fp8_sw<true>(c);
  else return (float)(int8_t)c;
}

template <int IN_DT, int OUT_DT>
__global__ __launch_bounds__(256) void scaled_gemm_generic_kernel(ScaledGemmArgs p) {
  const int64_t n = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
  const int64_t m = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
  if (m >= p.m || n >= p.n) return;
  const uint8_t* a = (const uint8_t*)p.a + m * p.a_stride_m;
  const uint8_t* b = (const uint8_t*)p.b + n * p.b_stride_n;
  float accf;
  if constexpr (IN_DT == CONCH_DT_INT8) {
    int acc = 0;  // exact
    for (int64_t k = 0; k < p.k; ++k)
      acc += (int)(int8_t)a[k * p.a_stride_k] * (int)(int8_t)b[k * p.b_stride_k];
    accf = (float)acc;
  } else {
    float acc = 0.f;
    for (int64_t k = 0; k < p.k; ++k)
      acc = fmaf(decode8<IN_DT>(a[k * p.a_stride_k]), decode8<IN_DT>(b[k * p.b_stride_k]), acc);
    accf = acc;
  }
  // reference/quantization/scaled_gemm.py:20-25: scale_a first, then scale_b, cast, then bias
  const float sa = p.scale_a[p.scale_a_numel == 1 ? 0 : m];
  const float sb = p.scale_b[p.scale_b_numel == 1 ? 0 : n];
  float v = sa * accf;
  v = pin_f32(sb * v);  // rounded to fp32 before the cast, like torch
  uint16_t o = float_to_bits16<OUT_DT>(v);
  if (p.bias) {
    const float bsum = bits16_to_float<OUT_DT>(o) + bits16_to_float<OUT_DT>(((const uint16_t*)p.bias)[n]);
    o = float_to_bits16<OUT_DT>(pin_f32(bsum));
  }
  ((uint16_t*)p.c)[m * p.c_stride_m + n * p.c_stride_n] = o;
}

template <int X_DT, int OUT_DT>
__global__ __launch_bounds__(256) void mixed_gemm_generic_kernel(MixedGemmArgs p) {
  const int64_t n = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
  const int64_t m = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
  if (m >= p.m || n >= p.n) return;
  const int per_word = 32 / p.bits;
  const uint32_t mask = (p.bits == 32) ? 0xffffffffu : ((1u << p.bits) - 1u);
  const uint16_t* x = (const uint16_t*)p.x + m * p.x_stride_m;
  const int zp_scalar = p.zp_mode == CONCH_ZP_SCALAR ? p.w_zp[0] : 0;
  float acc = 0.f;
  for (int64_t k = 0; k < p.k; ++k) {
    const uint32_t word = (uint32_t)p.w_q[(k / per_word) * p.wq_stride_k + n];
    int q = (int)((word >> ((k % per_word) * p.bits)) & mask) - p.weight_bias;
    const int64_t g = k / p.group_size;
    if (p.zp_mode == CONCH_ZP_TENSOR) q -= p.w_zp[g * p.wzp_stride_g + n];
    else q -= zp_scalar;
    const float s = bits16_to_float<X_DT>(((const uint16_t*)p.w_s)[g * p.ws_stride_g + n]);
    // (q exact) * s: the fp32 product is exact, so rounding it to X_DT is the single rounding of
    // the reference's fp16/bf16 multiply (kernels/quantization/gemm.py:201-210).
    // the difference is formed in the meta dtype (:205-210): bf16 rounds |q| > 256 (8-bit weights with a bias and a zero point)
    const float qd = bits16_to_float<X_DT>(float_to_bits16<X_DT>((float)q));
    const float w = bits16_to_float<X_DT>(float_to_bits16<X_DT>(qd * s));
    acc = fmaf(bits16_to_float<X_DT>(x[k]), w, acc);
  }
  ((uint16_t*)p.c)[m * p.c_stride_m + n] = float_to_bits16<OUT_DT>(acc);
}

// fp32 activations (the reference's benchmark offers --input-dtype fp32: benchmarks/mixed_precision_gemm_benchmark.py:99-105; its
// kernel then dequantises and accumulates in fp32, kernels/quantization/gemm.py:201-216): scales fp32, result fp32,
// w = float(q - bias - zp) * s rounded once to fp32 (= w_ref in fp32, quant_utils.py:74), fmaf accumulation in k order.  gfx950 has
// no fp32-input MFMA above the vector rate, so this is a completeness path, one thread per output: ~1 TFLOP/s.
__global__ __launch_bounds__(256) void mixed_gemm_generic_f32_kernel(MixedGemmArgs p) {
  const int64_t n = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
  const int64_t m = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
  if (m >= p.m || n >= p.n) return;
  const int per_word = 32 / p.bits;
  const uint32_t mask = (p.bits == 32) ? 0xffffffffu : ((1u << p.bits) - 1u);
  const float* x = (const float*)p.x + m * p.x_stride_m;
  const float* ws = (const float*)p.w_s;
  const int zp_scalar = p.zp_mode == CONCH_ZP_SCALAR ? p.w_zp[0] : 0;
  float acc = 0.f;
  for (int64_t k0 = 0; k0 < p.k; k0 += per_word) {
    const uint32_t word = (uint32_t)p.w_q[(k0 / per_word) * p.wq_stride_k + n];
    const int64_t g = k0 / p.group_size;  // group_size is a multiple of the words' span (checked by the caller)
    const int zp = p.zp_mode == CONCH_ZP_TENSOR ? p.w_zp[g * p.wzp_stride_g + n] : zp_scalar;
    const float s = ws[g * p.ws_stride_g + n];
    for (int j = 0; j < per_word; ++j) {
      const int q = (int)((word >> (j * p.bits)) & mask) - p.weight_bias - zp;
      const float w = pin_f32((float)q * s);
      acc = fmaf(x[k0 + j], w, acc);
    }
  }
  ((float*)p.c)[m * p.c_stride_m + n] = acc;
}

}  // namespace

int launch_scaled_gemm_generic(const ScaledGemmArgs& p, hipStream_t stream) {
  const dim3 grid((unsigned)((p.n + 15) / 16), (unsigned)((p.m + 15) / 16));
#define CONCH_LAUNCH(IN, OUT)                                                                  \
  hipLaunchKernelGGL((scaled_gemm_generic_kernel<IN, OUT>), grid, dim3(256), 0, stream, p);    \
  return check_launch("scaled_gemm_generic")
  if (p.out_dtype == CONCH_DT_FP16) {
    if (p.in_dtype == CONCH_DT_INT8) { CONCH_LAUNCH(CONCH_DT_INT8, CONCH_DT_FP16); }
    if (p.in_dtype == CONCH_DT_FP8_E4M3FN) { CONCH_LAUNCH(CONCH_DT_FP8_E4M3FN, CONCH_DT_FP16); }
    if (p.in_dtype == CONCH_DT_FP8_E4M3FNUZ) { CONCH_LAUNCH(CONCH_DT_FP8_E4M3FNUZ, CONCH_DT_FP16); }
  } else {
    if (p.in_dtype == CONCH_DT_INT8) { CONCH_LAUNCH(CONCH_DT_INT8, CONCH_DT_BF16); }
    if (p.in_dtype == CONCH_DT_FP8_E4M3FN) { CONCH_LAUNCH(CONCH_DT_FP8_E4M3FN, CONCH_DT_BF16); }
    if (p.in_dtype == CONCH_DT_FP8_E4M3FNUZ) { CONCH_LAUNCH(CONCH_DT_FP8_E4M3FNUZ, CONCH_DT_BF16); }
  }
#undef CONCH_LAUNCH
  set_error("scaled_gemm: unsupported dtype combination in=%d out=%d", p.in_dtype, p.out_dtype);
  return CONCH_ERR_UNSUPPORTED;
}

int launch_mixed_gemm_generic(const MixedGemmArgs& p, hipStream_t stream) {
  const dim3 grid((unsigned)((p.n + 15) / 16), (unsigned)((p.m + 15) / 16));
#define CONCH_LAUNCH(X, OUT)                                                                  \
  hipLaunchKernelGGL((mixed_gemm_generic_kernel<X, OUT>), grid, dim3(256), 0, stream, p);     \
  return check_launch("mixed_gemm_generic")
  if (p.x_dtype == CONCH_DT_FP32 && p.out_dtype == CONCH_DT_FP32) {
    hipLaunchKernelGGL(mixed_gemm_generic_f32_kernel, grid, dim3(256), 0, stream, p);
    return check_launch("mixed_gemm_generic_f32");
  }
  if (p.x_dtype == CONCH_DT_FP16 && p.out_dtype == CONCH_DT_FP16) { CONCH_LAUNCH(CONCH_DT_FP16, CONCH_DT_FP16); }
  if (p.x_dtype == CONCH_DT_FP16 && p.out_dtype == CONCH_DT_BF16) { CONCH_LAUNCH(CONCH_DT_FP16, CONCH_DT_BF16); }
  if (p.x_dtype == CONCH_DT_BF16 && p.out_dtype == CONCH_DT_BF16) { CONCH_LAUNCH(CONCH_DT_BF16, CONCH_DT_BF16); }
  if (p.x_dtype == CONCH_DT_BF16 && p.out_dtype == CONCH_DT_FP16) { CONCH_LAUNCH(CONCH_DT_BF16, CONCH_DT_FP16); }
#undef CONCH_LAUNCH
  set_error("mixed_precision_gemm: unsupported dtype combination x=%d out=%d", p.x_dtype, p.out_dtype);
  return CONCH_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// silu(x[:, :n]) * x[:, n:] on a 16-bit [m][2n] matrix: the unfused tail of scaled_gemm_silu_and_mul, run after the
// plain GEMM wherever the fused epilogue's layout contract is not met (and the comparator for the fused kernel).
// Same roundings as the fused epilogue: silu in fp32 rounded to the dtype, product rounded again.
// ---------------------------------------------------------------------------------------------
namespace {
template <int DT>
__global__ void silu_and_mul_kernel(uint16_t* out, const uint16_t* x, int64_t m, int64_t n, int64_t x_stride_m, int64_t out_stride_m, int act) {
  const int64_t total = m * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / n, col = i - r * n;
    const float g = bits16_to_float<DT>(x[r * x_stride_m + col]);
    const float u = bits16_to_float<DT>(x[r * x_stride_m + n + col]);
    const float s = bits16_to_float<DT>(float_to_bits16<DT>(pin_f32(act_f32(g, act))));
    out[r * out_stride_m + col] = float_to_bits16<DT>(pin_f32(s * u));
  }
}
// eight columns per thread (16-byte loads of the gate and the up half, one 16-byte store), blockIdx.y = row: no division, every
// load instruction of a wave one contiguous KiB.  Needs n % 8 == 0 and 16-byte aligned rows on both sides.
template <int DT>
__global__ __launch_bounds__(256) void silu_and_mul_vec_kernel(uint16_t* out, const uint16_t* x, int64_t n, int64_t x_stride_m, int64_t out_stride_m,
                                                               int act) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= n) return;
  const uint16_t* xr = x + (int64_t)blockIdx.y * x_stride_m;
  const u16x8 g8 = *(const u16x8*)(xr + c), u8 = *(const u16x8*)(xr + n + c);
  u16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float s = bits16_to_float<DT>(float_to_bits16<DT>(pin_f32(act_f32(bits16_to_float<DT>(g8[i]), act))));
    o[i] = float_to_bits16<DT>(pin_f32(s * bits16_to_float<DT>(u8[i])));
  }
  *(u16x8*)(out + (int64_t)blockIdx.y * out_stride_m + c) = o;
}
}  // namespace

int launch_silu_and_mul(void* out, const void* x, int64_t m, int64_t n, int64_t x_stride_m, int64_t out_stride_m, int dtype, int act,
                        hipStream_t stream) {
  if (m == 0 || n == 0) return CONCH_OK;
  if (n % 8 == 0 && x_stride_m % 8 == 0 && out_stride_m % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0 && m < 65536) {
    const dim3 grid((unsigned)((n / 8 + 255) / 256), (unsigned)m);
    if (dtype == CONCH_DT_FP16)
      hipLaunchKernelGGL((silu_and_mul_vec_kernel<CONCH_DT_FP16>), grid, dim3(256), 0, stream, (uint16_t*)out, (const uint16_t*)x, n, x_stride_m, out_stride_m, act);
    else
      hipLaunchKernelGGL((silu_and_mul_vec_kernel<CONCH_DT_BF16>), grid, dim3(256), 0, stream, (uint16_t*)out, (const uint16_t*)x, n, x_stride_m, out_stride_m, act);
    return check_launch("silu_and_mul");
  }
  const int64_t total = m * n;
  const unsigned blocks = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  if (dtype == CONCH_DT_FP16)
    hipLaunchKernelGGL((silu_and_mul_kernel<CONCH_DT_FP16>), dim3(blocks), dim3(256), 0, stream, (uint16_t*)out, (const uint16_t*)x, m, n, x_stride_m, out_stride_m, act);
  else
    hipLaunchKernelGGL((silu_and_mul_kernel<CONCH_DT_BF16>), dim3(blocks), dim3(256), 0, stream, (uint16_t*)out, (const uint16_t*)x, m, n, x_stride_m, out_stride_m, act);
  return check_launch("silu_and_mul");
}

}  // namespace conch
