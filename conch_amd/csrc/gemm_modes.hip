// The weight-group and channel-scale modes of the reference kernel that conch.ops never selects:
//   weight groups   NONE | SHIFT | SYMMETRIC_NO_SHIFT | SYMMETRIC_WITH_SHIFT | ASYMMETRIC      kernels/quantization/gemm.py:176-216
//   channel scales  NONE | WEIGHT_ONLY (mixed launcher)                                        :408-416
//   1-, 2-, 4-, 8- and 16-bit packed weights, integer or floating zero points, scalar or per group.
// The two modes conch.ops does produce on 4/8-bit weights keep their fused kernels (gemm_mixed.hip); everything else runs
//   1. dequant_modes_kernel: packed words -> W^T [N][K] in the activation dtype, every operation rounded to that dtype in
//      the kernel's order (bit-identical to the reference's Triton kernel: tests/golden/gemm_modes.npz), one HBM pass;
//   2. the 256x256-tile ping-pong MFMA kernel of gemm_mfma.hip on 16-bit operands (v_mfma_f32_16x16x32_{f16,bf16});
//   3. for WEIGHT_ONLY, colscale_kernel: C = out( meta(acc) * s[n] ), the accumulator rounded to the meta dtype first.
// K not a multiple of 64 (the tile contract) takes modes_generic_kernel, one thread per output, same arithmetic.
#include "common.hpp"
#include "gemm.hpp"

namespace conch {
namespace {

template <int XDT>
__device__ __forceinline__ float round_m(float v) {
  return bits16_to_float<XDT>(float_to_bits16<XDT>(pin_f32(v)));
}

struct ModesMeta {
  const int32_t* w_q;
  const void* scales;
  const void* zeros;
  int64_t wq_stride_k, scales_stride_g, zeros_stride_g;
  int bits, weight_bias, group_size, group_mode, zeros_kind;
};

// One weight: kernels/quantization/gemm.py:192-216 with every intermediate rounded to the meta dtype (= XDT).
template <int XDT>
__device__ __forceinline__ uint16_t dequant_one(const ModesMeta& q, uint32_t word, int e, int64_t k, int64_t n, float zero_scalar) {
  const uint32_t mask = q.bits == 32 ? 0xffffffffu : ((1u << q.bits) - 1u);
  float b = round_m<XDT>((float)((word >> (e * q.bits)) & mask));  // .to(meta)                       :193
  b = round_m<XDT>(b - (float)q.weight_bias);                      // b -= weight_bias                :194
  if (q.group_mode == CONCH_GROUP_NONE) return float_to_bits16<XDT>(b);
  const int64_t g = k / q.group_size;
  float z = zero_scalar;
  if (q.zeros_kind == CONCH_ZEROS_TENSOR_INT32) z = round_m<XDT>((float)((const int32_t*)q.zeros)[g * q.zeros_stride_g + n]);
  else if (q.zeros_kind == CONCH_ZEROS_TENSOR_FLOAT) z = bits16_to_float<XDT>(((const uint16_t*)q.zeros)[g * q.zeros_stride_g + n]);
  if (q.group_mode == CONCH_GROUP_SHIFT) return float_to_bits16<XDT>(b - z);                          // :197-198
  const float s = bits16_to_float<XDT>(((const uint16_t*)q.scales)[g * q.scales_stride_g + n]);
  if (q.group_mode == CONCH_GROUP_SYMMETRIC_NO_SHIFT) return float_to_bits16<XDT>(b * s);             // :201-202
  if (q.group_mode == CONCH_GROUP_SYMMETRIC_WITH_SHIFT) return float_to_bits16<XDT>(round_m<XDT>(b - z) * s);  // :205-210
  // ASYMMETRIC: fma(b, s, z) -- product exact in double, sum rounded to double, then to float, then to the dtype: the oracle's
  // definition of the fused form (oracle/reference.py dequantize_modes_ref; torch's double -> half / bfloat16 casts go
  // through float).  It differs from an ideally fused fp16 fma only where the float rounding lands exactly on a tie.
  const float r = (float)((double)b * (double)s + (double)z);                                        // :213-214
  return float_to_bits16<XDT>(r);
}

__device__ __forceinline__ float load_zero_scalar(const ModesMeta& q, int xdt) {
  if (q.zeros_kind == CONCH_ZEROS_SCALAR_INT32) {
    const float z = (float)((const int32_t*)q.zeros)[0];
    return xdt == CONCH_DT_FP16 ? round_m<CONCH_DT_FP16>(z) : round_m<CONCH_DT_BF16>(z);
  }
  if (q.zeros_kind == CONCH_ZEROS_SCALAR_FLOAT) {
    const uint16_t h = ((const uint16_t*)q.zeros)[0];
    return xdt == CONCH_DT_FP16 ? half_bits_to_float(h) : bf16_bits_to_float(h);
  }
  return 0.0f;
}

// W^T[n][k] for one column n and `kWords` consecutive packed words per thread (>= 32 contiguous output bytes per thread;
// the packed reads are coalesced across the threads' columns).
constexpr int kDqThreads = 256;
constexpr int kDqWords = 4;

template <int XDT>
__global__ __launch_bounds__(kDqThreads) void dequant_modes_kernel(uint16_t* __restrict__ wt, int64_t ldw, ModesMeta q, int64_t n_dim,
                                                                   int64_t k_dim) {
  const int64_t n = (int64_t)blockIdx.x * kDqThreads + threadIdx.x;
  if (n >= n_dim) return;
  const int pf = 32 / q.bits;
  const int64_t words = k_dim / pf;
  const float zs = load_zero_scalar(q, XDT);
  for (int w = 0; w < kDqWords; ++w) {
    const int64_t kw = (int64_t)blockIdx.y * kDqWords + w;
    if (kw >= words) return;
    const uint32_t word = (uint32_t)q.w_q[kw * q.wq_stride_k + n];
    uint16_t* dst = wt + n * ldw + kw * pf;
    if (pf % 8 == 0 && (((uintptr_t)dst) & 15) == 0) {
      // 16-byte stores of eight elements (1-, 2- and 4-bit weights; the per-thread rows are ldw apart whatever the store width, so
      // fewer, wider stores are all there is to gain here): round 4, one 2-byte store per element before
      for (int e0 = 0; e0 < pf; e0 += 8) {
        u16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = dequant_one<XDT>(q, word, e0 + e, kw * pf + e0 + e, n, zs);
        *(u16x8*)(dst + e0) = v;
      }
    } else {
      for (int e = 0; e < pf; ++e) dst[e] = dequant_one<XDT>(q, word, e, kw * pf + e, n, zs);
    }
  }
}

// out[m][n] = OUT( META(c[m][n]) * s[n] ), in place on a 16-bit matrix that holds META-typed accumulators  (:408-416, :457)
template <int MDT, int ODT>
__global__ __launch_bounds__(256) void colscale_kernel(uint16_t* __restrict__ c, int64_t m_dim, int64_t n_dim, int64_t ldc,
                                                       const uint16_t* __restrict__ s) {
  const int64_t total = m_dim * n_dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / n_dim, col = i - r * n_dim;
    const float v = round_m<MDT>(bits16_to_float<MDT>(c[r * ldc + col]) * bits16_to_float<MDT>(s[col]));
    c[r * ldc + col] = float_to_bits16<ODT>(v);
  }
}

// any shape: one thread per output element
template <int XDT, int ODT>
__global__ __launch_bounds__(256) void modes_generic_kernel(MixedGemmArgs p, ModesMeta q, const uint16_t* __restrict__ channel_scales) {
  const int64_t n = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
  const int64_t m = (int64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
  if (m >= p.m || n >= p.n) return;
  const int pf = 32 / q.bits;
  const uint16_t* x = (const uint16_t*)p.x + m * p.x_stride_m;
  const float zs = load_zero_scalar(q, XDT);
  float acc = 0.f;
  for (int64_t k = 0; k < p.k; ++k) {
    const uint32_t word = (uint32_t)q.w_q[(k / pf) * q.wq_stride_k + n];
    const float w = bits16_to_float<XDT>(dequant_one<XDT>(q, word, (int)(k % pf), k, n, zs));
    acc = fmaf(bits16_to_float<XDT>(x[k]), w, acc);
  }
  if (channel_scales) acc = round_m<XDT>(round_m<XDT>(acc) * bits16_to_float<XDT>(channel_scales[n]));
  ((uint16_t*)p.c)[m * p.c_stride_m + n] = float_to_bits16<ODT>(acc);
}

}  // namespace

// `p` carries the shapes / activations / output; `group_mode`, `zeros_kind`, `channel_scales` the modes.
int launch_mixed_gemm_modes(const MixedGemmArgs& p, const void* scales, const void* zeros, int64_t zeros_stride_g, int group_mode,
                            int zeros_kind, const void* channel_scales, hipStream_t stream) {
  ModesMeta q{p.w_q, scales, zeros, p.wq_stride_k, p.ws_stride_g, zeros_stride_g, p.bits, p.weight_bias, p.group_size, group_mode, zeros_kind};
  const bool fp16 = p.x_dtype == CONCH_DT_FP16;
  const int64_t lim = (int64_t)1 << 31;
  const bool tiled = p.k % 64 == 0 && p.k >= 128 && (p.x_stride_m * 2) % 16 == 0 && ((uintptr_t)p.x & 15) == 0 && p.m * p.x_stride_m * 2 < lim &&
                     p.n * p.k * 2 < lim && p.m < (1 << 24) && p.n < (1 << 24) && (int64_t)p.m * p.n * p.k >= ((int64_t)1 << 21);
  if (!tiled) {
    const dim3 grid((unsigned)((p.n + 15) / 16), (unsigned)((p.m + 15) / 16));
#define CONCH_LAUNCH(X, O) hipLaunchKernelGGL((modes_generic_kernel<X, O>), grid, dim3(256), 0, stream, p, q, (const uint16_t*)channel_scales)
    if (fp16 && p.out_dtype == CONCH_DT_FP16) CONCH_LAUNCH(CONCH_DT_FP16, CONCH_DT_FP16);
    else if (fp16) CONCH_LAUNCH(CONCH_DT_FP16, CONCH_DT_BF16);
    else if (p.out_dtype == CONCH_DT_BF16) CONCH_LAUNCH(CONCH_DT_BF16, CONCH_DT_BF16);
    else CONCH_LAUNCH(CONCH_DT_BF16, CONCH_DT_FP16);
#undef CONCH_LAUNCH
    return check_launch("mixed_gemm_modes_generic");
  }
  // 1. W^T [N][K] in the activation dtype
  void* wt = nullptr;
  if (int rc = get_scratch(stream, kScratchRepack, (size_t)p.n * p.k * 2, &wt)) return rc;
  {
    const int pf = 32 / p.bits;
    const dim3 grid((unsigned)((p.n + kDqThreads - 1) / kDqThreads), (unsigned)((p.k / pf + kDqWords - 1) / kDqWords));
    if (fp16) hipLaunchKernelGGL((dequant_modes_kernel<CONCH_DT_FP16>), grid, dim3(kDqThreads), 0, stream, (uint16_t*)wt, p.k, q, p.n, p.k);
    else hipLaunchKernelGGL((dequant_modes_kernel<CONCH_DT_BF16>), grid, dim3(kDqThreads), 0, stream, (uint16_t*)wt, p.k, q, p.n, p.k);
    if (int rc = check_launch("dequant_modes")) return rc;
  }
  // 2. C = X @ W on the 16-bit MFMA tile kernel (byte units; NULL scales = 1.0).  With a channel scale the accumulator is
  //    first rounded to the META dtype (= the activation dtype), so the GEMM writes that type and step 3 converts in place.
  ScaledGemmArgs g{};
  g.c = p.c;
  g.a = p.x;
  g.b = wt;
  g.scale_a = g.scale_b = nullptr;
  g.bias = nullptr;
  g.m = p.m;
  g.n = p.n;
  g.k = p.k * 2;
  g.a_stride_m = p.x_stride_m * 2;
  g.b_stride_n = p.k * 2;
  g.a_stride_k = g.b_stride_k = 1;
  g.c_stride_m = p.c_stride_m;
  g.c_stride_n = 1;
  g.scale_a_numel = g.scale_b_numel = 1;
  g.in_dtype = p.x_dtype;
  g.out_dtype = channel_scales ? p.x_dtype : p.out_dtype;
  if (int rc = launch_scaled_gemm_mfma_16bit(g, stream)) return rc;
  if (!channel_scales) return CONCH_OK;
  // 3. channel scale, in place
  const int64_t total = p.m * p.n;
  const unsigned blocks = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
#define CONCH_LAUNCH(X, O) \
  hipLaunchKernelGGL((colscale_kernel<X, O>), dim3(blocks), dim3(256), 0, stream, (uint16_t*)p.c, p.m, p.n, p.c_stride_m, (const uint16_t*)channel_scales)
  if (fp16 && p.out_dtype == CONCH_DT_FP16) CONCH_LAUNCH(CONCH_DT_FP16, CONCH_DT_FP16);
  else if (fp16) CONCH_LAUNCH(CONCH_DT_FP16, CONCH_DT_BF16);
  else if (p.out_dtype == CONCH_DT_BF16) CONCH_LAUNCH(CONCH_DT_BF16, CONCH_DT_BF16);
  else CONCH_LAUNCH(CONCH_DT_BF16, CONCH_DT_FP16);
#undef CONCH_LAUNCH
  return check_launch("colscale");
}

}  // namespace conch
