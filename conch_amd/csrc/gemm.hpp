// Argument blocks shared by the GEMM kernels and the C-ABI dispatcher.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace conch {

// scaled_gemm: C = out( sb * (sa * (A @ B)) ) + bias.  Strides in ELEMENTS.
struct ScaledGemmArgs {
  void* c;
  const void* a;
  const void* b;
  const float* scale_a;
  const float* scale_b;
  const void* bias;
  int64_t m, n, k;
  int64_t a_stride_m, a_stride_k;
  int64_t b_stride_k, b_stride_n;
  int64_t c_stride_m, c_stride_n;
  int64_t scale_a_numel, scale_b_numel;
  int in_dtype, out_dtype;
  // != 0 = fused gate/up FFN form; the value is the gate activation (1 = silu, 2 = gelu-tanh) (conch_scaled_gemm_silu_and_mul): B, scale_b and bias have 2n columns [gate | up], C has
  // n columns, C[i][j] = silu(gemm[i][j]) * gemm[i][n + j] with the reference's roundings (oracle: scaled_gemm_silu_and_mul_ref)
  int fuse_silu = 0;
  // != 0 (CONCH_DT_FP16 / CONCH_DT_BF16): `a` holds 16-bit ACTIVATIONS (a_stride_m in elements) that the kernel quantises to
  // in_dtype on the fly with the static per-tensor scale scale_a[0] (scale_a_numel == 1), exactly as
  // static_scaled_{int8,fp8}_quant would -- conch_static_quant_scaled_gemm, skinny-M split-K kernel only
  int a_src_dtype = 0;
  // 128 x 128-tile kernel: 1 = whole tiles leave row-major through LDS (epilogue_rows.hpp), 0 = direct stores from the accumulator
  // layout (set by the launcher from CONCH_TUNE_EPILOGUE; the 256 x 256-tile kernels pick by instantiation)
  int rows_epilogue = 1;
  // reciprocal of the tile raster's one runtime divisor (GROUP_M x tile columns), made by the launcher (set_raster_divisor):
  // the tile kernels otherwise open with three integer divisions by runtime values, ~0.6 us of every workgroup's life
  // (profiles/r03/probe_setup.txt).  raster_shift < 0: not set, the kernel divides.
  uint32_t raster_magic = 0;
  int raster_shift = -1;
  // The e4m3fnuz operands of the reference's AMD platform (conch/ops/quantization/fp8.py:27,54) on gfx950's OCP fp8 MFMA: every
  // fnuz code except 0x80 (NaN), 0x7F and 0xFF (+-240) is, read as OCP e4m3fn, exactly TWICE its fnuz value -- normal numbers
  // (bias 8 against 7) and subnormals (2^-7 m/8 against 2^-6 m/8) alike -- so the fp8 kernels run on the raw bytes and the
  // exact factor 1/4 rides on scale_a (`acc_scale`, a power of two: every rounding of the epilogue commutes with it).
  // `gate` points at a device word a scan of both operands has set to 1 if any of the three codes occurs; a gated launch
  // returns at once unless (*gate != 0) == (gate_run_if != 0) -- the fp8 kernel runs if none does, the exact bf16-expansion
  // path (repack.hip) if one does, and the host never reads the word (no synchronisation, graph-capturable).
  float acc_scale = 1.0f;
  const int* gate = nullptr;
  int gate_run_if = 0;
  // N-sharded multi-GPU form, direct mode (conch_scaled_gemm_multi; conch_amd/distributed.py): the 256 x 256 tile kernel stores every
  // tile into `n_more` further buffers besides `c` -- the peer-mapped [M, N] results of the other ranks, same element (row,
  // column) and row stride in each -- so that the all-gather happens in the epilogue, over all xGMI links at once, and
  // nothing is staged or unpacked afterwards.  A separate instantiation of the kernel: the one-destination code is untouched.
  void* c_more[7] = {};
  int n_more = 0;
  // 128 x 128-tile kernel, split-K form (few tiles, long K: gemm_mid.hip): blockIdx.y = K slice of `split_steps` 128-byte steps,
  // the raw fp32 / int32 partial tiles go to `slabs` [slice][M][N] and the split-K reduce kernel of gemm_skinny.hip sums them in
  // slice order and applies the epilogue.  0 = the whole K in one workgroup (no slabs).
  int split_steps = 0;
  void* slabs = nullptr;
};

#ifdef __HIPCC__
// workgroup-uniform: true = this gated launch is the branch NOT taken
__device__ __forceinline__ bool gated_off(const int* gate, int run_if) {
  return gate != nullptr && ((*gate != 0) != (run_if != 0));
}
#endif

// n / d for n < 2^31 as mulhi(n, magic) >> shift (d not a power of two: shift = ceil(log2 d) - 1, magic = ceil(2^(32 + shift) / d),
// exact because n d < 2^(32 + shift + 1)), or n >> shift with magic = 0 (d = 2^shift).
inline void set_raster_divisor(uint32_t d, uint32_t* magic, int* shift) {
  if (d < 1) {  // no divisor: leave "not set" (the kernel divides)
    *magic = 0;
    *shift = -1;
    return;
  }
  int s = 0;
  while ((1u << s) < d) ++s;
  if ((1u << s) == d) {
    *magic = 0;
    *shift = s;
  } else {
    *shift = s - 1;
    const uint64_t num = (uint64_t)1 << (32 + s - 1);
    *magic = (uint32_t)((num + d - 1) / d);
  }
}

// mixed_precision_gemm: C = out( X @ dequant(Wq) ).  Strides in ELEMENTS of the respective array.
struct MixedGemmArgs {
  void* c;
  const void* x;
  const int32_t* w_q;
  const void* w_s;
  const int32_t* w_zp;
  int64_t m, n, k;
  int64_t x_stride_m;
  int64_t wq_stride_k;
  int64_t ws_stride_g;
  int64_t wzp_stride_g;
  int64_t c_stride_m;
  int bits, weight_bias, group_size, zp_mode;
  int x_dtype, out_dtype;
  // 1 = fused gate/up FFN form (conch_mixed_precision_gemm_silu_and_mul): Wq / w_s / w_zp have 2n columns [gate | up],
  // C has n columns, C[i][j] = silu(G[i][j]) * G[i][n + j] with G = the plain product rounded to out_dtype
  int fuse_silu = 0;
  // != 0: w_q is the image conch_prepack_mixed_weights made for tiles of 64 x `prepacked` columns (2, 3 or 4)
  int prepacked = 0;
  // split-K form of the LDS-tiled kernel (few tiles, M > 256): blockIdx.y = K slice of `split_steps` 64-element steps (a
  // multiple of group_size / 64), fp32 partial sums to slabs [slice][M][N] instead of the cast-and-store epilogue
  int split_steps = 0;
  float* slabs = nullptr;
  // column-strip kernel, K-split form in ONE launch: arrival counters, one per (tile, wave), zero between launches (library scratch);
  // the wave that draws a tile strip's last ticket adds the slices in slice order and stores C.  NULL = the slab reduce kernel follows
  unsigned* counters = nullptr;
  // whole tiles of the LDS-tiled kernel: 1 = row-major epilogue through LDS (whole-line write-through stores), 0 = direct stores
  // from the accumulator layout (set by the launcher from CONCH_TUNE_EPILOGUE)
  int rows_epilogue = 1;
  uint32_t raster_magic = 0;  // see ScaledGemmArgs
  int raster_shift = -1;
};

// gemm_generic.hip
int launch_scaled_gemm_generic(const ScaledGemmArgs& p, hipStream_t stream);
int launch_mixed_gemm_generic(const MixedGemmArgs& p, hipStream_t stream);

// gemm_mfma.hip -- LDS-tiled MFMA kernels.  `variant`: 2 = simple double-buffered loop,
// 3 = 8-phase ping-pong pipeline.  *_supported() say whether the layout contract is met.
bool scaled_gemm_mfma_supported(const ScaledGemmArgs& p);
int launch_scaled_gemm_mfma(const ScaledGemmArgs& p, int variant, hipStream_t stream);
// gemm_asm.hip -- the one-wave-per-SIMD 256 x 352-tile fp8 kernel (hand-allocated assembly, csrc/asm/gen_gemm1w.py; variant 7)
bool scaled_gemm_asm1w_supported(const ScaledGemmArgs& p);
int launch_scaled_gemm_asm1w(const ScaledGemmArgs& p, hipStream_t stream);
bool scaled_asm1w_loadable();  // the embedded code object is (or can be) loaded on the current device
bool mixed_asm1w_loadable();
void preload_asm_modules();    // load both assembly modules for the current device (conch_reserve_scratch: before any capture)
int scaled_asm1w_tile_columns(const ScaledGemmArgs& p);  // the tile width it would use on `p` (352 / 288 / 224)
bool scaled_asm1w_beats_tiles(const ScaledGemmArgs& p);  // cost rule: its 256 x 352 tiles against the 256 x 256 tiles (full rounds, fewer epilogues)
// gemm_mid.hip -- 128x128 tiles, two workgroups per CU, for shapes with few 256x256 tiles (variant 6); same contract
int launch_scaled_gemm_mid(const ScaledGemmArgs& p, hipStream_t stream);
int mid_split_slices(const ScaledGemmArgs& p);  // K slices that launcher will use for `p` (1 = no split-K)
int64_t mid_split_slab_bytes(int64_t m, int64_t n);  // scratch bound of that form (workspace queries)
// the dispatcher's cost model (capi.hip, choose_scaled_kernel): 0 = 256 x 256 tiles, 1 = 128 x 128 tiles, 2 = split-K skinny
int scaled_kernel_choice(const ScaledGemmArgs& p);
double scaled_tiles_estimate_us(const ScaledGemmArgs& p);  // the same model's time (us) for the better of the two tile kernels
int unit_scale(const float** out);                          // one fp32 1.0 on the current device ("no scale on this side")
// gemm_skinny.hip -- M <= 256: 128x16 blocks, K split over the waves, register streaming (variant 4)
bool scaled_gemm_skinny_supported(const ScaledGemmArgs& p);
bool scaled_gemm_skinny_fnuz_supported(const ScaledGemmArgs& f);  // e4m3fnuz operands (dtype rewritten to e4m3fn in `f`) on the split-K kernel, all codes in registers
int launch_scaled_gemm_skinny(const ScaledGemmArgs& p, hipStream_t stream);
// the split-K reduce kernel on its own: C = epilogue(sum over `slices` slabs [slice][M][N] of fp32 / int32 partial sums, in slice
// order); N % 4 == 0.  Used by the 128 x 128-tile kernel's split-K form (gemm_mid.hip)
int launch_splitk_reduce(const ScaledGemmArgs& p, const void* slabs, int slices, hipStream_t stream);
bool scaled_gemm_skinny_fused_supported(const ScaledGemmArgs& wide);  // silu_and_mul fused into the split-K reduce kernel
// repack.hip -- copy operands into the MFMA layout contract (stream-ordered scratch)
int repack_for_mfma(const ScaledGemmArgs& p, ScaledGemmArgs* q, void** scratch, hipStream_t stream);
int expand_fnuz_to_bf16(const ScaledGemmArgs& p, ScaledGemmArgs* q, hipStream_t stream, const int* gate = nullptr);
bool fnuz_expansion_fits(const ScaledGemmArgs& p);  // expand_fnuz_to_bf16 will take the problem (sizes, C layout)
// *flag = 1 if A or B^T (K-contiguous e4m3fnuz rows, the tile contract) holds a code the OCP fp8 MFMA cannot take: 0x80, 0x7F, 0xFF
// sa_quarter (optional): also writes 0.25 x scale_a[i] there (scale_a_numel floats), for kernels without acc_scale
int launch_fnuz_scan(const ScaledGemmArgs& p, int* flag, hipStream_t stream, float* sa_quarter = nullptr);
int launch_scaled_gemm_mfma_bf16(const ScaledGemmArgs& p, hipStream_t stream);
int launch_scaled_gemm_mfma_16bit(const ScaledGemmArgs& p, hipStream_t stream);  // p.in_dtype = FP16 | BF16, byte units
// gemm_modes.hip -- weight-group / channel-scale modes beyond the two conch.ops produces (dequantise first, then 16-bit MFMA)
int launch_mixed_gemm_modes(const MixedGemmArgs& p, const void* scales, const void* zeros, int64_t zeros_stride_g, int group_mode,
                            int zeros_kind, const void* channel_scales, hipStream_t stream);
// quant.hip -- elementwise silu(x[:, :n]) * x[:, n:] on a 16-bit [m][2n] matrix (the unfused tail of the FFN pair)
int launch_silu_and_mul(void* out, const void* x, int64_t m, int64_t n, int64_t x_stride_m, int64_t out_stride_m, int dtype, int act,
                        hipStream_t stream);
bool mixed_gemm_mfma_supported(const MixedGemmArgs& p);
bool mixed_gemm_silu_fused_supported(const MixedGemmArgs& p);
// weight pre-pack for the tile kernel (gemm_mixed.hip): image size in 32-bit words, pack / unpack, the width auto would pick
int64_t prepacked_weight_words(int64_t k, int64_t n, int bits, int nt);
int prepack_mixed_weights(uint32_t* image, uint32_t* plain, int64_t k, int64_t n, int64_t plain_stride, int bits, int nt, bool unpack,
                          hipStream_t stream);
int mixed_gemm_tile_nt(const MixedGemmArgs& p);
double mixed_tiles_estimate_us(const MixedGemmArgs& p);
int mixed_tiles_auto_split(const MixedGemmArgs& p);  // K slices the LDS-tiled kernel would use on `p` (automatic tuning)  // cost model: the LDS-tiled kernel with the tile shape and K split it would pick
int launch_mixed_gemm_mfma(const MixedGemmArgs& p, hipStream_t stream);
// gemm_mixed_strip.hip: every wave dequantises its own column strip straight into MFMA operand registers
bool mixed_gemm_strip_supported(const MixedGemmArgs& p);
bool mixed_strip_beats_tiles(const MixedGemmArgs& p);
int launch_mixed_gemm_strip(const MixedGemmArgs& p, hipStream_t stream);
double mixed_strip_estimate_us(const MixedGemmArgs& p);  // cost model: its K-split forms on one row of tiles (M <= 256)
void mixed_strip_plan_query(const MixedGemmArgs& p, int* rows, int* cols, int* slices);  // strip_plan's pick (diagnostics)
double mixed_strip_short_or_split_us(const MixedGemmArgs& p);  // ... its time if strip_plan leaves the unsplit 256-row tile, else 1e30
// gemm_asm.hip: the one-wave-per-SIMD int4 x fp16 assembly kernel (csrc/asm/gen_mixed1w.py; CONCH_TUNE_MIXED_KERNEL = 3 forces it)
bool mixed_gemm_asm1w_supported(const MixedGemmArgs& p);
int launch_mixed_gemm_asm1w(const MixedGemmArgs& p, hipStream_t stream);
// gemm_mixed_skinny.hip -- decode batches (M <= 256 by a cost rule, N % 4 == 0, any K % 64 == 0): weights straight to MFMA registers, split-K (variant 4)
bool mixed_gemm_skinny_supported(const MixedGemmArgs& p);
int launch_mixed_gemm_skinny(const MixedGemmArgs& p, hipStream_t stream);
int launch_f32_slab_reduce(void* c, const float* slabs, int slices, int64_t m, int64_t n, int64_t c_stride_m, int out_dtype, hipStream_t stream);

}  // namespace conch
