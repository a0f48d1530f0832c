/*
 * conch_amd.h -- C ABI of the MI355X-native quantized-matmul hot path.
 *
 * This is the drop-in boundary beneath Conch's `conch.ops.quantization` Python API.  The reference
 * has no FFI of its own for this path: its device boundary is the Triton launchers
 *   conch/kernels/quantization/gemm.py:564  scaled_gemm_launcher
 *   conch/kernels/quantization/gemm.py:482  mixed_precision_gemm_launcher
 *   conch/kernels/quantization/int8.py:63   static_scaled_int8_quant_launcher
 *   conch/kernels/quantization/fp8.py:65    static_scaled_fp8_quant_launcher
 * Each entry point below replaces exactly one of those launchers (plus the bias `add_` of
 * conch/ops/quantization/gemm.py:249-250, which is fused here).  Plain pointers and sizes only:
 * no torch types, no exceptions.  All pointers are DEVICE pointers unless stated; every call is
 * asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream) and
 * re-entrant.  Launchers never allocate USER-VISIBLE memory (docs/conch/structure.md:14-15 of the reference).
 *
 * Library scratch.  Some paths need device scratch the caller does not see: the split-K slabs of the skinny-M kernels
 * (M <= 256 scaled, M <= 256 mixed, M <= 256 4-bit blockwise) and of the split-K form of the LDS-tiled mixed kernel (M > 256 with
 * few tiles), K-contiguous copies of operands in non-native layouts, the bf16 expansion of e4m3fnuz operands, the dequantised
 * weight of the dequantise-first paths (kernel modes, 4-bit matmul above 128 rows), the quantised activations and the [M][2d]
 * intermediate of a pair that cannot use its fused form.  It is owned by the
 * library, one buffer per (device, stream, slot), allocated with hipMalloc the first time a call needs more than the
 * slot holds and never freed or moved afterwards (enqueued and graph-captured launches keep using the old buffer).
 * Consequences: (1) a call that must grow a slot does a hipMalloc -- no stream or device synchronisation -- and so is
 * NOT legal inside hipStreamBeginCapture; it fails with CONCH_ERR_INVALID_ARGUMENT there instead of breaking the
 * capture; (2) after conch_reserve_scratch(stream, bytes) with bytes >= conch_*_workspace_bytes() of every shape that
 * will be used, every entry point is allocation-free, synchronisation-free and graph-capture safe on that stream.
 * (3) Scratch belongs to the STREAM the call was made on, and that includes the arrival counters and partial-sum slabs of
 * the one-launch split-K forms, whose addresses are baked into a captured graph.  Replay a captured graph on its CAPTURE
 * stream (or make sure no other conch call of that stream's is in flight while it replays elsewhere): two launches that
 * share counters and slabs and are not ordered by a stream race.  The counters are zero between launches by construction
 * (the last arriver of a tile puts its counter back); after a launch that did not complete (device fault / reset) call
 * conch_reset_scratch(stream) before using that stream's one-launch forms again.
 * The tiled scaled-GEMM MFMA kernels on native layouts (K-contiguous A and B^T, K % 128 == 0, M > 256) never use scratch.
 * The device a call runs on is the CURRENT device (hipSetDevice), which must be the one the pointers live on.
 *
 * Return value: CONCH_OK (0) or a conch_status_t error; conch_last_error() returns a
 * thread-local, human-readable description of the last failure on the calling thread.
 */
#ifndef CONCH_AMD_H
#define CONCH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CONCH_AMD_ABI_VERSION 3

typedef enum conch_status {
  CONCH_OK = 0,
  CONCH_ERR_INVALID_ARGUMENT = 1, /* NULL pointer, negative size, bad enum, misaligned stride */
  CONCH_ERR_UNSUPPORTED = 2,      /* valid request that no kernel implements (dtype combination ...) */
  CONCH_ERR_HIP = 3,              /* a HIP runtime call failed; see conch_last_error() */
  CONCH_ERR_NO_DEVICE = 4         /* no gfx950 device visible */
} conch_status_t;

/* Element types.  Values 0..8 follow the reference's DType enum
 * (conch/kernels/quantization/gemm.py:20-31); 9 is added because gfx950's native fp8 is OCP
 * e4m3fn while the reference's AMD choice (MI300X) was e4m3fnuz (conch/ops/quantization/fp8.py:27). */
typedef enum conch_dtype {
  CONCH_DT_FP32 = 0,
  CONCH_DT_FP16 = 1,
  CONCH_DT_BF16 = 2,
  CONCH_DT_FP8_E4M3FN = 3, /* OCP fp8: gfx950 MFMA-native */
  CONCH_DT_INT8 = 4,
  CONCH_DT_UINT8 = 5,
  CONCH_DT_INT32 = 6,
  CONCH_DT_UINT32 = 7,
  CONCH_DT_FP8_E5M2 = 8,
  CONCH_DT_FP8_E4M3FNUZ = 9 /* MI300-era fp8: accepted through an exact compat path */
} conch_dtype_t;

/* How the zero-point argument of the mixed-precision GEMM is to be read
 * (conch/ops/quantization/gemm.py:34-38,92: None -> SYMMETRIC_NO_SHIFT, tensor -> WITH_SHIFT,
 * numel()==1 -> zero_is_scalar). */
typedef enum conch_zp_mode {
  CONCH_ZP_NONE = 0,
  CONCH_ZP_SCALAR = 1, /* one int32 on the device */
  CONCH_ZP_TENSOR = 2  /* int32 [K/group_size, N] */
} conch_zp_mode_t;

/* Weight-group and channel-scale modes of the reference kernel (conch/kernels/quantization/gemm.py:34-41, :68-74; same
 * values).  conch.ops only ever selects SYMMETRIC_NO_SHIFT / SYMMETRIC_WITH_SHIFT with CHANNEL_NONE (mixed) and
 * WEIGHT_AND_ACTIVATION (scaled); the others are reachable through the launchers with hand-built metadata. */
typedef enum conch_group_mode {
  CONCH_GROUP_NONE = 0,                 /* W = q - bias */
  CONCH_GROUP_SHIFT = 1,                /* W = (q - bias) - zeros */
  CONCH_GROUP_SYMMETRIC_NO_SHIFT = 2,   /* W = (q - bias) * scales */
  CONCH_GROUP_SYMMETRIC_WITH_SHIFT = 3, /* W = ((q - bias) - zeros) * scales */
  CONCH_GROUP_ASYMMETRIC = 4            /* W = fma(q - bias, scales, zeros) */
} conch_group_mode_t;

typedef enum conch_channel_mode {
  CONCH_CHANNEL_NONE = 0,
  CONCH_CHANNEL_WEIGHT_ONLY = 1,
  CONCH_CHANNEL_ACTIVATION_ONLY = 2,
  CONCH_CHANNEL_WEIGHT_AND_ACTIVATION = 3
} conch_channel_mode_t;

/* What `zeros` points to in conch_mixed_precision_gemm_modes (the reference converts whatever it loads to the meta dtype,
 * kernels/quantization/gemm.py:363-371). */
typedef enum conch_zeros_kind {
  CONCH_ZEROS_NONE = 0,
  CONCH_ZEROS_SCALAR_INT32 = 1,
  CONCH_ZEROS_TENSOR_INT32 = 2, /* int32 [K/G][N] */
  CONCH_ZEROS_SCALAR_FLOAT = 3, /* one element of x_dtype */
  CONCH_ZEROS_TENSOR_FLOAT = 4  /* x_dtype [K/G][N] */
} conch_zeros_kind_t;

/* Kernel-selection knobs for tests and benchmarks (process-global, not part of the data path). */
typedef enum conch_tuning_key {
  CONCH_TUNE_GEMM_VARIANT = 0 /* 0 = auto, 1 = generic (any shape/stride), 2 = LDS-tiled MFMA with a plain
                                 double-buffered loop, 3 = ping-pong pipelined LDS-tiled MFMA, four phases per
                                 K step, 4 = skinny-M (M <= 256) split-K MFMA, 5 = ping-pong, two phases per K
                                 step (the default for M > 256), 6 = 128x128 tiles, two workgroups per CU (shapes with few
                                 256x256 tiles), 7 = 256x352 tiles, one wave per SIMD (hand-allocated assembly; e4m3fn or int8,
                                 K % 256 == 0, N % 16 == 0).  mixed_precision_gemm has one LDS-tiled MFMA
                                 kernel: 1 = generic, any other value = that kernel */
  ,
  CONCH_TUNE_MIXED_TILE_NT = 1 /* mixed_precision_gemm tile shape: 0 = auto, 2..4 = force 256 rows x 64 NT columns, 5 = force 512 x 128
                                  (LDS-tiled kernel only) */
  ,
  CONCH_TUNE_SKINNY_NO_SPLITK = 2 /* 1 = keep the skinny-M scaled GEMM's K split inside the workgroup */
  ,
  CONCH_TUNE_SKINNY_MODE = 3 /* split-K skinny-M scaled GEMM and the mixed decode-batch kernel: 0 = auto (one launch at GEMV sizes,
                                M <= 32, and for 64 < M <= 128 where it fills one round of the chip), 1 = two launches (partial
                                sums, then a reduce kernel), 2 = ONE launch (the last-arriving slice of a tile reduces it), 1024-byte
                                K slices, 3 = one launch, 2048-byte slices and <= 64-row blocks (scaled kernel only).  (The column-strip mixed kernel's K-split
                                forms always use the slab reduce kernel: their one-launch form is a build-time experiment.) */
  ,
  CONCH_TUNE_TILE_SCHEDULE = 4 /* 256x256-tile scaled GEMM: 0 = auto (= 1), 1 = uniform 256-column tiles, 2 = two tile
                                 widths (256-column tiles, then 192-column tiles: no idle last round), wide first on
                                 every XCD, 3 = the same with odd XCDs walking their narrow tiles first */
  ,
  CONCH_TUNE_PERSISTENT = 5 /* 256x256-tile scaled GEMM: 0 = auto (= 1), 1 = one workgroup per tile, 2 = persistent
                               workgroups (one per CU) that prefetch the next tile's first K steps under the epilogue,
                               n > 2 = persistent with n workgroups (test hook) */
  ,
  CONCH_TUNE_EPILOGUE = 6 /* 256x256-tile scaled GEMM (two-phase ping-pong): 0 = auto (= 2), 1 = direct 16-byte stores from the
                             accumulator layout (16 lines per quarter-wave), 2 = sub-tile passed through LDS sixteen rows at a
                             time and stored as whole 128-byte rows, write-through (whole tiles; others store directly) */
  ,
  CONCH_TUNE_DIAG = 7 /* diagnostic twin of the library only (-DCONCH_CLOCK_PROBE): timing experiments of the mixed K loop with
                         parts removed (wrong results); ignored by the product library */
  ,
  CONCH_TUNE_MID_STAGES = 8 /* 128x128-tile scaled GEMM: 0 = auto (the 4-stage ring, one workgroup per CU, when there is at
                               most one tile per CU; else the 2-stage loop, two workgroups per CU), 2 / 4 = force */
  ,
  CONCH_TUNE_MIXED_SPLITK = 9 /* LDS-tiled and column-strip mixed_precision_gemm: K slices per tile (fp32 slabs + reduce kernel) when
                                 the tiles leave most of the chip idle: 0 = auto, 1 = never, 2 / 4 / 8 = force (the strip kernel
                                 takes any count from 2 to 8) */
  ,
  CONCH_TUNE_SKINNY_GATHER = 10 /* fetch order of the decode kernels' operands (all forms bit-identical):
                                   - split-K skinny-M scaled GEMM (gemm_skinny.hip), the wave's B^T fragments: 0 = auto (2 when the
                                     launch is at most one workgroup per CU, or for 32-row blocks with 8-step slices), 1 = in MFMA
                                     operand order (16 rows x 16 bytes per quarter-wave), 2 = four lanes per 64-byte row piece, put
                                     into operand order by a cross-lane gather;
                                   - int4 decode-batch kernel of mixed_precision_gemm (gemm_mixed_skinny.hip), the packed words:
                                     1 = operand order into registers; 2 = 16 bytes per lane through LDS (LDS-DMA); 0 = auto = the
                                     LDS form when the launch is at most one workgroup per CU or the block has 64 rows.  The LDS form
                                     needs a 16-byte aligned w_q with wq_stride_k % 4 == 0 and int4 weights: otherwise registers,
                                     whatever the key says;
                                   - 4-bit decode GEMM of conch_bnb_gemm_4bit (bnb.hip, <= 32 rows): 1 = codes and absmax in operand
                                     order into registers, anything else = LDS-DMA staging + gather; a packed-weight pointer that is
                                     not 16-byte aligned takes the register form whatever the key says */
  ,
  CONCH_TUNE_MIXED_KERNEL = 11 /* mixed_precision_gemm above the decode sizes (any M with a non-zero CONCH_TUNE_GEMM_VARIANT other than 1 / 4):
                                  0 = auto (from 33 rows the strip kernel's tile height / width / K slices race the other kernels by
                                  cost model), 1 = the LDS-tiled kernel (dequantised weights pass
                                  through LDS: gemm_mixed.hip), 2 = the column-strip kernel (every wave dequantises its own 16 / 32
                                  columns straight into MFMA operand registers: gemm_mixed_strip.hip) wherever its contract holds, 3 = the
                                  one-wave-per-SIMD assembly form of it (csrc/asm/gen_mixed1w.py: fp16, 4-bit weights, no zero points,
                                  group 128; CONCH_TUNE_MIXED_TILE_NT 3 / 4 picks its 192- / 256-column tile) */
  ,
  CONCH_TUNE_MID_SPLITK = 12 /* 128x128-tile scaled GEMM: K slices per tile (fp32 / int32 slabs + the split-K reduce kernel) when the
                               tiles leave at least half the chip idle and K >= 4096: 0 = auto, 1 = never, 2..8 = that many */,
  CONCH_TUNE_SKINNY_CHUNKS = 13 /* split-K skinny-M scaled GEMM: passes of one 1024- / 2048-byte K unit a workgroup makes before it
                                  leaves its partial sums (fewer slabs for a long K): 0 = auto, n >= 1 = that many */,
  CONCH_TUNE_MIXED_STRIP_ROWS = 14 /* column-strip mixed_precision_gemm (gemm_mixed_strip.hip): rows of its tile: 0 = auto (64 / 128
                                      rows for M <= 64 / 128 in the K-split forms, else 256), 64 / 128 / 256 = force */,
  CONCH_TUNE__COUNT = 15 /* number of keys (array bound; not a key) */
} conch_tuning_key_t;

int conch_abi_version(void);
const char* conch_last_error(void);
int conch_set_tuning(int key, int value);
int conch_get_tuning(int key);

/* Number of gfx950 devices visible (0 if none); does not initialise a HIP context on failure. */
int conch_device_count(void);

/* Upper bound of the library scratch, per slot, any call with this shape may need (see "Library scratch" above).
 * For the fused FFN entry points pass n = 2 * n_out. */
int64_t conch_scaled_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k);
int64_t conch_mixed_precision_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k);
/* Grow every scratch slot of (current device, stream) to at least `bytes` now (hipMalloc; not inside capture). */
int conch_reserve_scratch(void* stream, int64_t bytes);
/* Zero the split-K arrival counters of (current device, stream) with a memset enqueued on `stream` (recovery after an
 * incomplete launch; see "Library scratch" (3)).  A stream that never used a one-launch form has none: CONCH_OK. */
int conch_reset_scratch(void* stream);

/*
 * static_scaled_int8_quant  (replaces kernels/quantization/int8.py:63-97;
 * semantics = reference/quantization/int8.py:12-18)
 *   out[t][h] = (int8) trunc( clamp( (float)x[t][h] * (1.0f / *scale), -128, 127 ) )
 * x: [tokens][hidden] of x_dtype (FP16/BF16/FP32), rows x_row_stride elements apart, hidden
 * contiguous; out: int8 [tokens][hidden], rows out_row_stride apart; scale: one fp32 on device.
 */
int conch_static_scaled_int8_quant(int8_t* out, const void* x, const float* scale, int64_t tokens,
                                   int64_t hidden, int64_t x_row_stride, int64_t out_row_stride,
                                   int x_dtype, void* stream);
/*
 * The same with the dtype of the PRODUCT made explicit: CONCH_DT_FP32 (the call above), or x_dtype (FP16 / BF16):
 *   out[t][h] = (int8) trunc( clamp( x_dtype( (float)x[t][h] * (1.0f / *scale) ), -128, 127 ) )
 * -- what reference/quantization/int8.py:16 computes when `scale` is a 0-dim tensor: torch's type promotion ignores a 0-dim
 * operand of the same category, so `x * scale.reciprocal()` is rounded to x's dtype (the fp32 reciprocal is not).  The
 * Python wrapper picks it by scale.dim() == 0; the (1,)-shaped scale of the reference's tests and benchmarks gives FP32.
 */
int conch_static_scaled_int8_quant_typed(int8_t* out, const void* x, const float* scale, int64_t tokens,
                                         int64_t hidden, int64_t x_row_stride, int64_t out_row_stride,
                                         int x_dtype, int product_dtype, void* stream);

/*
 * static_scaled_fp8_quant  (replaces kernels/quantization/fp8.py:65-97;
 * semantics = reference/quantization/fp8.py:12-18, i.e. WITH the clamp to +-max)
 *   out[t][h] = fp8_rne( clamp( (float)x[t][h] * (1.0f / *scale), -max, max ) )
 * fp8_dtype: CONCH_DT_FP8_E4M3FN (max 448) or CONCH_DT_FP8_E4M3FNUZ (max 240).
 */
int conch_static_scaled_fp8_quant(uint8_t* out, const void* x, const float* scale, int64_t tokens,
                                  int64_t hidden, int64_t x_row_stride, int64_t out_row_stride,
                                  int x_dtype, int fp8_dtype, void* stream);

/*
 * Dynamic per-token quantisation.  The reference's wrappers stop at
 * `scale is None -> NotImplementedError` (conch/ops/quantization/int8.py:42-44, fp8.py:46-48); these entry points are
 * what a launcher behind that branch would bind.  Per token row t:
 *   scale_out[t] = max_h |x[t][h]| / QMAX  (fp32; QMAX = 127 | 448 e4m3fn | 240 e4m3fnuz; 1.0 for an all-zero row)
 *   out[t][h]    = the static op above applied with scale_out[t]
 * scale_out: fp32 [tokens], contiguous -- viewed as (tokens, 1) it is the per-row scale_a of conch_scaled_gemm.
 */
int conch_dynamic_scaled_int8_quant(int8_t* out, float* scale_out, const void* x, int64_t tokens, int64_t hidden,
                                    int64_t x_row_stride, int64_t out_row_stride, int x_dtype, void* stream);
int conch_dynamic_scaled_fp8_quant(uint8_t* out, float* scale_out, const void* x, int64_t tokens, int64_t hidden,
                                   int64_t x_row_stride, int64_t out_row_stride, int x_dtype, int fp8_dtype,
                                   void* stream);

/*
 * scaled_gemm  (replaces kernels/quantization/gemm.py:564-627 + ops/quantization/gemm.py:249-250;
 * semantics = reference/quantization/scaled_gemm.py:12-27)
 *   C[m][n] = out_dtype( scale_b[n] * ( scale_a[m] * (float) sum_k A[m][k]*B[k][n] ) ) (+ bias[n])
 * A: [M][K], B: [K][N], both of in_dtype (INT8 | FP8_E4M3FN | FP8_E4M3FNUZ), arbitrary element
 * strides (the reference passes B as a transposed view, strides (1, K)); C: [M][N] of out_dtype
 * (FP16 | BF16); scale_a: fp32, scale_a_numel = 1 or M (element stride 1); scale_b likewise with N;
 * bias: NULL or [N] of out_dtype, added in out_dtype after the cast exactly as the reference does.
 * int8 accumulates exactly in int32; fp8 accumulates in fp32.
 * scale_a == NULL / scale_b == NULL mean "no scale on that side" (a constant 1): the launcher's channel-scale modes
 * ACTIVATION_ONLY (scale_b NULL), WEIGHT_ONLY (scale_a NULL) and NONE (both NULL), kernels/quantization/gemm.py:408-440.
 */
int conch_scaled_gemm(void* c, const void* a, const void* b, const float* scale_a,
                      const float* scale_b, const void* bias, int64_t m, int64_t n, int64_t k,
                      int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                      int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n,
                      int64_t scale_a_numel, int64_t scale_b_numel, int in_dtype, int out_dtype,
                      void* stream);

/*
 * The same product stored into 1 + n_more buffers: `c` and c_more[0 .. n_more-1] (n_more <= 7), every one a [M][N] result with the
 * SAME row stride, written at the same (row, column).  No reference counterpart (the reference has no multi-GPU code): this is the
 * N-sharded form of BASELINE config C5 with the all-gather done by the GEMM's epilogue -- rank r passes its column block of its own
 * result as `c` and the same block of every peer's result, mapped into its address space (HIP IPC / symmetric memory), as c_more;
 * the stores to peers leave over xGMI from the epilogue, nothing is staged or unpacked (conch_amd/distributed.py, direct mode).
 * Tile-kernel layout only (K-contiguous int8 / e4m3fn A and B^T, K % 128 == 0, K >= 256, 16-byte aligned rows, unit-stride C rows):
 * CONCH_ERR_UNSUPPORTED otherwise.  n_more == 0 is conch_scaled_gemm.  Every destination receives bit-identical values.
 */
int conch_scaled_gemm_multi(void* c, const void* a, const void* b, const float* scale_a, const float* scale_b, const void* bias,
                            int64_t m, int64_t n, int64_t k, int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                            int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n, int64_t scale_a_numel,
                            int64_t scale_b_numel, int in_dtype, int out_dtype, void* const* c_more, int n_more, void* stream);

/*
 * scaled_gemm_silu_and_mul  (the FFN pair `silu_and_mul(scaled_gemm(a, b, ...))` in one launch;
 * replaces scaled_gemm_launcher, kernels/quantization/gemm.py:564-627, followed by silu_and_mul_launcher,
 * conch/ops/activation/silu_and_mul.py:11-29; semantics = reference/quantization/scaled_gemm.py:12-27 then
 * reference/activation/silu_and_mul.py:13-16, every intermediate rounded to out_dtype as torch does)
 *   G = conch_scaled_gemm(...) with N = 2 * n_out columns [gate | up]          (never written to memory)
 *   C[m][j] = out_dtype( out_dtype( silu((float) G[m][j]) ) * (float) G[m][n_out + j] ),   j < n_out
 * B: [K][2 n_out]; scale_b_numel = 1 or 2 n_out; bias NULL or [2 n_out]; C: [M][n_out], unit column stride.
 * Layouts outside the tiled kernel's contract run the plain GEMM into library scratch followed by an elementwise pass.
 */
int conch_scaled_gemm_silu_and_mul(void* c, const void* a, const void* b, const float* scale_a,
                                   const float* scale_b, const void* bias, int64_t m, int64_t n_out, int64_t k,
                                   int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                                   int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n,
                                   int64_t scale_a_numel, int64_t scale_b_numel, int in_dtype, int out_dtype,
                                   void* stream);

/*
 * scaled_gemm_gelu_tanh_and_mul  (second activation: the GeGLU pair
 * `gelu_tanh_and_mul(scaled_gemm(a, b, ...))`, conch/ops/activation/gelu_tanh_and_mul.py after scaled_gemm; semantics =
 * reference/activation/gelu_tanh_and_mul.py:13-16, F.gelu(gate, approximate="tanh") * up, every intermediate rounded to out_dtype).
 * Arguments exactly as conch_scaled_gemm_silu_and_mul.
 */
int conch_scaled_gemm_gelu_tanh_and_mul(void* c, const void* a, const void* b, const float* scale_a,
                                        const float* scale_b, const void* bias, int64_t m, int64_t n_out, int64_t k,
                                        int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                                        int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n,
                                        int64_t scale_a_numel, int64_t scale_b_numel, int in_dtype, int out_dtype,
                                        void* stream);

/*
 * mixed_precision_gemm  (replaces kernels/quantization/gemm.py:482-545;
 * dequantisation semantics = kernels/quantization/gemm.py:176-216, bit-identical to the w_ref of
 * third_party/vllm/quant_utils.py:74)
 *   W[k][n] = x_dtype( (q[k][n] - weight_bias [- zp[k/G][n]]) ) * w_s[k/G][n]      (one rounding)
 *   C[m][n] = out_dtype( sum_k X[m][k] * W[k][n] )   accumulated in fp32
 * X: [M][K] of x_dtype (FP16 | BF16), row stride x_stride_m elements, K contiguous;
 * w_q_packed: int32 [K*bits/32][N] row-major (row stride wq_stride_k), element k of column n in
 * word k/(32/bits) at bit (k%(32/bits))*bits (quant_utils.py:119-122); w_s: [K/G][N] of x_dtype
 * (row stride ws_stride_g); w_zp per zp_mode (row stride wzp_stride_g); C: [M][N] of out_dtype.
 */
int conch_mixed_precision_gemm(void* c, const void* x, const int32_t* w_q_packed, const void* w_s,
                               const int32_t* w_zp, int64_t m, int64_t n, int64_t k,
                               int64_t x_stride_m, int64_t wq_stride_k, int64_t ws_stride_g,
                               int64_t wzp_stride_g, int64_t c_stride_m, int weight_bits,
                               int weight_bias, int group_size, int zp_mode, int x_dtype,
                               int out_dtype, void* stream);

/*
 * mixed_precision_gemm with an explicit weight-group mode, zero-point kind and channel-scale mode (what
 * mixed_precision_gemm_launcher, kernels/quantization/gemm.py:482-545, does when its metadata carries a mode conch.ops never
 * sets).  Every step of the dequantisation is rounded to x_dtype in the kernel's order (kernels/quantization/gemm.py:192-216):
 *   b = x_dtype(q) - weight_bias;  NONE: W = b;  SHIFT: W = b - z;  SYMMETRIC_NO_SHIFT: W = b * s;
 *   SYMMETRIC_WITH_SHIFT: W = (b - z) * s;  ASYMMETRIC: W = fma(b, s, z)
 * weight_bits in {1, 2, 4, 8, 16}; scales: x_dtype [K/G][N] (unused by NONE / SHIFT); zeros per zeros_kind;
 * channel_mode NONE, or WEIGHT_ONLY with channel_scales = x_dtype [N]:  C = out_dtype( x_dtype(acc) * channel_scales[n] )
 * (:408-416).  ACTIVATION_ONLY / WEIGHT_AND_ACTIVATION need activation scales the mixed launcher never passes: UNSUPPORTED.
 */
int conch_mixed_precision_gemm_modes(void* c, const void* x, const int32_t* w_q_packed, const void* scales, const void* zeros,
                                     const void* channel_scales, int64_t m, int64_t n, int64_t k, int64_t x_stride_m,
                                     int64_t wq_stride_k, int64_t scales_stride_g, int64_t zeros_stride_g, int64_t c_stride_m,
                                     int weight_bits, int weight_bias, int group_size, int group_mode, int zeros_kind,
                                     int channel_mode, int x_dtype, int out_dtype, void* stream);

/*
 * Weight pre-pack for the tile kernel (the reference's comparator pre-packs for ITS kernel,
 * benchmarks/mixed_precision_gemm_benchmark.py:59-75).  conch_prepack_mixed_weights rewrites the [K*bits/32][N] packed tensor
 * (kernels/quantization/gemm.py:320-326) ONCE, offline, into the order the K loop consumes it for tiles of 64 * tile_nt output
 * columns: per (tile column, K step of 64) one contiguous record of 512 threads x tile_nt x (bits/4) words, nibbles re-ordered so
 * that the in-register dequantisation needs no permute.  Same number of bytes as the source (rounded up to whole tiles).
 *   conch_mixed_gemm_tile_nt               the tile width (2, 3 or 4) the dispatcher picks for (M, N, K): pre-pack for it
 *   conch_prepacked_mixed_weights_bytes    size of the image (-1 on bad arguments)
 *   conch_unpack_mixed_weights             the inverse (round trip: unpack(prepack(w)) == w)
 *   conch_mixed_precision_gemm_prepacked   conch_mixed_precision_gemm on an image; bit-identical results; weight_bits 4 | 8,
 *                                          out_dtype == x_dtype, per-group zero points need tile_nt <= 3
 */
int64_t conch_prepacked_mixed_weights_bytes(int64_t k, int64_t n, int weight_bits, int tile_nt);
int conch_mixed_gemm_tile_nt(int64_t m, int64_t n, int64_t k, int weight_bits, int zp_mode);
int conch_prepack_mixed_weights(int32_t* image, const int32_t* w_q_packed, int64_t k, int64_t n, int64_t wq_stride_k,
                                int weight_bits, int tile_nt, void* stream);
int conch_unpack_mixed_weights(int32_t* w_q_packed, const int32_t* image, int64_t k, int64_t n, int64_t wq_stride_k,
                               int weight_bits, int tile_nt, void* stream);
int conch_mixed_precision_gemm_prepacked(void* c, const void* x, const int32_t* image, const void* w_s, const int32_t* w_zp,
                                         int64_t m, int64_t n, int64_t k, int64_t x_stride_m, int64_t ws_stride_g,
                                         int64_t wzp_stride_g, int64_t c_stride_m, int weight_bits, int weight_bias,
                                         int group_size, int zp_mode, int tile_nt, int x_dtype, int out_dtype, void* stream);

/*
 * mixed_precision_gemm_silu_and_mul  (SURVEY.md 8(f) N3 for the int4 / int8-weight FFN: the reference's pair
 * mixed_precision_gemm (kernels/quantization/gemm.py:482-545) + silu_and_mul (ops/activation/silu_and_mul.py:11-29)
 * in one launch).  w_q_packed / w_s / w_zp have 2 n_out columns [gate | up]; C: [M][n_out]:
 *   C[m][j] = out_dtype( out_dtype( silu((float) G[m][j]) ) * (float) G[m][n_out + j] ),  G = conch_mixed_precision_gemm(...)
 */
int conch_mixed_precision_gemm_silu_and_mul(void* c, const void* x, const int32_t* w_q_packed, const void* w_s,
                               const int32_t* w_zp, int64_t m, int64_t n_out, int64_t k,
                               int64_t x_stride_m, int64_t wq_stride_k, int64_t ws_stride_g,
                               int64_t wzp_stride_g, int64_t c_stride_m, int weight_bits,
                               int weight_bias, int group_size, int zp_mode, int x_dtype,
                               int out_dtype, void* stream);

/* The same pair with the gelu-tanh gate (arguments exactly as conch_mixed_precision_gemm_silu_and_mul). */
int conch_mixed_precision_gemm_gelu_tanh_and_mul(void* c, const void* x, const int32_t* w_q_packed, const void* w_s,
                               const int32_t* w_zp, int64_t m, int64_t n_out, int64_t k,
                               int64_t x_stride_m, int64_t wq_stride_k, int64_t ws_stride_g,
                               int64_t wzp_stride_g, int64_t c_stride_m, int weight_bits,
                               int weight_bias, int group_size, int zp_mode, int x_dtype,
                               int out_dtype, void* stream);

/*
 * bitsandbytes-style blockwise quantisation (replaces quantize_blockwise_launcher /
 * dequantize_blockwise_launcher of conch/kernels/quantization/bitsandbytes/{quantize,dequantize}_blockwise.py; semantics = the reference's PyTorch
 * implementation, conch/reference/quantization/bitsandbytes/).  quant_type: 0 = nf4, 1 = fp4 (two codes per byte, the
 * first element in the high nibble), 2 = 8-bit with a 256-entry fp32 code book `code` (NULL otherwise).  blocksize: a power
 * of two in 64..4096; block b covers elements [b * blocksize, (b+1) * blocksize) of the flat tensor of `n` elements.
 *   quantise:    absmax[b] = max |x| (stored in absmax_dtype);  code(x * (1 / absmax[b]))
 *   dequantise:  out[e] = out_dtype( absmax_dtype( value[code] * absmax[e / blocksize] ) )
 */
int conch_bnb_quantize_blockwise(uint8_t* out, void* absmax, const void* x, const float* code, int64_t n, int blocksize,
                                 int quant_type, int x_dtype, int absmax_dtype, void* stream);
int conch_bnb_dequantize_blockwise(void* out, const uint8_t* xq, const void* absmax, const float* code, int64_t n, int blocksize,
                                   int quant_type, int out_dtype, int absmax_dtype, void* stream);
/* C[M][N] = X[M][K] @ dequantise(W)^T for a 4-bit (nf4 / fp4) weight W[N][K]: dequantised into library scratch in x_dtype, then
 * the 16-bit MFMA tile kernel (fp32 accumulation).  K % 64 == 0. */
int64_t conch_bnb_gemm_4bit_workspace_bytes(int64_t m, int64_t n, int64_t k); /* library-scratch upper bound (conch_reserve_scratch) */
int conch_bnb_gemm_4bit(void* c, const void* x, const uint8_t* w_packed, const void* absmax, int64_t m, int64_t n, int64_t k,
                        int64_t x_stride_m, int64_t c_stride_m, int blocksize, int quant_type, int absmax_dtype, int x_dtype,
                        int out_dtype, void* stream);

/*
 * scaled_gemm on 16-bit ACTIVATIONS with a static per-tensor activation scale (second half):
 *   c = conch_scaled_gemm( static_scaled_{int8,fp8}_quant(x, scale_x), b, scale_a = scale_x, scale_b, bias )
 * bit for bit.  Where the skinny-M split-K kernel runs the shape (decode batches: M <= 256, N % 4 == 0, 16-byte aligned
 * K-contiguous rows) the quantisation happens on A's way into LDS -- no quantised copy of A in HBM, no second launch; other
 * shapes run the unfused pair through library scratch.  Replaces the pair of reference launchers
 * static_scaled_{int8,fp8}_quant_launcher (conch/kernels/quantization/int8.py:63-97, fp8.py:65-97) + scaled_gemm_launcher
 * (conch/kernels/quantization/gemm.py:564-627).  x_dtype: FP16 / BF16; quant_dtype: INT8 / FP8_E4M3FN (= b's dtype).
 */
int conch_static_quant_scaled_gemm(void* c, const void* x, const void* b, const float* scale_x, const float* scale_b, const void* bias,
                                   int64_t m, int64_t n, int64_t k, int64_t x_stride_m, int64_t x_stride_k, int64_t b_stride_k,
                                   int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n, int64_t scale_b_numel, int x_dtype,
                                   int quant_dtype, int out_dtype, void* stream);

/*
 * Timing helper used by bench.py: launches `iters` back-to-back scaled_gemm calls on `stream`
 * bracketed by HIP events recorded ON THAT STREAM and returns the average milliseconds per call
 * in *avg_ms (synchronises the stream; not for use inside graph capture).
 *
 * conch_set_timing_rotation(count, stride_bytes): launch i of conch_time_scaled_gemm / conch_time_mixed_precision_gemm[_prepacked]
 * reads its WEIGHTS (b / w_q_packed / image) at base + (i % count) * stride_bytes.  With count x (weight bytes) beyond the
 * 256 MiB Infinity Cache every launch streams its weights from HBM, as a decode step of a model does; count = 1 (the default)
 * re-reads one cache-resident buffer.  The caller owns the count buffers.  (tools/dispatch_cold_sweep.py)
 */
int conch_set_timing_rotation(int count, int64_t stride_bytes);
int conch_time_scaled_gemm(void* c, const void* a, const void* b, const float* scale_a,
                           const float* scale_b, const void* bias, int64_t m, int64_t n, int64_t k,
                           int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                           int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n,
                           int64_t scale_a_numel, int64_t scale_b_numel, int in_dtype,
                           int out_dtype, void* stream, int iters, float* avg_ms);
int conch_time_scaled_gemm_silu_and_mul(void* c, const void* a, const void* b, const float* scale_a,
                           const float* scale_b, const void* bias, int64_t m, int64_t n_out, int64_t k,
                           int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                           int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n,
                           int64_t scale_a_numel, int64_t scale_b_numel, int in_dtype,
                           int out_dtype, void* stream, int iters, float* avg_ms);

int conch_time_mixed_precision_gemm_prepacked(void* c, const void* x, const int32_t* image, const void* w_s,
                                              const int32_t* w_zp, int64_t m, int64_t n, int64_t k, int64_t x_stride_m,
                                              int64_t ws_stride_g, int64_t wzp_stride_g, int64_t c_stride_m, int weight_bits,
                                              int weight_bias, int group_size, int zp_mode, int tile_nt, int x_dtype,
                                              int out_dtype, void* stream, int iters, float* avg_ms);
int conch_time_mixed_precision_gemm(void* c, const void* x, const int32_t* w_q_packed,
                                    const void* w_s, const int32_t* w_zp, int64_t m, int64_t n,
                                    int64_t k, int64_t x_stride_m, int64_t wq_stride_k,
                                    int64_t ws_stride_g, int64_t wzp_stride_g, int64_t c_stride_m,
                                    int weight_bits, int weight_bias, int group_size, int zp_mode,
                                    int x_dtype, int out_dtype, void* stream, int iters,
                                    float* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* CONCH_AMD_H */
