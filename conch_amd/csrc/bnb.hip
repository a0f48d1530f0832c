// bitsandbytes-style blockwise quantisation: NF4 / FP4 (two codes per byte, high nibble first) and the
// general 8-bit form with a 256-entry code book, per-block absmax scaling.
//
// Replaces the Triton kernels of conch/kernels/quantization/bitsandbytes/{quantize,dequantize}_blockwise.py (one program per
// block walking its elements ONE AT A TIME -- the reference's slowest kernels, README.md:66-67); arithmetic = the reference's
// PyTorch implementation (conch/reference/quantization/bitsandbytes/*.py; oracle/bnb.py pins it):
//   quantise    absmax[b] = max |x| over the block (exact, stored in the absmax tensor's dtype);  v = float(x) * (1.0f / float(absmax));
//               code = position of v among the format's decision thresholds (strict >), fp4 on |v| with the sign in bit 3;
//               8-bit: seven bisection steps over the code book from pivot 127, then the nearer neighbour
//   dequantise  out = out_dtype( absmax_dtype( value[code] * float(absmax[b]) ) )
// gfx950 design: 16-byte loads (8 fp16 / bf16 elements per lane), a block = a group of blocksize/8 lanes of one wave (up to
// 512 elements; larger blocks take 2 / 4 / 8 register-resident passes of the wave), segmented wave reduction by xor shuffles,
// one dword (4-bit) or two dwords (8-bit) stored per lane: one pass over HBM.  The code book sits in LDS.
#include <algorithm>

#include "common.hpp"
#include "gemm.hpp"
#include "mfma_tile.hpp"

namespace conch {
namespace {

enum { kQtNf4 = 0, kQtFp4 = 1, kQt8bit = 2 };

__device__ const float kNf4Values[16] = {(float)-1.0, (float)-0.6961928009986877, (float)-0.5250730514526367, (float)-0.39491748809814453,
                                         (float)-0.28444138169288635, (float)-0.18477343022823334, (float)-0.09105003625154495, (float)0.0,
                                         (float)0.07958029955625534, (float)0.16093020141124725, (float)0.24611230194568634, (float)0.33791524171829224,
                                         (float)0.44070982933044434, (float)0.5626170039176941, (float)0.7229568362236023, (float)1.0};
__device__ const float kFp4Values[16] = {(float)0.0, (float)0.0052083333, (float)0.6666666, (float)1.0, (float)0.333333, (float)0.5, (float)0.166666, (float)0.25,
                                         (float)-0.0, (float)-0.0052083333, (float)-0.666666, (float)-1.0, (float)-0.333333, (float)-0.5, (float)-0.166666, (float)-0.25};

// Constants are written as DOUBLE literals cast to float: the reference's Python floats reach fp32 through a double (several
// thresholds are exact midpoints of two fp32 NF4 values, i.e. fp32 rounding ties: a decimal -> float literal rounds some of
// them the other way, caught by the golden edge vectors)
__device__ __forceinline__ uint32_t nf4_code(float v) {
  // reference/.../quantize_blockwise.py:38-74: a binary search with strict > over these ascending thresholds; the code is the
  // number of thresholds below v.  Searched as a 4-level tree (4 compares + 11 selects instead of 15 compares + 15 adds).
  constexpr float t0 = (float)-0.8480964004993439, t1 = (float)-0.6106329262256622, t2 = (float)-0.4599952697753906,
                  t3 = (float)-0.33967943489551544, t4 = (float)-0.23460740596055984, t5 = (float)-0.13791173323988914,
                  t6 = (float)-0.045525018125772476, t7 = (float)0.03979014977812767, t8 = (float)0.1202552504837513,
                  t9 = (float)0.2035212516784668, t10 = (float)0.2920137718319893, t11 = (float)0.3893125355243683,
                  t12 = (float)0.5016634166240692, t13 = (float)0.6427869200706482, t14 = (float)0.8614784181118011;
  const bool b3 = v > t7;
  const bool b2 = v > (b3 ? t11 : t3);
  const float l3 = b3 ? (b2 ? t13 : t9) : (b2 ? t5 : t1);
  const bool b1 = v > l3;
  const float l4 = b3 ? (b2 ? (b1 ? t14 : t12) : (b1 ? t10 : t8)) : (b2 ? (b1 ? t6 : t4) : (b1 ? t2 : t0));
  const bool b0 = v > l4;
  return (b3 ? 8u : 0u) + (b2 ? 4u : 0u) + (b1 ? 2u : 0u) + (b0 ? 1u : 0u);
}

__device__ __forceinline__ uint32_t fp4_code(float v) {
  // :12-35: sign in bit 3, the magnitude's rank among seven thresholds (a 3-level tree) mapped through the format's code order
  const float a = fabsf(v);
  constexpr float t0 = (float)0.00260417, t1 = (float)0.0859375, t2 = (float)0.208333334, t3 = (float)0.29166667, t4 = (float)0.4166667,
                  t5 = (float)0.5833334, t6 = (float)0.83333334;
  const bool b2 = a > t3;
  const bool b1 = a > (b2 ? t5 : t1);
  const bool b0 = a > (b2 ? (b1 ? t6 : t4) : (b1 ? t2 : t0));
  const uint32_t r = (b2 ? 4u : 0u) + (b1 ? 2u : 0u) + (b0 ? 1u : 0u);
  const uint32_t by_rank = 0x32547610u;  // rank r -> code: 0,1,6,7,4,5,2,3 (nibble r)
  return ((by_rank >> (4 * r)) & 0xfu) + (v < 0.0f ? 8u : 0u);
}

__device__ __forceinline__ uint32_t code8(float x, const float* code) {
  // :77-119
  int pivot = 127, upper_pivot = 255, lower_pivot = 0;
  float lower = -1.0f, upper = 1.0f;
  float val = code[pivot];
#pragma unroll
  for (int step = 64; step >= 1; step >>= 1) {
    if (x > val) {
      lower_pivot = pivot;
      lower = val;
      pivot += step;
    } else {
      upper_pivot = pivot;
      upper = val;
      pivot -= step;
    }
    val = code[pivot];
  }
  if (upper_pivot == 255) upper = code[upper_pivot];
  if (lower_pivot == 0) lower = code[lower_pivot];
  if (x > val) {
    const float mid = pin_f32(upper + val) * 0.5f;
    return (uint32_t)(x > mid ? upper_pivot : pivot);
  }
  const float mid = pin_f32(lower + val) * 0.5f;
  return (uint32_t)(x < mid ? lower_pivot : pivot);
}

template <int DT>
__device__ __forceinline__ void store_scalar(void* p, int64_t i, float v) {
  if constexpr (DT == CONCH_DT_FP32) ((float*)p)[i] = v;
  else ((uint16_t*)p)[i] = float_to_bits16<DT>(v);
}

// 8 consecutive elements starting at `e0` (a multiple of 8); elements at or beyond `end` read as absent (valid = false).
// `wide`: the base pointer is 16-byte aligned (grid-uniform), so the 8-element group is too; a slice such as t[1:] of a
// contiguous tensor is not, and takes the per-element path like the ragged tail (the reference accepts any pointer)
template <int XDT>
__device__ __forceinline__ void load8(const void* x, int64_t e0, int64_t end, bool wide, float (&f)[8]) {
  if (wide && e0 + 8 <= end) {
    if constexpr (XDT == CONCH_DT_FP32) {
      const f32x4 a = *(const f32x4*)((const float*)x + e0), b = *(const f32x4*)((const float*)x + e0 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f[i] = a[i];
        f[4 + i] = b[i];
      }
    } else {
      const u16x8 v = *(const u16x8*)((const uint16_t*)x + e0);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = bits16_to_float<XDT>(v[i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = e0 + i < end ? load_as_float<XDT>(x, e0 + i) : 0.0f;
  }
}

constexpr int kBnbThreads = 256;

// PASSES x 512 = elements a wave covers per block sweep.  blocksize <= 512: PASSES = 1 and a wave holds 512 / blocksize blocks
// side by side (lane group = blocksize / 8 lanes); larger blocks: PASSES = blocksize / 512, one block per wave.
template <int XDT, int ADT, int QT, int PASSES>
__global__ __launch_bounds__(kBnbThreads) void bnb_quantize_kernel(uint8_t* __restrict__ out, void* __restrict__ absmax,
                                                                    const void* __restrict__ x, const float* __restrict__ code,
                                                                    int64_t n, int blocksize) {
  __shared__ float lcode[256];
  if constexpr (QT == kQt8bit) {
    lcode[threadIdx.x] = code[threadIdx.x];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (kBnbThreads / 64) + (threadIdx.x >> 6);
  const int64_t span = (int64_t)PASSES * 512;           // elements per wave
  const int64_t w0 = wave * span;                       // first element of this wave
  if (w0 >= n) return;
  const int bs_shift = __builtin_ctz((unsigned)blocksize);  // a power of two (check_common): the block of an element is a shift, not
                                                            // a 64-bit division per lane and pass (round 3: ~100 instructions each)
  const bool wide = ((uintptr_t)x & 15) == 0;
  float f[PASSES][8];
  float amax = 0.0f;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int64_t e0 = w0 + ps * 512 + lane * 8;
    // the block this lane's elements belong to ends at blk_end: elements of the NEXT block never enter this one's maximum
    const int64_t blk = e0 >> bs_shift;
    const int64_t blk_end = min((blk + 1) << bs_shift, n);
    load8<XDT>(x, e0, blk_end, wide, f[ps]);
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(f[ps][i]));
  }
  // reduce over the lane group that shares a block (all 64 lanes when the block is >= 512 elements)
  const int group = PASSES > 1 ? 64 : blocksize / 8;
  for (int off = 1; off < group; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  const int64_t e_first = w0 + lane * 8;
  const int64_t blk = e_first >> bs_shift;
  if (e_first < n && (lane & (group - 1)) == 0) {
    if constexpr (ADT == CONCH_DT_FP32) ((float*)absmax)[blk] = amax;
    else ((uint16_t*)absmax)[blk] = float_to_bits16<ADT>(amax);  // exact: amax is a value of the input dtype
  }
  const float inv = 1.0f / amax;  // IEEE division, like torch; an all-zero block gives inf and v = NaN (every comparison false)
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int64_t e0 = w0 + ps * 512 + lane * 8;
    if (e0 >= n) continue;
    const int64_t b = e0 >> bs_shift;
    const int64_t blk_end = min((b + 1) << bs_shift, n);
    const int64_t valid = blk_end - e0;  // elements of this lane that exist (> 0)
    uint32_t c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = f[ps][i] * inv;
      if constexpr (QT == kQtNf4) c[i] = nf4_code(v);
      else if constexpr (QT == kQtFp4) c[i] = fp4_code(v);
      else c[i] = code8(v, lcode);
    }
    if constexpr (QT == kQt8bit) {
      uint8_t* dst = out + e0;
      if (valid >= 8) {
        i32x2 pk;
        pk[0] = (int)(c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24));
        pk[1] = (int)(c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24));
        *(i32x2*)dst = pk;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i < valid) dst[i] = (uint8_t)c[i];
      }
    } else {
      // pair j of a block = elements 2j, 2j+1 -> one byte, first element in the high nibble; a block of odd length drops its
      // last element (quantize_blockwise.py:162); output byte of element e of block b: b * blocksize/2 + (e - b*blocksize)/2
      uint8_t* dst = out + (b << (bs_shift - 1)) + ((e0 - (b << bs_shift)) >> 1);
      const uint32_t word = ((c[0] << 4) | c[1]) | (((c[2] << 4) | c[3]) << 8) | (((c[4] << 4) | c[5]) << 16) | (((c[6] << 4) | c[7]) << 24);
      if (valid >= 8) {
        *(uint32_t*)dst = word;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (2 * j + 1 < valid) dst[j] = (uint8_t)(word >> (8 * j));
      }
    }
  }
}

// one thread = 8 consecutive output elements (4 packed bytes, or 8 code bytes)
template <int ODT, int ADT, int QT>
__global__ __launch_bounds__(kBnbThreads) void bnb_dequantize_kernel(void* __restrict__ out, const uint8_t* __restrict__ xq,
                                                                      const void* __restrict__ absmax, const float* __restrict__ code,
                                                                      int64_t n, int blocksize, int64_t out_stride_row, int64_t row_len) {
  __shared__ float lut[256];
  if constexpr (QT == kQt8bit) lut[threadIdx.x] = code[threadIdx.x];
  else if (threadIdx.x < 16) lut[threadIdx.x] = QT == kQtNf4 ? kNf4Values[threadIdx.x] : kFp4Values[threadIdx.x];
  __syncthreads();
  const int64_t e0 = ((int64_t)blockIdx.x * kBnbThreads + threadIdx.x) * 8;
  if (e0 >= n) return;
  const int64_t b = e0 >> __builtin_ctz((unsigned)blocksize);  // a power of two; 8 | blocksize: the eight elements share a block
  float am;
  if constexpr (ADT == CONCH_DT_FP32) am = ((const float*)absmax)[b];
  else am = bits16_to_float<ADT>(((const uint16_t*)absmax)[b]);
  // 4-bit inputs only ever hold an even number of elements (dequantize_blockwise.py: output_size even)
  const int64_t valid = min((int64_t)8, n - e0);
  uint32_t c[8];
  if constexpr (QT == kQt8bit) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = i < valid ? xq[e0 + i] : 0u;
  } else {
    // one 4-byte load of the thread's four packed bytes (64 lanes: 256 contiguous bytes per instruction) where the tensor allows it,
    // instead of four 1-byte loads at a 4-byte lane stride
    uint32_t word;
    if (valid == 8 && (((uintptr_t)xq) & 3) == 0) {
      word = *(const uint32_t*)(xq + e0 / 2);
    } else {
      word = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (2 * j < valid) word |= (uint32_t)xq[e0 / 2 + j] << (8 * j);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t byte = (word >> (8 * j)) & 0xffu;
      c[2 * j] = byte >> 4;
      c[2 * j + 1] = byte & 0xfu;
    }
  }
  // row-strided output (the GEMM's W^T scratch has the same rows): element e -> row e / row_len, column e % row_len
  const int64_t row = row_len >= n ? 0 : e0 / row_len, col = e0 - row * row_len;  // flat output: one "row" (no division)
  const int64_t o0 = row * out_stride_row + col;
  float r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r[i] = pin_f32(lut[c[i]] * am);
    if constexpr (ADT != CONCH_DT_FP32) r[i] = bits16_to_float<ADT>(float_to_bits16<ADT>(r[i]));  // the product lives in absmax's dtype
  }
  constexpr int kEltBytes = ODT == CONCH_DT_FP32 ? 4 : 2;
  char* dst = (char*)out + o0 * kEltBytes;
  if (valid == 8 && ((uintptr_t)dst & 15) == 0) {  // whole group, aligned: 16-byte stores
    if constexpr (ODT == CONCH_DT_FP32) {
      ((f32x4*)dst)[0] = f32x4{r[0], r[1], r[2], r[3]};
      ((f32x4*)dst)[1] = f32x4{r[4], r[5], r[6], r[7]};
    } else {
      i32x4 pk;
#pragma unroll
      for (int j = 0; j < 4; ++j) pk[j] = (int)pack2_bits16<ODT>(f32x2{r[2 * j], r[2 * j + 1]});
      *(i32x4*)dst = pk;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < valid) store_scalar<ODT>(out, o0 + i, r[i]);
  }
}

template <int XDT, int ADT, int QT>
int launch_quantize(uint8_t* out, void* absmax, const void* x, const float* code, int64_t n, int blocksize, hipStream_t stream) {
  const int passes = blocksize <= 512 ? 1 : blocksize / 512;
  const int64_t waves = (n + (int64_t)passes * 512 - 1) / ((int64_t)passes * 512);
  const dim3 grid((unsigned)((waves + 3) / 4)), block(kBnbThreads);
  switch (passes) {
    case 1: hipLaunchKernelGGL((bnb_quantize_kernel<XDT, ADT, QT, 1>), grid, block, 0, stream, out, absmax, x, code, n, blocksize); break;
    case 2: hipLaunchKernelGGL((bnb_quantize_kernel<XDT, ADT, QT, 2>), grid, block, 0, stream, out, absmax, x, code, n, blocksize); break;
    case 4: hipLaunchKernelGGL((bnb_quantize_kernel<XDT, ADT, QT, 4>), grid, block, 0, stream, out, absmax, x, code, n, blocksize); break;
    default: hipLaunchKernelGGL((bnb_quantize_kernel<XDT, ADT, QT, 8>), grid, block, 0, stream, out, absmax, x, code, n, blocksize); break;
  }
  return check_launch("bnb_quantize_blockwise");
}

template <int XDT, int ADT>
int quantize_qt(uint8_t* out, void* absmax, const void* x, const float* code, int64_t n, int blocksize, int qt, hipStream_t stream) {
  if (qt == kQtNf4) return launch_quantize<XDT, ADT, kQtNf4>(out, absmax, x, code, n, blocksize, stream);
  if (qt == kQtFp4) return launch_quantize<XDT, ADT, kQtFp4>(out, absmax, x, code, n, blocksize, stream);
  return launch_quantize<XDT, ADT, kQt8bit>(out, absmax, x, code, n, blocksize, stream);
}

template <int XDT>
int quantize_adt(uint8_t* out, void* absmax, const void* x, const float* code, int64_t n, int blocksize, int qt, int adt, hipStream_t stream) {
  if (adt == CONCH_DT_FP32) return quantize_qt<XDT, CONCH_DT_FP32>(out, absmax, x, code, n, blocksize, qt, stream);
  if (adt == CONCH_DT_FP16) return quantize_qt<XDT, CONCH_DT_FP16>(out, absmax, x, code, n, blocksize, qt, stream);
  return quantize_qt<XDT, CONCH_DT_BF16>(out, absmax, x, code, n, blocksize, qt, stream);
}

template <int ODT, int ADT>
int dequantize_qt(void* out, const uint8_t* xq, const void* absmax, const float* code, int64_t n, int blocksize, int qt,
                  int64_t out_stride_row, int64_t row_len, hipStream_t stream) {
  const dim3 grid((unsigned)((n + 8 * kBnbThreads - 1) / (8 * kBnbThreads))), block(kBnbThreads);
  if (qt == kQtNf4) hipLaunchKernelGGL((bnb_dequantize_kernel<ODT, ADT, kQtNf4>), grid, block, 0, stream, out, xq, absmax, code, n, blocksize, out_stride_row, row_len);
  else if (qt == kQtFp4) hipLaunchKernelGGL((bnb_dequantize_kernel<ODT, ADT, kQtFp4>), grid, block, 0, stream, out, xq, absmax, code, n, blocksize, out_stride_row, row_len);
  else hipLaunchKernelGGL((bnb_dequantize_kernel<ODT, ADT, kQt8bit>), grid, block, 0, stream, out, xq, absmax, code, n, blocksize, out_stride_row, row_len);
  return check_launch("bnb_dequantize_blockwise");
}

template <int ODT>
int dequantize_adt(void* out, const uint8_t* xq, const void* absmax, const float* code, int64_t n, int blocksize, int qt, int adt,
                   int64_t out_stride_row, int64_t row_len, hipStream_t stream) {
  if (adt == CONCH_DT_FP32) return dequantize_qt<ODT, CONCH_DT_FP32>(out, xq, absmax, code, n, blocksize, qt, out_stride_row, row_len, stream);
  if (adt == CONCH_DT_FP16) return dequantize_qt<ODT, CONCH_DT_FP16>(out, xq, absmax, code, n, blocksize, qt, out_stride_row, row_len, stream);
  return dequantize_qt<ODT, CONCH_DT_BF16>(out, xq, absmax, code, n, blocksize, qt, out_stride_row, row_len, stream);
}

// ---------------------------------------------------------------------------------------------
// Decode-batch GEMM on the 4-bit weight (M <= 256 by a cost rule): y = x @ dequantise(W)^T without materialising W.
// Same construction as gemm_mixed_skinny.hip: a workgroup owns (64 columns, one 1024-element K slice, a block of 16 / 32 / 64
// rows); the X slice is staged once by LDS-DMA; a lane of the MFMA's weight operand is (column n, 8-element k-group), and those
// eight codes are exactly ONE 32-bit word of row n of the packed tensor (byte j = code 2j << 4 | code 2j+1).  A code becomes a
// number through a table in LDS (256 entries: packed byte -> its two fp32 numbers) and the block's absmax, with the roundings of bnb_dequantize_kernel -- absmax dtype, then the activation dtype -- so the
// weights multiplied here are bit for bit the ones the dequantise-first path multiplies.  fp32 partial sums go to slabs; the
// fp32-slab reduce of gemm_mixed_skinny.hip adds the slices in slice order and casts.
// ---------------------------------------------------------------------------------------------
struct BnbGemmArgs {
  void* c;
  const void* x;
  const uint8_t* w;
  const void* absmax;
  int64_t m, n, k, x_stride_m, c_stride_m;
  int blocksize;
};

constexpr int kBgThreads = 256;
constexpr int kBgN = 64;        // columns per workgroup
constexpr int kBgStepK = 64;    // k elements per step (128 bytes of fp16 / bf16)
constexpr int kBgSteps = 16;    // steps per slice
constexpr int kBgSliceK = kBgSteps * kBgStepK;

// WLDS (rows <= 32): the packed codes take the LDS-DMA path too.  A lane of the MFMA's weight operand is (row n, k-group g), and
// W's rows are K-contiguous -- the four lanes of a quad sit on four different rows, four different 128-byte lines: the texture
// path takes a quad one line at a time, and a wave instruction brought 16 bytes from each of 16 lines (every line touched by
// eight instructions).  Here lane L fetches row L >> 2, one 16-byte chunk of the row's 64 bytes of two K steps (a quad = 64
// contiguous bytes; 8 instructions per wave instead of 32), straight into the wave's 8 KiB of LDS, and the operand lane reads its
// word back with ds_read_b32.  The chunk a lane fetches is rotated by (row >> 2) so that the read -- lane (r, g) wants chunk c of
// row r: dword 16 (r & 3) + 4 ((c + (r >> 2)) & 3) + g of its 64-dword row group -- touches 64 different banks.
// Blocks of 64 (bitsandbytes' default): the absmax of a row's 16 steps are 16 consecutive elements -- four loads of 16 rows x 4
// elements and a ds_bpermute per step instead of 16 loads of one element from each of 16 lines.
// B64 = blocks of 64 elements, a template parameter although the blocksize is a kernel argument: as a run-time branch in the issue
// phase the two absmax forms shared their registers, and hipcc waited for the code loads in flight before it let the 64-form
// write "its" copy of them (s_waitcnt vmcnt(4) ahead of the X slice's loads: a drain of the whole code stream per launch).
template <int X_DT, int ADT, int QT, int ROWS, bool WLDS, bool B64>
__global__ __launch_bounds__(kBgThreads) void bnb_decode_gemm_kernel(BnbGemmArgs p, float* __restrict__ slabs) {
  using namespace tile;
  constexpr int kUnit = ROWS * 128;  // one K step of X
  // pairs of steps whose codes are staged through LDS: all eight beside 16 / 32 rows of X; beside 64 rows (128 KiB) seven -- 28 KiB
  // + the 2 KiB table = 158 of the 160 KiB -- and the last pair takes the register path
  constexpr int kDmaPairs = !WLDS ? 0 : ROWS <= 32 ? kBgSteps / 2 : kBgSteps / 2 - 1;
  constexpr int kWBytes = 4 * kDmaPairs * 1024;  // per wave: one KiB per pair of steps
  __shared__ __attribute__((aligned(1024))) char lds[kBgSteps * kUnit + 2048 + kWBytes];
  // byte -> its two numbers (code byte >> 4 first): ONE 8-byte LDS read per packed byte instead of two 4-byte reads and twice the
  // address arithmetic (the loop is bound by vector instructions, not by the LDS)
  f32x2* lut = (f32x2*)(lds + kBgSteps * kUnit);  // (filled behind the slice's loads: its two table reads are loads too)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const int nw = blockIdx.x * kBgN + wave * 16;
  const int n = min(nw + r, (int)p.n - 1);  // this lane's weight row
  const int k0 = blockIdx.y * kBgSliceK;
  const int m0 = blockIdx.z * ROWS;
  const int kk = (int)p.k;
  // a ragged last slice: the steps past K load from an out-of-range VGPR offset (zeros for X in LDS; absmax 0 -> weights 0)
  const int valid_steps = min(kBgSteps, (kk - k0) / kBgStepK);
#define CONCH_KILL(s) ((s) < valid_steps ? 0 : (int)0x80000000)

  // ---- packed codes and absmax of the whole slice (first: they come from HBM, the X slice from L2); all 32-bit arithmetic
  constexpr int kAbsBytes = ADT == CONCH_DT_FP32 ? 4 : 2;
  const int bs_shift = __builtin_ctz((unsigned)p.blocksize);  // a power of two (check_common)
  const uint32_t w_bytes = (uint32_t)(p.n * p.k / 2);         // below 2 GiB (bnb_decode_supported)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.absmax, 0, (uint32_t)(((uint64_t)w_bytes * 2) >> bs_shift) * kAbsBytes, 0x00020000);
  char* wl = lds + kBgSteps * kUnit + 2048 + wave * (kWBytes / 4);  // this wave's code staging (WLDS)
  uint32_t wq[kBgSteps][2];  // register path only (a dependent bound makes hipcc's host pass drop the kernel's stub)
  if constexpr (kDmaPairs > 0) {
    const int rho = lane >> 2;
    const int chunk = ((lane & 3) - (rho >> 2)) & 3;
    const int vw = min(nw + rho, (int)p.n - 1) * (kk / 2) + 16 * chunk;
#pragma unroll
    for (int i = 0; i < kDmaPairs; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(wl + i * 1024), 16, vw | CONCH_KILL(2 * i), k0 / 2 + i * 64, 0, 0);
  }
  if constexpr (2 * kDmaPairs < kBgSteps) {
    const int vw = n * (kk / 2) + 4 * g;  // byte offset of the lane's k-group inside a half step
#pragma unroll
    for (int s = 2 * kDmaPairs; s < kBgSteps; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        wq[s][h] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rw, vw | CONCH_KILL(s), (k0 + s * kBgStepK + 32 * h) / 2, 0);
  }
  float am[kBgSteps];
  uint32_t am_raw[4] = {0, 0, 0, 0};
  constexpr bool blocks_of_64 = B64;
  if constexpr (blocks_of_64) {
    // lane L: row L >> 2, blocks 4 j + (L & 3) of the slice
    const int va = min(nw + (lane >> 2), (int)p.n - 1) * (kk >> 6) * kAbsBytes + (lane & 3) * kAbsBytes;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kill = 4 * j + (lane & 3) < valid_steps ? 0 : (int)0x80000000;
      if constexpr (ADT == CONCH_DT_FP32) am_raw[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ra, va | kill, ((k0 >> 6) + 4 * j) * 4, 0);
      else am_raw[j] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(ra, va | kill, ((k0 >> 6) + 4 * j) * 2, 0);
    }
  } else {
    const int va = n * (kk >> bs_shift) * kAbsBytes;  // ... of the row's first block
#pragma unroll
    for (int s = 0; s < kBgSteps; ++s) {
      const int blk = (k0 + s * kBgStepK) >> bs_shift;
      if constexpr (ADT == CONCH_DT_FP32)
        am[s] = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ra, va | CONCH_KILL(s), blk * 4, 0));
      else
        am[s] = bits16_to_float<ADT>((uint16_t)__builtin_amdgcn_raw_buffer_load_b16(ra, va | CONCH_KILL(s), blk * 2, 0));
    }
  }

  // ---- X slice -> LDS (8-row x 128-byte subtiles, source-side swizzle): the kBgSteps x kPieces (step, piece) items dealt
  // round-robin to the four waves without a branch (item t = wave + 4 j is step t / kPieces, piece t % kPieces)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.x, 0, (((uint32_t)p.m - 1) * (uint32_t)p.x_stride_m + (uint32_t)kk) * 2, 0x00020000);
  constexpr int kPieces = ROWS / 8;
  constexpr int kMine = kPieces > 4 ? 2 : 1;
  int voff_x[2];  // the first kMine are used (a dependent array bound makes hipcc's host pass drop the kernel's stub)
#pragma unroll
  for (int v = 0; v < kMine; ++v) {
    const int q = (int)(((unsigned)wave + 4u * v) % kPieces);
    const int row = 8 * q + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    voff_x[v] = min(m0 + row, (int)p.m - 1) * (int)p.x_stride_m * 2 + chunk * 16;
  }
#pragma unroll
  for (int j = 0; j < kBgSteps * kPieces / 4; ++j) {
    const unsigned t = (unsigned)wave + 4u * j;
    const int s = (int)(t / kPieces), q = (int)(t % kPieces);  // wave-uniform
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t*)(lds + s * kUnit + q * 1024), 16, voff_x[kMine == 2 ? (j & 1) : 0] | CONCH_KILL(s),
                                             (k0 + s * kBgStepK) * 2, 0, 0);
  }
#undef CONCH_KILL

  f32x4 acc[ROWS / 16];
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) acc[i] = f32x4{0, 0, 0, 0};

  {
    // the table, AFTER the slice's loads are in flight: filled first, its two reads from the constant table put an s_waitcnt
    // vmcnt(0) -- a memory round trip -- ahead of the first code load of every launch
    const float* t16 = QT == kQtNf4 ? kNf4Values : kFp4Values;
    lut[threadIdx.x] = f32x2{t16[threadIdx.x >> 4], t16[threadIdx.x & 15]};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // the X pieces and the codes of every wave and the table have landed

  if constexpr (blocks_of_64) {
    // step s = block s of the slice: element s & 3 of load s >> 2, held by lane 4 r + (s & 3)
#pragma unroll
    for (int s = 0; s < kBgSteps; ++s) {
      const uint32_t raw = (uint32_t)__builtin_amdgcn_ds_bpermute((4 * r + (s & 3)) * 4, (int)am_raw[s >> 2]);
      if constexpr (ADT == CONCH_DT_FP32) am[s] = __builtin_bit_cast(float, raw);
      else am[s] = bits16_to_float<ADT>((uint16_t)raw);
    }
  }

  // eight codes of one word -> eight numbers of the activation dtype: output dword j = (k 2j, k 2j + 1) = (high, low nibble of byte j)
  // the table reads of a step are all in flight together (lookup), then the arithmetic (scale8); pin_f32 keeps the fp32 product
  // and the conversion two roundings
  auto lookup = [&](uint32_t w, f32x2 (&t)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t addr = j == 0 ? (w << 3) & 0x7f8u : (w >> (8 * j - 3)) & 0x7f8u;  // byte j x 8
      t[j] = *(const f32x2*)((const char*)lut + addr);
    }
  };
  auto scale8 = [&](const f32x2 (&t)[4], float a) {
    i32x4 out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v0 = pin_f32(t[j][0] * a), v1 = pin_f32(t[j][1] * a);
      if constexpr (ADT != CONCH_DT_FP32) {  // the product lives in absmax's dtype (bnb_dequantize_kernel)
        v0 = bits16_to_float<ADT>(float_to_bits16<ADT>(v0));
        v1 = bits16_to_float<ADT>(float_to_bits16<ADT>(v1));
      }
      out[j] = (int)pack2_bits16<X_DT>(f32x2{v0, v1});
    }
    return out;
  };

  const int lane_off = (r >> 3) * 1024 + (r & 7) * 128 + ((g ^ ((r >> 1) & 7)) * 16);
  // WLDS: byte address, inside a pair of steps' KiB, of this lane's word in chunk c = 2 (s & 1) + h
  int w_read[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) w_read[c] = (4 * r + ((c + (r >> 2)) & 3)) * 16 + 4 * g;
#pragma unroll
  for (int s = 0; s < kBgSteps; ++s) {
    uint32_t w0, w1;
    if ((s >> 1) < kDmaPairs) {  // (a constant in every unrolled copy)
      w0 = *(const uint32_t*)(wl + (s >> 1) * 1024 + w_read[2 * (s & 1)]);
      w1 = *(const uint32_t*)(wl + (s >> 1) * 1024 + w_read[2 * (s & 1) + 1]);
    } else {
      w0 = wq[s][0];
      w1 = wq[s][1];
    }
    f32x2 t_lo[4], t_hi[4];
    lookup(w0, t_lo);
    lookup(w1, t_hi);
    const i32x4 w_lo = scale8(t_lo, am[s]);
    const i32x4 w_hi = scale8(t_hi, am[s]);
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
      const Frag fx = read_frag(lds, s * kUnit + lane_off + i * 2048);
      if constexpr (X_DT == CONCH_DT_FP16) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w_lo), __builtin_bit_cast(f16x8, fx.lo), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w_hi), __builtin_bit_cast(f16x8, fx.hi), acc[i], 0, 0, 0);
      } else {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w_lo), __builtin_bit_cast(bf16x8, fx.lo), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w_hi), __builtin_bit_cast(bf16x8, fx.hi), acc[i], 0, 0, 0);
      }
    }
  }

  // D rows = n (4g + e), D columns = m (lane % 16): four consecutive n of one row per lane
  float* slab = slabs + (int64_t)blockIdx.y * p.m * p.n;
#pragma unroll
  for (int i = 0; i < ROWS / 16; ++i) {
    const int m = m0 + i * 16 + r;
    const int nn = nw + 4 * g;
    if (m < p.m && nn + 4 <= p.n) *(f32x4*)(slab + (int64_t)m * p.n + nn) = acc[i];
  }
}

// CONCH_TUNE_SKINNY_GATHER 1 (the same question as in gemm_skinny.hip: how a wave fetches its weight operand) = the register path
// for the codes at every row count
template <int X_DT, int ADT, int QT, bool B64>
void launch_decode_rows_b(const BnbGemmArgs& p, int rows, dim3 grid, float* ws, hipStream_t stream) {
  const dim3 block(kBgThreads);
  const bool wlds = tuning(CONCH_TUNE_SKINNY_GATHER) != 1 && (((uintptr_t)p.w) & 15) == 0;  // 16-byte LDS-DMA pieces (K % 32 == 0 holds)
  if (rows == 16) {
    if (wlds) hipLaunchKernelGGL((bnb_decode_gemm_kernel<X_DT, ADT, QT, 16, true, B64>), grid, block, 0, stream, p, ws);
    else hipLaunchKernelGGL((bnb_decode_gemm_kernel<X_DT, ADT, QT, 16, false, B64>), grid, block, 0, stream, p, ws);
  } else if (rows == 32) {
    if (wlds) hipLaunchKernelGGL((bnb_decode_gemm_kernel<X_DT, ADT, QT, 32, true, B64>), grid, block, 0, stream, p, ws);
    else hipLaunchKernelGGL((bnb_decode_gemm_kernel<X_DT, ADT, QT, 32, false, B64>), grid, block, 0, stream, p, ws);
  } else {
    if (wlds) hipLaunchKernelGGL((bnb_decode_gemm_kernel<X_DT, ADT, QT, 64, true, B64>), grid, block, 0, stream, p, ws);
    else hipLaunchKernelGGL((bnb_decode_gemm_kernel<X_DT, ADT, QT, 64, false, B64>), grid, block, 0, stream, p, ws);
  }
}

// CONCH_TUNE_SKINNY_GATHER 1 (the same question as in gemm_skinny.hip: how a wave fetches its weight operand) = the register path
// for the codes at every row count
template <int X_DT, int ADT, int QT>
void launch_decode_rows(const BnbGemmArgs& p, int rows, dim3 grid, float* ws, hipStream_t stream) {
  if (p.blocksize == 64) launch_decode_rows_b<X_DT, ADT, QT, true>(p, rows, grid, ws, stream);
  else launch_decode_rows_b<X_DT, ADT, QT, false>(p, rows, grid, ws, stream);
}

template <int X_DT, int ADT>
void launch_decode_qt(const BnbGemmArgs& p, int qt, int rows, dim3 grid, float* ws, hipStream_t stream) {
  if (qt == kQtNf4) launch_decode_rows<X_DT, ADT, kQtNf4>(p, rows, grid, ws, stream);
  else launch_decode_rows<X_DT, ADT, kQtFp4>(p, rows, grid, ws, stream);
}

template <int X_DT>
void launch_decode_adt(const BnbGemmArgs& p, int qt, int adt, int rows, dim3 grid, float* ws, hipStream_t stream) {
  if (adt == CONCH_DT_FP32) launch_decode_qt<X_DT, CONCH_DT_FP32>(p, qt, rows, grid, ws, stream);
  else if (adt == CONCH_DT_FP16) launch_decode_qt<X_DT, CONCH_DT_FP16>(p, qt, rows, grid, ws, stream);
  else launch_decode_qt<X_DT, CONCH_DT_BF16>(p, qt, rows, grid, ws, stream);
}

// decode batches: up to four 64-row blocks, rows of four columns, whole blocks inside a row.  Above 64 rows every row block
// decodes the weight again, so the decode kernel runs only while that is cheaper than materialising the weight once and running the
// 16-bit tile kernel on it (fitted to profiles/r02/bnb_decode_times.txt, us: 5 + 0.85e-6 N K per row block against
// 15 K / 1024 + 0.7e-6 N K); `forced` = CONCH_TUNE_GEMM_VARIANT 4.
bool bnb_decode_supported(int64_t m, int64_t n, int64_t k, int blocksize, bool forced) {
  if (m > 256 || n % 4 || k % blocksize || n * k / 2 >= ((int64_t)1 << 31)) return false;
  if (m <= 64 || forced) return true;
  const double nk = (double)n * (double)k;
  // round 4: the 16-bit tile kernels split K too (gemm_mid.hip), so the tile side is the 8-bit dispatcher's model on the same problem
  // in byte units, not 15 us per 1024 of K (256 x 11008 x 4096: 96 us here against 64 for dequantise + GEMM: profiles/r04/fused_ops_sweep.txt)
  ScaledGemmArgs g{};
  g.m = m;
  g.n = n;
  g.k = k * 2;
  g.in_dtype = CONCH_DT_FP16;
  g.c_stride_n = 1;
  const double tiles_us = std::min(15.0 * (double)k / 1024.0, scaled_tiles_estimate_us(g));
  return (double)((m + 63) / 64) * (5.0 + 0.56e-6 * nk) < tiles_us + 0.7e-6 * nk;  // round 3: 0.85 -> 0.56 (64 x 4096 x 11008 42.3 -> 29.2 us)
}

int launch_bnb_decode_gemm(const BnbGemmArgs& p, int qt, int adt, int x_dtype, int out_dtype, hipStream_t stream) {
  const int slices = (int)((p.k + kBgSliceK - 1) / kBgSliceK);
  const int rows = p.m <= 16 ? 16 : p.m <= 32 ? 32 : 64;
  void* ws = nullptr;
  if (int rc = get_scratch(stream, kScratchMixedSplitK, (size_t)slices * p.m * p.n * 4, &ws)) return rc;
  const dim3 grid((unsigned)((p.n + kBgN - 1) / kBgN), (unsigned)slices, (unsigned)((p.m + rows - 1) / rows));
  if (x_dtype == CONCH_DT_FP16) launch_decode_adt<CONCH_DT_FP16>(p, qt, adt, rows, grid, (float*)ws, stream);
  else launch_decode_adt<CONCH_DT_BF16>(p, qt, adt, rows, grid, (float*)ws, stream);
  if (int rc = check_launch("bnb_decode_gemm")) return rc;
  return launch_f32_slab_reduce(p.c, (const float*)ws, slices, p.m, p.n, p.c_stride_m, out_dtype, stream);
}

bool float_dt(int dt) { return dt == CONCH_DT_FP32 || dt == CONCH_DT_FP16 || dt == CONCH_DT_BF16; }

int check_common(int64_t n, int blocksize, int qt, int adt, const void* code, const char* what) {
  CONCH_CHECK_ARG(n >= 0, "%s: negative size", what);
  CONCH_CHECK_ARG(blocksize >= 64 && blocksize <= 4096 && (blocksize & (blocksize - 1)) == 0,
                  "%s: blocksize %d (want a power of two in 64..4096)", what, blocksize);
  CONCH_CHECK_ARG(qt >= kQtNf4 && qt <= kQt8bit, "%s: quant type %d (0 = nf4, 1 = fp4, 2 = 8-bit code book)", what, qt);
  CONCH_CHECK_ARG(float_dt(adt), "%s: absmax dtype %d (want FP32 / FP16 / BF16)", what, adt);
  CONCH_CHECK_ARG(qt != kQt8bit || code, "%s: the 8-bit form needs its 256-entry code book", what);
  return CONCH_OK;
}

int dequantize_any(void* out, const uint8_t* xq, const void* absmax, const float* code, int64_t n, int blocksize, int qt, int adt, int odt,
                   int64_t out_stride_row, int64_t row_len, hipStream_t stream) {
  if (odt == CONCH_DT_FP32) return dequantize_adt<CONCH_DT_FP32>(out, xq, absmax, code, n, blocksize, qt, adt, out_stride_row, row_len, stream);
  if (odt == CONCH_DT_FP16) return dequantize_adt<CONCH_DT_FP16>(out, xq, absmax, code, n, blocksize, qt, adt, out_stride_row, row_len, stream);
  return dequantize_adt<CONCH_DT_BF16>(out, xq, absmax, code, n, blocksize, qt, adt, out_stride_row, row_len, stream);
}

}  // namespace
}  // namespace conch

using namespace conch;

extern "C" int conch_bnb_quantize_blockwise(uint8_t* out, void* absmax, const void* x, const float* code, int64_t n, int blocksize,
                                            int quant_type, int x_dtype, int absmax_dtype, void* stream) {
  if (int rc = check_common(n, blocksize, quant_type, absmax_dtype, code, "bnb_quantize_blockwise")) return rc;
  CONCH_CHECK_ARG(float_dt(x_dtype), "bnb_quantize_blockwise: input dtype %d (want FP32 / FP16 / BF16)", x_dtype);
  if (n == 0) return CONCH_OK;
  CONCH_CHECK_ARG(out && absmax && x, "bnb_quantize_blockwise: NULL pointer");
  CONCH_CHECK_ARG(((uintptr_t)out & 7) == 0, "bnb_quantize_blockwise: out must be 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == CONCH_DT_FP32) return quantize_adt<CONCH_DT_FP32>(out, absmax, x, code, n, blocksize, quant_type, absmax_dtype, s);
  if (x_dtype == CONCH_DT_FP16) return quantize_adt<CONCH_DT_FP16>(out, absmax, x, code, n, blocksize, quant_type, absmax_dtype, s);
  return quantize_adt<CONCH_DT_BF16>(out, absmax, x, code, n, blocksize, quant_type, absmax_dtype, s);
}

extern "C" int conch_bnb_dequantize_blockwise(void* out, const uint8_t* xq, const void* absmax, const float* code, int64_t n, int blocksize,
                                              int quant_type, int out_dtype, int absmax_dtype, void* stream) {
  if (int rc = check_common(n, blocksize, quant_type, absmax_dtype, code, "bnb_dequantize_blockwise")) return rc;
  CONCH_CHECK_ARG(float_dt(out_dtype), "bnb_dequantize_blockwise: output dtype %d (want FP32 / FP16 / BF16)", out_dtype);
  CONCH_CHECK_ARG(quant_type == kQt8bit || n % 2 == 0, "bnb_dequantize_blockwise: 4-bit outputs have an even number of elements");
  if (n == 0) return CONCH_OK;
  CONCH_CHECK_ARG(out && absmax && xq, "bnb_dequantize_blockwise: NULL pointer");
  return dequantize_any(out, xq, absmax, code, n, blocksize, quant_type, absmax_dtype, out_dtype, n, n, (hipStream_t)stream);
}

// Upper bound of the library scratch conch_bnb_gemm_4bit may need for this shape (see "Library scratch" in conch_amd.h): the
// fp32 slabs of the decode kernel (M <= 256) or the dequantised weight of the dequantise-first path.
extern "C" int64_t conch_bnb_gemm_4bit_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  int64_t need = n * k * 2;
  if (m <= 256) need = std::max(need, (k / kBgSliceK + 1) * m * n * 4);
  need = std::max(need, mid_split_slab_bytes(m, n));  // the 16-bit tile kernel's split-K slabs (gemm_mid.hip)
  return need + ((int64_t)1 << 20);
}

// y = x @ dequantise(W)^T for a bitsandbytes-style 4-bit weight W [N][K] (flattened, blocks of `blocksize` along the flat index,
// two codes per byte): W is dequantised into library scratch in x's dtype -- its rows ARE the K-contiguous B^T the tile kernel
// wants -- and multiplied on the 16-bit MFMA tile kernel of gemm_mfma.hip (fp32 accumulation).  Decode batches (M <= 256 by a cost rule) skip
// the materialised weight: bnb_decode_gemm_kernel.
extern "C" int conch_bnb_gemm_4bit(void* c, const void* x, const uint8_t* w_packed, const void* absmax, int64_t m, int64_t n, int64_t k,
                                   int64_t x_stride_m, int64_t c_stride_m, int blocksize, int quant_type, int absmax_dtype, int x_dtype,
                                   int out_dtype, void* stream) {
  if (int rc = check_common(n * k, blocksize, quant_type, absmax_dtype, nullptr, "bnb_gemm_4bit")) return rc;
  CONCH_CHECK_ARG(m >= 0 && n >= 0 && k >= 0, "bnb_gemm_4bit: negative shape");
  if (quant_type == kQt8bit || (x_dtype != CONCH_DT_FP16 && x_dtype != CONCH_DT_BF16) || (out_dtype != CONCH_DT_FP16 && out_dtype != CONCH_DT_BF16)) {
    set_error("bnb_gemm_4bit: nf4 / fp4 weights with fp16 / bf16 activations and outputs only");
    return CONCH_ERR_UNSUPPORTED;
  }
  if (m == 0 || n == 0) return CONCH_OK;
  CONCH_CHECK_ARG(c && x && w_packed && absmax, "bnb_gemm_4bit: NULL pointer");
  const int64_t lim = (int64_t)1 << 31;
  if (k % 64 || k < 128 || (x_stride_m * 2) % 16 || ((uintptr_t)x & 15) || m * x_stride_m * 2 >= lim || n * k * 2 >= lim || m >= (1 << 24) || n >= (1 << 24)) {
    set_error("bnb_gemm_4bit: needs K %% 64 == 0, K >= 128, 16-byte aligned activation rows and operands below 2 GiB");
    return CONCH_ERR_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  // decode batches: the weights go from the packed tensor straight into MFMA operand registers (CONCH_TUNE_GEMM_VARIANT = 2
  // forces the dequantise-first path, 4 the decode kernel)
  const int variant = tuning(CONCH_TUNE_GEMM_VARIANT);
  if (variant != 2 && bnb_decode_supported(m, n, k, blocksize, variant == 4) && (((uintptr_t)c & 1) == 0) && c_stride_m >= n) {
    const BnbGemmArgs d{c, x, w_packed, absmax, m, n, k, x_stride_m, c_stride_m, blocksize};
    return launch_bnb_decode_gemm(d, quant_type, absmax_dtype, x_dtype, out_dtype, s);
  }
  if (variant == 4) {
    set_error("bnb_gemm_4bit: decode kernel forced but its contract is not met (M <= 256, N %% 4 == 0, K %% blocksize == 0)");
    return CONCH_ERR_UNSUPPORTED;
  }
  void* wt = nullptr;
  if (int rc = get_scratch(s, kScratchRepack, (size_t)n * k * 2, &wt)) return rc;
  if (int rc = dequantize_any(wt, w_packed, absmax, nullptr, n * k, blocksize, quant_type, absmax_dtype, x_dtype, k, k, s)) return rc;
  ScaledGemmArgs g{};
  g.c = c;
  g.a = x;
  g.b = wt;
  g.m = m;
  g.n = n;
  g.k = k * 2;
  g.a_stride_m = x_stride_m * 2;
  g.b_stride_n = k * 2;
  g.a_stride_k = g.b_stride_k = 1;
  g.c_stride_m = c_stride_m;
  g.c_stride_n = 1;
  g.scale_a_numel = g.scale_b_numel = 1;
  g.in_dtype = x_dtype;
  g.out_dtype = out_dtype;
  return launch_scaled_gemm_mfma_16bit(g, s);
}
