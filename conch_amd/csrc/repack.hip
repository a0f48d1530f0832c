// Operand repacking for the MFMA kernels' layout contract.
//
// The MFMA GEMMs want both operands K-contiguous with K a multiple of 128 bytes and 16-byte aligned
// rows -- what the reference's own tests and benchmarks pass (b = (N, K).T).  Any other legal input
// (row-major [K][N] weights, a transposed activation view, K = 200, a misaligned slice) is first
// copied into that layout in library-owned scratch memory (get_scratch: per stream, grown on demand,
// the call stays asynchronous), K zero-padded: zeros add nothing to int8 or fp8 dot products.  One 64x64-byte LDS tile per workgroup, reads coalesced along whichever source
// dimension has the smaller stride, writes coalesced along K.  HBM-bound, one pass.
#include "common.hpp"
#include "gemm.hpp"

namespace conch {
namespace {

constexpr int kRpTile = 64;

// dst[r][k] (row stride kp, k < kp, zero for k >= K) = src[r * sr + k * sk]
__global__ __launch_bounds__(256) void repack_rows_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src,
                                                          int64_t rows, int64_t k_dim, int64_t kp, int64_t sr,
                                                          int64_t sk) {
  __shared__ uint8_t tile[kRpTile][kRpTile + 4];  // [r][k]
  const int64_t k0 = (int64_t)blockIdx.x * kRpTile, r0 = (int64_t)blockIdx.y * kRpTile;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  const bool k_fast = sk <= sr;
#pragma unroll 4
  for (int i = ty; i < kRpTile; i += 4) {
    const int rr = k_fast ? i : tx, kk = k_fast ? tx : i;
    const int64_t r = r0 + rr, k = k0 + kk;
    uint8_t v = 0;
    if (r < rows && k < k_dim) v = src[r * sr + k * sk];
    tile[rr][kk] = v;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = ty; i < kRpTile; i += 4) {
    const int64_t r = r0 + i, k = k0 + tx;
    if (r < rows && k < kp) dst[r * kp + k] = tile[i][tx];
  }
}

// The common non-native case -- row-major [K][N] weights handed over as b: element (r = n, k) at src[k * sk + r], sr == 1, 16-byte
// aligned rows of the source -- as a 64 x 64-byte transpose with 16-byte global accesses on both sides (the generic kernel above moves
// one byte per lane and instruction: 20 us for a 4096 x 4096 weight where this takes a third; round 4, tools/time_layouts.py).
// Thread t loads 16 consecutive r of source row k0 + (t >> 2), then gathers 16 consecutive k of result row r0 + (t >> 2) from LDS.
__global__ __launch_bounds__(256) void repack_transpose_vec_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int64_t rows,
                                                                   int64_t k_dim, int64_t kp, int64_t sk) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[kRpTile][kRpTile + 16];  // [k][r], rows 16-byte aligned
  const int64_t k0 = (int64_t)blockIdx.x * kRpTile, r0 = (int64_t)blockIdx.y * kRpTile;
  const int line = threadIdx.x >> 2, piece = threadIdx.x & 3;
  {
    const int64_t k = k0 + line, r = r0 + 16 * piece;
    i32x4 v = {0, 0, 0, 0};
    if (k < k_dim) {
      if (r + 16 <= rows) {
        v = *(const i32x4*)(src + k * sk + r);
      } else {
        uint8_t* b = (uint8_t*)&v;
        for (int i = 0; i < 16; ++i)
          if (r + i < rows) b[i] = src[k * sk + r + i];
      }
    }
    // row block b = line >> 4 keeps its columns rotated by 16 b: the four lanes of a quad below read rows 16 apart and would
    // otherwise all hit one bank group
    *(i32x4*)&tile[line][16 * ((piece + (line >> 4)) & 3)] = v;
  }
  __syncthreads();
  const int64_t r = r0 + line, k = k0 + 16 * piece;
  if (r >= rows || k >= kp) return;  // kp % 64 == 0 (callers pad K to 128 or 1024): whole 16-byte groups
  i32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) w |= (uint32_t)tile[16 * piece + 4 * j + i][(line + 16 * piece) & 63] << (8 * i);
    o[j] = (int)w;
  }
  *(i32x4*)(dst + r * kp + k) = o;
}

// e4m3fnuz -> bf16 (every e4m3 value is exactly representable), K-contiguous, zero padded:
// dst[r][k] (bf16, row stride kp elements) = decode_fnuz(src[r * sr + k * sk])
__global__ __launch_bounds__(256) void fnuz_to_bf16_rows_kernel(uint16_t* __restrict__ dst, const uint8_t* __restrict__ src,
                                                                int64_t rows, int64_t k_dim, int64_t kp, int64_t sr,
                                                                int64_t sk, const int* __restrict__ gate) {
  __shared__ uint8_t tile[kRpTile][kRpTile + 4];  // [r][k]
  if (gated_off(gate, 1)) return;  // the operands hold none of the codes the fp8 MFMA cannot take: the fp8 kernel runs instead
  const int64_t k0 = (int64_t)blockIdx.x * kRpTile, r0 = (int64_t)blockIdx.y * kRpTile;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const bool k_fast = sk <= sr;
#pragma unroll 4
  for (int i = ty; i < kRpTile; i += 4) {
    const int rr = k_fast ? i : tx, kk = k_fast ? tx : i;
    const int64_t r = r0 + rr, k = k0 + kk;
    uint8_t v = 0;
    if (r < rows && k < k_dim) v = src[r * sr + k * sk];
    tile[rr][kk] = v;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = ty; i < kRpTile; i += 4) {
    const int64_t r = r0 + i, k = k0 + tx;
    if (r < rows && k < kp) dst[r * kp + k] = float_to_bf16_bits(decode_fp8_sw<true>(tile[i][tx]));
  }
}

// The same for K-contiguous rows (stride 1 along K, 8-byte aligned rows, K % 8 == 0 -- every operand the MFMA contract admits):
// eight codes per thread, one 8-byte load and one 16-byte store, both contiguous across the lanes (the transposing kernel above
// moves one byte per lane and instruction).
__global__ __launch_bounds__(256) void fnuz_to_bf16_vec_kernel(uint16_t* __restrict__ dst, const uint8_t* __restrict__ src, int64_t rows, int64_t k_dim,
                                                               int64_t kp, int64_t sr, const int* __restrict__ gate) {
  if (gated_off(gate, 1)) return;
  // at most 2048 workgroups walk the rows: in the common case the launch is gated off, and an empty launch costs by its
  // workgroup count (30 000 empty workgroups at C3 size took 12 us)
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    for (int64_t k = (int64_t)threadIdx.x * 8; k < kp; k += 256 * 8) {
      i32x2 w = {0, 0};
      if (k < k_dim) w = *(const i32x2*)(src + r * sr + k);  // K % 8 == 0: a group of eight is inside K or in the zero padding
      u16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = float_to_bf16_bits(decode_fp8_sw<true>((uint8_t)((uint32_t)w[i >> 2] >> (8 * (i & 3)))));
      *(u16x8*)(dst + r * kp + k) = o;
    }
  }
}

// both operands in ONE launch (blockIdx.y = 0: A, 1: B^T): the flow's launches are usually gated off, and an empty launch costs
// ~4.7 us of stream time whatever it would have done (profiles/r06/fnuz_c3_flow_trace.txt)
__global__ __launch_bounds__(256) void fnuz_to_bf16_vec2_kernel(uint16_t* __restrict__ dst_a, const uint8_t* __restrict__ src_a, int64_t rows_a, int64_t sr_a,
                                                                uint16_t* __restrict__ dst_b, const uint8_t* __restrict__ src_b, int64_t rows_b, int64_t sr_b,
                                                                int64_t k_dim, int64_t kp, const int* __restrict__ gate) {
  if (gated_off(gate, 1)) return;
  uint16_t* dst = blockIdx.y ? dst_b : dst_a;
  const uint8_t* src = blockIdx.y ? src_b : src_a;
  const int64_t rows = blockIdx.y ? rows_b : rows_a, sr = blockIdx.y ? sr_b : sr_a;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    for (int64_t k = (int64_t)threadIdx.x * 8; k < kp; k += 256 * 8) {
      i32x2 w = {0, 0};
      if (k < k_dim) w = *(const i32x2*)(src + r * sr + k);
      u16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = float_to_bf16_bits(decode_fp8_sw<true>((uint8_t)((uint32_t)w[i >> 2] >> (8 * (i & 3)))));
      *(u16x8*)(dst + r * kp + k) = o;
    }
  }
}

bool k_major_ok(const void* ptr, int64_t stride_k, int64_t stride_row, int64_t k_dim, int64_t k_mult) {
  return stride_k == 1 && stride_row % 16 == 0 && ((uintptr_t)ptr & 15) == 0 && k_dim % k_mult == 0;
}

}  // namespace

// Decide whether repacking makes `p` eligible for the MFMA kernels; if so fill `q` (a copy of p that
// points into scratch) and launch the copies.  *scratch is library-owned (get_scratch), nothing to release.
int repack_for_mfma(const ScaledGemmArgs& p, ScaledGemmArgs* q, void** scratch, hipStream_t stream) {
  *scratch = nullptr;
  *q = p;
  if (p.in_dtype != CONCH_DT_INT8 && p.in_dtype != CONCH_DT_FP8_E4M3FN) return CONCH_ERR_UNSUPPORTED;
  if (p.c_stride_n != 1 || p.k < 1) return CONCH_ERR_UNSUPPORTED;
  const int64_t k_mult = p.m <= 256 ? 1024 : 128;  // skinny kernel: K/4 per wave in 256-byte pairs
  const int64_t kp = (p.k + k_mult - 1) / k_mult * k_mult;
  const bool fix_a = !k_major_ok(p.a, p.a_stride_k, p.a_stride_m, p.k, k_mult);
  const bool fix_b = !k_major_ok(p.b, p.b_stride_k, p.b_stride_n, p.k, k_mult);
  if (!fix_a && !fix_b) return CONCH_ERR_UNSUPPORTED;  // something else is wrong with the request
  const bool pad = kp != p.k;
  const bool copy_a = fix_a || pad, copy_b = fix_b || pad;
  const size_t a_bytes = copy_a ? (size_t)(p.m * kp + 255) / 256 * 256 : 0;
  const size_t b_bytes = copy_b ? (size_t)(p.n * kp + 255) / 256 * 256 : 0;
  if (p.m * kp >= ((int64_t)1 << 31) || p.n * kp >= ((int64_t)1 << 31)) return CONCH_ERR_UNSUPPORTED;
  void* ws = nullptr;
  if (int rc = get_scratch(stream, kScratchRepack, a_bytes + b_bytes, &ws)) return rc;
  *scratch = ws;
  const dim3 block(256);
  if (copy_a) {
    const dim3 grid((unsigned)((kp + kRpTile - 1) / kRpTile), (unsigned)((p.m + kRpTile - 1) / kRpTile));
    if (p.a_stride_m == 1 && p.a_stride_k % 16 == 0 && ((uintptr_t)p.a & 15) == 0)  // a transposed activation view
      hipLaunchKernelGGL(repack_transpose_vec_kernel, grid, block, 0, stream, (uint8_t*)ws, (const uint8_t*)p.a, p.m, p.k, kp, p.a_stride_k);
    else
      hipLaunchKernelGGL(repack_rows_kernel, grid, block, 0, stream, (uint8_t*)ws, (const uint8_t*)p.a, p.m, p.k, kp,
                         p.a_stride_m, p.a_stride_k);
    q->a = ws;
    q->a_stride_m = kp;
    q->a_stride_k = 1;
  }
  if (copy_b) {
    uint8_t* bt = (uint8_t*)ws + a_bytes;
    const dim3 grid((unsigned)((kp + kRpTile - 1) / kRpTile), (unsigned)((p.n + kRpTile - 1) / kRpTile));
    if (p.b_stride_n == 1 && p.b_stride_k % 16 == 0 && ((uintptr_t)p.b & 15) == 0)  // row-major [K][N] weights
      hipLaunchKernelGGL(repack_transpose_vec_kernel, grid, block, 0, stream, bt, (const uint8_t*)p.b, p.n, p.k, kp, p.b_stride_k);
    else
      hipLaunchKernelGGL(repack_rows_kernel, grid, block, 0, stream, bt, (const uint8_t*)p.b, p.n, p.k, kp,
                         p.b_stride_n, p.b_stride_k);
    q->b = bt;
    q->b_stride_n = kp;
    q->b_stride_k = 1;
  }
  q->k = kp;
  return check_launch("repack_rows");
}

// e4m3fnuz operands: expand both to bf16 in scratch (exact) and describe them to the tiled MFMA kernel in
// BYTE units (a 128-byte K step = 64 elements).  The MI300-era flavour cannot use gfx950's OCP fp8 MFMA
// (different bias, 0x7F/0xFF are numbers, 0x80 is NaN), so it runs at the bf16 MFMA rate instead.
// Does A or B^T hold 0x80 (fnuz NaN; OCP -0), 0x7F or 0xFF (fnuz +-240; OCP NaN)?  One pass over both operands, 16 bytes per
// lane (fnuz_scan_kernel says how the verdict reaches the dispatch word).  Rows are K-contiguous and 16-byte aligned, K % 16 == 0 (the tile contract).
__device__ __forceinline__ uint32_t fnuz_special_in(uint32_t x) {
  const uint32_t t = x & 0x7f7f7f7fu;                                   // a byte is 0x7F or 0xFF  <=>  its low seven bits are all set
  const uint32_t top = (t + 0x01010101u) & 0x80808080u;                 // (no carry between bytes: 0x7F + 1 = 0x80)
  const uint32_t y = x ^ 0x80808080u;                                   // a byte is 0x80  <=>  it is zero here
  const uint32_t nan = (y - 0x01010101u) & ~y & 0x80808080u;            // the classic zero-byte test (false positives impossible
  return top | nan;                                                     // for the lowest zero byte; any hit means "some byte")
}

// `flag` was zeroed on the stream in front of the launch.  ONE atomicOr per workgroup that found a code (round 6: one per THREAD
// put up to half a million atomics on one address when every row holds +-240).  (A form without the 4-byte
// memset -- last-arriving workgroup publishes the verdict and resets the bookkeeping -- was tried: its agent-scope fences write
// the L2s back, 67 us per scan behind a GEMM's 90 MB of dirty C lines.)
__global__ __launch_bounds__(256) void fnuz_scan_kernel(const uint8_t* __restrict__ a, int rows_a, int64_t stride_a,
                                                        const uint8_t* __restrict__ b, int rows_b, int64_t stride_b, int k_vecs,
                                                        int* __restrict__ flag, const float* __restrict__ sa, int sa_numel,
                                                        float* __restrict__ sa_quarter) {
  // the exact factor 1/4 of the raw-bits product (gemm.hpp, acc_scale) as a scaled copy of scale_a: kernels that know nothing
  // of acc_scale (split-K forms, their reduce kernel) then run the e4m3fnuz problem unchanged.  0.25 x sa is exact (a power of two)
  if (sa_quarter && blockIdx.x == 0)
    for (int i = threadIdx.x; i < sa_numel; i += 256) sa_quarter[i] = 0.25f * sa[i];
  uint32_t found = 0;
  const int rows = rows_a + rows_b;
  for (int r0 = blockIdx.x * 4; r0 < rows; r0 += gridDim.x * 4) {
    for (int c = threadIdx.x; c < k_vecs; c += 256) {
      u32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // four rows' loads in flight
        const int r = min(r0 + j, rows - 1);
        const uint8_t* row = r < rows_a ? a + (int64_t)r * stride_a : b + (int64_t)(r - rows_a) * stride_b;
        v[j] = *(const u32x4*)(row + (int64_t)c * 16);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) found |= fnuz_special_in(v[j][e]);
    }
  }
  const int any = __syncthreads_or(found != 0);
  // (... and none once the word is set)
  if (any && threadIdx.x == 0 && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flag, 1);
}

int launch_fnuz_scan(const ScaledGemmArgs& p, int* flag, hipStream_t stream, float* sa_quarter) {
  const int rows = (int)(p.m + p.n);
  int blocks = (rows + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(fnuz_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const uint8_t*)p.a, (int)p.m, p.a_stride_m,
                     (const uint8_t*)p.b, (int)p.n, p.b_stride_n, (int)(p.k / 16), flag, p.scale_a, (int)p.scale_a_numel, sa_quarter);
  return check_launch("fnuz_scan");
}

bool fnuz_expansion_fits(const ScaledGemmArgs& p) {
  if (p.in_dtype != CONCH_DT_FP8_E4M3FNUZ || p.c_stride_n != 1 || p.k < 1) return false;
  const int64_t kp = (p.k + 63) / 64 * 64;  // elements; 128-byte steps
  if (p.m * kp * 2 >= ((int64_t)1 << 31) || p.n * kp * 2 >= ((int64_t)1 << 31)) return false;
  if (p.m >= (1 << 24) || p.n >= (1 << 24) || ((uintptr_t)p.c & 1)) return false;
  return true;
}

int expand_fnuz_to_bf16(const ScaledGemmArgs& p, ScaledGemmArgs* q, hipStream_t stream, const int* gate) {
  *q = p;
  if (!fnuz_expansion_fits(p)) return CONCH_ERR_UNSUPPORTED;
  const int64_t kp = (p.k + 63) / 64 * 64;  // elements; 128-byte steps
  const size_t a_bytes = (size_t)(p.m * kp * 2 + 255) / 256 * 256;
  const size_t b_bytes = (size_t)(p.n * kp * 2 + 255) / 256 * 256;
  void* ws = nullptr;
  if (int rc = get_scratch(stream, kScratchRepack, a_bytes + b_bytes, &ws)) return rc;
  const dim3 block(256);
  auto expand = [&](uint16_t* out, const void* src, int64_t rows, int64_t stride_row, int64_t stride_k) {
    if (stride_k == 1 && stride_row % 8 == 0 && ((uintptr_t)src & 7) == 0 && p.k % 8 == 0) {
      const dim3 grid((unsigned)std::min<int64_t>(rows, 2048));
      hipLaunchKernelGGL(fnuz_to_bf16_vec_kernel, grid, block, 0, stream, out, (const uint8_t*)src, rows, p.k, kp, stride_row, gate);
    } else {
      const dim3 grid((unsigned)((kp + kRpTile - 1) / kRpTile), (unsigned)((rows + kRpTile - 1) / kRpTile));
      hipLaunchKernelGGL(fnuz_to_bf16_rows_kernel, grid, block, 0, stream, out, (const uint8_t*)src, rows, p.k, kp, stride_row, stride_k, gate);
    }
  };
  uint8_t* bt = (uint8_t*)ws + a_bytes;
  auto vec_ok = [&](const void* src, int64_t stride_row, int64_t stride_k) {
    return stride_k == 1 && stride_row % 8 == 0 && ((uintptr_t)src & 7) == 0 && p.k % 8 == 0;
  };
  if (vec_ok(p.a, p.a_stride_m, p.a_stride_k) && vec_ok(p.b, p.b_stride_n, p.b_stride_k)) {
    const dim3 grid((unsigned)std::min<int64_t>(std::max(p.m, p.n), 1024), 2);
    hipLaunchKernelGGL(fnuz_to_bf16_vec2_kernel, grid, block, 0, stream, (uint16_t*)ws, (const uint8_t*)p.a, p.m, p.a_stride_m, (uint16_t*)bt,
                       (const uint8_t*)p.b, p.n, p.b_stride_n, p.k, kp, gate);
  } else {
    expand((uint16_t*)ws, p.a, p.m, p.a_stride_m, p.a_stride_k);
    expand((uint16_t*)bt, p.b, p.n, p.b_stride_n, p.b_stride_k);
  }
  q->a = ws;
  q->b = bt;
  q->a_stride_m = q->b_stride_n = kp * 2;  // bytes
  q->a_stride_k = q->b_stride_k = 1;
  q->k = kp * 2;                            // bytes
  return check_launch("fnuz_to_bf16_rows");
}

}  // namespace conch
