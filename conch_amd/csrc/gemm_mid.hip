// scaled_gemm for shapes with few 256x256 tiles (M of a few hundred rows): 128x128 tiles, 4 waves, two workgroups per CU.
//
// A lone 256x256 tile is bound by its CU's L2 -> LDS rate (64 KiB per K step at ~60 GB/s: 35 us at K = 4096 whatever
// N is), so a problem with 43 such tiles leaves 213 CUs idle.  Here a tile is 128 rows x 128 columns (32 KiB per K
// step): four times as many workgroups for the same problem, 64 KiB of LDS and 4 waves each, so two fit on a CU and one
// workgroup's LDS-DMA overlaps the other's MFMAs without any ping-pong choreography -- a plain double-buffered loop,
// one barrier per K step.  Same LDS image (8-row x 128-byte subtiles, (row>>1)&7 source-side swizzle, conflict-free
// ds_read_b128), same swapped MFMA operands and V-row permutation (a lane owns 8 consecutive n: 16-byte stores), same
// epilogue arithmetic as gemm_mfma.hip.  Replaces the same reference kernel (kernels/quantization/gemm.py:219-457).
#include "common.hpp"
#include "gemm.hpp"
#include "mfma_tile.hpp"
#include "epilogue_rows.hpp"

namespace conch {
namespace {

using namespace tile;

constexpr int kMidThreads = 256;
constexpr int kMidTile = 128;
constexpr int kMidBuf = 2 * kUnitBytes;                 // U (128 rows of A) + V (128 rows of B^T)
// STAGES buffers, then float sa[128] | sb[128] | bias[128].  STAGES = 2 (64 KiB + 1.5 KiB): two workgroups per CU, one K step in
// flight each.  STAGES = 4 (128 KiB + 1.5 KiB): ONE workgroup per CU with three K steps (96 KiB) in flight, for problems of at
// most one tile per CU -- there a CU holds a single 2-stage workgroup anyway, its one step in flight is issued after the barrier
// and waited for before the next (32 KiB per memory latency: 1 MiB of operands in 31 us at K = 4096 on weights streamed from HBM,
// whatever N and M are; profiles/r02/dispatch_cold_sweep_before.txt).  (A FIVE-stage ring -- all 160 KiB as operand buffers, the
// epilogue constants in the dead ones behind the loop -- measured 2-8 % slower than four stages on nine shapes, warm and on
// weights streamed from HBM; reading the fragments of step t + 1 under the MFMAs of step t -- which makes the wave wait for step
// t + 1, one step less of lead -- 15-20 % slower: profiles/r03/mid_stages_ab.txt.)

struct MidOffsets {
  int u[4], v[4];  // byte offset of this lane's 16-byte source chunk for the wave's four pieces of a unit
};

// SPLIT (round 4): blockIdx.y is a K slice of p.split_steps steps; the workgroup leaves its raw fp32 / int32 partial tile in
// p.slabs [slice][M][N] and the split-K reduce kernel (gemm_skinny.hip) finishes the op.  For few tiles and a long K -- e.g.
// 256 x 11008 x 4096, the Llama-7B down projection at a prefill chunk of 256 rows: 64 tiles of 86 steps each left three
// quarters of the chip idle (51.5 us against 24 for the vendor's GEMM: profiles/r04/vendor_grid_fp8.txt).  A separate
// instantiation: the whole-K kernels are unchanged.
template <int MMA, int OUT_DT, int STAGES, bool SPLIT = false>
__global__ __launch_bounds__(kMidThreads, 2) void scaled_gemm_mid_kernel(ScaledGemmArgs p) {
  constexpr int kMidEpi = STAGES * kMidBuf;
  constexpr int kMidLds = kMidEpi + 3 * 128 * 4;
  __shared__ __attribute__((aligned(1024))) char lds[kMidLds];
  if (gated_off(p.gate, p.gate_run_if)) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_m = ((int)p.m + kMidTile - 1) / kMidTile;
  const int tiles_n = ((int)p.n + kMidTile - 1) / kMidTile;
  const TileCoord tc = map_tile(blockIdx.x, tiles_m, tiles_n, p.raster_magic, p.raster_shift);
  const int bm0 = tc.tm * kMidTile, bn0 = tc.tn * kMidTile;

  Srcs src;
  src.a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (uint32_t)((p.m - 1) * p.a_stride_m + p.k), 0x00020000);
  src.b = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, (uint32_t)((p.n - 1) * p.b_stride_n + p.k), 0x00020000);
  // staging: wave w feeds unit rows [32w, 32w + 32) = four 8-row subtiles of both units
  MidOffsets so;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rho = 32 * wave + 8 * j + (lane >> 3);
    const int chunk = (lane & 7) ^ ((rho >> 1) & 7);
    const int r5 = rho & 31, r = r5 & 15, t = r5 >> 4;
    const int nrow = (rho >> 5) * 32 + 8 * (r >> 2) + (r & 3) + 4 * t;  // MFMA D rows 4g+e of tile t <-> n = 8g+e+4t
    so.u[j] = min(bm0 + rho, (int)p.m - 1) * (int)p.a_stride_m + chunk * 16;
    so.v[j] = min(bn0 + nrow, (int)p.n - 1) * (int)p.b_stride_n + chunk * 16;
  }
  const int total_steps = (int)(p.k / kStepBytes);
  const int step0 = SPLIT ? (int)blockIdx.y * p.split_steps : 0;
  const int steps = SPLIT ? min(p.split_steps, total_steps - step0) : total_steps;
  auto stage = [&](int step) {
    char* dst = lds + (step % STAGES) * kMidBuf + wave * 4096;
    const int koff = (step0 + step) * kStepBytes;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(src.a, (lds_void_t*)(dst + j * 1024), 16, so.u[j], koff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(src.b, (lds_void_t*)(dst + kUnitBytes + j * 1024), 16, so.v[j], koff, 0, 0);
    }
  };

  // epilogue constants: threads 0-127 fetch scale_a[row], 128-255 scale_b[col] and the bias -- requested ahead of the operands,
  // kept in registers through the K loop and put into LDS behind it.  (Round 3, tools/isa_issue_waits.py: stored to LDS right
  // after the first stages were requested, the ds_write made hipcc wait for ALL the LDS-DMA in flight -- the ring's three steps
  // instead of its first -- and a bias loaded under `if (p.bias)` was waited for, a memory round trip, before the first operand
  // load was requested.)
  const bool epi_is_b = threadIdx.x >= 128;
  [[maybe_unused]] float epi_v0 = 0.0f;
  [[maybe_unused]] uint32_t epi_bits = 0;
  if constexpr (SPLIT) {
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0)
      if (s0 < steps) stage(s0);
  } else {
    const int tt = threadIdx.x & 127;
    const int idx = epi_is_b ? min(bn0 + tt, (int)p.n - 1) : min(bm0 + tt, (int)p.m - 1);
    const float* base = epi_is_b ? p.scale_b : p.scale_a;
    const bool vec = (epi_is_b ? p.scale_b_numel : p.scale_a_numel) != 1;
    epi_v0 = base ? base[vec ? idx : 0] : 1.0f;  // NULL scale pointer = 1 (the 16-bit operand paths of gemm_modes.hip / bnb.hip)
    if (!epi_is_b) epi_v0 *= p.acc_scale;  // 1, or the exact 1/4 of e4m3fnuz operands on the OCP fp8 MFMA (gemm.hpp)
    // unconditional (without a bias: two readable bytes of A, never used)
    const uint16_t* bias_src = p.bias ? (const uint16_t*)p.bias + min(bn0 + tt, (int)p.n - 1) : (const uint16_t*)p.a;
    epi_bits = *bias_src;
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0)
      if (s0 < steps) stage(s0);
  }

  const int r = lane & 15, g = lane >> 4;
  const int lane_off = (r >> 3) * 1024 + (r & 7) * 128 + ((g ^ ((r >> 1) & 7)) * 16);
  const int m_base = (wr * 8) * 1024 + lane_off;               // the wave's 64 rows of U
  const int n_base = kUnitBytes + (wc * 8) * 1024 + lane_off;  // the wave's 64 rows of V

  typename AccT<MMA>::type acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = typename AccT<MMA>::type{0, 0, 0, 0};

  for (int t = 0; t < steps; ++t) {
    // this wave's pieces of step t have landed (the steps behind it -- up to STAGES - 2, eight LDS-DMA instructions each, the
    // only vector-memory work in the loop -- may still be in flight; the last steps simply drain) ...
    if (STAGES > 2 && steps - 1 - t >= STAGES - 2) {
      wait_vmcnt_n<8 * (STAGES - 2)>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                // ... and every wave is done with the buffer of step t - 1
    // fp8: the step's LDS-DMA requests ahead of its fragment reads; int8 (twice the MFMA instructions per step): the reads first --
    // measured (profiles/r03/mid_reads_first_ab.txt): int8 -6 % with the reads first, fp8 +1 %
    constexpr bool kReadsFirst = MMA != kMmaFp8;
    if constexpr (!kReadsFirst) {
      if (t + STAGES - 1 < steps) stage(t + STAGES - 1);
    }
    const int buf = (t % STAGES) * kMidBuf;
    Frag fn[4], fm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) fn[q] = read_frag(lds, buf + n_base + q * 2048);
#pragma unroll
    for (int i = 0; i < 4; ++i) fm[i] = read_frag(lds, buf + m_base + i * 2048);
    if constexpr (kReadsFirst) {
      __builtin_amdgcn_sched_barrier(0);
      if (t + STAGES - 1 < steps) stage(t + STAGES - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // D rows = n, D cols = m.  Boustrophedon over (m tile, n tile) so that consecutive MFMAs share an operand, the int8 form's two
    // instructions per tile pair as two passes (gemm_mfma.hip, cluster_mma); every accumulator keeps its order of products
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int q = (i & 1) ? 3 - qq : qq;
        mma_part<MMA, 0>(acc[i][q], fn[q], fm[i]);
      }
    if constexpr (MMA != kMmaFp8) {
#pragma unroll
      for (int i = 3; i >= 0; --i)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int q = (i & 1) ? qq : 3 - qq;
          mma_part<MMA, 1>(acc[i][q], fn[q], fm[i]);
        }
    }
  }

  if constexpr (SPLIT) {
    int* slab = (int*)p.slabs + (int64_t)blockIdx.y * p.m * p.n;
#ifndef CONCH_EXP_MID_SLAB_DIRECT
    // The raw partial tile leaves in WHOLE rows: from the accumulator layout a store instruction covers 16 rows x 64 bytes -- half
    // lines, each line written by two instructions.  A wave's 64 x 64 sub-tile goes through a private 32-row x 68-dword LDS image
    // (the operand ring is dead by now), two halves, and every store instruction writes 4 rows x 256 contiguous bytes: 1-3 % of
    // the op (profiles/r04/mid_slab_store_ab.txt).  The slabs cost 6-11 us of a 25-32 us op whatever their store shape (measured by
    // leaving them out): the traffic itself, which a second kernel sees only after the L2 write-back at the kernel boundary.
    const bool whole_cols = bn0 + wc * 64 + 64 <= (int)p.n;  // wave-uniform; ragged columns keep the direct stores below
    __syncthreads();                                          // every wave is past its last operand read
    if (whole_cols) {
      int* wl = (int*)(lds + wave * (32 * 68 * 4));
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *(i32x4*)(wl + (mt2 * 16 + r) * 68 + (q >> 1) * 32 + 8 * g + 4 * (q & 1)) = __builtin_bit_cast(i32x4, acc[half * 2 + mt2][q]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = 4 * i + (lane >> 4), col = 4 * (lane & 15);
          const i32x4 v = *(const i32x4*)(wl + row * 68 + col);
          const int m = bm0 + wr * 64 + half * 32 + row;
          if (m < p.m) *(i32x4*)(slab + (int64_t)m * p.n + bn0 + wc * 64 + col) = v;
        }
      }
      return;
    }
#endif
    // the raw partial tile, row-major: lane (r, g) holds eight consecutive columns of row r of every (m tile, column half) --
    // 32 bytes, a 128-byte line per quad of g
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const int n0 = bn0 + wc * 64 + nh * 32 + 8 * g;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int m = bm0 + wr * 64 + mt * 16 + r;
        if (m >= p.m) continue;
        int* dst = slab + (int64_t)m * p.n + n0;
#pragma unroll
        for (int h = 0; h < 2; ++h)  // N % 4 == 0 (launcher): whole 16-byte quads
          if (n0 + 4 * h + 4 <= p.n) *(i32x4*)(dst + 4 * h) = __builtin_bit_cast(i32x4, acc[mt][nh * 2 + h]);
      }
    }
    return;
  }
  // epilogue: sb * (sa * acc), RNE cast, bias in the output dtype (reference/quantization/scaled_gemm.py:21-25).
  // Whole tiles: row-major through 2 KiB of LDS per wave, whole-line write-through stores (epilogue_rows.hpp; a wave's 64
  // columns are one 128-byte line per row); CONCH_TUNE_EPILOGUE = 1 and ragged tiles store from the accumulator layout.
  {
    float* f = (float*)(lds + kMidEpi);
    f[threadIdx.x] = epi_v0;
    if (epi_is_b) f[threadIdx.x + 128] = bits16_to_float<OUT_DT>((uint16_t)epi_bits);
  }
  __syncthreads();  // the constants are in LDS; every wave is past its last operand read; nothing is in flight (the last steps drained)
  if (p.rows_epilogue && tile_stores_whole_lines(p, bm0, bn0, kMidTile, kMidTile)) {  // workgroup-uniform
    if (p.bias) epilogue_rows_body<MMA, OUT_DT, true, 4, kMidTile>(acc, p, lds, 0, bm0, bn0, wr, wc, lane, wave, kMidEpi);
    else epilogue_rows_body<MMA, OUT_DT, false, 4, kMidTile>(acc, p, lds, 0, bm0, bn0, wr, wc, lane, wave, kMidEpi);
    return;
  }
  const bool vec_store = (p.c_stride_m % 8 == 0) && (((uintptr_t)p.c & 15) == 0);
  const bool has_bias = p.bias != nullptr;
  const float* lsa = (const float*)(lds + kMidEpi);
  const float* lsb = lsa + 128;
  const float* lbias = lsa + 256;
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    const int nl = wc * 64 + nh * 32 + 8 * g;
    const int n0 = bn0 + nl;
    const f32x4 sb_lo = *(const f32x4*)(lsb + nl), sb_hi = *(const f32x4*)(lsb + nl + 4);
    const f32x4 bs_lo = *(const f32x4*)(lbias + nl), bs_hi = *(const f32x4*)(lbias + nl + 4);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int ml = wr * 64 + mt * 16 + r;
      const int m = bm0 + ml;
      const float sa = lsa[ml];
      i32x4 pk;
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const int e = 2 * e2;
        const f32x2 a2 = {(float)acc[mt][nh * 2 + (e >> 2)][e & 3], (float)acc[mt][nh * 2 + (e >> 2)][(e & 3) + 1]};
        const f32x2 sb2 = e < 4 ? f32x2{sb_lo[e & 3], sb_lo[(e & 3) + 1]} : f32x2{sb_hi[e & 3], sb_hi[(e & 3) + 1]};
        f32x2 v = f32x2{sa, sa} * a2;
        v = pin_f32x2(sb2 * v);
        uint32_t h = pack2_bits16<OUT_DT>(v);
        if (has_bias) {
          const f32x2 b2 = e < 4 ? f32x2{bs_lo[e & 3], bs_lo[(e & 3) + 1]} : f32x2{bs_hi[e & 3], bs_hi[(e & 3) + 1]};
          h = pack2_bits16<OUT_DT>(pin_f32x2(unpack2_bits16<OUT_DT>(h) + b2));
        }
        pk[e2] = (int)h;
      }
      if (m < p.m) {
        uint16_t* dst = (uint16_t*)p.c + (int64_t)m * p.c_stride_m + n0;
        if (vec_store && n0 + 8 <= p.n) {
          *(i32x4*)dst = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (n0 + e < p.n) dst[e] = (uint16_t)((uint32_t)pk[e >> 1] >> (16 * (e & 1)));
        }
      }
    }
  }
}

}  // namespace

// K slices the 128 x 128-tile kernel splits a problem into (1 = whole K per workgroup).  Plain fp8 / int8 problems only (the
// reduce kernel applies scale_a / scale_b / bias: no acc_scale -- the e4m3fnuz flow hands these forms a scaled copy of scale_a --, no 16-bit operand forms); N % 4 == 0 (16-byte slab quads).
// Measured (profiles/r04/mid_splitk_sweep.txt, 18 shapes x 1..8 slices x both ring depths): the launch wants EITHER about one
// workgroup per CU (the 4-stage ring) OR about two (the 2-stage loop) -- 1.5 per CU leaves half the chip with twice the work
// (256 x 28672 x 8192: 73.5 us in 2 slices, 99 in 3, 83 in 4) -- so: up to half a chip of tiles, as many slices as fit one
// workgroup per CU; up to ~0.7 of a chip, as many as fit two; beyond that the reduce pass costs what the idle CUs would give.
// A slice is at least eight K steps, and K at least 8192 bytes (4096 for at most a quarter chip of tiles), at most 8 slabs.
// CONCH_TUNE_MID_SPLITK: 0 = auto, 1 = never, 2..8 = that many slices where the form applies.
int mid_split_slices(const ScaledGemmArgs& p) {
  const int forced = tuning(CONCH_TUNE_MID_SPLITK);
  if (forced == 1) return 1;
  const bool bits16 = p.in_dtype == CONCH_DT_FP16 || p.in_dtype == CONCH_DT_BF16;  // the dequantise-first paths (NULL scales = 1)
  if (p.in_dtype != CONCH_DT_FP8_E4M3FN && p.in_dtype != CONCH_DT_INT8 && !bits16) return 1;
  if (p.acc_scale != 1.0f || (!bits16 && (!p.scale_a || !p.scale_b)) || p.fuse_silu || p.n_more || p.n % 4 || p.c_stride_n != 1) return 1;
  const int64_t tiles = ((p.m + kMidTile - 1) / kMidTile) * ((p.n + kMidTile - 1) / kMidTile);
  const int64_t steps = p.k / kStepBytes;
  if ((p.m * p.n * 4) >= ((int64_t)1 << 28) || tiles < 1) return 1;
  int64_t slices;
  if (forced >= 2) slices = forced;
  else {
    const int64_t cus = device_cu_count();
    if (tiles * 10 > cus * 7) return 1;
    if (steps < (tiles * 4 <= cus ? 32 : 64)) return 1;
    slices = tiles * 2 <= cus ? cus / tiles : (2 * cus) / tiles;
  }
  slices = std::min<int64_t>(std::min<int64_t>(slices, 8), steps / 8);
  if (slices < 2) return 1;
  const int64_t per = (steps + slices - 1) / slices;
  return (int)((steps + per - 1) / per);  // no empty slice
}

// upper bound of the slabs the split-K form may ask for on an m x n result (auto rule: at most 0.7 of a chip of tiles, 8 slices;
// a FORCED slice count -- CONCH_TUNE_MID_SPLITK >= 2 -- bypasses the tile rule in mid_split_slices, so it does here too: a caller
// that sized its scratch by this query and then captures a graph with the forced split must not run out)
int64_t mid_split_slab_bytes(int64_t m, int64_t n) {
  const int forced = tuning(CONCH_TUNE_MID_SPLITK);
  if (forced == 1) return 0;
  if (forced >= 2) return (int64_t)std::min(forced, 8) * m * n * 4;
  const int64_t tiles = ((m + kMidTile - 1) / kMidTile) * ((n + kMidTile - 1) / kMidTile);
  return tiles * 10 > (int64_t)device_cu_count() * 7 ? 0 : 8 * m * n * 4;
}

// Same layout contract as the 256x256 kernels (scaled_gemm_mfma_supported).
int launch_scaled_gemm_mid(const ScaledGemmArgs& p_in, hipStream_t stream) {
  ScaledGemmArgs p = p_in;
  p.rows_epilogue = tuning(CONCH_TUNE_EPILOGUE) != 1;
  set_raster_divisor((uint32_t)(kGroupM * ((p.n + kMidTile - 1) / kMidTile)), &p.raster_magic, &p.raster_shift);
  const int tiles_m = (int)((p.m + kMidTile - 1) / kMidTile);
  const int tiles_n = (int)((p.n + kMidTile - 1) / kMidTile);
  const int slices = mid_split_slices(p);
  const dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)slices);
  // at most one workgroup per CU: the deep ring (one workgroup per CU either way); CONCH_TUNE_MID_STAGES forces 2 or 4
  const int forced = tuning(CONCH_TUNE_MID_STAGES);
  const bool deep = forced == 4 || (forced != 2 && (int64_t)tiles_m * tiles_n * slices <= device_cu_count());
  if (slices > 1) {
    const int64_t steps = p.k / kStepBytes;
    p.split_steps = (int)((steps + slices - 1) / slices);
    if (int rc = get_scratch(stream, kScratchSplitK, (size_t)slices * (size_t)p.m * (size_t)p.n * 4, &p.slabs)) return rc;
    if (!p.scale_a || !p.scale_b) {  // the reduce kernel multiplies by both
      const float* one = nullptr;
      if (int rc = unit_scale(&one)) return rc;
      if (!p.scale_a) { p.scale_a = one; p.scale_a_numel = 1; }
      if (!p.scale_b) { p.scale_b = one; p.scale_b_numel = 1; }
    }
#define CONCH_LAUNCH_MID_SPLIT(MMA)                                                                                      \
  do {                                                                                                                   \
    if (deep) hipLaunchKernelGGL((scaled_gemm_mid_kernel<MMA, CONCH_DT_BF16, 4, true>), grid, dim3(kMidThreads), 0, stream, p); \
    else hipLaunchKernelGGL((scaled_gemm_mid_kernel<MMA, CONCH_DT_BF16, 2, true>), grid, dim3(kMidThreads), 0, stream, p);  \
  } while (0)
    if (p.in_dtype == CONCH_DT_FP8_E4M3FN) CONCH_LAUNCH_MID_SPLIT(kMmaFp8);
    else if (p.in_dtype == CONCH_DT_FP16) CONCH_LAUNCH_MID_SPLIT(kMmaF16);
    else if (p.in_dtype == CONCH_DT_BF16) CONCH_LAUNCH_MID_SPLIT(kMmaBf16);
    else CONCH_LAUNCH_MID_SPLIT(kMmaInt8);
#undef CONCH_LAUNCH_MID_SPLIT
    if (int rc = check_launch("scaled_gemm_mid_split")) return rc;
    return launch_splitk_reduce(p, p.slabs, slices, stream);
  }
#define CONCH_LAUNCH_MID(MMA, OUT)                                                                         \
  do {                                                                                                     \
    if (deep) hipLaunchKernelGGL((scaled_gemm_mid_kernel<MMA, OUT, 4>), grid, dim3(kMidThreads), 0, stream, p); \
    else hipLaunchKernelGGL((scaled_gemm_mid_kernel<MMA, OUT, 2>), grid, dim3(kMidThreads), 0, stream, p);  \
    return check_launch("scaled_gemm_mid");                                                                \
  } while (0)
  if (p.in_dtype == CONCH_DT_FP8_E4M3FN) {
    if (p.out_dtype == CONCH_DT_BF16) CONCH_LAUNCH_MID(kMmaFp8, CONCH_DT_BF16);
    CONCH_LAUNCH_MID(kMmaFp8, CONCH_DT_FP16);
  }
  // 16-bit operands in byte units (launch_scaled_gemm_mfma_16bit: the dequantise-first paths of gemm_modes.hip and bnb.hip, the
  // e4m3fnuz expansion): the same loop on v_mfma_f32_16x16x32_{f16,bf16}, two instructions per 128-byte K step like int8
  if (p.in_dtype == CONCH_DT_FP16) {
    if (p.out_dtype == CONCH_DT_BF16) CONCH_LAUNCH_MID(kMmaF16, CONCH_DT_BF16);
    CONCH_LAUNCH_MID(kMmaF16, CONCH_DT_FP16);
  }
  if (p.in_dtype == CONCH_DT_BF16) {
    if (p.out_dtype == CONCH_DT_BF16) CONCH_LAUNCH_MID(kMmaBf16, CONCH_DT_BF16);
    CONCH_LAUNCH_MID(kMmaBf16, CONCH_DT_FP16);
  }
  if (p.out_dtype == CONCH_DT_BF16) CONCH_LAUNCH_MID(kMmaInt8, CONCH_DT_BF16);
  CONCH_LAUNCH_MID(kMmaInt8, CONCH_DT_FP16);
#undef CONCH_LAUNCH_MID
}

}  // namespace conch
