// Row-major epilogue of the scaled-GEMM tile kernels (gemm_mfma.hip: 256 x 256 tiles, 128 x 64 per wave; gemm_mid.hip: 128 x 128
// tiles, 64 x 64 per wave): whole 128-byte lines, write-through.
#pragma once

#include "common.hpp"
#include "gemm.hpp"
#include "mfma_tile.hpp"

namespace conch {
namespace tile {

// The accumulator layout gives a lane 8 consecutive columns of ONE row, so a 16-lane quarter of a direct store instruction touches 16 different 128-byte lines, and that access
// shape -- not the chip, not the XCD -- is what bounds the direct epilogue: tools/micro/store_rate.hip
// (profiles/r03/store_rate.txt) stores one 128 KiB tile per CU in 4.0 us (13.6 B/clk/CU) whether 1 or 256 CUs do it, while
// 8 rows x 128 bytes per instruction (whole lines, two per quarter-wave) take 1.0 us on a lone CU and 2.4 us when all 256 CUs
// burst together.  Here a wave passes its 128 x 64 sub-tile (already scaled, cast and biased: the same arithmetic, bit for bit)
// through ONE 2 KiB LDS buffer of its own, sixteen rows at a time: conflict-free ds_write_b128 in accumulator order (16-byte
// chunk index XOR row & 7), conflict-free ds_read_b128 row-wise, `buffer_store_dwordx4` of whole lines.  A wave's LDS
// instructions execute in issue order, so one buffer is enough (block b is written behind the read of block b - 1) and no
// barrier is needed; the loop is a software pipeline -- read back block b - 1, convert and park block b, store block b - 1 --
// so the store queue fills from the second block on.  Stores are write-through (`sc1`): C is never re-read by this kernel,
// and lines that do not stay in the XCD's L2 leave it to the operand panels (C3: K loop 36.5 -> 35.2 us, held clock 2.07 ->
// 2.14 GHz; profiles/r03/probe_epilogue_first.txt).  The per-store address is ONE VGPR offset (see the note at the stores).
// `stage_off`: byte offset of 16 KiB of LDS nobody else touches during the epilogue (plain launch: the dead operand buffers,
// behind the kernel's last barrier; persistent walk: the spare LDS behind the parked scales, the operand ring being refilled).
// Tiles that are not whole (ragged M / N), unaligned C rows and C buffers beyond 4 GiB take the direct epilogue.
#ifdef CONCH_EXP_C_AUX  // experiment builds: another cache policy for the whole-line C stores
constexpr int kCStoreAux = CONCH_EXP_C_AUX;
#else
constexpr int kCStoreAux = 16;  // cache-policy bits of the C stores: 16 = sc1 (write-through), 0 = default, 2 = nt
#endif

__device__ __forceinline__ bool tile_stores_whole_lines(const ScaledGemmArgs& p, int bm0, int bn0, int tile_m = kTileM, int tile_n = kTileN) {
  return (p.c_stride_m % 8 == 0) && (((uintptr_t)p.c & 15) == 0) && bm0 + tile_m <= (int)p.m && bn0 + tile_n <= (int)p.n &&
         ((p.m - 1) * p.c_stride_m + p.n) * 2 < ((int64_t)1 << 32);
}

// Vector-memory operations ONE wave issues in epilogue_rows_body: two whole-line stores per 16-row block, nothing else (the
// scales come from LDS).  The persistent kernel's hand-counted vmcnt waits (gemm_mfma.hip) are derived from this number.
constexpr int epilogue_rows_vm_ops(int mt) { return 2 * mt; }

// MT = 16-row blocks per wave (8: 128 rows, 4: 64 rows), TILE = rows = columns of the workgroup tile (the parked constants are
// float sa[TILE] | sb[TILE] | bias[TILE] at epi_off); a wave owns 64 columns = one line per row.
template <int MMA, int OUT_DT, bool BIAS, int MT = 8, int TILE = 256, bool MULTI = false>
__device__ __forceinline__ void epilogue_rows_body(const typename AccT<MMA>::type (&acc)[MT][4], const ScaledGemmArgs& p, char* lds,
                                                   int stage_off, int bm0, int bn0, int wr, int wc, int lane, int wave, int epi_off) {
  asm volatile("" ::: "memory");  // nothing of this is moved above the caller's closing barrier
  asm volatile("" : "+v"(lane));  // lane-constant addresses are formed HERE, not hoisted above the K loop (and spilled across it)
  const int g = lane >> 4, jm = lane & 15;
  const int rr = lane >> 3, ch = lane & 7;
  const float* lsa = (const float*)(lds + epi_off) + wr * (16 * MT) + jm;
  const float* lsb = (const float*)(lds + epi_off) + TILE;
  const float* lbias = lsb + TILE;
  const int region = stage_off + wave * 2048;                  // byte offsets from the 1024-aligned LDS base
  const int woff = region + jm * 128 + ((g ^ (jm & 7)) * 16);  // column half 0; half 1 is chunk + 4 = this offset ^ 64
  const int roff = region + rr * 128 + ((ch ^ rr) * 16);       // rows rr (and rr + 8: + 1024), logical chunk ch
  f32x4 sbv[2][2], bsv[2][2];  // [column half nh][low / high four columns]
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    const int nl = wc * 64 + nh * 32 + 8 * g;
    sbv[nh][0] = *(const f32x4*)(lsb + nl);
    sbv[nh][1] = *(const f32x4*)(lsb + nl + 4);
    if constexpr (BIAS) {
      bsv[nh][0] = *(const f32x4*)(lbias + nl);
      bsv[nh][1] = *(const f32x4*)(lbias + nl + 4);
    }
  }
  // byte offsets into C are unsigned 32-bit quantities (C is below 4 GiB here, not below 2 GiB)
  const uint32_t stride_b = (uint32_t)p.c_stride_m * 2u;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.c, 0, (uint32_t)(((p.m - 1) * p.c_stride_m + p.n) * 2), 0x00020000);
  const uint32_t voff = (uint32_t)(bm0 + wr * (16 * MT) + rr) * stride_b + (uint32_t)(bn0 + wc * 64 + ch * 8) * 2u;
  const uint32_t step8 = 8u * stride_b;  // scalar: eight rows further down
  i32x4 rd0, rd1;
  // exactly two buffer stores per block below and no other vector-memory instruction in this function: counted waits depend on it
  static_assert(epilogue_rows_vm_ops(MT) == 2 * MT, "epilogue_rows_body issues two stores per 16-row block");
#pragma unroll
  for (int b = 0; b < MT + 1; ++b) {
    if (b >= 1) {  // block b - 1 back, row-wise: lands under the conversion below
      rd0 = *(const i32x4*)(lds + roff);
      rd1 = *(const i32x4*)(lds + roff + 1024);
    }
    if (b < MT) {  // convert block b and park it (behind the read above in LDS order: same buffer)
      const float sa = lsa[b * 16];
      i32x4 pk[2];
      // the eight column pairs are independent chains of {multiply, multiply, convert}: one straight-line block, so that
      // hipcc interleaves them instead of padding every dependent pair with s_nop
      f32x2 v[2][4];
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const int e = 2 * e2;
          const f32x2 a2 = {(float)acc[b][nh * 2 + (e >> 2)][e & 3], (float)acc[b][nh * 2 + (e >> 2)][(e & 3) + 1]};
          v[nh][e2] = f32x2{sa, sa} * a2;  // scaled_gemm.py:21
        }
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const int e = 2 * e2;
          const f32x2 sb2 = {sbv[nh][e >> 2][e & 3], sbv[nh][e >> 2][(e & 3) + 1]};
          v[nh][e2] = pin_f32x2(sb2 * v[nh][e2]);  // :22 (rounded to fp32 before the cast, like torch)
        }
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const int e = 2 * e2;
          uint32_t h = pack2_bits16<OUT_DT>(v[nh][e2]);  // :23
          if constexpr (BIAS) {                          // :24-25, added in the output dtype
            const f32x2 b2 = {bsv[nh][e >> 2][e & 3], bsv[nh][e >> 2][(e & 3) + 1]};
            h = pack2_bits16<OUT_DT>(pin_f32x2(unpack2_bits16<OUT_DT>(h) + b2));
          }
          pk[nh][e2] = (int)h;
        }
      *(i32x4*)(lds + woff) = pk[0];
      *(i32x4*)(lds + (woff ^ 64)) = pk[1];
    }
    if (b >= 1) {
      // the row-group offset goes into the VGPR offset, soffset stays the constant 0: with a REGISTER soffset hipcc (ROCm 7.2)
      // assumes there is no write-after-read hazard on the data registers of a 16-byte buffer store and may overwrite
      // them with its next VALU instruction before the store has read them -- observed on gfx950 as a wrong first dword in
      // a few lanes of some launches (gemm_mixed.hip's row-major epilogue, round 3)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rd0), rc, (int)(voff + (uint32_t)(2 * b - 2) * step8), 0, kCStoreAux);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rd1), rc, (int)(voff + (uint32_t)(2 * b - 1) * step8), 0, kCStoreAux);
      if constexpr (MULTI) {  // the same two row groups into the other ranks' results (peer-mapped: the stores leave over xGMI)
        for (int d = 0; d < p.n_more; ++d) {
          const __amdgpu_buffer_rsrc_t rd = make_uniform_rsrc(p.c_more[d], (uint32_t)(((p.m - 1) * p.c_stride_m + p.n) * 2));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rd0), rd, (int)(voff + (uint32_t)(2 * b - 2) * step8), 0, kCStoreAux);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rd1), rd, (int)(voff + (uint32_t)(2 * b - 1) * step8), 0, kCStoreAux);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Workgroup-built image: for tiles whose waves own LESS than a 128-byte line of a row (the mixed GEMM's 192- and 128-column
// tiles, the fused gate/up tiles: 64 or 96 bytes per wave and row).  Every wave parks its cast 16-byte chunks at
// image_chunk_offset() in a [ROWS][PITCH bytes] image in dead operand LDS, the workgroup synchronises, and image_store_rows()
// stores the image as whole lines, write-through: 4 rows x 256 bytes per instruction where PITCH is a multiple of 256 (a
// 16-lane group reads one row's 16 chunks: 16 different bank slots), 8 rows x 128 bytes otherwise (384-byte rows).  The chunk
// index is XORed with row & 7 inside its 128-byte line: conflict-free ds_write_b128 for 8 lanes on 8 consecutive rows of one
// chunk column, conflict-free ds_read_b128 for either piece shape (derivation: gemm_mixed.hip, mixed_epilogue_rows).
// ---------------------------------------------------------------------------------------------------------------------
template <int PITCH>
__device__ __forceinline__ int image_chunk_offset(int row, int chunk) {
  return row * PITCH + (((chunk & ~7) | ((chunk & 7) ^ (row & 7))) * 16);
}

// `n_cols`: columns of C (elements); c_stride_m in elements; the tile is whole (caller checked) and starts at (bm0, bn0).
template <int ROWS, int PITCH>
__device__ __forceinline__ void image_store_rows(const char* lds, void* c, int64_t m, int64_t n_cols, int64_t c_stride_m, int bm0, int bn0,
                                                 int lane, int wave) {
  const uint32_t stride_b = (uint32_t)c_stride_m * 2u;  // unsigned byte offsets: C is below 4 GiB, not below 2 GiB
  const __amdgpu_buffer_rsrc_t rc =
      __builtin_amdgcn_make_buffer_rsrc(c, 0, (uint32_t)(((m - 1) * c_stride_m + n_cols) * 2), 0x00020000);
  constexpr bool kWide = PITCH % 256 == 0;
  constexpr int kPieceRows = kWide ? 4 : 8;
  constexpr int kPieceCols = PITCH / (kWide ? 256 : 128);  // pieces side by side in a row group
  constexpr int kPerWave = (ROWS / kPieceRows) * kPieceCols / 8;
  static_assert((ROWS / kPieceRows) * kPieceCols % 8 == 0, "pieces divide over the eight waves");
  const int lr = kWide ? lane >> 4 : lane >> 3;  // row of the piece
  const int lc = kWide ? lane & 15 : lane & 7;   // 16-byte chunk of the piece
  const uint32_t voff = (uint32_t)(bm0 + lr) * stride_b + (uint32_t)(bn0 * 2 + lc * 16);
#pragma unroll
  for (int i = 0; i < kPerWave; ++i) {
    const int q = wave * kPerWave + i;  // wave-uniform
    const int rg = q / kPieceCols, pc = q % kPieceCols;
    const int row = rg * kPieceRows + lr;
    const int chunk = pc * (kWide ? 16 : 8) + lc;
    const i32x4 v = *(const i32x4*)(lds + image_chunk_offset<PITCH>(row, chunk));
    // offset in the VGPR, soffset the constant 0 (see epilogue_rows_body)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, (int)(voff + (uint32_t)(rg * kPieceRows) * stride_b + (uint32_t)(pc * (kWide ? 256 : 128))), 0,
                                           kCStoreAux);
  }
}

}  // namespace tile
}  // namespace conch
