// Every FITTED constant of the kernel dispatchers, in one place, with the sweep it was fitted on.
// The formulas stay where they are used (capi.hip choose_scaled_kernel / mixed_decode_beats_tiles, gemm_mixed.hip pick_split /
// pick_tile_and_split, gemm_mixed_strip.hip, gemm_asm.hip); they read their numbers from here.  A kernel change that moves a kernel's
// speed invalidates the rows that model it: `python tools/refit_dispatch.py` (GPU box) times every candidate the dispatcher
// chooses between on the sweep shapes and prints, per family, the worst automatic pick against the best candidate -- anything
// above ~1.10 means a row of this table needs refitting from the sweep script named beside it.
#pragma once

namespace conch {
namespace fit {

// ---- scaled_gemm (capi.hip, choose_scaled_kernel) -- us; kscale = K / 4096; t = tiles per round -------------------------------
// profiles/r02/dispatch_cold_sweep_after.txt, refit profiles/r03/dispatch_cold_sweep.txt (tools/dispatch_cold_sweep.py: weights cycled
// through > 600 MB, M 8..1024, six (K, N) pairs, int8 and fp8)
struct ScaledTiles {
  double t256_base = 41.0, t256_per_tile = 0.06;            // 256x256 tiles: rounds x (base + per_tile t) kscale
  double t128_base = 31.0, t128_per_tile = 0.02;            // 128x128 tiles, two per CU: rounds x (base + per_tile t) kscale
  double ring_base = 23.0, ring_per_tile = 0.08;            // ... at most one tile per CU (4-stage ring): (base + per_tile max(0, t - 128)) kscale
  // split-K form of the 128x128 tiles: profiles/r04/mid_splitk_sweep.txt, dispatch_cold_sweep.txt (tools/sweep_mid_split.py)
  double split_fixed = 8.0, split_step = 0.52, split_step_fill = 0.16, split_slab_per_elem = 1e-6;
};
// split-K skinny kernel: fixed + c(M) 1e-7 N K; steps at the 32 / 64 / 128-row forms and the second row block
struct ScaledSkinny {
  double fixed = 5.3;
  double c8 = 2.0, c16 = 2.1, c32 = 2.5, c48 = 2.7, c64 = 3.0, c96 = 4.5, c128 = 5.15, c192 = 9.3, c256 = 10.2;
};
// one-wave-per-SIMD 256x352 tiles against 256x256 tiles: gemm_asm.hip keeps its own table (kAsm1wFit) beside the rule, fitted on
// profiles/r05/asm1w_persistent.txt (tools/try_asm1w.py)
// kAsm1wFit (csrc/gemm_asm.hip) and int8: int8 runs BOTH tilings ~12.5 % slower than fp8 (C3: assembly 119.0 against 105.8 us,
// 256 x 256 tiles 133.4 against 118.8: profiles/r06/asm1w_bias_ab.txt; round 5: 124.4 / 139.8), so the ratio the rule compares is
// the fp8 one and the table serves both dtypes; a bias adds ~1.4 us to the assembly kernel, inside the rule's margin.

// ---- mixed_precision_gemm -----------------------------------------------------------------------------------------------------
// decode-batch kernel against one row of tiles (capi.hip, mixed_decode_beats_tiles): profiles/r02/dispatch_cold_sweep_after.txt,
// refit round 3; int8 weights: int8_factor below (1.85 = the byte ratio was the model until round 5 measured 1.07-1.19x)
struct MixedDecode {
  double fixed = 5.0, per_nk_block = 0.41e-6, int8_factor = 1.2;   // (int8: 1.07-1.19x measured in round 5, mixed_mid_sweep_int8.txt; 1.85 until then)
  double unsplit_tile_us_per_1024k = 16.5;                  // an UNSPLIT row of the narrowest tiles (split off / not allowed)
};
// K split of the LDS-tiled kernel at a given tile shape (gemm_mixed.hip, pick_split): profiles/r04/dispatch_cold_sweep.txt
struct MixedSplit {
  double wg_fixed = 3.0, step = 0.97, launch2 = 5.0, slab_per_elem = 1.0e-6;
};
// tile shape and K split together, M > 256 (gemm_mixed.hip, pick_tile_and_split): tools/sweep_mixed_nt_split.py, 120 timings, RMS
// error 4.5 % (profiles/r04/mixed_nt_split_sweep.txt)
struct MixedTileSplit {
  double step = 0.24, width_offset = 1.2, tall_width = 5.4, fill_slowdown = 0.2, launch2 = 5.0, slab_per_elem = 1.5e-6;
};
// column-strip kernel against the LDS-tiled one (gemm_mixed_strip.hip, mixed_strip_beats_tiles / pick_strip_nt): tools/sweep_mixed_strip.py,
// 48 shapes (profiles/r04/mixed_strip_sweep.txt)
struct MixedStrip {
  long long min_n = 8192;                                   // the strip kernel wins from this N on
  double min_fill = 0.65;                                   // ... with at least this fraction of a chip of tiles
  double width_offset = 0.6;                                // cost of a tile of 64 nt columns ~ nt + this
};

// Tile height, width and K slices of the column-strip kernel (gemm_mixed_strip.hip, strip_plan): 64- / 128- / 256-row tiles (mt = 4 /
// 8 / 16 m tiles per wave) x 128 / 192 / 256 columns (nt = 2..4) x 1..8 slices.  Least squares over 2 457 timings on weights
// streamed from HBM -- tools/sweep_mixed_mid.py (M = 32..512, profiles/r05/mixed_mid_sweep.txt) and tools/archive/
// sweep_mixed_rows_tall.py --full (M = 288..1536, every tile height, profiles/r05/mixed_rows_tall_sweep.txt): RMS error 5.0 %, the
// model's pick within 9 % of the fastest form on all 109 shapes (0.7 % on average), the pick's own time predicted within -15 .. +11 %.
struct MixedStripSplit {
  double wg_fixed = 5.98;                                   // launch, prologue, epilogue (us)
  double step = 0.0623, step_mt = 0.0132, step_nt = 0.0494, step_mt_nt = 0.0089;  // a K step of 64: us = step + ... (mt, nt as above)
  double fill_slowdown = 0.3;                               // ... x (1 + this x the fraction of the chip the launch fills)
  double launch2 = 1.385, slab_per_elem = 1.4115e-6;        // the reduce launch; fp32 partial sums written and read again (us / element)
  double margin = 1.08;                                     // the dispatcher takes these forms when this x their time beats the others'
  long long min_m = 33, max_m = 256;                        // ... one row of tiles: for this many rows (below: the one-launch decode forms)
  long long max_m_search = 1536;                            // several rows of tiles: short tiles / K slices are candidates up to here
  double tall_margin = 1.05;                                // ... and replace the unsplit 256-row tile for a modelled gain beyond this
  double tall_vs_tiles = 1.0;                               // ... and the LDS-tiled kernel when this x their time beats its estimate (the two
                                                            // models err in opposite directions there: 320 x 4096 x 4096 26.0 for 24.4 measured
                                                            // against 27.6 for 29.8)
};

inline constexpr ScaledTiles kScaledTiles{};
inline constexpr ScaledSkinny kScaledSkinny{};
inline constexpr MixedDecode kMixedDecode{};
inline constexpr MixedSplit kMixedSplit{};
inline constexpr MixedTileSplit kMixedTileSplit{};
inline constexpr MixedStrip kMixedStrip{};
inline constexpr MixedStripSplit kMixedStripSplit{};

}  // namespace fit
}  // namespace conch
