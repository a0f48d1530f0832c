// Launcher of the ONE-WAVE-PER-SIMD fp8 scaled-GEMM kernel (csrc/asm/gen_gemm1w.py writes its assembly; the build assembles
// it for gfx950 and embeds the code object below).  Same place in the call stack as launch_scaled_gemm_mfma: replaces the
// reference's Triton `_gemm_kernel` + `scaled_gemm_launcher` (conch/kernels/quantization/gemm.py:219-457, :564-627) for large
// fp8 problems -- 256 x 352 output tiles, four waves of 128 x 176, every register of the SIMD in one wave.
//
// The kernel is loaded from memory with hipModuleLoadData on first use ON EACH DEVICE (HIP modules and their functions belong to
// the device that was current at load time: one table per device id, under a mutex, like unit_scale / device_cu_count) and
// launched with hipModuleLaunchKernel on the caller's stream, so it is stream-ordered and graph-capturable like every other
// kernel of the library.  A load allocates and uploads the code object, which a stream capture may not contain:
// conch_reserve_scratch (the call every capture is preceded by) loads both modules for the current device
// (preload_asm_modules).  A failed load is remembered per device; under the automatic dispatcher the caller falls back to the
// HIP kernels (capi.hip), only a FORCED assembly variant reports the error.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <mutex>

#include "common.hpp"
#include "gemm.hpp"

namespace conch {
namespace {

alignas(4096) const unsigned char kGemm1wCodeObject[] = {
#include "gemm1w_hsaco.inc"
};

alignas(4096) const unsigned char kMixed1wCodeObject[] = {
#include "mixed1w_hsaco.inc"
};

// kernel argument block of conch_mixed1w_* (gen_mixed1w.py)
struct Mixed1wArgs {
  const void* x;
  const void* wq;
  const void* ws;
  void* c;
  uint32_t m, n, k, ldx, ldq, lds, ldc;
  int32_t off;  // weight bias (+ scalar zero point): q - off is the dequantised integer
  uint32_t tiles_m, tiles_n, nwg, magic_pg, grid, magic_last;
  unsigned long long* probe;
};
static_assert(sizeof(Mixed1wArgs) == 96 && offsetof(Mixed1wArgs, m) == 32 && offsetof(Mixed1wArgs, tiles_m) == 64 && offsetof(Mixed1wArgs, probe) == 88,
              "kernarg block of conch_mixed1w_*");

constexpr int kMaxDevices = 64;

// Per-device table of lazily loaded modules: `load(M&)` runs once per device id (under the mutex), on the thread that first
// needs the module there.  Returns NULL (error text set) when the current device id is out of range.
template <class M, class Load>
M* module_of_current_device(M (&table)[kMaxDevices], bool (&loaded)[kMaxDevices], std::mutex& mu, Load load) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    set_error("assembly kernels: device id %d out of range", dev);
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(mu);
  if (!loaded[dev]) {
    load(table[dev]);
    loaded[dev] = true;
  }
  return &table[dev];
}

struct Mixed1wModule {
  hipModule_t mod = nullptr;
  hipFunction_t w3 = nullptr, w4 = nullptr, w3_probe = nullptr, w4_probe = nullptr;
  int rc = CONCH_OK;
};

Mixed1wModule* mixed1w_module() {
  static Mixed1wModule table[kMaxDevices];
  static bool loaded[kMaxDevices] = {};
  static std::mutex mu;
  return module_of_current_device(table, loaded, mu, [](Mixed1wModule& m) {
    hipError_t e = hipModuleLoadData(&m.mod, kMixed1wCodeObject);
    if (e == hipSuccess) e = hipModuleGetFunction(&m.w3, m.mod, "conch_mixed1w_f16_i4_w3");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.w4, m.mod, "conch_mixed1w_f16_i4_w4");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.w3_probe, m.mod, "conch_mixed1w_f16_i4_w3_probe");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.w4_probe, m.mod, "conch_mixed1w_f16_i4_w4_probe");
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("mixed_precision_gemm (one-wave-per-SIMD kernel): loading the embedded code object failed: %s", hipGetErrorString(e));
      m.rc = CONCH_ERR_HIP;
    }
  });
}

std::atomic<unsigned long long*> g_mixed1w_probe{nullptr};

// kernel argument block: the layout S_* of gen_gemm1w.py reads with s_load_dword*
struct Gemm1wArgs {
  const void* a;
  const void* b;
  void* c;
  const float* sa;
  const float* sb;
  uint32_t m, n, k, lda, ldb, ldc;
  uint32_t tiles_m, tiles_n, nwg, magic_pg, grid, magic_last, sa_vec, sb_vec;
  float acc_scale;
  uint32_t out_dtype;
  unsigned long long* probe;  // diagnostic kernels: [tile][5][2] 64-bit stamps (s_memtime, s_memrealtime); *_bias kernels: the bias
                              // vector (N values of the output dtype); else ignored
  const int* gate;            // NULL, or: run only if (*gate != 0) == (gate_run_if != 0)  (ScaledGemmArgs::gate)
  uint32_t gate_run_if;
  uint32_t pad;
};
static_assert(sizeof(Gemm1wArgs) == 128 && offsetof(Gemm1wArgs, gate) == 112 && offsetof(Gemm1wArgs, gate_run_if) == 120, "kernarg block of conch_gemm1w_*");
static_assert(offsetof(Gemm1wArgs, m) == 40 && offsetof(Gemm1wArgs, tiles_m) == 64 && offsetof(Gemm1wArgs, acc_scale) == 96, "kernarg offsets");

constexpr int kTileM = 256;
constexpr int kWidths[3] = {11, 9, 7};  // n tiles of 16 columns per wave: tile widths 352, 288, 224 (gen_gemm1w.py configure())

struct Gemm1wModule {
  hipModule_t mod = nullptr;
  hipFunction_t fn[3][2][2][2] = {};  // [width index][int8][f16 output][bias in the epilogue]
  hipFunction_t bf16_probe = nullptr, bf16_alt = nullptr, bf16_alt_probe = nullptr, i8_probe = nullptr;
  int rc = CONCH_OK;
};

Gemm1wModule* gemm1w_module() {
  static Gemm1wModule table[kMaxDevices];
  static bool loaded[kMaxDevices] = {};
  static std::mutex mu;
  return module_of_current_device(table, loaded, mu, [](Gemm1wModule& m) {
    hipError_t e = hipModuleLoadData(&m.mod, kGemm1wCodeObject);
    for (int w = 0; w < 3 && e == hipSuccess; ++w)
      for (int i8 = 0; i8 < 2 && e == hipSuccess; ++i8)
        for (int f16 = 0; f16 < 2 && e == hipSuccess; ++f16)
          for (int bias = 0; bias < 2 && e == hipSuccess; ++bias) {
            char name[64];
            snprintf(name, sizeof(name), "conch_gemm1w_%s_%s%s%s", i8 ? "i8" : "fp8", f16 ? "f16" : "bf16", w == 0 ? "" : w == 1 ? "_n9" : "_n7",
                     bias ? "_bias" : "");
            e = hipModuleGetFunction(&m.fn[w][i8][f16][bias], m.mod, name);
          }
    if (e == hipSuccess) e = hipModuleGetFunction(&m.bf16_probe, m.mod, "conch_gemm1w_fp8_bf16_probe");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.bf16_alt, m.mod, "conch_gemm1w_fp8_bf16_alt");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.bf16_alt_probe, m.mod, "conch_gemm1w_fp8_bf16_alt_probe");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.i8_probe, m.mod, "conch_gemm1w_i8_bf16_probe");
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("scaled_gemm (one-wave-per-SIMD kernel): loading the embedded code object failed: %s", hipGetErrorString(e));
      m.rc = CONCH_ERR_HIP;
    }
  });
}

std::atomic<unsigned long long*> g_gemm1w_probe{nullptr};
std::atomic<int> g_gemm1w_alt{0};
std::atomic<int> g_gemm1w_grid{0};  // diagnostic: caps the persistent grid (a multiple of 8), 0 = one workgroup per CU
std::atomic<int> g_gemm1w_nt{0};  // diagnostic: 11 / 9 / 7 forces that tile width, 0 = the cost rule's pick

}  // namespace
}  // namespace conch

// Diagnostic: with a non-NULL buffer of (workgroups x 10) 64-bit words the NEXT bf16 launches run the stamped twin of the kernel
// (the product kernel contains no stamp) and wave 0 of every workgroup writes (shader clock, 100 MHz clock) at kernel entry, K loop
// start, K loop end, epilogue stores issued, stores retired.  NULL switches back.  Used by tools/try_asm1w.py only.
extern "C" int conch_debug_gemm1w_probe(unsigned long long* buffer) {
  conch::g_gemm1w_probe.store(buffer);
  return CONCH_OK;
}

// Diagnostic: the same for the mixed-precision assembly kernel ([workgroup][4][2] stamps: entry, K loop start, stores issued, retired)
extern "C" int conch_debug_mixed1w_probe(unsigned long long* buffer) {
  conch::g_mixed1w_probe.store(buffer);
  return CONCH_OK;
}

// Diagnostic: force the assembly scaled GEMM's tile width (n tiles per wave: 11, 9 or 7 = 352, 288, 224 columns); 0 = automatic
extern "C" int conch_debug_gemm1w_width(int nt) {
  conch::g_gemm1w_nt.store(nt == 11 || nt == 9 || nt == 7 ? nt : 0);
  return CONCH_OK;
}

// Diagnostic: cap the persistent grid of the assembly scaled GEMM at `wgs` workgroups (rounded down to a multiple of 8); 0 = automatic
extern "C" int conch_debug_gemm1w_grid(int wgs) {
  conch::g_gemm1w_grid.store(wgs >= 8 ? (wgs & ~7) : 0);
  return CONCH_OK;
}

// Diagnostic: 1 = the bf16 launches run the A/B twin of the kernel (the generator's `alt` choice; bit-identical results), 0 = back.
extern "C" int conch_debug_gemm1w_alt(int on) {
  conch::g_gemm1w_alt.store(on);
  return CONCH_OK;
}

namespace conch {

// Contract on top of scaled_gemm_mfma_supported (K-contiguous A and B^T, 16-byte aligned rows): OCP fp8, K a multiple of 256 bytes
// and >= 512, N a multiple of 16, unit-stride C rows of bf16 / fp16 (a bias vector of that dtype is added in the epilogue by the
// *_bias kernels), no fused gate/up form, one destination, every
// array below 2 GiB (32-bit buffer offsets with the sign bit kept for "row out of range"), at most 65535 tiles.
bool scaled_gemm_asm1w_supported(const ScaledGemmArgs& p) {
  if (!scaled_gemm_mfma_supported(p)) return false;
  if (p.in_dtype != CONCH_DT_FP8_E4M3FN && p.in_dtype != CONCH_DT_INT8) return false;
  if (p.out_dtype != CONCH_DT_BF16 && p.out_dtype != CONCH_DT_FP16) return false;
  if (p.fuse_silu || p.n_more || p.split_steps || p.a_src_dtype) return false;
  if (p.bias && (((uintptr_t)p.bias) & 1)) return false;
  if (p.k < 512 || p.k % 256 || p.n % 16 || p.c_stride_n != 1 || (((uintptr_t)p.c) & 15)) return false;
  if (!p.scale_a || !p.scale_b) return false;
  const int64_t lim = (int64_t)1 << 31;
  if ((p.m - 1) * p.a_stride_m + p.k >= lim || (p.n - 1) * p.b_stride_n + p.k >= lim) return false;
  if (((p.m - 1) * p.c_stride_m + p.n) * 2 >= lim) return false;
  if (p.a_stride_m >= lim || p.b_stride_n >= lim || p.c_stride_m >= lim) return false;
  const int64_t cols = (p.n + 223) / 224;  // the narrowest tile: the most tiles
  return ((p.m + kTileM - 1) / kTileM) * cols < 65536 && 8 * cols < 65536;
}

// ---- which tiling: 256 x (352 | 288 | 224) of this kernel, or the 256 x 256 tiles of gemm_mfma.hip ------------------------------
// One table, with where each number comes from (refit: tools/try_asm1w.py prints the per-round times this table models; the other
// dispatchers' fitted constants are in dispatch_fit.hpp, and tools/refit_dispatch.py reports every family's worst automatic pick).
struct Asm1wFit {
  double fixed_base_us, fixed_per_nt_us;   // a tile of NT n tiles per wave on a full chip: fixed part = base + per_nt NT ...
  double step_per_nt_us, step_base_us;     // ... + per 128-byte K step: base + per_nt NT
  double dma_bytes_per_us;                 // ... but not faster than the workgroup's (256 + 32 NT) x 128 bytes arrive by LDS-DMA
  double fixed256_us, step256_us;          // a 256 x 256 tile of gemm_mfma.hip
  double partial_base;                     // a round on a fraction f of the CUs costs (partial_base + (1 - partial_base) f) of a full one
  double margin;                           // take this kernel only when it wins by this factor
};
// profiles/r05/asm1w_persistent.txt, asm1w_widths.txt (one box each, interleaved): NT = 11: C3 2 rounds 110.6 us / 256-wide 3 rounds
// (2.69) 119.5; C5 10.25 rounds 1111.9 / 14 rounds 1135.1; full-chip round 55.3 us at K = 4096, 102.1 at 8192 -> fixed 8.5, step 1.4625;
// 256-wide 40.5 / 81.1.  The fixed part is ~3 us of tile start + an epilogue proportional to the tile width; a K step is 8 NT MFMAs
// of 32 cycles + ~45 cycles of barriers; the 224-column tile (NT = 7: 8192 x 8192 x 3584 in two full rounds, 142.1 us against 147.0
// on 256-wide tiles) runs its step in 1.01 us where the MFMAs need 0.94: the L2 -> LDS stream (60 KiB per step and CU) sets it.
constexpr Asm1wFit kAsm1wFit = {3.0, 0.5, 0.131, 0.0215, 61000.0, 0.5, 1.269, 0.85, 0.98};

static double weighted_rounds(int64_t tiles, int cus, double base) {
  const int64_t full = tiles / cus, rem = tiles % cus;
  return (double)full + (rem ? base + (1.0 - base) * (double)rem / (double)cus : 0.0);
}

static double asm1w_estimate_us(const ScaledGemmArgs& p, int nt, int cus) {
  const Asm1wFit& f = kAsm1wFit;
  const int64_t tiles = ((p.m + kTileM - 1) / kTileM) * ((p.n + 32 * nt - 1) / (32 * nt));
  const double steps = (double)(p.k / 128);
  const double step_us = std::max(f.step_base_us + f.step_per_nt_us * nt, (256.0 + 32.0 * nt) * 128.0 / f.dma_bytes_per_us);
  return weighted_rounds(tiles, cus, f.partial_base) * (f.fixed_base_us + f.fixed_per_nt_us * nt + steps * step_us);
}

// the tile width (n tiles per wave) this kernel would use on `p`: the debug override, else the cheapest by the model
static int asm1w_width(const ScaledGemmArgs& p, double* us_out = nullptr) {
  const int cus = device_cu_count();
  const int forced = g_gemm1w_nt.load();
  int best = 0;
  double best_us = 1e30;
  for (int nt : kWidths) {
    if (forced && nt != forced) continue;
    const double us = asm1w_estimate_us(p, nt, cus);
    if (us < best_us) {
      best_us = us;
      best = nt;
    }
  }
  if (us_out) *us_out = best_us;
  return best;
}

int scaled_asm1w_tile_columns(const ScaledGemmArgs& p) { return 32 * asm1w_width(p); }

bool scaled_asm1w_beats_tiles(const ScaledGemmArgs& p) {
  if (!scaled_gemm_asm1w_supported(p)) return false;
  const Asm1wFit& f = kAsm1wFit;
  const int cus = device_cu_count();
  double us = 0.0;
  asm1w_width(p, &us);
  const int64_t t256 = ((p.m + 255) / 256) * ((p.n + 255) / 256);
  const double us256 = weighted_rounds(t256, cus, f.partial_base) * (f.fixed256_us + f.step256_us * (double)(p.k / 128));
  return us < f.margin * us256;
}

// Both assembly modules loaded for the CURRENT device (conch_reserve_scratch: before any capture; a load inside a capture would
// put an allocation + upload into the graph).  Load failures are not errors here: they are remembered and the dispatchers fall back.
void preload_asm_modules() {
  (void)gemm1w_module();
  (void)mixed1w_module();
}

bool scaled_asm1w_loadable() {
  const Gemm1wModule* m = gemm1w_module();
  return m && m->rc == CONCH_OK;
}

bool mixed_asm1w_loadable() {
  const Mixed1wModule* m = mixed1w_module();
  return m && m->rc == CONCH_OK;
}

int launch_scaled_gemm_asm1w(const ScaledGemmArgs& p, hipStream_t stream) {
  Gemm1wModule* modp = gemm1w_module();
  if (!modp) return CONCH_ERR_HIP;
  if (modp->rc != CONCH_OK) {
    set_error("scaled_gemm (one-wave-per-SIMD kernel): the embedded code object did not load on this device");
    return modp->rc;
  }
  Gemm1wModule& mod = *modp;
  Gemm1wArgs a{};
  a.a = p.a;
  a.b = p.b;
  a.c = p.c;
  a.sa = p.scale_a;
  a.sb = p.scale_b;
  a.m = (uint32_t)p.m;
  a.n = (uint32_t)p.n;
  a.k = (uint32_t)p.k;
  a.lda = (uint32_t)p.a_stride_m;
  a.ldb = (uint32_t)p.b_stride_n;
  a.ldc = (uint32_t)p.c_stride_m;
  const int nt = asm1w_width(p);
  const int tile_n = 32 * nt;
  a.tiles_m = (uint32_t)((p.m + kTileM - 1) / kTileM);
  a.tiles_n = (uint32_t)((p.n + tile_n - 1) / tile_n);
  a.nwg = a.tiles_m * a.tiles_n;
  // n / d = mulhi(n, ceil(2^32 / d)) for n, d < 2^16 (the error term n e / 2^32 with e < d stays below 1 / d)
  const uint32_t per_group = 8 * a.tiles_n;
  a.magic_pg = (uint32_t)((((uint64_t)1 << 32) + per_group - 1) / per_group);
  // persistent walk: workgroup w runs tiles w, w + grid, ...; grid = one workgroup per CU (a multiple of 8: a tile id keeps its XCD)
  const uint32_t cus = (uint32_t)device_cu_count();
  a.grid = a.nwg <= cus ? a.nwg : (cus & ~7u);
  if (const uint32_t cap = (uint32_t)g_gemm1w_grid.load(); cap && cap < a.grid) a.grid = cap;
  const uint32_t last = a.tiles_m % 8;  // height of the last raster group when it is not a full one
  a.magic_last = last ? (uint32_t)((((uint64_t)1 << 32) + last - 1) / last) : 0;
  a.sa_vec = p.scale_a_numel != 1;
  a.sb_vec = p.scale_b_numel != 1;
  a.acc_scale = p.acc_scale;
  a.out_dtype = (uint32_t)p.out_dtype;
  a.gate = p.gate;
  a.gate_run_if = (uint32_t)p.gate_run_if;
  size_t size = sizeof(a);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  a.probe = p.out_dtype == CONCH_DT_BF16 ? g_gemm1w_probe.load() : nullptr;
  const bool fp8 = p.in_dtype == CONCH_DT_FP8_E4M3FN;
  const bool diag_ok = p.out_dtype == CONCH_DT_BF16 && nt == 11 && !p.bias;  // the stamped twins (fp8, int8) and the A/B twin (fp8) exist for that form only
  const bool alt = g_gemm1w_alt.load() != 0 && diag_ok && fp8;
  if (!diag_ok) a.probe = nullptr;
  const int wi = nt == 11 ? 0 : nt == 9 ? 1 : 2;
  const hipFunction_t f = a.probe ? (!fp8 ? mod.i8_probe : alt ? mod.bf16_alt_probe : mod.bf16_probe)
                                  : alt ? mod.bf16_alt : mod.fn[wi][fp8 ? 0 : 1][p.out_dtype == CONCH_DT_BF16 ? 0 : 1][p.bias ? 1 : 0];
  if (p.bias) a.probe = (unsigned long long*)p.bias;  // the *_bias kernels read their vector from the diagnostic buffer's slot
  CONCH_HIP(hipModuleLaunchKernel(f, a.grid, 1, 1, 256, 1, 1, 0, stream, nullptr, extra));
  return CONCH_OK;
}


// ---- mixed_precision_gemm: the one-wave-per-SIMD int4 x fp16 kernel (csrc/asm/gen_mixed1w.py) --------------------------------
// Contract on top of mixed_gemm_mfma_supported: fp16 activations, scales and output, 4-bit weights with a bias and NO zero points
// (the reference benchmark's uint4b8 form), group_size 128, K % 128 == 0, N % 16 == 0, plain [K/8][N] words, no fused / pre-packed /
// split form, 16-byte aligned activation rows, 8-byte aligned C rows, every array below 2 GiB, fewer than 65536 tiles.
bool mixed_gemm_asm1w_supported(const MixedGemmArgs& p) {
  if (!mixed_gemm_mfma_supported(p)) return false;
  if (p.x_dtype != CONCH_DT_FP16 || p.out_dtype != CONCH_DT_FP16 || p.bits != 4 || p.zp_mode != CONCH_ZP_NONE) return false;
  if (p.group_size != 128 || p.k % 128 || p.k < 256 || p.n % 16) return false;
  if (p.fuse_silu || p.prepacked || p.split_steps || p.slabs) return false;
  if ((((uintptr_t)p.x) & 15) || p.x_stride_m % 8 || (((uintptr_t)p.c) & 7) || p.c_stride_m % 4) return false;
  if ((((uintptr_t)p.w_q) & 3) || (((uintptr_t)p.w_s) & 1)) return false;
  const int64_t lim = (int64_t)1 << 31;
  if (((p.m - 1) * p.x_stride_m + p.k) * 2 >= lim || ((p.k / 8 - 1) * p.wq_stride_k + p.n) * 4 >= lim) return false;
  if (((p.k / 128 - 1) * p.ws_stride_g + p.n) * 2 >= lim || ((p.m - 1) * p.c_stride_m + p.n) * 2 >= lim) return false;
  return ((p.m + 255) / 256) * ((p.n + 191) / 192) < 65536;
}

// tile width: 4 waves x 16 W columns, W = 3 or 4 -- the one that needs fewer (rounds of workgroups) x (work per workgroup)
int mixed_asm1w_width(const MixedGemmArgs& p) {
  const int forced = tuning(CONCH_TUNE_MIXED_TILE_NT);
  if (forced == 3 || forced == 4) return forced;
  const int cus = device_cu_count();
  const int64_t tm = (p.m + 255) / 256;
  const int64_t t3 = tm * ((p.n + 191) / 192), t4 = tm * ((p.n + 255) / 256);
  const double c3 = (double)((t3 + cus - 1) / cus) * 3.3, c4 = (double)((t4 + cus - 1) / cus) * 4.3;
  return c3 < c4 ? 3 : 4;
}

int launch_mixed_gemm_asm1w(const MixedGemmArgs& p, hipStream_t stream) {
  Mixed1wModule* modp = mixed1w_module();
  if (!modp) return CONCH_ERR_HIP;
  if (modp->rc != CONCH_OK) {
    set_error("mixed_precision_gemm (one-wave-per-SIMD kernel): the embedded code object did not load on this device");
    return modp->rc;
  }
  Mixed1wModule& mod = *modp;
  const int w = mixed_asm1w_width(p);
  Mixed1wArgs a{};
  a.x = p.x;
  a.wq = p.w_q;
  a.ws = p.w_s;
  a.c = p.c;
  a.m = (uint32_t)p.m;
  a.n = (uint32_t)p.n;
  a.k = (uint32_t)p.k;
  a.ldx = (uint32_t)p.x_stride_m;
  a.ldq = (uint32_t)p.wq_stride_k;
  a.lds = (uint32_t)p.ws_stride_g;
  a.ldc = (uint32_t)p.c_stride_m;
  a.off = p.weight_bias;
  a.tiles_m = (uint32_t)((p.m + 255) / 256);
  a.tiles_n = (uint32_t)((p.n + 64 * w - 1) / (64 * w));
  a.nwg = a.tiles_m * a.tiles_n;
  const uint32_t per_group = 8 * a.tiles_n;
  a.magic_pg = (uint32_t)((((uint64_t)1 << 32) + per_group - 1) / per_group);
  a.grid = a.nwg;
  const uint32_t last = a.tiles_m % 8;
  a.magic_last = last ? (uint32_t)((((uint64_t)1 << 32) + last - 1) / last) : 0;
  a.probe = g_mixed1w_probe.load();
  size_t size = sizeof(a);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  const hipFunction_t f = a.probe ? (w == 3 ? mod.w3_probe : mod.w4_probe) : (w == 3 ? mod.w3 : mod.w4);
  CONCH_HIP(hipModuleLaunchKernel(f, a.nwg, 1, 1, 256, 1, 1, 0, stream, nullptr, extra));
  return CONCH_OK;
}

}  // namespace conch
