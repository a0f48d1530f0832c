// Launcher of the ONE-WAVE-PER-SIMD fp8 scaled-GEMM kernel (csrc/asm/gen_gemm1w.py writes its assembly; the build assembles
// it for gfx950 and embeds the code object below).  Same place in the call stack as launch_scaled_gemm_mfma: replaces the
// reference's Triton `_gemm_kernel` + `scaled_gemm_launcher` (conch/kernels/quantization/gemm.py:219-457, :564-627) for large
// fp8 problems -- 256 x 352 output tiles, four waves of 128 x 176, every register of the SIMD in one wave.
//
// The kernel is loaded from memory with hipModuleLoadData on first use (one module per process; one process per GPU) and
// launched with hipModuleLaunchKernel on the caller's stream, so it is stream-ordered and graph-capturable like every other
// kernel of the library.
#include <atomic>
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

struct Mixed1wModule {
  hipModule_t mod = nullptr;
  hipFunction_t w3 = nullptr, w4 = nullptr, w3_probe = nullptr, w4_probe = nullptr;
  int rc = CONCH_OK;
};

Mixed1wModule& mixed1w_module() {
  static Mixed1wModule m;
  static std::once_flag once;
  std::call_once(once, [] {
    hipError_t e = hipModuleLoadData(&m.mod, kMixed1wCodeObject);
    if (e == hipSuccess) e = hipModuleGetFunction(&m.w3, m.mod, "conch_mixed1w_f16_i4_w3");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.w4, m.mod, "conch_mixed1w_f16_i4_w4");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.w3_probe, m.mod, "conch_mixed1w_f16_i4_w3_probe");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.w4_probe, m.mod, "conch_mixed1w_f16_i4_w4_probe");
    if (e != hipSuccess) {
      set_error("mixed_precision_gemm (one-wave-per-SIMD kernel): loading the embedded code object failed: %s", hipGetErrorString(e));
      m.rc = CONCH_ERR_HIP;
    }
  });
  return m;
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
  unsigned long long* probe;  // diagnostic kernel only: [tile][5][2] 64-bit stamps (s_memtime, s_memrealtime); else ignored
  const int* gate;            // NULL, or: run only if (*gate != 0) == (gate_run_if != 0)  (ScaledGemmArgs::gate)
  uint32_t gate_run_if;
  uint32_t pad;
};
static_assert(sizeof(Gemm1wArgs) == 128 && offsetof(Gemm1wArgs, gate) == 112 && offsetof(Gemm1wArgs, gate_run_if) == 120, "kernarg block of conch_gemm1w_*");
static_assert(offsetof(Gemm1wArgs, m) == 40 && offsetof(Gemm1wArgs, tiles_m) == 64 && offsetof(Gemm1wArgs, acc_scale) == 96, "kernarg offsets");

constexpr int kTileM = 256, kTileN = 352;

struct Gemm1wModule {
  hipModule_t mod = nullptr;
  hipFunction_t bf16 = nullptr, f16 = nullptr, bf16_probe = nullptr, bf16_alt = nullptr, bf16_alt_probe = nullptr, i8_bf16 = nullptr, i8_f16 = nullptr;
  int rc = CONCH_OK;
};

Gemm1wModule& gemm1w_module() {
  static Gemm1wModule m;
  static std::once_flag once;
  std::call_once(once, [] {
    hipError_t e = hipModuleLoadData(&m.mod, kGemm1wCodeObject);
    if (e == hipSuccess) e = hipModuleGetFunction(&m.bf16, m.mod, "conch_gemm1w_fp8_bf16");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.f16, m.mod, "conch_gemm1w_fp8_f16");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.bf16_probe, m.mod, "conch_gemm1w_fp8_bf16_probe");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.bf16_alt, m.mod, "conch_gemm1w_fp8_bf16_alt");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.bf16_alt_probe, m.mod, "conch_gemm1w_fp8_bf16_alt_probe");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.i8_bf16, m.mod, "conch_gemm1w_i8_bf16");
    if (e == hipSuccess) e = hipModuleGetFunction(&m.i8_f16, m.mod, "conch_gemm1w_i8_f16");
    if (e != hipSuccess) {
      set_error("scaled_gemm (one-wave-per-SIMD kernel): loading the embedded code object failed: %s", hipGetErrorString(e));
      m.rc = CONCH_ERR_HIP;
    }
  });
  return m;
}

std::atomic<unsigned long long*> g_gemm1w_probe{nullptr};
std::atomic<int> g_gemm1w_alt{0};

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

// Diagnostic: 1 = the bf16 launches run the A/B twin of the kernel (the generator's `alt` choice; bit-identical results), 0 = back.
extern "C" int conch_debug_gemm1w_alt(int on) {
  conch::g_gemm1w_alt.store(on);
  return CONCH_OK;
}

namespace conch {

// Contract on top of scaled_gemm_mfma_supported (K-contiguous A and B^T, 16-byte aligned rows): OCP fp8, K a multiple of 256 bytes
// and >= 512, N a multiple of 16, unit-stride C rows of bf16 / fp16, no bias, no fused gate/up form, one destination, every
// array below 2 GiB (32-bit buffer offsets with the sign bit kept for "row out of range"), at most 65535 tiles.
bool scaled_gemm_asm1w_supported(const ScaledGemmArgs& p) {
  if (!scaled_gemm_mfma_supported(p)) return false;
  if (p.in_dtype != CONCH_DT_FP8_E4M3FN && p.in_dtype != CONCH_DT_INT8) return false;
  if (p.out_dtype != CONCH_DT_BF16 && p.out_dtype != CONCH_DT_FP16) return false;
  if (p.bias || p.fuse_silu || p.n_more || p.split_steps || p.a_src_dtype) return false;
  if (p.k < 512 || p.k % 256 || p.n % 16 || p.c_stride_n != 1 || (((uintptr_t)p.c) & 15)) return false;
  if (!p.scale_a || !p.scale_b) return false;
  const int64_t lim = (int64_t)1 << 31;
  if ((p.m - 1) * p.a_stride_m + p.k >= lim || (p.n - 1) * p.b_stride_n + p.k >= lim) return false;
  if (((p.m - 1) * p.c_stride_m + p.n) * 2 >= lim) return false;
  if (p.a_stride_m >= lim || p.b_stride_n >= lim || p.c_stride_m >= lim) return false;
  const int64_t tiles = ((p.m + kTileM - 1) / kTileM) * ((p.n + kTileN - 1) / kTileN);
  return tiles >= 1 && tiles < 65536 && 8 * ((p.n + kTileN - 1) / kTileN) < 65536;
}

// ---- when the 256 x 352 tiles beat the 256 x 256 tiles -----------------------------------------------------------------------
// One table, with where each number comes from (refit: tools/try_asm1w.py prints the per-round times this table models; the other
// dispatchers' fitted constants are in dispatch_fit.hpp, and tools/refit_dispatch.py reports every family's worst automatic pick).
struct Asm1wFit {
  double fixed352_us, step352_us;   // a 256 x 352 tile on a full chip: fixed part + per 128-byte K step
  double fixed256_us, step256_us;   // a 256 x 256 tile of gemm_mfma.hip
  double partial_base;              // a round on a fraction f of the CUs costs (partial_base + (1 - partial_base) f) of a full one
  double margin;                    // take the 352-wide tiles only when they win by this factor
};
// profiles/r05/asm1w_persistent.txt (the persistent form, one box, interleaved): C3 2 rounds 110.6 us / 3 rounds (2.69) 119.5 us;
// C5 10.25 rounds 1111.9 us / 14 rounds 1135.1 us; 8192 x 8192 x 3584 1.375 rounds 178.7 / 1.75 rounds 149.7; 4096 x 8192 x 4096 0.75
// rounds 89.8 / 1 round 79.9.  Full-chip round: 352-wide 55.3 us at K = 4096, 102.1 at 8192; 256-wide 40.5 / 81.1.
constexpr Asm1wFit kAsm1wFit = {8.5, 1.4625, 0.5, 1.269, 0.85, 0.98};

static double weighted_rounds(int64_t tiles, int cus, double base) {
  const int64_t full = tiles / cus, rem = tiles % cus;
  return (double)full + (rem ? base + (1.0 - base) * (double)rem / (double)cus : 0.0);
}

bool scaled_asm1w_beats_tiles(const ScaledGemmArgs& p) {
  if (!scaled_gemm_asm1w_supported(p)) return false;
  const Asm1wFit& f = kAsm1wFit;
  const int cus = device_cu_count();
  const double steps = (double)(p.k / 128);
  const int64_t t352 = ((p.m + kTileM - 1) / kTileM) * ((p.n + kTileN - 1) / kTileN);
  const int64_t t256 = ((p.m + 255) / 256) * ((p.n + 255) / 256);
  const double us352 = weighted_rounds(t352, cus, f.partial_base) * (f.fixed352_us + f.step352_us * steps);
  const double us256 = weighted_rounds(t256, cus, f.partial_base) * (f.fixed256_us + f.step256_us * steps);
  return us352 < f.margin * us256;
}

int launch_scaled_gemm_asm1w(const ScaledGemmArgs& p, hipStream_t stream) {
  Gemm1wModule& mod = gemm1w_module();
  if (mod.rc != CONCH_OK) return mod.rc;
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
  a.tiles_m = (uint32_t)((p.m + kTileM - 1) / kTileM);
  a.tiles_n = (uint32_t)((p.n + kTileN - 1) / kTileN);
  a.nwg = a.tiles_m * a.tiles_n;
  // n / d = mulhi(n, ceil(2^32 / d)) for n, d < 2^16 (the error term n e / 2^32 with e < d stays below 1 / d)
  const uint32_t per_group = 8 * a.tiles_n;
  a.magic_pg = (uint32_t)((((uint64_t)1 << 32) + per_group - 1) / per_group);
  // persistent walk: workgroup w runs tiles w, w + grid, ...; grid = one workgroup per CU (a multiple of 8: a tile id keeps its XCD)
  const uint32_t cus = (uint32_t)device_cu_count();
  a.grid = a.nwg <= cus ? a.nwg : (cus & ~7u);
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
  const bool alt = g_gemm1w_alt.load() != 0 && p.out_dtype == CONCH_DT_BF16 && fp8;
  if (!fp8) a.probe = nullptr;
  const hipFunction_t f = !fp8 ? (p.out_dtype == CONCH_DT_BF16 ? mod.i8_bf16 : mod.i8_f16)
                          : a.probe ? (alt ? mod.bf16_alt_probe : mod.bf16_probe) : alt ? mod.bf16_alt : p.out_dtype == CONCH_DT_BF16 ? mod.bf16 : mod.f16;
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
  Mixed1wModule& mod = mixed1w_module();
  if (mod.rc != CONCH_OK) return mod.rc;
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
