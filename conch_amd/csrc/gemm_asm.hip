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

// kernel argument block: the layout S_* of gen_gemm1w.py reads with s_load_dword*
struct Gemm1wArgs {
  const void* a;
  const void* b;
  void* c;
  const float* sa;
  const float* sb;
  uint32_t m, n, k, lda, ldb, ldc;
  uint32_t tiles_m, tiles_n, nwg, magic_pg, shift_pg, magic_last, sa_vec, sb_vec;
  float acc_scale;
  uint32_t out_dtype;
  unsigned long long* probe;  // diagnostic kernel only: [workgroup][5][2] 64-bit stamps (s_memtime, s_memrealtime); else ignored
};
static_assert(sizeof(Gemm1wArgs) == 112, "kernarg block of conch_gemm1w_*: 112 bytes");
static_assert(offsetof(Gemm1wArgs, m) == 40 && offsetof(Gemm1wArgs, tiles_m) == 64 && offsetof(Gemm1wArgs, acc_scale) == 96, "kernarg offsets");

constexpr int kTileM = 256, kTileN = 352;

struct Gemm1wModule {
  hipModule_t mod = nullptr;
  hipFunction_t bf16 = nullptr, f16 = nullptr, bf16_probe = nullptr;
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
    if (e != hipSuccess) {
      set_error("scaled_gemm (one-wave-per-SIMD kernel): loading the embedded code object failed: %s", hipGetErrorString(e));
      m.rc = CONCH_ERR_HIP;
    }
  });
  return m;
}

std::atomic<unsigned long long*> g_gemm1w_probe{nullptr};

}  // namespace
}  // namespace conch

// Diagnostic: with a non-NULL buffer of (workgroups x 10) 64-bit words the NEXT bf16 launches run the stamped twin of the kernel
// (the product kernel contains no stamp) and wave 0 of every workgroup writes (shader clock, 100 MHz clock) at kernel entry, K loop
// start, K loop end, epilogue stores issued, stores retired.  NULL switches back.  Used by tools/try_asm1w.py only.
extern "C" int conch_debug_gemm1w_probe(unsigned long long* buffer) {
  conch::g_gemm1w_probe.store(buffer);
  return CONCH_OK;
}

namespace conch {

// Contract on top of scaled_gemm_mfma_supported (K-contiguous A and B^T, 16-byte aligned rows): OCP fp8, K a multiple of 256 bytes
// and >= 512, N a multiple of 16, unit-stride C rows of bf16 / fp16, no bias, no fused gate/up form, one destination, every
// array below 2 GiB (32-bit buffer offsets with the sign bit kept for "row out of range"), at most 65535 tiles.
bool scaled_gemm_asm1w_supported(const ScaledGemmArgs& p) {
  if (!scaled_gemm_mfma_supported(p)) return false;
  if (p.in_dtype != CONCH_DT_FP8_E4M3FN) return false;
  if (p.out_dtype != CONCH_DT_BF16 && p.out_dtype != CONCH_DT_FP16) return false;
  if (p.bias || p.fuse_silu || p.n_more || p.split_steps || p.gate || p.a_src_dtype) return false;
  if (p.k < 512 || p.k % 256 || p.n % 16 || p.c_stride_n != 1 || (((uintptr_t)p.c) & 15)) return false;
  if (!p.scale_a || !p.scale_b) return false;
  const int64_t lim = (int64_t)1 << 31;
  if ((p.m - 1) * p.a_stride_m + p.k >= lim || (p.n - 1) * p.b_stride_n + p.k >= lim) return false;
  if (((p.m - 1) * p.c_stride_m + p.n) * 2 >= lim) return false;
  if (p.a_stride_m >= lim || p.b_stride_n >= lim || p.c_stride_m >= lim) return false;
  const int64_t tiles = ((p.m + kTileM - 1) / kTileM) * ((p.n + kTileN - 1) / kTileN);
  return tiles >= 1 && tiles < 65536 && 8 * ((p.n + kTileN - 1) / kTileN) < 65536;
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
  a.shift_pg = 0;
  const uint32_t last = a.tiles_m % 8;  // height of the last raster group when it is not a full one
  a.magic_last = last ? (uint32_t)((((uint64_t)1 << 32) + last - 1) / last) : 0;
  a.sa_vec = p.scale_a_numel != 1;
  a.sb_vec = p.scale_b_numel != 1;
  a.acc_scale = p.acc_scale;
  a.out_dtype = (uint32_t)p.out_dtype;
  size_t size = sizeof(a);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  a.probe = p.out_dtype == CONCH_DT_BF16 ? g_gemm1w_probe.load() : nullptr;
  const hipFunction_t f = a.probe ? mod.bf16_probe : p.out_dtype == CONCH_DT_BF16 ? mod.bf16 : mod.f16;
  CONCH_HIP(hipModuleLaunchKernel(f, a.nwg, 1, 1, 256, 1, 1, 0, stream, nullptr, extra));
  return CONCH_OK;
}

}  // namespace conch
