// extern "C" entry points of libconch_amd.so (declared in include/conch_amd.h): argument
// validation, kernel selection, error strings.  No torch types, no exceptions across the boundary.
#include <algorithm>
#include <cmath>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.hpp"
#include "gemm.hpp"
#include "dispatch_fit.hpp"

namespace conch {

static thread_local char g_error[512] = "";
static std::atomic<int> g_tuning[CONCH_TUNE__COUNT] = {};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

int tuning(int key) { return (key >= 0 && key < CONCH_TUNE__COUNT) ? g_tuning[key].load() : 0; }

int device_cu_count() {
  static std::atomic<int> cache[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cache[dev].load();
  if (n == 0) {
    int v = 0;
    n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    cache[dev].store(n);
  }
  return n;
}

// ---------------------------------------------------------------------------------------------
// Library-owned device scratch.  One buffer per (device, stream, slot).  A buffer that has been handed to a kernel is
// NEVER freed or moved while the process lives: a grown slot gets a fresh allocation (at least twice the old size, so
// the retired ones sum to less than the live one) and the old pointer is kept on a retired list, because launches
// already enqueued -- or captured into a hipGraph -- still read it.  Growth allocates with hipMalloc, which is not
// legal inside stream capture: a call that would have to grow a slot while `stream` is capturing fails with
// CONCH_ERR_INVALID_ARGUMENT instead (reserve first: conch_reserve_scratch).  No host synchronisation anywhere.
// ---------------------------------------------------------------------------------------------
namespace {
struct ScratchKey {
  int device;
  hipStream_t stream;
  int slot;
  bool operator<(const ScratchKey& o) const {
    if (device != o.device) return device < o.device;
    if (stream != o.stream) return stream < o.stream;
    return slot < o.slot;
  }
};
struct ScratchBuf {
  void* ptr = nullptr;
  size_t size = 0;
};
std::mutex g_scratch_mu;
std::map<ScratchKey, ScratchBuf> g_scratch;
std::vector<void*> g_scratch_retired;
}  // namespace

int get_scratch(hipStream_t stream, int slot, size_t bytes, void** out, bool zero_on_alloc) {
  int device = 0;
  CONCH_HIP(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  ScratchBuf& b = g_scratch[ScratchKey{device, stream, slot}];
  if (b.size < bytes) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
      set_error("scratch slot %d needs %zu bytes (has %zu) while the stream is being captured: call "
                "conch_reserve_scratch(stream, bytes) before capture", slot, bytes, b.size);
      return CONCH_ERR_INVALID_ARGUMENT;
    }
    (void)hipGetLastError();
    size_t want = bytes > 2 * b.size ? bytes : 2 * b.size;
    want = (want + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    void* fresh = nullptr;
    CONCH_HIP(hipMalloc(&fresh, want));
    if (zero_on_alloc) CONCH_HIP(hipMemset(fresh, 0, want));
    if (b.ptr) g_scratch_retired.push_back(b.ptr);  // still referenced by enqueued / captured work
    b.ptr = fresh;
    b.size = want;
  }
  *out = b.ptr;
  return CONCH_OK;
}

namespace {

bool is_in8(int dt) {
  return dt == CONCH_DT_INT8 || dt == CONCH_DT_FP8_E4M3FN || dt == CONCH_DT_FP8_E4M3FNUZ;
}
bool is_out16(int dt) { return dt == CONCH_DT_FP16 || dt == CONCH_DT_BF16; }

int check_scaled(const ScaledGemmArgs& p) {
  CONCH_CHECK_ARG(p.m >= 0 && p.n >= 0 && p.k >= 0, "scaled_gemm: negative shape (%lld,%lld,%lld)",
                  (long long)p.m, (long long)p.n, (long long)p.k);
  if (!is_in8(p.in_dtype)) {
    set_error("scaled_gemm: unsupported input dtype %d (want INT8, FP8_E4M3FN or FP8_E4M3FNUZ)", p.in_dtype);
    return CONCH_ERR_UNSUPPORTED;
  }
  if (!is_out16(p.out_dtype)) {
    set_error("scaled_gemm: unsupported output dtype %d (want FP16 or BF16)", p.out_dtype);
    return CONCH_ERR_UNSUPPORTED;
  }
  if (p.m == 0 || p.n == 0) return CONCH_OK;
  CONCH_CHECK_ARG(p.c, "scaled_gemm: NULL pointer (c=%p)", p.c);
  CONCH_CHECK_ARG(p.k == 0 || (p.a && p.b), "scaled_gemm: NULL operand (a=%p b=%p)", p.a, p.b);
  CONCH_CHECK_ARG(!p.scale_a || p.scale_a_numel == 1 || p.scale_a_numel == p.m,
                  "scaled_gemm: scale_a has %lld elements, want 1 or M=%lld", (long long)p.scale_a_numel,
                  (long long)p.m);
  CONCH_CHECK_ARG(!p.scale_b || p.scale_b_numel == 1 || p.scale_b_numel == p.n,
                  "scaled_gemm: scale_b has %lld elements, want 1 or N=%lld", (long long)p.scale_b_numel,
                  (long long)p.n);
  return CONCH_OK;
}

enum ScaledKernel { kKernelTiled = 0, kKernelMid = 1, kKernelSkinny = 2 };

// The cheapest of three kernels by a cost model, in us.  Round 1 fitted it to back-to-back launches on ONE weight buffer, which
// stays in L2 and the 256 MiB Infinity Cache; a decode step streams every layer's weights from HBM, where the tile kernels with few
// workgroups are bound by what ONE CU can pull from HBM (a 128x128 tile's 256 K bytes at ~33 GB/s, a 256x256 tile's 512 K bytes
// at ~48 GB/s) while the split-K form spreads the stream over the chip.  Round 2 refits it to that regime
// (tools/dispatch_cold_sweep.py: every candidate timed while cycling through > 600 MB of weight buffers, M in 8..1024,
// six (K, N) pairs, int8 and fp8; profiles/r02/dispatch_cold_sweep_before.txt / _after.txt):
//   split-K skinny  5.3 + c(M) N K        (re-reads the A slice per 64-column block; two row blocks above M = 128)
//   256x256 tiles   rounds x (41 + 0.06 t) K/4096,  t = tiles per round of 256 workgroups
//   128x128 tiles   rounds x (31 + 0.02 t) K/4096,  t = tiles per round of 512 workgroups (two per CU); with at most one tile
//                   per CU (the 4-stage ring): (23 + 0.08 max(0, t - 128)) K/4096
// Round 3 re-ran the sweep after the split-K and decode kernels got faster (profiles/r03/dispatch_cold_sweep.txt: "before" =
// the round-2 constants): c(M <= 16) 2.1 / 2.2 -> 2.0 / 2.1 and the ring's slope 0.04 -> 0.08 (8-16 x 4096 x 28672 had gone to
// the 128x128 tiles at 30.5 us where the split-K form now takes 27.4-28.9).
// The warm fit (round 1) was 4.5-5.5 + c N K with c 1.7 .. 9.7, rounds x (34 + 0.05 t), rounds x (19 + 0.0176 t): with it the
// dispatcher picked the 128x128 tiles where the split-K form is 15-47 % faster on cold weights (96-128 x 4096 x 11008,
// 8-32 x 4096 x 28672, 256 x 4096 x 4096).
ScaledKernel choose_scaled_kernel(const ScaledGemmArgs& p, double* tiles_us_out = nullptr) {
  const double kscale = (double)p.k / 4096.0;
  const int64_t tiles256 = ((p.m + 255) / 256) * ((p.n + 255) / 256);
  const int64_t rounds256 = (tiles256 + 255) / 256;
  const fit::ScaledTiles& ft = fit::kScaledTiles;  // (every fitted number: dispatch_fit.hpp)
  const double tiled_us = (double)rounds256 * (ft.t256_base + ft.t256_per_tile * (double)tiles256 / (double)rounds256) * kscale;
  const int64_t tiles128 = ((p.m + 127) / 128) * ((p.n + 127) / 128);
  const int64_t rounds128 = (tiles128 + 511) / 512;
  // at most one tile per CU: the 4-stage ring (gemm_mid.hip), three K steps in flight per CU instead of one
  double mid_us = tiles128 <= device_cu_count() ? (ft.ring_base + ft.ring_per_tile * (double)std::max<int64_t>(0, tiles128 - 128)) * kscale
                                                : (double)rounds128 * (ft.t128_base + ft.t128_per_tile * (double)tiles128 / (double)rounds128) * kscale;
  // split-K form of the 128 x 128 tiles (gemm_mid.hip, mid_split_slices; round 4, profiles/r04/mid_splitk_sweep.txt): two launches
  // (~8 us of fixed cost: with 7 the fp8 sibling of C2, 128 x 4096 x 4096, left the one-launch split-K kernel's 10.1 us for 15.1; with 9, 32-48 x 11008 x 4096 stayed on the skinny kernel at 13-22 % more on weights streamed from HBM: profiles/r04/dispatch_cold_sweep.txt), a K step 0.52-0.68 us with up to one workgroup per CU and ~1.0 us with two, the slabs written
  // and read back at ~8 bytes per us and CU-free MB
  if (const int msl = mid_split_slices(p); msl > 1) {
    const double cus = (double)device_cu_count();
    const double wgs = (double)tiles128 * msl;
    const double steps_slice = std::ceil((double)(p.k / 128) / msl);
    const double per_step = wgs <= cus ? ft.split_step + ft.split_step_fill * wgs / cus : std::ceil(wgs / (2.0 * cus));
    mid_us = ft.split_fixed + steps_slice * per_step + (double)msl * (double)p.m * (double)p.n * ft.split_slab_per_elem;
  }
  ScaledKernel pick = kKernelTiled;
  double best = tiled_us;
  if (mid_us < best) {
    best = mid_us;
    pick = kKernelMid;
  }
  if (tiles_us_out) *tiles_us_out = best;
  if (scaled_gemm_skinny_supported(p)) {
    const fit::ScaledSkinny& fs = fit::kScaledSkinny;
    const double c = p.m <= 8 ? fs.c8 : p.m <= 16 ? fs.c16 : p.m <= 32 ? fs.c32 : p.m <= 48 ? fs.c48 : p.m <= 64 ? fs.c64 : p.m <= 96 ? fs.c96
                     : p.m <= 128 ? fs.c128 : p.m <= 192 ? fs.c192 : fs.c256;  // steps at the 32- / 64- / 128-row forms and at the second row block
    const double skinny_us = fs.fixed + c * 1e-7 * (double)p.n * (double)p.k;
    if (skinny_us < best) pick = kKernelSkinny;
  }
  return pick;
}

}  // namespace
int scaled_kernel_choice(const ScaledGemmArgs& p) { return (int)choose_scaled_kernel(p); }
double scaled_tiles_estimate_us(const ScaledGemmArgs& p) {
  double us = 0.0;
  choose_scaled_kernel(p, &us);
  return us;
}
namespace {

// Which kernel the MFMA-contract path of scaled_gemm runs for `p` at GEMM variant `variant` -- no launch (run_scaled_fast launches
// exactly this; conch_debug_scaled_plan reports it).  kScaledPickError: a forced kernel's contract is not met (+ the error text).
// Diagnostic A/B switch (conch_debug_fnuz_in_registers; tools/time_fnuz_decode.py): 0 = e4m3fnuz decode shapes take round 5's scan +
// gated kernels again instead of the split-K kernel that splits the special codes off in registers
std::atomic<int> g_fnuz_in_registers{1};

enum ScaledPick { kScaledPickError = -1, kScaledPickTiles = 0, kScaledPickMid = 1, kScaledPickSkinny = 2, kScaledPickAsm = 3 };

ScaledPick pick_scaled_fast(const ScaledGemmArgs& p, int variant) {
  if (variant == 4 && !scaled_gemm_skinny_supported(p)) {
    set_error("scaled_gemm: skinny variant forced but its contract is not met (M <= 256; N %% 4 == 0 or K %% 1024 == 0)");
    return kScaledPickError;
  }
  if (variant == 7) {
    if (!scaled_gemm_asm1w_supported(p)) {
      set_error("scaled_gemm: variant 7 (one-wave-per-SIMD assembly kernel) forced but its contract is not met (e4m3fn or int8, K %% 256 == 0, K >= 512, "
                "N %% 16 == 0, bf16 / fp16 row-major C, arrays below 2 GiB)");
      return kScaledPickError;
    }
    return kScaledPickAsm;
  }
  const ScaledKernel pick = variant == 4 ? kKernelSkinny : variant == 6 ? kKernelMid : variant == 0 ? choose_scaled_kernel(p) : kKernelTiled;
  if (pick == kKernelSkinny && scaled_gemm_skinny_supported(p)) return kScaledPickSkinny;
  if (pick == kKernelMid) return kScaledPickMid;
  // large fp8 problems whose 256 x 352 tiling fills the chip in fewer, fuller rounds: the one-wave-per-SIMD assembly kernel
  if (variant == 0 && pick == kKernelTiled && scaled_asm1w_beats_tiles(p)) return kScaledPickAsm;
  return kScaledPickTiles;
}

int run_scaled_fast(const ScaledGemmArgs& p, int variant, hipStream_t stream) {
  switch (pick_scaled_fast(p, variant)) {
    case kScaledPickSkinny: return launch_scaled_gemm_skinny(p, stream);
    case kScaledPickMid: return launch_scaled_gemm_mid(p, stream);
    case kScaledPickAsm:
      // the automatic pick on a device where the embedded code object did not load: the 256 x 256 HIP tiles serve (bit-identical
      // results); a FORCED assembly variant reports the load error instead
      if (variant == 0 && !scaled_asm1w_loadable()) return launch_scaled_gemm_mfma(p, 5, stream);
      return launch_scaled_gemm_asm1w(p, stream);
    case kScaledPickTiles: return launch_scaled_gemm_mfma(p, (variant == 2 || variant == 3) ? variant : 5, stream);
    default: return CONCH_ERR_UNSUPPORTED;
  }
}

}  // namespace
// One fp32 1.0 per device, for "no scale on this side" (NULL scale pointer): every kernel then runs unchanged.
int unit_scale(const float** out) {
  static std::mutex mu;
  static float* ones[64] = {};
  int dev = 0;
  CONCH_HIP(hipGetDevice(&dev));
  CONCH_CHECK_ARG(dev >= 0 && dev < 64, "device id %d out of range", dev);
  std::lock_guard<std::mutex> lock(mu);
  if (!ones[dev]) {
    const float one = 1.0f;
    float* ptr = nullptr;
    CONCH_HIP(hipMalloc((void**)&ptr, 256));
    CONCH_HIP(hipMemcpy(ptr, &one, sizeof(one), hipMemcpyHostToDevice));
    ones[dev] = ptr;
  }
  *out = ones[dev];
  return CONCH_OK;
}
namespace {

int run_scaled(const ScaledGemmArgs& p_in, hipStream_t stream) {
  ScaledGemmArgs p = p_in;
  if (int rc = check_scaled(p)) return rc;
  if (p.m == 0 || p.n == 0) return CONCH_OK;
  if (!p.scale_a || !p.scale_b) {  // channel-scale modes NONE / WEIGHT_ONLY / ACTIVATION_ONLY
    const float* one = nullptr;
    if (int rc = unit_scale(&one)) return rc;
    if (!p.scale_a) { p.scale_a = one; p.scale_a_numel = 1; }
    if (!p.scale_b) { p.scale_b = one; p.scale_b_numel = 1; }
  }
  const int variant = tuning(CONCH_TUNE_GEMM_VARIANT);
  if (variant != 1 && scaled_gemm_mfma_supported(p)) return run_scaled_fast(p, variant, stream);
  if (variant == 0 && p.in_dtype == CONCH_DT_FP8_E4M3FNUZ && !p.fuse_silu) {
    // e4m3fnuz (the dtype the reference gives AMD users: conch/ops/quantization/fp8.py:27,54) at the fp8 rate: see gemm.hpp,
    // ScaledGemmArgs::acc_scale.  Tile kernels only (one launch each; the split-K forms keep the expansion path).  On the
    // stream: zero the flag, scan both operands, the fp8 kernel (runs if no special code), the expansion + bf16 kernel (run
    // if one is there).  Three of the launches return at once; the host never learns which.
    ScaledGemmArgs f = p;
    f.in_dtype = CONCH_DT_FP8_E4M3FN;
    // decode sizes (round 6): the split-K skinny kernel passes both operands through registers and splits the special codes off
    // there (gemm_skinny.hip, fnuz_split) -- no scan, no flag, no expansion, saturated or not
    // Taken where the dispatcher gives the e4m3fn twin of the problem to that kernel, and up to 64 rows whatever it picks: there the
    // 128 x 128 tiles it may prefer for e4m3fn cost a scan + (saturated data) the expansion on top -- 64 x 8192 x 8192: 28.8 / 33.0 us
    // clean / saturated on this kernel against 42.4 / 103.5; from 96 rows the per-fragment test of eight A fragments per step loses
    // to the scan on clean data (profiles/r06/fnuz_decode_ab.txt).  (g_fnuz_in_registers: 0 = never, 2 = wherever the kernel runs.)
    const int in_regs = g_fnuz_in_registers.load();
    if (in_regs && scaled_gemm_mfma_supported(f) && (in_regs == 2 || p.m <= 64 || choose_scaled_kernel(f) == kKernelSkinny) &&
        scaled_gemm_skinny_fnuz_supported(f)) {
      ScaledGemmArgs g = p;  // (in_dtype stays e4m3fnuz: launch_scaled_gemm_skinny picks the splitting instantiation by it)
      return launch_scaled_gemm_skinny(g, stream);
    }
    if (scaled_gemm_mfma_supported(f) && fnuz_expansion_fits(p) && p.scale_a_numel <= kFnuzMaxScales) {
      void* flag = nullptr;
      if (int rc = get_scratch(stream, kScratchFlags, kFlagsBytes, &flag)) return rc;
      CONCH_HIP(hipMemsetAsync(flag, 0, 4, stream));
      const ScaledKernel pick = choose_scaled_kernel(f);
      if (pick == kKernelTiled) {  // the 256 x 256 tiles carry the factor themselves (acc_scale) and honour the gate
        if (int rc = launch_fnuz_scan(p, (int*)flag, stream)) return rc;
        f.acc_scale = 0.25f;
      } else {
        // the 128 x 128 tiles' split-K form and the split-K skinny kernels (decode batches: 68 us through the expansion where
        // e4m3fn takes 6.4, round 4) know neither acc_scale nor -- their reduce kernels -- the gate: they get a scaled COPY of
        // scale_a, made by the scan, and run whatever the flag says; if a special code is there, the gated expansion path below
        // overwrites every element of C afterwards.
        float* sa_quarter = (float*)((char*)flag + 256);
        if (int rc = launch_fnuz_scan(p, (int*)flag, stream, sa_quarter)) return rc;
        f.scale_a = sa_quarter;
      }
      f.gate = (const int*)flag;
      f.gate_run_if = 0;
      if (int rc = run_scaled_fast(f, 0, stream)) return rc;
      ScaledGemmArgs e;
      if (int rc = expand_fnuz_to_bf16(p, &e, stream, (const int*)flag)) return rc;
      e.gate = (const int*)flag;
      e.gate_run_if = 1;
      return launch_scaled_gemm_mfma_bf16(e, stream);
    }
  }
  if (variant >= 2 && variant <= 7) {
    set_error("scaled_gemm: MFMA variant %d forced but the layout contract is not met "
              "(need K-contiguous A and B^T, K %% 128 == 0, 16-byte aligned rows)", variant);
    return CONCH_ERR_UNSUPPORTED;
  }
  if (variant == 0) {
    // legal but non-native layout (row-major [K][N] weights, odd K, misaligned views): repack, then MFMA
    ScaledGemmArgs q;
    void* scratch = nullptr;
    if (repack_for_mfma(p, &q, &scratch, stream) == CONCH_OK && scaled_gemm_mfma_supported(q))
      return run_scaled_fast(q, 0, stream);
    // MI300-era e4m3fnuz operands: exact expansion to bf16, then the tiled kernel on bf16 MFMA
    if (p.in_dtype == CONCH_DT_FP8_E4M3FNUZ && expand_fnuz_to_bf16(p, &q, stream) == CONCH_OK)
      return launch_scaled_gemm_mfma_bf16(q, stream);
  }
  return launch_scaled_gemm_generic(p, stream);
}

// scaled_gemm on 16-bit activations with a static per-tensor activation scale: c = scaled_gemm(static_quant(x, scale_x), b, scale_x,
// scale_b) in one launch where the skinny-M split-K kernel runs the shape (it quantises A on its way into LDS, bit-identically
// to quant.hip), else the unfused pair through library scratch.  p.a = x, p.a_src_dtype = its dtype, p.a_stride_m in elements.

int run_static_quant_scaled(const ScaledGemmArgs& p, hipStream_t stream) {
  CONCH_CHECK_ARG(p.a_src_dtype == CONCH_DT_FP16 || p.a_src_dtype == CONCH_DT_BF16,
                  "static_quant_scaled_gemm: activations must be FP16 or BF16 (got dtype %d)", p.a_src_dtype);
  CONCH_CHECK_ARG(p.in_dtype == CONCH_DT_INT8 || p.in_dtype == CONCH_DT_FP8_E4M3FN,
                  "static_quant_scaled_gemm: quantised dtype %d (want INT8 or FP8_E4M3FN)", p.in_dtype);
  CONCH_CHECK_ARG(p.scale_a && p.scale_a_numel == 1, "static_quant_scaled_gemm: one per-tensor activation scale");
  ScaledGemmArgs q = p;  // the problem as the GEMM kernels see it: A = m x k quantised bytes
  q.a_src_dtype = 0;
  if (int rc = check_scaled(q)) return rc;
  if (p.m == 0 || p.n == 0) return CONCH_OK;
  CONCH_CHECK_ARG(p.scale_b, "static_quant_scaled_gemm: NULL scale_b");
  const int variant = tuning(CONCH_TUNE_GEMM_VARIANT);
  ScaledGemmArgs probe = q;      // layout contract of the fused path: 16-byte aligned activation rows, K-contiguous
  probe.a_stride_m = p.a_stride_m * 2;
  // Every (64-column block, K slice) workgroup quantises its own A slice again, N / 64 times over: worth it while that work
  // (measured ~11 elements per ns and CU, profiles/r02/fused_quant_times.txt) stays under the ~8 us a separate quantiser launch
  // costs at these sizes.
  const int rows = p.m <= 32 ? 32 : p.m <= 64 ? 64 : 128;
  const int64_t wgs = ((p.n + 63) / 64) * ((p.k + 1023) / 1024) * ((p.m + rows - 1) / rows);
  const double requant_us = (double)((wgs + device_cu_count() - 1) / device_cu_count()) * (double)rows * 1024.0 / 11.0e3;
  const bool fused_ok = p.a_stride_k == 1 && p.n % 4 == 0 && scaled_gemm_skinny_supported(probe) &&
                        (variant == 4 || (variant == 0 && choose_scaled_kernel(probe) == kKernelSkinny && requant_us < 8.0));
  if (fused_ok) return launch_scaled_gemm_skinny(p, stream);
  // the unfused pair: quantise into scratch (row stride K), then the ordinary op
  void* aq = nullptr;
  if (int rc = get_scratch(stream, kScratchWide, (size_t)p.m * (size_t)p.k, &aq)) return rc;
  CONCH_CHECK_ARG(p.a_stride_k == 1, "static_quant_scaled_gemm: activations must be K-contiguous");
  const int rc = p.in_dtype == CONCH_DT_INT8
                     ? conch_static_scaled_int8_quant((int8_t*)aq, p.a, p.scale_a, p.m, p.k, p.a_stride_m, p.k, p.a_src_dtype, stream)
                     : conch_static_scaled_fp8_quant((uint8_t*)aq, p.a, p.scale_a, p.m, p.k, p.a_stride_m, p.k, p.a_src_dtype, CONCH_DT_FP8_E4M3FN, stream);
  if (rc) return rc;
  q.a = aq;
  q.a_stride_m = p.k;
  return run_scaled(q, stream);
}


// scaled_gemm_silu_and_mul: `p` describes the OUTPUT (n columns); B / scale_b / bias have 2n columns [gate | up].
int run_scaled_silu(const ScaledGemmArgs& p, hipStream_t stream) {
  ScaledGemmArgs wide = p;  // the plain GEMM the fused op contains: checks and the unfused fallback run on it
  wide.fuse_silu = 0;
  wide.n = 2 * p.n;
  if (int rc = check_scaled(wide)) return rc;
  CONCH_CHECK_ARG(p.c_stride_n == 1, "scaled_gemm_silu_and_mul: C must have unit column stride (got %lld)", (long long)p.c_stride_n);
  if (p.m == 0 || p.n == 0) return CONCH_OK;
  const int variant = tuning(CONCH_TUNE_GEMM_VARIANT);
  // decode batches: the split-K kernel on the wide problem with the silu fused into its reduce kernel (the fused tile of
  // gemm_mfma.hip is a 256-row tile: 35 us at K = 4096 whatever M is)
  bool unfused_pair = variant == 1 || variant == 2;
  if (variant == 0 && scaled_gemm_mfma_supported(wide)) {
    const ScaledKernel pick = choose_scaled_kernel(wide);
    if (pick == kKernelSkinny && scaled_gemm_skinny_fused_supported(wide)) {
      ScaledGemmArgs q = wide;  // n = 2d for the partial sums; c / c_stride_m describe the d-column result
      q.fuse_silu = p.fuse_silu;
      return launch_scaled_gemm_skinny(q, stream);
    }
    // few 256-row tiles: the 128x128-tile kernel on the wide problem plus the elementwise pass beats the fused 256-row tile
    if (pick != kKernelTiled) unfused_pair = true;
  }
  if (!unfused_pair && scaled_gemm_mfma_supported(p)) return launch_scaled_gemm_mfma(p, 5, stream);
  // any other layout / dtype / K: the plain GEMM into stream-ordered scratch, then the elementwise tail
  void* tmp = nullptr;
  if (int rc = get_scratch(stream, kScratchWide, (size_t)p.m * (size_t)wide.n * 2, &tmp)) return rc;
  wide.c = tmp;
  wide.c_stride_m = wide.n;
  wide.c_stride_n = 1;
  if (int rc = run_scaled(wide, stream)) return rc;
  return launch_silu_and_mul(p.c, tmp, p.m, p.n, wide.n, p.c_stride_m, p.out_dtype, p.fuse_silu, stream);
}

int check_mixed(const MixedGemmArgs& p) {
  CONCH_CHECK_ARG(p.m >= 0 && p.n >= 0 && p.k >= 0, "mixed_precision_gemm: negative shape");
  const bool f32 = p.x_dtype == CONCH_DT_FP32;  // fp32 activations, scales and result: the one-thread-per-output kernel only
  if (p.x_dtype != CONCH_DT_FP16 && p.x_dtype != CONCH_DT_BF16 && !f32) {
    set_error("mixed_precision_gemm: unsupported activation dtype %d (want FP16, BF16 or FP32)", p.x_dtype);
    return CONCH_ERR_UNSUPPORTED;
  }
  if (f32 ? p.out_dtype != CONCH_DT_FP32 : !is_out16(p.out_dtype)) {
    set_error("mixed_precision_gemm: unsupported output dtype %d for activation dtype %d (FP16 / BF16 activations give FP16 / BF16, "
              "FP32 activations give FP32)", p.out_dtype, p.x_dtype);
    return CONCH_ERR_UNSUPPORTED;
  }
  if (f32) CONCH_CHECK_ARG(p.group_size % (32 / p.bits) == 0, "mixed_precision_gemm (fp32): group_size %d must hold whole packed words", p.group_size);
  CONCH_CHECK_ARG(p.bits == 1 || p.bits == 2 || p.bits == 4 || p.bits == 8 || p.bits == 16 || p.bits == 32,
                  "mixed_precision_gemm: weight_bits %d does not divide 32", p.bits);
  CONCH_CHECK_ARG(p.group_size > 0, "mixed_precision_gemm: group_size %d", p.group_size);
  CONCH_CHECK_ARG(p.zp_mode >= CONCH_ZP_NONE && p.zp_mode <= CONCH_ZP_TENSOR,
                  "mixed_precision_gemm: bad zp_mode %d", p.zp_mode);
  if (p.m == 0 || p.n == 0) return CONCH_OK;
  CONCH_CHECK_ARG(p.c, "mixed_precision_gemm: NULL output");
  CONCH_CHECK_ARG(p.k == 0 || (p.x && p.w_q && p.w_s), "mixed_precision_gemm: NULL operand");
  CONCH_CHECK_ARG(p.zp_mode == CONCH_ZP_NONE || p.w_zp, "mixed_precision_gemm: zp_mode %d but w_zp NULL",
                  p.zp_mode);
  CONCH_CHECK_ARG(p.k % (32 / p.bits) == 0, "mixed_precision_gemm: K=%lld not a multiple of %d",
                  (long long)p.k, 32 / p.bits);
  return CONCH_OK;
}

// 64 < M <= 256: two to four row blocks of the decode kernel (each streams the weights; the later ones mostly from L2) against ONE
// row of tiles of the LDS-tiled kernel, which for so few rows is bound by what a CU can pull per K step, not by N.  Decode: fitted on
// weights streamed from HBM (profiles/r02/dispatch_cold_sweep_after.txt; refitted in round 3 after the decode kernel's prologue
// was rewritten -- 0.475e-6 -> 0.41e-6), us: 5 + 0.41e-6 N K per row block.  Tiles: round 4 lets the one row of tiles split K
// (gemm_mixed.hip, pick_split) and takes that model's time -- before, the comparison was against an unsplit row of tiles
// (16.5 us per 1024 of K) and the decode kernel kept 192-256 rows at up to 3.8x the split tiles' time
// (profiles/r04/mixed_splitk_sweep.txt: 256 x 28672 x 8192 458 -> 122 us, 256 x 8192 x 8192 110 -> 46, 128 x 11008 x 4096 44 -> 31).
// `tiles_may_split` = false for the fused gate/up form, whose 256-row tile does not split.
double mixed_decode_estimate_us(const MixedGemmArgs& p) {
  const double blocks = (double)((p.m + 63) / 64);
  const fit::MixedDecode& fd = fit::kMixedDecode;
  return fd.fixed + fd.per_nk_block * (double)p.n * (double)p.k * blocks * (p.bits == 8 ? fd.int8_factor : 1.0);
}

bool mixed_decode_beats_tiles(const MixedGemmArgs& p, bool tiles_may_split = true) {
  const fit::MixedDecode& fd = fit::kMixedDecode;
  const double decode_us = mixed_decode_estimate_us(p);
  if (tiles_may_split && mixed_gemm_mfma_supported(p) && tuning(CONCH_TUNE_MIXED_SPLITK) == 0) return decode_us < mixed_tiles_estimate_us(p);
  const int64_t tiles = (p.n + 127) / 128;  // the narrowest tile: the most workgroups a single row of tiles can have
  const double tile_us = fd.unsplit_tile_us_per_1024k * (double)p.k / 1024.0 * (double)((tiles + 255) / 256);
  return decode_us < tile_us;
}

// Which kernel mixed_precision_gemm runs for `p` under the current tuning keys -- no launch, no device access beyond the CU count.
// A forced kernel whose contract `p` does not meet: kMixedPickError (+ the error text).  run_mixed launches exactly this pick;
// conch_debug_mixed_plan reports it (tests/test_host_helpers.py pins the picks of the benchmark shapes without a GPU).
enum MixedPick { kMixedPickError = -1, kMixedPickGeneric = 0, kMixedPickDecode = 1, kMixedPickTiles = 2, kMixedPickStrip = 3, kMixedPickAsm = 4 };

MixedPick pick_mixed(const MixedGemmArgs& p) {
  // variant 1 forces the generic kernel, 4 the decode-batch kernel (gemm_mixed_skinny.hip: M <= 256), any other non-zero
  // value the LDS-tiled MFMA kernel (gemm_mixed.hip); auto = decode-batch kernel when its contract is met, else tiled
  const int variant = tuning(CONCH_TUNE_GEMM_VARIANT);
  const bool fast_ok = mixed_gemm_mfma_supported(p);
  if (variant == 4 && !mixed_gemm_skinny_supported(p)) {
    set_error("mixed_precision_gemm: skinny variant forced but its contract is not met (M <= 256, N %% 4 == 0)");
    return kMixedPickError;
  }
  if (p.x_dtype == CONCH_DT_FP32) {
    if (variant >= 2) {
      set_error("mixed_precision_gemm: MFMA variant %d forced but fp32 activations run on the generic kernel only", variant);
      return kMixedPickError;
    }
    return kMixedPickGeneric;
  }
  // auto: the decode-batch kernel for every shape it takes (M <= 64; up to four row blocks = 256 rows where mixed_decode_beats_tiles says so).  Round 1 stopped at 32 rows for N x K >= 9e7, where the two
  // kernels tie on a cache-resident weight (64x4096x28672: 58 us each); on weights streamed from HBM the decode kernel is 6-13 %
  // ahead there too (48-64 x 4096/8192 x 28672: profiles/r02/dispatch_cold_sweep_before.txt)
  bool decode = variant == 4 || variant == 0;
  if (variant == 0 && p.m > 64) decode = mixed_decode_beats_tiles(p);
  // batched-decode sizes (33..256 rows, round 5): the column-strip kernel on 64- / 128- / 256-row tiles with K split over the chip
  // dequantises every weight once whatever M is, where the decode kernel does it per 64-row block and the LDS-tiled kernel runs the
  // MFMAs of 256 rows -- 1.3-1.9x at 48..128 rows on wide or deep problems (64 x 8192 x 28672: 97.7 -> 50.8 us; 128 x 4096 x 11008:
  // 43.0 -> 25.9; profiles/r05/mixed_mid_sweep.txt), nothing at N = K = 4096, which the model leaves where it was
  if (variant == 0 && tuning(CONCH_TUNE_MIXED_KERNEL) == 0 && tuning(CONCH_TUNE_MIXED_SPLITK) == 0 && tuning(CONCH_TUNE_MIXED_TILE_NT) == 0 &&
      p.m >= fit::kMixedStripSplit.min_m && p.m <= fit::kMixedStripSplit.max_m && fast_ok && mixed_gemm_strip_supported(p)) {
    const bool decode_ok = decode && mixed_gemm_skinny_supported(p);
    const double other_us = decode_ok ? mixed_decode_estimate_us(p) : mixed_tiles_estimate_us(p);
    const double margin = decode_ok ? fit::kMixedStripSplit.margin : fit::kMixedStripSplit.tall_vs_tiles;
    if (mixed_strip_estimate_us(p) * margin < other_us) return kMixedPickStrip;
  }
  if (decode && mixed_gemm_skinny_supported(p)) return kMixedPickDecode;
  if (variant == 1 || !fast_ok) {
    if (variant >= 2 && !fast_ok) {
      set_error("mixed_precision_gemm: MFMA variant %d forced but the layout contract is not met", variant);
      return kMixedPickError;
    }
    return kMixedPickGeneric;
  }
  // CONCH_TUNE_MIXED_KERNEL: 2 = the column-strip kernel (gemm_mixed_strip.hip) wherever its contract holds, 1 = never,
  // 0 = auto (see mixed_strip_beats_tiles)
  const int kernel = tuning(CONCH_TUNE_MIXED_KERNEL);
  if (kernel == 3) {  // the one-wave-per-SIMD assembly kernel, forced
    if (!mixed_gemm_asm1w_supported(p)) {
      set_error("mixed_precision_gemm: kernel 3 (one-wave-per-SIMD assembly kernel) forced but its contract is not met (fp16, 4-bit weights without "
                "zero points, group 128, K %% 128 == 0, N %% 16 == 0)");
      return kMixedPickError;
    }
    return kMixedPickAsm;
  }
  if (kernel != 1 && mixed_gemm_strip_supported(p) && (kernel == 2 || mixed_strip_beats_tiles(p))) return kMixedPickStrip;
  // a few rows of tiles that leave CUs idle (round 5): the strip kernel on 128- / 64-row tiles or in K slices, where its model beats
  // the LDS-tiled kernel's (mixed_strip_short_or_split_us is finite only when strip_plan left the unsplit 256-row tile)
  if (variant == 0 && kernel == 0 && tuning(CONCH_TUNE_MIXED_SPLITK) == 0 && tuning(CONCH_TUNE_MIXED_TILE_NT) == 0 && p.m > fit::kMixedStripSplit.max_m &&
      p.m <= fit::kMixedStripSplit.max_m_search && mixed_gemm_strip_supported(p) &&
      mixed_strip_short_or_split_us(p) * fit::kMixedStripSplit.tall_vs_tiles < mixed_tiles_estimate_us(p))
    return kMixedPickStrip;
  return kMixedPickTiles;
}

int run_mixed(const MixedGemmArgs& p, hipStream_t stream) {
  if (int rc = check_mixed(p)) return rc;
  if (p.m == 0 || p.n == 0) return CONCH_OK;
  switch (pick_mixed(p)) {
    case kMixedPickGeneric: return launch_mixed_gemm_generic(p, stream);
    case kMixedPickDecode: return launch_mixed_gemm_skinny(p, stream);
    case kMixedPickStrip: return launch_mixed_gemm_strip(p, stream);
    case kMixedPickAsm: return launch_mixed_gemm_asm1w(p, stream);
    case kMixedPickTiles: return launch_mixed_gemm_mfma(p, stream);
    default: return CONCH_ERR_UNSUPPORTED;
  }
}

int run_mixed_modes(const MixedGemmArgs& p, const void* scales, const void* zeros, int64_t zeros_stride_g, int group_mode, int zeros_kind,
                    int channel_mode, const void* channel_scales, hipStream_t stream) {
  CONCH_CHECK_ARG(group_mode >= CONCH_GROUP_NONE && group_mode <= CONCH_GROUP_ASYMMETRIC, "mixed_precision_gemm: bad group mode %d", group_mode);
  CONCH_CHECK_ARG(zeros_kind >= CONCH_ZEROS_NONE && zeros_kind <= CONCH_ZEROS_TENSOR_FLOAT, "mixed_precision_gemm: bad zeros kind %d", zeros_kind);
  if (channel_mode != CONCH_CHANNEL_NONE && channel_mode != CONCH_CHANNEL_WEIGHT_ONLY) {
    set_error("mixed_precision_gemm: channel-scale mode %d needs activation scales, which the mixed-precision launcher never passes", channel_mode);
    return CONCH_ERR_UNSUPPORTED;
  }
  MixedGemmArgs chk = p;
  chk.w_s = p.w_s ? p.w_s : (const void*)p.w_q;  // NONE / SHIFT carry no group scales: any non-NULL pointer passes the shared checks
  chk.zp_mode = CONCH_ZP_NONE;
  if (int rc = check_mixed(chk)) return rc;
  if (p.m == 0 || p.n == 0) return CONCH_OK;
  const bool needs_zeros = group_mode == CONCH_GROUP_SHIFT || group_mode == CONCH_GROUP_SYMMETRIC_WITH_SHIFT || group_mode == CONCH_GROUP_ASYMMETRIC;
  const bool needs_scales = group_mode >= CONCH_GROUP_SYMMETRIC_NO_SHIFT;
  CONCH_CHECK_ARG(!needs_zeros || (zeros && zeros_kind != CONCH_ZEROS_NONE), "mixed_precision_gemm: group mode %d needs zero points", group_mode);
  CONCH_CHECK_ARG(!needs_scales || scales, "mixed_precision_gemm: group mode %d needs group scales", group_mode);
  CONCH_CHECK_ARG(channel_mode == CONCH_CHANNEL_NONE || channel_scales, "mixed_precision_gemm: WEIGHT_ONLY needs channel scales");
  CONCH_CHECK_ARG(p.bits <= 16, "mixed_precision_gemm: %d-bit weights do not fit a 16-bit activation dtype", p.bits);
  // the two modes conch.ops produces, on the widths its fused kernels take: the fused path (bit-identical arithmetic)
  const bool int_zeros = zeros_kind == CONCH_ZEROS_SCALAR_INT32 || zeros_kind == CONCH_ZEROS_TENSOR_INT32;
  if (channel_mode == CONCH_CHANNEL_NONE && (p.bits == 4 || p.bits == 8) &&
      (group_mode == CONCH_GROUP_SYMMETRIC_NO_SHIFT || (group_mode == CONCH_GROUP_SYMMETRIC_WITH_SHIFT && int_zeros))) {
    MixedGemmArgs q = p;
    q.w_s = scales;
    q.w_zp = group_mode == CONCH_GROUP_SYMMETRIC_NO_SHIFT ? nullptr : (const int32_t*)zeros;
    q.wzp_stride_g = zeros_stride_g;
    q.zp_mode = group_mode == CONCH_GROUP_SYMMETRIC_NO_SHIFT ? CONCH_ZP_NONE
                : zeros_kind == CONCH_ZEROS_SCALAR_INT32    ? CONCH_ZP_SCALAR
                                                            : CONCH_ZP_TENSOR;
    return run_mixed(q, stream);
  }
  return launch_mixed_gemm_modes(p, scales, zeros, zeros_stride_g, group_mode, zeros_kind,
                                 channel_mode == CONCH_CHANNEL_WEIGHT_ONLY ? channel_scales : nullptr, stream);
}

// mixed_precision_gemm_silu_and_mul: `p` describes the OUTPUT (n columns); Wq / w_s / w_zp have 2n columns [gate | up].
int run_mixed_silu(const MixedGemmArgs& p, hipStream_t stream) {
  MixedGemmArgs wide = p;
  wide.fuse_silu = 0;
  wide.n = 2 * p.n;
  if (int rc = check_mixed(wide)) return rc;
  if (p.m == 0 || p.n == 0) return CONCH_OK;
  const int variant = tuning(CONCH_TUNE_GEMM_VARIANT);
  // decode batches: the unfused pair on the decode-batch GEMM beats the fused 256-row tile
  // decode batches: the decode-batch kernel on the wide problem with the silu fused into its reduce kernel
  MixedGemmArgs probe = wide;  // the contract check wants a C row that holds n columns; the fused reduce writes only d
  probe.c_stride_m = wide.n;
  bool decode = variant == 0 && mixed_gemm_skinny_supported(probe) && wide.n % 8 == 0 &&
                (p.m <= 64 || mixed_decode_beats_tiles(wide, /*tiles_may_split=*/false));  // the plain op's rule, against the fused tile
  // 65-256 rows (round 4, profiles/r04/fused_ops_sweep.txt): the plain op now splits K over one row of tiles, and the pair
  // "split tiles into scratch + the elementwise tail" beats both fused forms there (128 x 4096 x 22016: 88 us fused, 62 as two ops)
  bool pair_wins = false;
  // (round 5: from 33 rows on, and the plain op's time is the better of its LDS-tiled and K-split strip forms -- run_mixed's own rule)
  // ... and above 256 rows while the fused tiles (256 x 128 of the result, never split) leave CUs idle: 512 x 4096 x 2*4096 ran 88.7 us
  // fused on 64 tiles against 57.1 us as the plain op in K slices + the elementwise tail (profiles/r05/mixed_silu_mid.txt)
  const int64_t fused_tiles = ((p.m + 255) / 256) * ((p.n + 127) / 128);
  if (variant == 0 && p.m >= fit::kMixedStripSplit.min_m && (p.m <= 256 || fused_tiles < device_cu_count()) && mixed_gemm_mfma_supported(wide) &&
      tuning(CONCH_TUNE_MIXED_SPLITK) == 0) {
    const bool strip_ok = tuning(CONCH_TUNE_MIXED_KERNEL) == 0 && tuning(CONCH_TUNE_MIXED_TILE_NT) == 0 && mixed_gemm_strip_supported(wide);
    const double pair_us = std::min(mixed_tiles_estimate_us(wide), strip_ok ? mixed_strip_estimate_us(wide) * fit::kMixedStripSplit.margin : 1e30) + 4.0;
    const double blocks = (double)((p.m + 63) / 64);
    const fit::MixedDecode& fd = fit::kMixedDecode;
    const double decode_us = decode ? fd.fixed + fd.per_nk_block * (double)wide.n * (double)p.k * blocks * (p.bits == 8 ? fd.int8_factor : 1.0) : 1e30;
    // the fused tile is 256 gate/up columns wide (128 of the result) and never splits: 1.35 us per 64 of K at one tile per CU
    // (86 us at K = 4096 whatever M <= 256 is; profiles/r04/fused_ops_sweep.txt)
    const double fused_rounds = (double)((fused_tiles + device_cu_count() - 1) / device_cu_count());
    const double fused_us = mixed_gemm_silu_fused_supported(p) ? 1.35 * (double)(p.k / 64) * fused_rounds : 1e30;
    pair_wins = pair_us < std::min(decode_us, fused_us);
    if (pair_wins) decode = false;
  }
  if (decode) {
    MixedGemmArgs q = wide;  // n = 2d for the partial sums; c / c_stride_m describe the d-column result
    q.fuse_silu = p.fuse_silu;
    return launch_mixed_gemm_skinny(q, stream);
  }
  if (!pair_wins && variant != 1 && variant != 2 && mixed_gemm_silu_fused_supported(p)) return launch_mixed_gemm_mfma(p, stream);
  void* tmp = nullptr;
  if (int rc = get_scratch(stream, kScratchWide, (size_t)p.m * (size_t)wide.n * 2, &tmp)) return rc;
  wide.c = tmp;
  wide.c_stride_m = wide.n;
  if (int rc = run_mixed(wide, stream)) return rc;
  return launch_silu_and_mul(p.c, tmp, p.m, p.n, wide.n, p.c_stride_m, p.out_dtype, p.fuse_silu, stream);
}

// Weight rotation of the timing helpers (conch_set_timing_rotation): launch i reads its weights at base + (i % count) * stride, so
// that with count x (weight bytes) beyond the 256 MiB Infinity Cache every launch streams its weights from HBM, as a decode step
// does -- back-to-back launches on ONE buffer time the cache-resident case only.
std::atomic<int> g_rotate_count{1};
std::atomic<int64_t> g_rotate_stride{0};

inline const void* rotated(const void* base, int i) {
  const int count = g_rotate_count.load();
  return count > 1 ? (const void*)((const char*)base + (int64_t)(i % count) * g_rotate_stride.load()) : base;
}

struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
};

// `launch(i)`: enqueue launch number i
template <class F>
int time_loop(F&& launch, hipStream_t stream, int iters, float* avg_ms) {
  CONCH_CHECK_ARG(iters > 0 && avg_ms, "timing: iters=%d avg_ms=%p", iters, (void*)avg_ms);
  EventPair ev;  // destroyed on every return path
  CONCH_HIP(hipEventCreate(&ev.e0));
  CONCH_HIP(hipEventCreate(&ev.e1));
  int rc = CONCH_OK;
  CONCH_HIP(hipEventRecord(ev.e0, stream));
  for (int i = 0; i < iters && rc == CONCH_OK; ++i) rc = launch(i);
  CONCH_HIP(hipEventRecord(ev.e1, stream));
  CONCH_HIP(hipEventSynchronize(ev.e1));
  float ms = 0.f;
  CONCH_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  *avg_ms = ms / (float)iters;
  return rc;
}

}  // namespace
}  // namespace conch

using namespace conch;

extern "C" int conch_abi_version(void) { return CONCH_AMD_ABI_VERSION; }
extern "C" const char* conch_last_error(void) { return g_error; }

// Diagnostic (not in include/conch_amd.h; tests/test_host_helpers.py, no GPU needed): the kernel scaled_gemm would run for K-contiguous,
// aligned operands of this shape (per-row / per-column scales, bf16 result, no bias) under the current GEMM variant.
// out[0] = 0 256 x 256 tiles / 1 128 x 128 tiles / 2 split-K skinny / 3 the assembly kernel (-1: forced outside its contract),
// out[1] = the assembly kernel's tile width in columns (0 otherwise).  Nothing is launched, no pointer is dereferenced.
extern "C" int conch_debug_scaled_plan(int64_t m, int64_t n, int64_t k, int in_dtype, int* out) {
  CONCH_CHECK_ARG(out && m > 0 && n > 0 && k > 0, "conch_debug_scaled_plan: bad arguments");
  ScaledGemmArgs p{};
  p.c = (void*)(uintptr_t)0x10000000;
  p.a = (const void*)(uintptr_t)0x20000000;
  p.b = (const void*)(uintptr_t)0x30000000;
  p.scale_a = (const float*)(uintptr_t)0x40000000;
  p.scale_b = (const float*)(uintptr_t)0x50000000;
  p.m = m, p.n = n, p.k = k;
  p.a_stride_m = k, p.a_stride_k = 1, p.b_stride_n = k, p.b_stride_k = 1, p.c_stride_m = n, p.c_stride_n = 1;
  p.scale_a_numel = m, p.scale_b_numel = n;
  p.in_dtype = in_dtype, p.out_dtype = CONCH_DT_BF16;
  CONCH_CHECK_ARG(scaled_gemm_mfma_supported(p), "conch_debug_scaled_plan: the shape is outside the MFMA kernels' contract");
  out[0] = (int)pick_scaled_fast(p, tuning(CONCH_TUNE_GEMM_VARIANT));
  out[1] = out[0] == kScaledPickAsm ? scaled_asm1w_tile_columns(p) : 0;
  return CONCH_OK;
}

// Diagnostic (not in include/conch_amd.h; tests/test_host_helpers.py, no GPU needed): the kernel mixed_precision_gemm would run for a
// contiguous, aligned problem of this shape under the current tuning keys, and the strip kernel's plan when that is the pick.
// out[0] = 0 generic / 1 decode-batch / 2 LDS-tiled / 3 column-strip / 4 assembly (-1: a forced kernel's contract is not met),
// out[1..3] = tile rows, tile columns, K slices of the strip plan (0 otherwise).  Nothing is launched, no pointer is dereferenced.
extern "C" int conch_debug_mixed_plan(int64_t m, int64_t n, int64_t k, int bits, int x_dtype, int per_group_zero_points, int* out) {
  CONCH_CHECK_ARG(out && m > 0 && n > 0 && k > 0 && (bits == 4 || bits == 8), "conch_debug_mixed_plan: bad arguments");
  MixedGemmArgs p{};
  p.c = (void*)(uintptr_t)0x10000000;
  p.x = (const void*)(uintptr_t)0x20000000;
  p.w_q = (const int32_t*)(uintptr_t)0x30000000;
  p.w_s = (const void*)(uintptr_t)0x40000000;
  p.w_zp = per_group_zero_points ? (const int32_t*)(uintptr_t)0x50000000 : nullptr;
  p.m = m, p.n = n, p.k = k;
  p.x_stride_m = k, p.wq_stride_k = n, p.ws_stride_g = n, p.wzp_stride_g = n, p.c_stride_m = n;
  p.bits = bits, p.weight_bias = per_group_zero_points ? 0 : 1 << (bits - 1), p.group_size = 128;
  p.zp_mode = per_group_zero_points ? CONCH_ZP_TENSOR : CONCH_ZP_NONE;
  p.x_dtype = x_dtype, p.out_dtype = x_dtype;
  if (int rc = check_mixed(p)) return rc;
  out[0] = (int)pick_mixed(p);
  out[1] = out[2] = out[3] = 0;
  if (out[0] == kMixedPickStrip) mixed_strip_plan_query(p, &out[1], &out[2], &out[3]);
  return CONCH_OK;
}

extern "C" int conch_debug_fnuz_in_registers(int on) {
  g_fnuz_in_registers.store(on < 0 ? 0 : on > 2 ? 2 : on);
  return CONCH_OK;
}

extern "C" int conch_set_tuning(int key, int value) {
  CONCH_CHECK_ARG(key >= 0 && key < CONCH_TUNE__COUNT, "conch_set_tuning: unknown key %d", key);
  g_tuning[key].store(value);
  return CONCH_OK;
}
extern "C" int conch_get_tuning(int key) { return tuning(key); }

extern "C" int conch_set_timing_rotation(int count, int64_t stride_bytes) {
  CONCH_CHECK_ARG(count >= 1 && stride_bytes >= 0, "conch_set_timing_rotation: count=%d stride=%lld", count, (long long)stride_bytes);
  g_rotate_count.store(count);
  g_rotate_stride.store(stride_bytes);
  return CONCH_OK;
}

// Worst case over every kernel the dispatcher may pick for this shape (any layout, either fp8 flavour, fused FFN form
// included): split-K slabs (M <= 256), K-contiguous / bf16-expanded operand copies, the [M][N] intermediate of an unfused pair.
extern "C" int64_t conch_scaled_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m <= 0 || n <= 0 || k < 0) return 0;
  const int64_t kpad = (k + 127) / 128 * 128;
  int64_t need = 2 * (m + n) * kpad + 512;                      // repack.hip: both operands, 2 bytes per element at most
  if (m <= 256) need = std::max(need, (k / 1024 + 1) * m * n * 4);  // gemm_skinny.hip slabs
  need = std::max(need, mid_split_slab_bytes(m, n));            // gemm_mid.hip split-K slabs (few tiles)
  need = std::max(need, m * n * 2);                             // unfused silu pair: `n` = the wide width there
  return need + ((int64_t)1 << 20);
}

extern "C" int64_t conch_mixed_precision_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  if (m <= 0 || n <= 0 || k < 0) return 0;
  int64_t need = m * n * 2;
  if (m <= 256) need = std::max(need, (k / 1024 + 1) * m * n * 4);  // gemm_mixed_skinny.hip slabs
  need = std::max(need, (int64_t)8 * m * n * 4);                     // gemm_mixed.hip / gemm_mixed_strip.hip split-K slabs (at most 8 slices, any M)
  return need + ((int64_t)1 << 20);
}

extern "C" int conch_reserve_scratch(void* stream, int64_t bytes) {
  CONCH_CHECK_ARG(bytes >= 0, "conch_reserve_scratch: negative size");
  void* ignored = nullptr;
  for (int slot = 0; slot < kScratchCounters; ++slot)
    if (int rc = get_scratch((hipStream_t)stream, slot, (size_t)bytes, &ignored)) return rc;
  if (int rc = get_scratch((hipStream_t)stream, kScratchFlags, kFlagsBytes, &ignored)) return rc;  // e4m3fnuz dispatch word + scaled scale_a
  preload_asm_modules();  // the assembly kernels' code objects for this device: never loaded inside a capture
  return get_scratch((hipStream_t)stream, kScratchCounters, (size_t)64 * 1024, &ignored, /*zero_on_alloc=*/true);
}

// The one-launch split-K forms (gemm_skinny.hip, gemm_mixed_skinny.hip, bnb.hip) count arrivals in the 64 KiB counter slot of
// (device, stream): zero at allocation, put back to zero by the last arriver of every tile.  A launch that did not run to
// completion (a fault, a reset) can leave them non-zero, and every later one-launch call on that stream would then reduce a
// tile too early or never.  This puts them back: a memset enqueued ON `stream`, behind whatever is already there.
extern "C" int conch_reset_scratch(void* stream) {
  int device = 0;
  CONCH_HIP(hipGetDevice(&device));
  void* ptr = nullptr;
  size_t size = 0;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    auto it = g_scratch.find(ScratchKey{device, (hipStream_t)stream, kScratchCounters});
    if (it != g_scratch.end()) {
      ptr = it->second.ptr;
      size = it->second.size;
    }
  }
  if (ptr) CONCH_HIP(hipMemsetAsync(ptr, 0, size, (hipStream_t)stream));
  return CONCH_OK;
}

extern "C" int conch_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int conch_scaled_gemm(void* c, const void* a, const void* b, const float* scale_a,
                                 const float* scale_b, const void* bias, int64_t m, int64_t n, int64_t k,
                                 int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                                 int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n,
                                 int64_t scale_a_numel, int64_t scale_b_numel, int in_dtype, int out_dtype,
                                 void* stream) {
  const ScaledGemmArgs p{c, a, b, scale_a, scale_b, bias, m, n, k, a_stride_m, a_stride_k, b_stride_k,
                         b_stride_n, c_stride_m, c_stride_n, scale_a_numel, scale_b_numel, in_dtype, out_dtype};
  return run_scaled(p, (hipStream_t)stream);
}

extern "C" int conch_scaled_gemm_multi(void* c, const void* a, const void* b, const float* scale_a, const float* scale_b, const void* bias,
                                       int64_t m, int64_t n, int64_t k, int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                                       int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n, int64_t scale_a_numel,
                                       int64_t scale_b_numel, int in_dtype, int out_dtype, void* const* c_more, int n_more, void* stream) {
  ScaledGemmArgs p{c, a, b, scale_a, scale_b, bias, m, n, k, a_stride_m, a_stride_k, b_stride_k,
                   b_stride_n, c_stride_m, c_stride_n, scale_a_numel, scale_b_numel, in_dtype, out_dtype};
  if (n_more == 0) return run_scaled(p, (hipStream_t)stream);
  CONCH_CHECK_ARG(n_more > 0 && n_more <= 7 && c_more, "scaled_gemm_multi: %d further destinations (0 .. 7)", n_more);
  for (int d = 0; d < n_more; ++d) {
    CONCH_CHECK_ARG(c_more[d], "scaled_gemm_multi: NULL destination %d", d + 1);
    p.c_more[d] = c_more[d];
  }
  p.n_more = n_more;
  if (int rc = check_scaled(p)) return rc;
  if (p.m == 0 || p.n == 0) return CONCH_OK;
  if (!p.scale_a || !p.scale_b) {
    const float* one = nullptr;
    if (int rc = unit_scale(&one)) return rc;
    if (!p.scale_a) { p.scale_a = one; p.scale_a_numel = 1; }
    if (!p.scale_b) { p.scale_b = one; p.scale_b_numel = 1; }
  }
  // the 256 x 256 tile kernel only (its row-major epilogue is where the further stores live): native layout, K >= 256 bytes
  if (!scaled_gemm_mfma_supported(p) || p.k < 256 || p.c_stride_n != 1) {
    set_error("scaled_gemm_multi: needs the tile kernel's layout (K-contiguous int8 / e4m3fn A and B^T, K %% 128 == 0, K >= 256, "
              "16-byte aligned rows, unit-stride C rows)");
    return CONCH_ERR_UNSUPPORTED;
  }
  return launch_scaled_gemm_mfma(p, 5, (hipStream_t)stream);
}

extern "C" int conch_static_quant_scaled_gemm(void* c, const void* x, const void* b, const float* scale_x, const float* scale_b,
                                              const void* bias, int64_t m, int64_t n, int64_t k, int64_t x_stride_m,
                                              int64_t x_stride_k, int64_t b_stride_k, int64_t b_stride_n, int64_t c_stride_m,
                                              int64_t c_stride_n, int64_t scale_b_numel, int x_dtype, int quant_dtype, int out_dtype,
                                              void* stream) {
  ScaledGemmArgs p{c, x, b, scale_x, scale_b, bias, m, n, k, x_stride_m, x_stride_k, b_stride_k, b_stride_n, c_stride_m, c_stride_n,
                   1, scale_b_numel, quant_dtype, out_dtype};
  p.a_src_dtype = x_dtype;
  return run_static_quant_scaled(p, (hipStream_t)stream);
}

extern "C" int conch_scaled_gemm_silu_and_mul(void* c, const void* a, const void* b, const float* scale_a,
                                              const float* scale_b, const void* bias, int64_t m, int64_t n_out, int64_t k,
                                              int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                                              int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n,
                                              int64_t scale_a_numel, int64_t scale_b_numel, int in_dtype, int out_dtype,
                                              void* stream) {
  ScaledGemmArgs p{c, a, b, scale_a, scale_b, bias, m, n_out, k, a_stride_m, a_stride_k, b_stride_k,
                   b_stride_n, c_stride_m, c_stride_n, scale_a_numel, scale_b_numel, in_dtype, out_dtype};
  p.fuse_silu = 1;
  return run_scaled_silu(p, (hipStream_t)stream);
}

extern "C" int conch_scaled_gemm_gelu_tanh_and_mul(void* c, const void* a, const void* b, const float* scale_a,
                                                   const float* scale_b, const void* bias, int64_t m, int64_t n_out, int64_t k,
                                                   int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                                                   int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n,
                                                   int64_t scale_a_numel, int64_t scale_b_numel, int in_dtype, int out_dtype,
                                                   void* stream) {
  ScaledGemmArgs p{c, a, b, scale_a, scale_b, bias, m, n_out, k, a_stride_m, a_stride_k, b_stride_k,
                   b_stride_n, c_stride_m, c_stride_n, scale_a_numel, scale_b_numel, in_dtype, out_dtype};
  p.fuse_silu = 2;  // gelu-tanh gate
  return run_scaled_silu(p, (hipStream_t)stream);
}

extern "C" int conch_time_scaled_gemm_silu_and_mul(void* c, const void* a, const void* b, const float* scale_a,
                                                   const float* scale_b, const void* bias, int64_t m, int64_t n_out,
                                                   int64_t k, int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                                                   int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n,
                                                   int64_t scale_a_numel, int64_t scale_b_numel, int in_dtype,
                                                   int out_dtype, void* stream, int iters, float* avg_ms) {
  ScaledGemmArgs p{c, a, b, scale_a, scale_b, bias, m, n_out, k, a_stride_m, a_stride_k, b_stride_k,
                   b_stride_n, c_stride_m, c_stride_n, scale_a_numel, scale_b_numel, in_dtype, out_dtype};
  p.fuse_silu = 1;
  hipStream_t s = (hipStream_t)stream;
  return time_loop([&](int) { return run_scaled_silu(p, s); }, s, iters, avg_ms);
}

extern "C" int conch_time_scaled_gemm(void* c, const void* a, const void* b, const float* scale_a,
                                      const float* scale_b, const void* bias, int64_t m, int64_t n,
                                      int64_t k, int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_k,
                                      int64_t b_stride_n, int64_t c_stride_m, int64_t c_stride_n,
                                      int64_t scale_a_numel, int64_t scale_b_numel, int in_dtype,
                                      int out_dtype, void* stream, int iters, float* avg_ms) {
  const ScaledGemmArgs p{c, a, b, scale_a, scale_b, bias, m, n, k, a_stride_m, a_stride_k, b_stride_k,
                         b_stride_n, c_stride_m, c_stride_n, scale_a_numel, scale_b_numel, in_dtype, out_dtype};
  hipStream_t s = (hipStream_t)stream;
  return time_loop(
      [&](int i) {
        ScaledGemmArgs q = p;
        q.b = rotated(p.b, i);
        return run_scaled(q, s);
      },
      s, iters, avg_ms);
}

extern "C" int conch_mixed_precision_gemm(void* c, const void* x, const int32_t* w_q_packed, const void* w_s,
                                          const int32_t* w_zp, int64_t m, int64_t n, int64_t k,
                                          int64_t x_stride_m, int64_t wq_stride_k, int64_t ws_stride_g,
                                          int64_t wzp_stride_g, int64_t c_stride_m, int weight_bits,
                                          int weight_bias, int group_size, int zp_mode, int x_dtype,
                                          int out_dtype, void* stream) {
  const MixedGemmArgs p{c, x, w_q_packed, w_s, w_zp, m, n, k, x_stride_m, wq_stride_k, ws_stride_g,
                        wzp_stride_g, c_stride_m, weight_bits, weight_bias, group_size, zp_mode, x_dtype,
                        out_dtype};
  return run_mixed(p, (hipStream_t)stream);
}

// ---- weight pre-pack ----
static int check_prepack(int64_t k, int64_t n, int bits, int nt, const char* what) {
  CONCH_CHECK_ARG(bits == 4 || bits == 8, "%s: %d-bit weights (the tile kernel takes 4 and 8)", what, bits);
  CONCH_CHECK_ARG(nt >= 2 && nt <= 4, "%s: tile width %d x 64 columns (want 2, 3 or 4)", what, nt);
  CONCH_CHECK_ARG(k >= 64 && k % 64 == 0 && n >= 1, "%s: K=%lld must be a multiple of 64, N=%lld positive", what, (long long)k, (long long)n);
  CONCH_CHECK_ARG(prepacked_weight_words(k, n, bits, nt) * 4 < ((int64_t)1 << 31), "%s: image of 2 GiB or more", what);
  return CONCH_OK;
}

extern "C" int64_t conch_prepacked_mixed_weights_bytes(int64_t k, int64_t n, int weight_bits, int tile_nt) {
  if (check_prepack(k, n, weight_bits, tile_nt, "prepacked_mixed_weights_bytes")) return -1;
  return prepacked_weight_words(k, n, weight_bits, tile_nt) * 4;
}

extern "C" int conch_mixed_gemm_tile_nt(int64_t m, int64_t n, int64_t k, int weight_bits, int zp_mode) {
  MixedGemmArgs p{};
  p.m = m;
  p.n = n;
  p.k = k;
  p.bits = weight_bits;
  p.zp_mode = zp_mode;
  return mixed_gemm_tile_nt(p);
}

extern "C" int conch_prepack_mixed_weights(int32_t* image, const int32_t* w_q_packed, int64_t k, int64_t n, int64_t wq_stride_k,
                                           int weight_bits, int tile_nt, void* stream) {
  if (int rc = check_prepack(k, n, weight_bits, tile_nt, "prepack_mixed_weights")) return rc;
  CONCH_CHECK_ARG(image && w_q_packed && wq_stride_k >= n, "prepack_mixed_weights: NULL pointer or row stride below N");
  return prepack_mixed_weights((uint32_t*)image, (uint32_t*)w_q_packed, k, n, wq_stride_k, weight_bits, tile_nt, false, (hipStream_t)stream);
}

extern "C" int conch_unpack_mixed_weights(int32_t* w_q_packed, const int32_t* image, int64_t k, int64_t n, int64_t wq_stride_k,
                                          int weight_bits, int tile_nt, void* stream) {
  if (int rc = check_prepack(k, n, weight_bits, tile_nt, "unpack_mixed_weights")) return rc;
  CONCH_CHECK_ARG(image && w_q_packed && wq_stride_k >= n, "unpack_mixed_weights: NULL pointer or row stride below N");
  return prepack_mixed_weights((uint32_t*)image, (uint32_t*)w_q_packed, k, n, wq_stride_k, weight_bits, tile_nt, true, (hipStream_t)stream);
}

static int run_mixed_prepacked(MixedGemmArgs p, int tile_nt, hipStream_t stream) {
  if (int rc = check_prepack(p.k, p.n > 0 ? p.n : 1, p.bits, tile_nt, "mixed_precision_gemm_prepacked")) return rc;
  p.wq_stride_k = p.n;  // unused by the kernel; keeps the shared checks meaningful
  if (int rc = check_mixed(p)) return rc;
  if (p.m == 0 || p.n == 0) return CONCH_OK;
  p.prepacked = tile_nt;
  if (p.zp_mode == CONCH_ZP_TENSOR && tile_nt == 4) {
    set_error("mixed_precision_gemm_prepacked: per-group zero points run on tiles of at most 192 columns (tile_nt <= 3)");
    return CONCH_ERR_UNSUPPORTED;
  }
  if (!mixed_gemm_mfma_supported(p) || p.x_dtype != p.out_dtype) {
    set_error("mixed_precision_gemm_prepacked: outside the tile kernel's contract (K %% 64 == 0, group_size %% 64 == 0, 16-byte "
              "aligned activation rows, output dtype = activation dtype)");
    return CONCH_ERR_UNSUPPORTED;
  }
  return launch_mixed_gemm_mfma(p, stream);
}

extern "C" int conch_mixed_precision_gemm_prepacked(void* c, const void* x, const int32_t* image, const void* w_s, const int32_t* w_zp,
                                                    int64_t m, int64_t n, int64_t k, int64_t x_stride_m, int64_t ws_stride_g,
                                                    int64_t wzp_stride_g, int64_t c_stride_m, int weight_bits, int weight_bias,
                                                    int group_size, int zp_mode, int tile_nt, int x_dtype, int out_dtype, void* stream) {
  const MixedGemmArgs p{c, x, image, w_s, w_zp, m, n, k, x_stride_m, n, ws_stride_g, wzp_stride_g, c_stride_m, weight_bits, weight_bias,
                        group_size, zp_mode, x_dtype, out_dtype};
  return run_mixed_prepacked(p, tile_nt, (hipStream_t)stream);
}

extern "C" int conch_time_mixed_precision_gemm_prepacked(void* c, const void* x, const int32_t* image, const void* w_s,
                                                         const int32_t* w_zp, int64_t m, int64_t n, int64_t k, int64_t x_stride_m,
                                                         int64_t ws_stride_g, int64_t wzp_stride_g, int64_t c_stride_m, int weight_bits,
                                                         int weight_bias, int group_size, int zp_mode, int tile_nt, int x_dtype,
                                                         int out_dtype, void* stream, int iters, float* avg_ms) {
  const MixedGemmArgs p{c, x, image, w_s, w_zp, m, n, k, x_stride_m, n, ws_stride_g, wzp_stride_g, c_stride_m, weight_bits, weight_bias,
                        group_size, zp_mode, x_dtype, out_dtype};
  hipStream_t s = (hipStream_t)stream;
  return time_loop(
      [&](int i) {
        MixedGemmArgs q = p;
        q.w_q = (const int32_t*)rotated(p.w_q, i);
        return run_mixed_prepacked(q, tile_nt, s);
      },
      s, iters, avg_ms);
}

extern "C" int conch_mixed_precision_gemm_modes(void* c, const void* x, const int32_t* w_q_packed, const void* scales, const void* zeros,
                                                const void* channel_scales, int64_t m, int64_t n, int64_t k, int64_t x_stride_m,
                                                int64_t wq_stride_k, int64_t scales_stride_g, int64_t zeros_stride_g,
                                                int64_t c_stride_m, int weight_bits, int weight_bias, int group_size, int group_mode,
                                                int zeros_kind, int channel_mode, int x_dtype, int out_dtype, void* stream) {
  const MixedGemmArgs p{c, x, w_q_packed, scales, nullptr, m, n, k, x_stride_m, wq_stride_k, scales_stride_g, zeros_stride_g,
                        c_stride_m, weight_bits, weight_bias, group_size, CONCH_ZP_NONE, x_dtype, out_dtype};
  return run_mixed_modes(p, scales, zeros, zeros_stride_g, group_mode, zeros_kind, channel_mode, channel_scales, (hipStream_t)stream);
}

extern "C" int conch_mixed_precision_gemm_silu_and_mul(void* c, const void* x, const int32_t* w_q_packed, const void* w_s,
                                                       const int32_t* w_zp, int64_t m, int64_t n_out, int64_t k,
                                                       int64_t x_stride_m, int64_t wq_stride_k, int64_t ws_stride_g,
                                                       int64_t wzp_stride_g, int64_t c_stride_m, int weight_bits,
                                                       int weight_bias, int group_size, int zp_mode, int x_dtype,
                                                       int out_dtype, void* stream) {
  MixedGemmArgs p{c, x, w_q_packed, w_s, w_zp, m, n_out, k, x_stride_m, wq_stride_k, ws_stride_g,
                  wzp_stride_g, c_stride_m, weight_bits, weight_bias, group_size, zp_mode, x_dtype, out_dtype};
  p.fuse_silu = 1;
  return run_mixed_silu(p, (hipStream_t)stream);
}

extern "C" int conch_mixed_precision_gemm_gelu_tanh_and_mul(void* c, const void* x, const int32_t* w_q_packed, const void* w_s,
                                                            const int32_t* w_zp, int64_t m, int64_t n_out, int64_t k,
                                                            int64_t x_stride_m, int64_t wq_stride_k, int64_t ws_stride_g,
                                                            int64_t wzp_stride_g, int64_t c_stride_m, int weight_bits,
                                                            int weight_bias, int group_size, int zp_mode, int x_dtype,
                                                            int out_dtype, void* stream) {
  MixedGemmArgs p{c, x, w_q_packed, w_s, w_zp, m, n_out, k, x_stride_m, wq_stride_k, ws_stride_g,
                  wzp_stride_g, c_stride_m, weight_bits, weight_bias, group_size, zp_mode, x_dtype, out_dtype};
  p.fuse_silu = 2;  // gelu-tanh gate
  return run_mixed_silu(p, (hipStream_t)stream);
}

extern "C" int conch_time_mixed_precision_gemm(void* c, const void* x, const int32_t* w_q_packed,
                                               const void* w_s, const int32_t* w_zp, int64_t m, int64_t n,
                                               int64_t k, int64_t x_stride_m, int64_t wq_stride_k,
                                               int64_t ws_stride_g, int64_t wzp_stride_g, int64_t c_stride_m,
                                               int weight_bits, int weight_bias, int group_size, int zp_mode,
                                               int x_dtype, int out_dtype, void* stream, int iters,
                                               float* avg_ms) {
  const MixedGemmArgs p{c, x, w_q_packed, w_s, w_zp, m, n, k, x_stride_m, wq_stride_k, ws_stride_g,
                        wzp_stride_g, c_stride_m, weight_bits, weight_bias, group_size, zp_mode, x_dtype,
                        out_dtype};
  hipStream_t s = (hipStream_t)stream;
  return time_loop(
      [&](int i) {
        MixedGemmArgs q = p;
        q.w_q = (const int32_t*)rotated(p.w_q, i);
        return run_mixed(q, s);
      },
      s, iters, avg_ms);
}
