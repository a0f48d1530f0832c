// Shared device/host helpers for the gfx950 quantized-matmul kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conch_amd.h"

namespace conch {

// ---------------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int tuning(int key);

#define CONCH_CHECK_ARG(cond, ...)           \
  do {                                       \
    if (!(cond)) {                           \
      ::conch::set_error(__VA_ARGS__);       \
      return CONCH_ERR_INVALID_ARGUMENT;     \
    }                                        \
  } while (0)

#define CONCH_HIP(expr)                                                                   \
  do {                                                                                    \
    hipError_t err__ = (expr);                                                            \
    if (err__ != hipSuccess) {                                                            \
      ::conch::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), __FILE__, \
                         __LINE__);                                                       \
      return CONCH_ERR_HIP;                                                               \
    }                                                                                     \
  } while (0)

// A buffer descriptor that hipcc can see is wave-uniform: the base pointer (two halves) and the byte count go through
// v_readfirstlane first.  Without it a descriptor whose inputs hipcc computed on the vector ALU (SGPR pressure, a select, a value
// carried in a struct with per-lane members) is kept in VGPRs and EVERY buffer instruction that uses it is wrapped in a
// "waterfall" loop -- four v_readfirstlane, two v_cmp, s_and_saveexec, the instruction, a branch -- found around half of the
// LDS-DMA instructions of the scaled GEMM's K loop in round 3 (cdna_hip_programming.md T20: `grep -c s_and_saveexec` next to
// buffer ops in the .s; tools/isa_waterfalls.py).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_uniform_rsrc(const void* base, uint32_t bytes) {
#ifdef CONCH_EXP_PLAIN_RSRC  // A/B variant: the descriptor as rounds 1-2 built it
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
#endif
  const uint64_t a = (uint64_t)base;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

// Library-owned device scratch, one buffer per (device, stream, slot), grown on demand and kept for the life of the
// process (stream-ordered hipMallocAsync/hipFreeAsync per call measured ~10 us on ROCm 7.2).  A buffer is only ever used
// by work enqueued on its own stream, so reuse is ordered by the stream; a grown slot never frees the old buffer (work
// already enqueued or captured into a graph may still use it).  Growth is refused while the stream is being captured
// (conch_reserve_scratch beforehand makes every entry point allocation-free).  `zero_on_alloc`: the buffer holds
// self-resetting arrival counters (all zero between launches).
enum ScratchSlot {
  kScratchRepack = 0,     // repack.hip: K-contiguous / bf16-expanded operand copies
  kScratchSplitK = 1,     // gemm_skinny.hip: split-K partial slabs
  kScratchMixedSplitK = 2,  // gemm_mixed_skinny.hip: split-K partial slabs
  kScratchWide = 3,       // unfused FFN pair: the [M][2d] intermediate
  kScratchCounters = 4,   // split-K arrival counters (zeroed at allocation, reset by the last arriver)
  kScratchFlags = 5,      // device-side dispatch words (e4m3fnuz: "an operand holds a code the fp8 MFMA cannot take")
  kScratchSlots = 6
};
// kScratchFlags layout: the e4m3fnuz dispatch word at byte 0, a 0.25 x scale_a copy of up to kFnuzMaxScales floats at byte 256
constexpr int kFnuzMaxScales = 65536;
constexpr size_t kFlagsBytes = 256 + 4 * (size_t)kFnuzMaxScales;

int get_scratch(hipStream_t stream, int slot, size_t bytes, void** out, bool zero_on_alloc = false);

// Compute units of the CURRENT device (cached per device id; 256 if the query fails).
int device_cu_count();

inline int check_launch(const char* what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error("launch of %s failed: %s", what, hipGetErrorString(err));
    return CONCH_ERR_HIP;
  }
  return CONCH_OK;
}

// ---------------------------------------------------------------------------------------------
// vector register types (wave64 MFMA operand shapes)
// ---------------------------------------------------------------------------------------------
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

#ifdef CONCH_CLOCK_PROBE
// Diagnostic build only (tools/clock_probe.py builds a second library with -DCONCH_CLOCK_PROBE; the shipped
// library contains none of this).  Thread 0 of a workgroup stamps the shader clock (s_memtime) and the constant
// 100 MHz reference clock (s_memrealtime) at kernel entry (2), before (0) and after (1) the K loop and after the
// epilogue (3) into a buffer nothing else reads:
// in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
constexpr int kProbeBlocks = 4096;
__device__ __forceinline__ void probe_stamp(unsigned long long* slot) {
  slot[0] = __builtin_amdgcn_s_memtime();
  slot[1] = __builtin_amdgcn_s_memrealtime();
}
#define CONCH_PROBE(buf, which)                                                                    \
  do {                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x < kProbeBlocks) probe_stamp(&(buf)[blockIdx.x * 8 + 2 * (which)]); \
  } while (0)
#define CONCH_PROBE_AT(buf, which, idx)                                                           \
  do {                                                                                             \
    if (threadIdx.x == 0 && (idx) < kProbeBlocks) probe_stamp(&(buf)[(idx) * 8 + 2 * (which)]);    \
  } while (0)
// eight 100 MHz stamps per workgroup (slot 0..7), workgroup index given by the kernel
#define CONCH_STAMP(buf, slot, idx)                                                                \
  do {                                                                                             \
    if (threadIdx.x == 0 && (idx) < kProbeBlocks) (buf)[(idx) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define CONCH_PROBE_READER(fn, buf)                                                                \
  extern "C" int fn(unsigned long long* out, int n_blocks) {                                       \
    if (n_blocks > conch::kProbeBlocks) n_blocks = conch::kProbeBlocks;                            \
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(buf), sizeof(unsigned long long) * 8 * n_blocks); \
  }
#else
#define CONCH_PROBE(buf, which) do { } while (0)
#define CONCH_PROBE_AT(buf, which, idx) do { } while (0)
#define CONCH_STAMP(buf, slot, idx) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// scalar conversions (device).  float -> half/bf16 are round-to-nearest-even, NaN preserving
// (hipcc lowers the casts to v_cvt_f16_f32 / v_cvt_pk_bf16_f32 on gfx950).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bits_to_float(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t float_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ float half_bits_to_float(uint16_t h) {
  return (float)__builtin_bit_cast(_Float16, h);
}
__device__ __forceinline__ float bf16_bits_to_float(uint16_t h) {
  return bits_to_float(((uint32_t)h) << 16);
}
__device__ __forceinline__ uint16_t float_to_half_bits(float f) {
  return __builtin_bit_cast(uint16_t, (_Float16)f);
}
__device__ __forceinline__ uint16_t float_to_bf16_bits(float f) {
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}

// Make an fp32 value opaque to the optimiser.  hipcc otherwise folds `(_Float16)(a * b)` into
// v_fma_mixlo_f16, which rounds the exact product ONCE to fp16; the reference rounds to fp32 first
// and then to fp16 (torch), and the two differ in rare double-rounding cases.
// NOT volatile: the value must be opaque, the statement need not be ordered -- a volatile asm is a barrier for every load and
// store around it (bnb.hip's table lookups completed one LDS round trip at a time behind one).
#ifdef CONCH_EXP_VOLATILE_PIN  // A/B variant (tools/ab_lib.py): the ordered form the tree had before
#define CONCH_PIN_ASM asm volatile
#else
#define CONCH_PIN_ASM asm
#endif
__device__ __forceinline__ float pin_f32(float v) {
  CONCH_PIN_ASM("" : "+v"(v));
  return v;
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pin_f32x2(f32x2 v) {
  CONCH_PIN_ASM("" : "+v"(v));
  return v;
}

// Two fp32 -> two fp16 / bf16 (RNE) in one dword, element 0 in the low half: v_cvt_pk_{f16,bf16}_f32 on gfx950.
template <int DT>
__device__ __forceinline__ uint32_t pack2_bits16(f32x2 v) {
  if constexpr (DT == CONCH_DT_FP16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
template <int DT>
__device__ __forceinline__ f32x2 unpack2_bits16(uint32_t w) {
  if constexpr (DT == CONCH_DT_FP16) return __builtin_convertvector(__builtin_bit_cast(f16x2, w), f32x2);
  else return f32x2{bits_to_float(w << 16), bits_to_float(w & 0xffff0000u)};
}

// silu(g) = g / (1 + exp(-g)) in fp32 on the transcendental unit (v_exp_f32, v_rcp_f32: 1 ulp, no denormals).  The
// denominator is scaled by 2^-32 before the reciprocal and the quotient by 2^-32 after it (both exact), so that the
// reciprocal of a denominator near FLT_MAX (g ~ -88, silu ~ 1e-37: still a normal number) is not flushed to zero.
__device__ __forceinline__ float silu_f32(float g) {
  const float e = __builtin_amdgcn_exp2f(g * -1.44269504088896340736f);  // exp(-g); +inf for g < -88.7 -> silu = -0
  return (g * __builtin_amdgcn_rcpf((1.0f + e) * 0x1p-32f)) * 0x1p-32f;
}

// gelu(g, approximate="tanh") = 0.5 g (1 + tanh(z)), z = sqrt(2/pi) (g + 0.044715 g^3)
// (conch/kernels/activation/gelu_tanh_and_mul.py:66-68; reference/activation/gelu_tanh_and_mul.py:13-16 = F.gelu).
// 0.5 (1 + tanh z) is sigmoid(2z), so this is g * sigmoid(2z) and shares silu's form -- and its care for the denominator.
__device__ __forceinline__ float gelu_tanh_f32(float g) {
  const float z = 0.7978845608028654f * (g + 0.044715f * (g * g * g));
  const float e = __builtin_amdgcn_exp2f(z * -2.88539008177792681472f);  // exp(-2z)
  return (g * __builtin_amdgcn_rcpf((1.0f + e) * 0x1p-32f)) * 0x1p-32f;
}

// gate activation of the fused FFN epilogues: the value of the args' `fuse_silu` field (1 = silu, 2 = gelu-tanh)
enum { kActSilu = 1, kActGeluTanh = 2 };
__device__ __forceinline__ float act_f32(float g, int act) { return act == kActGeluTanh ? gelu_tanh_f32(g) : silu_f32(g); }

template <int DT>
__device__ __forceinline__ float load_as_float(const void* p, int64_t idx) {
  if constexpr (DT == CONCH_DT_FP32) {
    return ((const float*)p)[idx];
  } else if constexpr (DT == CONCH_DT_FP16) {
    return half_bits_to_float(((const uint16_t*)p)[idx]);
  } else {
    static_assert(DT == CONCH_DT_BF16, "unsupported dtype");
    return bf16_bits_to_float(((const uint16_t*)p)[idx]);
  }
}

template <int DT>
__device__ __forceinline__ float bits16_to_float(uint16_t h) {
  if constexpr (DT == CONCH_DT_FP16) return half_bits_to_float(h);
  else return bf16_bits_to_float(h);
}
template <int DT>
__device__ __forceinline__ uint16_t float_to_bits16(float f) {
  if constexpr (DT == CONCH_DT_FP16) return float_to_half_bits(f);
  else return float_to_bf16_bits(f);
}

// ---------------------------------------------------------------------------------------------
// FP8 E4M3 codec (device).  FNUZ=false: OCP e4m3fn (bias 7, NaN 0x7F/0xFF, has -0);
// FNUZ=true: e4m3fnuz (bias 8, NaN 0x80, no -0).  Encoding is round-to-nearest-even of a value
// already clamped to +-max (so no overflow handling is needed beyond NaN propagation).
// ---------------------------------------------------------------------------------------------
template <bool FNUZ>
__device__ __forceinline__ float fp8_max() { return FNUZ ? 240.0f : 448.0f; }

template <bool FNUZ>
__device__ __forceinline__ uint32_t encode_fp8_sw(float v) {
  constexpr int BIAS = FNUZ ? 8 : 7;
  const uint32_t u = float_to_bits(v);
  const uint32_t sign = u >> 31;
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return FNUZ ? 0x80u : (0x7fu | (sign << 7));
  uint32_t code;
  const float av = bits_to_float(a);
  const float min_normal = bits_to_float((uint32_t)(127 + 1 - BIAS) << 23);
  if (av < min_normal) {
    // subnormal grid: multiples of 2^(1-BIAS-3); rint is round-half-even.  8 -> smallest normal.
    code = (uint32_t)__builtin_rintf(av * bits_to_float((uint32_t)(127 + BIAS + 2) << 23));
  } else {
    const uint32_t r = a + 0x7ffffu + ((a >> 20) & 1u);  // RNE on the 20 dropped mantissa bits
    code = (r >> 20) - ((uint32_t)(127 - BIAS) << 3);
  }
  if (FNUZ) return code == 0 ? 0u : (code | (sign << 7));
  return code | (sign << 7);
}

template <bool FNUZ>
__device__ __forceinline__ float decode_fp8_sw(uint32_t c) {
  constexpr int BIAS = FNUZ ? 8 : 7;
  const uint32_t sign = (c >> 7) & 1u;
  const uint32_t e = (c >> 3) & 0xfu;
  const uint32_t m = c & 7u;
  if (FNUZ) {
    if (c == 0x80u) return bits_to_float(0x7fc00000u);
  } else {
    if ((c & 0x7fu) == 0x7fu) return bits_to_float(0x7fc00000u | (sign << 31));
  }
  float v;
  if (e == 0) {
    v = (float)m * bits_to_float((uint32_t)(127 + 1 - BIAS - 3) << 23);
  } else {
    v = bits_to_float(((e + 127 - BIAS) << 23) | (m << 20));
  }
  return sign ? -v : v;
}

}  // namespace conch
