/*
 * oracle_c.c -- plain-C restatement of the byte/integer parts of the quantized-matmul path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): never linked into conch_amd.  It exists so that
 * the Python oracle's use of torch casts is cross-checked by code that shares nothing with torch:
 * scalar loops, explicit IEEE bit manipulation, round-to-nearest-even written out by hand.
 *
 *   static int8 quantisation      conch/reference/quantization/int8.py:12-18
 *   static fp8 quantisation       conch/reference/quantization/fp8.py:12-18 (e4m3fn and e4m3fnuz)
 *   scaled GEMM, int8 operands    conch/reference/quantization/scaled_gemm.py:12-27
 *   packed-weight dequantisation  conch/kernels/quantization/gemm.py:176-216 (== quant_utils.py:74 w_ref)
 *
 * Parity status: pinned -- tests/test_oracle_c.py runs it against tests/golden/ *.npz (outputs of the
 * real reference) bit-for-bit.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

enum { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2 };

static float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static float half_to_f32(uint16_t h) {
  uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 0x1f, m = h & 0x3ff;
  if (e == 0) {
    if (m == 0) return bits_f32(s);
    float v = ldexpf((float)m, -24);
    return s ? -v : v;
  }
  if (e == 31) return bits_f32(s | 0x7f800000u | (m << 13));
  return bits_f32(s | ((e + 112) << 23) | (m << 13));
}
static float bf16_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }

/* float -> bf16 / fp16, round-to-nearest-even, NaN preserved */
static uint16_t f32_to_bf16(float f) {
  uint32_t u = f32_bits(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static uint16_t f32_to_half(float f) {
  uint32_t u = f32_bits(f), s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(s | 0x7e00u);
  if (a >= 0x477ff000u) return (uint16_t)(s | 0x7c00u);             /* rounds to >= 65520 -> inf */
  if (a < 0x33000001u) return (uint16_t)s;                           /* < 2^-25 (or == with tie to even) -> 0 */
  int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u;
  int shift = e < -14 ? (13 + (-14 - e)) : 13;                       /* subnormal halves lose more bits */
  uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) q++;
  if (e < -14) return (uint16_t)(s | q);                             /* q may carry into the exponent: still right */
  return (uint16_t)(s | (((uint32_t)(e + 15) << 10) + (q - 0x400u)));
}
static float load_elem(const void* p, int64_t i, int dt) {
  if (dt == DT_F32) return ((const float*)p)[i];
  if (dt == DT_F16) return half_to_f32(((const uint16_t*)p)[i]);
  return bf16_to_f32(((const uint16_t*)p)[i]);
}
static uint16_t to16(float f, int dt) { return dt == DT_F16 ? f32_to_half(f) : f32_to_bf16(f); }
static float from16(uint16_t h, int dt) { return dt == DT_F16 ? half_to_f32(h) : bf16_to_f32(h); }

/* q = (int8) trunc(clamp(x * (1/scale), -128, 127)), product in fp32 */
void oracle_int8_quant(int8_t* out, const void* x, int x_dtype, float scale, int64_t n) {
  const float inv = 1.0f / scale;
  for (int64_t i = 0; i < n; ++i) {
    float p = load_elem(x, i, x_dtype) * inv;
    if (p < -128.0f) p = -128.0f;
    if (p > 127.0f) p = 127.0f;
    out[i] = (int8_t)(int)p; /* C conversion truncates toward zero */
  }
}

/* e4m3 encoder, RNE, input already clamped to +-max; fnuz: bias 8, NaN 0x80, no -0 */
static uint8_t encode_e4m3(float v, int fnuz) {
  const int bias = fnuz ? 8 : 7;
  uint32_t u = f32_bits(v), sign = u >> 31, a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return fnuz ? 0x80 : (uint8_t)(0x7f | (sign << 7));
  uint32_t code;
  float av = bits_f32(a), min_normal = ldexpf(1.0f, 1 - bias);
  if (av < min_normal) {
    code = (uint32_t)nearbyintf(av * ldexpf(1.0f, bias + 2)); /* default rounding mode: ties to even */
  } else {
    uint32_t r = a + 0x7ffffu + ((a >> 20) & 1u);
    code = (r >> 20) - ((uint32_t)(127 - bias) << 3);
  }
  if (fnuz) return code == 0 ? 0 : (uint8_t)(code | (sign << 7));
  return (uint8_t)(code | (sign << 7));
}

void oracle_fp8_quant(uint8_t* out, const void* x, int x_dtype, float scale, int64_t n, int fnuz) {
  const float inv = 1.0f / scale, mx = fnuz ? 240.0f : 448.0f;
  for (int64_t i = 0; i < n; ++i) {
    float p = load_elem(x, i, x_dtype) * inv;
    if (p == p) { if (p < -mx) p = -mx; if (p > mx) p = mx; }
    out[i] = encode_e4m3(p, fnuz);
  }
}

/* C[m][n] = out16( sb[n] * (sa[m] * (float) sum_k A[m][k] * Bt[n][k]) ) (+ bias[n], added in out dtype) */
void oracle_scaled_gemm_int8(uint16_t* c, const int8_t* a, const int8_t* bt, const float* sa, int64_t sa_n,
                             const float* sb, int64_t sb_n, const uint16_t* bias, int64_t m, int64_t n, int64_t k,
                             int out_dtype) {
  for (int64_t i = 0; i < m; ++i)
    for (int64_t j = 0; j < n; ++j) {
      int32_t acc = 0;
      for (int64_t kk = 0; kk < k; ++kk) acc += (int32_t)a[i * k + kk] * (int32_t)bt[j * k + kk];
      float v = sa[sa_n == 1 ? 0 : i] * (float)acc;
      v = sb[sb_n == 1 ? 0 : j] * v;
      uint16_t h = to16(v, out_dtype);
      if (bias) h = to16(from16(h, out_dtype) + from16(bias[j], out_dtype), out_dtype);
      c[i * n + j] = h;
    }
}

/* w[k][n] = dt( (q[k][n] - bias - zp[k/G][n]) ) * s[k/G][n], ONE rounding (the product is exact in fp32) */
void oracle_dequant_packed(uint16_t* w, const int32_t* packed, const uint16_t* scales, const int32_t* zp, int zp_scalar,
                           int64_t k, int64_t n, int bits, int bias, int group, int dtype) {
  const int pf = 32 / bits;
  const uint32_t mask = bits == 32 ? 0xffffffffu : ((1u << bits) - 1u);
  for (int64_t kk = 0; kk < k; ++kk)
    for (int64_t j = 0; j < n; ++j) {
      uint32_t word = (uint32_t)packed[(kk / pf) * n + j];
      int q = (int)((word >> ((kk % pf) * bits)) & mask) - bias;
      if (zp) q -= zp_scalar ? zp[0] : zp[(kk / group) * n + j];
      w[kk * n + j] = to16((float)q * from16(scales[(kk / group) * n + j], dtype), dtype);
    }
}
