// Host-side shim of the hot ops: argument checks, output allocation and the C-ABI call in C++ (round 4).
//
// A decode-size op call is host-bound: 13-14 us of Python per call in round 3, ~9 us with the short Python path of
// kernels/quantization/_fast.py (torch.empty 1.8, ~30 tensor attribute calls, the 20-argument ctypes marshalling), against 6-9 us
// of GPU time.  This module does the same steps with ATen from C++ -- the SAME entry points of libconch_amd.so (include/conch_amd.h),
// bound by dlsym from the library the package has already loaded, the same checks as _fast.py, None for anything but the plain case
// (the Python paths then handle or refuse it) -- so the ops keep their behaviour and lose ~3 us per call.  PyTorch is used for what
// the task uses it for: device memory and the current stream.
#include <dlfcn.h>

#include <string>
#include <type_traits>
#include <vector>

#include <c10/hip/HIPStream.h>
#include <torch/csrc/Dtype.h>
#include <torch/extension.h>

#include "conch_amd.h"

namespace {

// The entry points are typed FROM the header (decltype), so a changed signature in include/conch_amd.h fails to compile here instead of
// calling through a mismatched function pointer; bind_library also compares conch_abi_version() with the header's number.
typedef decltype(&conch_scaled_gemm) scaled_gemm_fn;                      // also the two fused gate/up forms (same signature)
typedef decltype(&conch_mixed_precision_gemm) mixed_gemm_fn;              // ditto
typedef decltype(&conch_static_scaled_int8_quant_typed) int8_quant_fn;
typedef decltype(&conch_static_scaled_fp8_quant) fp8_quant_fn;
typedef decltype(&conch_dynamic_scaled_int8_quant) dyn_int8_fn;
typedef decltype(&conch_dynamic_scaled_fp8_quant) dyn_fp8_fn;
typedef decltype(&conch_static_quant_scaled_gemm) quant_gemm_fn;
typedef decltype(&conch_last_error) last_error_fn;
typedef decltype(&conch_abi_version) abi_version_fn;
static_assert(std::is_same<decltype(&conch_scaled_gemm_silu_and_mul), scaled_gemm_fn>::value && std::is_same<decltype(&conch_scaled_gemm_gelu_tanh_and_mul), scaled_gemm_fn>::value,
              "the fused scaled forms share conch_scaled_gemm's signature");
static_assert(std::is_same<decltype(&conch_mixed_precision_gemm_silu_and_mul), mixed_gemm_fn>::value &&
                  std::is_same<decltype(&conch_mixed_precision_gemm_gelu_tanh_and_mul), mixed_gemm_fn>::value,
              "the fused mixed forms share conch_mixed_precision_gemm's signature");

scaled_gemm_fn g_scaled = nullptr, g_scaled_silu = nullptr, g_scaled_gelu = nullptr;
mixed_gemm_fn g_mixed = nullptr, g_mixed_silu = nullptr, g_mixed_gelu = nullptr;
int8_quant_fn g_int8 = nullptr;
fp8_quant_fn g_fp8 = nullptr;
quant_gemm_fn g_quant_gemm = nullptr;
dyn_int8_fn g_dyn_int8 = nullptr;
dyn_fp8_fn g_dyn_fp8 = nullptr;
last_error_fn g_last_error = nullptr;

// conch_dtype_t of include/conch_amd.h
enum { DT_FP32 = 0, DT_FP16 = 1, DT_BF16 = 2, DT_FP8_E4M3FN = 3, DT_INT8 = 4, DT_FP8_E4M3FNUZ = 9 };

int dtype_code(at::ScalarType t) {
  switch (t) {
    case at::kFloat: return DT_FP32;
    case at::kHalf: return DT_FP16;
    case at::kBFloat16: return DT_BF16;
    case at::kFloat8_e4m3fn: return DT_FP8_E4M3FN;
    case at::kFloat8_e4m3fnuz: return DT_FP8_E4M3FNUZ;
    case at::kChar: return DT_INT8;
    default: return -1;
  }
}

void bind_library(const std::string& path) {
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);  // already loaded by ctypes: the same handle, the same library state
  if (!h) throw std::runtime_error(std::string("conch_amd host shim: cannot open ") + path + ": " + dlerror());
  const abi_version_fn abi = (abi_version_fn)dlsym(h, "conch_abi_version");
  if (!abi || abi() != CONCH_AMD_ABI_VERSION)
    throw std::runtime_error("conch_amd host shim: " + path + " reports ABI version " + (abi ? std::to_string(abi()) : std::string("(none)")) +
                             ", this shim was compiled against " + std::to_string(CONCH_AMD_ABI_VERSION) + " (rebuild: python -m conch_amd._build)");
  g_scaled = (scaled_gemm_fn)dlsym(h, "conch_scaled_gemm");
  g_mixed = (mixed_gemm_fn)dlsym(h, "conch_mixed_precision_gemm");
  // the fused gate/up FFN forms share their plain op's signature (n = the OUTPUT width, half of b's columns)
  g_scaled_silu = (scaled_gemm_fn)dlsym(h, "conch_scaled_gemm_silu_and_mul");
  g_scaled_gelu = (scaled_gemm_fn)dlsym(h, "conch_scaled_gemm_gelu_tanh_and_mul");
  g_mixed_silu = (mixed_gemm_fn)dlsym(h, "conch_mixed_precision_gemm_silu_and_mul");
  g_mixed_gelu = (mixed_gemm_fn)dlsym(h, "conch_mixed_precision_gemm_gelu_tanh_and_mul");
  g_int8 = (int8_quant_fn)dlsym(h, "conch_static_scaled_int8_quant_typed");
  g_fp8 = (fp8_quant_fn)dlsym(h, "conch_static_scaled_fp8_quant");
  g_quant_gemm = (quant_gemm_fn)dlsym(h, "conch_static_quant_scaled_gemm");
  g_dyn_int8 = (dyn_int8_fn)dlsym(h, "conch_dynamic_scaled_int8_quant");
  g_dyn_fp8 = (dyn_fp8_fn)dlsym(h, "conch_dynamic_scaled_fp8_quant");
  g_last_error = (last_error_fn)dlsym(h, "conch_last_error");
  if (!g_scaled_silu || !g_scaled_gelu || !g_mixed_silu || !g_mixed_gelu) throw std::runtime_error("conch_amd host shim: " + path + " lacks a fused FFN entry point");
  if (!g_scaled || !g_mixed || !g_int8 || !g_fp8 || !g_quant_gemm || !g_dyn_int8 || !g_dyn_fp8 || !g_last_error) throw std::runtime_error("conch_amd host shim: " + path + " lacks an entry point");
}

[[noreturn]] void raise_status(int status, const char* what) {
  const std::string msg = std::string(what) + ": " + (g_last_error ? g_last_error() : "");
  if (status == 1) throw py::value_error(msg);
  if (status == 2) {
    PyErr_SetString(PyExc_NotImplementedError, msg.c_str());
    throw py::error_already_set();
  }
  throw std::runtime_error(std::string(what) + " failed with status " + std::to_string(status) + ": " + (g_last_error ? g_last_error() : ""));
}

inline bool on_current_device(const at::Tensor& t, c10::DeviceIndex dev) { return t.is_cuda() && t.get_device() == dev; }

at::ScalarType dtype_of(const py::object& o) {
  if (!THPDtype_Check(o.ptr())) throw py::type_error("expected a torch.dtype");
  return reinterpret_cast<THPDtype*>(o.ptr())->scalar_type;
}

// conch_scaled_gemm for 2-D a, b of one 8-bit dtype on the current device, contiguous float32 scales, contiguous bias in the output
// dtype; None = not that case (kernels/quantization/_fast.py has the same contract in Python)
// act: 0 = the plain op; 1 / 2 = the fused gate/up forms (silu / gelu-tanh): b, scale_b and bias have 2d columns [gate | up], the result d
py::object scaled_gemm_act(const at::Tensor& a, const at::Tensor& b, const at::Tensor& sa, const at::Tensor& sb, const py::object& out_dtype_obj,
                           const c10::optional<at::Tensor>& bias, int act) {
  if (!a.is_cuda()) return py::none();
  const c10::DeviceIndex dev = a.get_device();
  if (dev != c10::hip::current_device() || !on_current_device(b, dev) || !on_current_device(sa, dev) || !on_current_device(sb, dev)) return py::none();
  const at::ScalarType out_dtype = dtype_of(out_dtype_obj);
  const int code = dtype_code(a.scalar_type()), out_code = dtype_code(out_dtype);
  if (code < 0 || out_code < 0 || b.scalar_type() != a.scalar_type() || sa.scalar_type() != at::kFloat || sb.scalar_type() != at::kFloat) return py::none();
  if (a.dim() != 2 || b.dim() != 2 || a.size(1) != b.size(0) || !sa.is_contiguous() || !sb.is_contiguous()) return py::none();
  const int64_t m = a.size(0), k = a.size(1), n = b.size(1);
  if (act && (n % 2 || (out_code != DT_FP16 && out_code != DT_BF16))) return py::none();
  const void* bias_ptr = nullptr;
  if (bias.has_value()) {
    const at::Tensor& bt = *bias;
    if (bt.scalar_type() != out_dtype || !on_current_device(bt, dev) || bt.numel() != n || !bt.is_contiguous()) return py::none();
    bias_ptr = bt.data_ptr();
  }
  const int64_t n_out = act ? n / 2 : n;
  at::Tensor out = at::empty({m, n_out}, a.options().dtype(out_dtype));
  const scaled_gemm_fn fn = act == 1 ? g_scaled_silu : act == 2 ? g_scaled_gelu : g_scaled;
  const int status = fn(out.data_ptr(), a.data_ptr(), b.data_ptr(), (const float*)sa.data_ptr(), (const float*)sb.data_ptr(), bias_ptr, m, n_out, k,
                        a.stride(0), a.stride(1), b.stride(0), b.stride(1), n_out, 1, sa.numel(), sb.numel(), code, out_code,
                        (void*)c10::hip::getCurrentHIPStream(dev).stream());
  if (status) raise_status(status, act == 1 ? "scaled_gemm_silu_and_mul" : act == 2 ? "scaled_gemm_gelu_tanh_and_mul" : "scaled_gemm");
  return py::cast(out);
}

py::object scaled_gemm(const at::Tensor& a, const at::Tensor& b, const at::Tensor& sa, const at::Tensor& sb, const py::object& out_dtype_obj,
                       const c10::optional<at::Tensor>& bias) {
  return scaled_gemm_act(a, b, sa, sb, out_dtype_obj, bias, 0);
}

py::object mixed_precision_gemm_act(const at::Tensor& x, const at::Tensor& wq, const at::Tensor& ws, const c10::optional<at::Tensor>& wzp, int64_t bits,
                                    int64_t weight_bias, int64_t group_size, int act) {
  if (!x.is_cuda()) return py::none();
  const c10::DeviceIndex dev = x.get_device();
  if (dev != c10::hip::current_device() || !on_current_device(wq, dev) || !on_current_device(ws, dev)) return py::none();
  const at::ScalarType dt = x.scalar_type();
  if ((dt != at::kHalf && dt != at::kBFloat16) || ws.scalar_type() != dt || wq.scalar_type() != at::kInt || (bits != 4 && bits != 8)) return py::none();
  if (x.dim() != 2 || wq.dim() != 2 || ws.dim() != 2 || x.stride(1) != 1 || wq.stride(1) != 1 || ws.stride(1) != 1) return py::none();
  const int64_t m = x.size(0), k = x.size(1), n = wq.size(1);
  if (wq.size(0) * (32 / bits) != k || group_size <= 0 || k % group_size || ws.size(0) * group_size != k || ws.size(1) != n) return py::none();
  const int32_t* zp_ptr = nullptr;
  int zp_mode = 0;
  int64_t zp_stride = 0;
  if (wzp.has_value()) {
    const at::Tensor& z = *wzp;
    if (!on_current_device(z, dev) || z.scalar_type() != at::kInt) return py::none();
    if (z.numel() == 1) zp_mode = 1;
    else if (z.dim() == 2 && z.size(0) == ws.size(0) && z.size(1) == n && z.stride(1) == 1) {
      zp_mode = 2;
      zp_stride = z.stride(0);
    } else return py::none();
    zp_ptr = (const int32_t*)z.data_ptr();
  }
  if (act && n % 2) return py::none();
  const int64_t n_out = act ? n / 2 : n;
  at::Tensor out = at::empty({m, n_out}, x.options());
  const int code = dtype_code(dt);
  const mixed_gemm_fn fn = act == 1 ? g_mixed_silu : act == 2 ? g_mixed_gelu : g_mixed;
  const int status = fn(out.data_ptr(), x.data_ptr(), (const int32_t*)wq.data_ptr(), ws.data_ptr(), zp_ptr, m, n_out, k, x.stride(0), wq.stride(0),
                        ws.stride(0), zp_stride, n_out, (int)bits, (int)weight_bias, (int)group_size, zp_mode, code, code,
                        (void*)c10::hip::getCurrentHIPStream(dev).stream());
  if (status) raise_status(status, act == 1 ? "mixed_precision_gemm_silu_and_mul" : act == 2 ? "mixed_precision_gemm_gelu_tanh_and_mul" : "mixed_precision_gemm");
  return py::cast(out);
}

py::object mixed_precision_gemm(const at::Tensor& x, const at::Tensor& wq, const at::Tensor& ws, const c10::optional<at::Tensor>& wzp, int64_t bits,
                                int64_t weight_bias, int64_t group_size) {
  return mixed_precision_gemm_act(x, wq, ws, wzp, bits, weight_bias, group_size, 0);
}

// static per-tensor quantisation of a contiguous tensor: int8 (kind 0), e4m3fn (1), e4m3fnuz (2); None = not the plain case
py::object static_quant(const at::Tensor& x, const at::Tensor& scale, int64_t kind) {
  if (kind < 0 || kind > 2) return py::none();  // an output dtype this path does not know: the general path raises the proper error
  if (!x.is_cuda() || x.dim() < 1 || !x.is_contiguous()) return py::none();
  const c10::DeviceIndex dev = x.get_device();
  if (dev != c10::hip::current_device() || !on_current_device(scale, dev) || scale.scalar_type() != at::kFloat || scale.numel() != 1) return py::none();
  const int xcode = dtype_code(x.scalar_type());
  if (xcode != DT_FP32 && xcode != DT_FP16 && xcode != DT_BF16) return py::none();
  const int64_t hidden = x.size(-1), tokens = hidden ? x.numel() / hidden : 0;
  void* stream = (void*)c10::hip::getCurrentHIPStream(dev).stream();
  if (kind == 0) {
    at::Tensor out = at::empty_like(x, x.options().dtype(at::kChar));
    // a 0-dim scale makes torch round the product to x's dtype (kernels/quantization/int8.py)
    const int product = (scale.dim() == 0 && xcode != DT_FP32) ? xcode : DT_FP32;
    const int status = g_int8((int8_t*)out.data_ptr(), x.data_ptr(), (const float*)scale.data_ptr(), tokens, hidden, hidden, hidden, xcode, product, stream);
    if (status) raise_status(status, "static_scaled_int8_quant");
    return py::cast(out);
  }
  const at::ScalarType odt = kind == 2 ? at::kFloat8_e4m3fnuz : at::kFloat8_e4m3fn;
  at::Tensor out = at::empty_like(x, x.options().dtype(odt));
  const int status = g_fp8((uint8_t*)out.data_ptr(), x.data_ptr(), (const float*)scale.data_ptr(), tokens, hidden, hidden, hidden, xcode, dtype_code(odt), stream);
  if (status) raise_status(status, "static_scaled_fp8_quant");
  return py::cast(out);
}

// dynamic per-token quantisation of a contiguous tensor (>= 1 dimension): (codes, scales of shape x.shape[:-1] + (1,)) -- what
// ops/quantization/_static_quant.py quantize_new returns for scale=None; kind 0 = int8, 1 = e4m3fn, 2 = e4m3fnuz; None = not the plain case
py::object dynamic_quant(const at::Tensor& x, int64_t kind) {
  if (kind < 0 || kind > 2) return py::none();
  if (!x.is_cuda() || x.dim() < 1 || !x.is_contiguous() || x.numel() == 0) return py::none();
  const c10::DeviceIndex dev = x.get_device();
  if (dev != c10::hip::current_device()) return py::none();
  const int xcode = dtype_code(x.scalar_type());
  if (xcode != DT_FP32 && xcode != DT_FP16 && xcode != DT_BF16) return py::none();
  const int64_t hidden = x.size(-1), tokens = x.numel() / hidden;
  std::vector<int64_t> sshape(x.sizes().begin(), x.sizes().end());
  sshape.back() = 1;
  at::Tensor scales = at::empty(sshape, x.options().dtype(at::kFloat));
  void* stream = (void*)c10::hip::getCurrentHIPStream(dev).stream();
  if (kind == 0) {
    at::Tensor out = at::empty_like(x, x.options().dtype(at::kChar));
    const int status = g_dyn_int8((int8_t*)out.data_ptr(), (float*)scales.data_ptr(), x.data_ptr(), tokens, hidden, hidden, hidden, xcode, stream);
    if (status) raise_status(status, "dynamic_scaled_int8_quant");
    return py::make_tuple(out, scales);
  }
  const at::ScalarType odt = kind == 2 ? at::kFloat8_e4m3fnuz : at::kFloat8_e4m3fn;
  at::Tensor out = at::empty_like(x, x.options().dtype(odt));
  const int status = g_dyn_fp8((uint8_t*)out.data_ptr(), (float*)scales.data_ptr(), x.data_ptr(), tokens, hidden, hidden, hidden, xcode, dtype_code(odt), stream);
  if (status) raise_status(status, "dynamic_scaled_fp8_quant");
  return py::make_tuple(out, scales);
}

// conch_static_quant_scaled_gemm for 2-D fp16 / bf16 activations, 2-D int8 / e4m3fn weights, one float32 activation scale (not 0-dim
// with int8 weights: that form runs the unfused pair, ops/quantization/gemm.py), contiguous float32 scale_b, contiguous bias in the
// output dtype; None = not that case
py::object static_quant_scaled_gemm(const at::Tensor& x, const at::Tensor& b, const at::Tensor& sx, const at::Tensor& sb, const py::object& out_dtype_obj,
                                    const c10::optional<at::Tensor>& bias) {
  if (!x.is_cuda()) return py::none();
  const c10::DeviceIndex dev = x.get_device();
  if (dev != c10::hip::current_device() || !on_current_device(b, dev) || !on_current_device(sx, dev) || !on_current_device(sb, dev)) return py::none();
  const at::ScalarType out_dtype = dtype_of(out_dtype_obj);
  const int xcode = dtype_code(x.scalar_type()), bcode = dtype_code(b.scalar_type()), out_code = dtype_code(out_dtype);
  if ((xcode != DT_FP16 && xcode != DT_BF16) || (bcode != DT_INT8 && bcode != DT_FP8_E4M3FN) || (out_code != DT_FP16 && out_code != DT_BF16)) return py::none();
  if (x.dim() != 2 || b.dim() != 2 || x.size(1) != b.size(0)) return py::none();
  if (sx.scalar_type() != at::kFloat || sx.numel() != 1 || (sx.dim() == 0 && bcode == DT_INT8)) return py::none();
  if (sb.scalar_type() != at::kFloat || !sb.is_contiguous()) return py::none();
  const int64_t m = x.size(0), k = x.size(1), n = b.size(1);
  if (sb.numel() != 1 && sb.numel() != n) return py::none();
  const void* bias_ptr = nullptr;
  if (bias.has_value()) {
    const at::Tensor& bt = *bias;
    if (bt.scalar_type() != out_dtype || !on_current_device(bt, dev) || bt.numel() != n || !bt.is_contiguous()) return py::none();
    bias_ptr = bt.data_ptr();
  }
  at::Tensor out = at::empty({m, n}, x.options().dtype(out_dtype));
  const int status = g_quant_gemm(out.data_ptr(), x.data_ptr(), b.data_ptr(), (const float*)sx.data_ptr(), (const float*)sb.data_ptr(), bias_ptr, m, n, k,
                                  x.stride(0), x.stride(1), b.stride(0), b.stride(1), n, 1, sb.numel(), xcode, bcode, out_code,
                                  (void*)c10::hip::getCurrentHIPStream(dev).stream());
  if (status) raise_status(status, "static_quant_scaled_gemm");
  return py::cast(out);
}

}  // namespace

PYBIND11_MODULE(_conch_host, mod) {
  mod.doc() = "conch_amd host shim: checks + allocation + C-ABI call of the hot ops in C++";
  mod.def("bind_library", &bind_library);
  mod.def("scaled_gemm", &scaled_gemm);
  mod.def("mixed_precision_gemm", &mixed_precision_gemm);
  mod.def("static_quant", &static_quant);
  mod.def("static_quant_scaled_gemm", &static_quant_scaled_gemm);
  mod.def("dynamic_quant", &dynamic_quant);
  mod.def("scaled_gemm_act", &scaled_gemm_act);
  mod.def("mixed_precision_gemm_act", &mixed_precision_gemm_act);
}
