// Mixed-precision (int4/int8 packed weights x fp16/bf16 activations) MFMA kernels.
#include "common.hpp"
#include "gemm.hpp"

namespace conch {

bool mixed_gemm_mfma_supported(const MixedGemmArgs&) { return false; }

int launch_mixed_gemm_mfma(const MixedGemmArgs&, int, hipStream_t) {
  set_error("mixed_precision_gemm: MFMA kernel not built yet");
  return CONCH_ERR_UNSUPPORTED;
}

}  // namespace conch
