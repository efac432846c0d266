// Kernel instantiations of the multi-tap row-shifted GEMM for float operands and bf16 outputs.
#include "convgemm_impl.h"

hipError_t q3_convgemm_f32_bf16(const ConvGemmArgs& a, hipStream_t s) { return launch_typed<float, bf16, false>(a, s); }
