// Kernel instantiations of the multi-tap row-shifted GEMM for bf16 operands and float outputs.
#include "convgemm_impl.h"

hipError_t q3_convgemm_bf16_f32(const ConvGemmArgs& a, hipStream_t s) { return launch_typed<bf16, float, false>(a, s); }
