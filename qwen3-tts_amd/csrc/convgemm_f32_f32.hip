// Kernel instantiations of the multi-tap row-shifted GEMM for float operands and float outputs.
#include "convgemm_impl.h"

hipError_t q3_convgemm_f32_f32(const ConvGemmArgs& a, hipStream_t s) { return launch_typed<float, float, true>(a, s); }
