// Dispatch of the multi-tap row-shifted GEMM (convgemm_impl.h) by operand / output element type.  Each type pair is
// its own translation unit (convgemm_<in>_<out>.hip) so that the kernel instantiations compile in parallel.
#include "convgemm_impl.h"

hipError_t q3_convgemm_f32_f32(const ConvGemmArgs& a, hipStream_t s);
hipError_t q3_convgemm_bf16_bf16(const ConvGemmArgs& a, hipStream_t s);
hipError_t q3_convgemm_f32_bf16(const ConvGemmArgs& a, hipStream_t s);
hipError_t q3_convgemm_bf16_f32(const ConvGemmArgs& a, hipStream_t s);

// the h tile of the fused residual unit (256 rows x N channels of the operand type) must fit LDS beside the weight buffers
// 384 channels fit one launch too (shape "resunit 128x384"), but measured equal to the two-launch unit in bulk (22.1 ms
// against 14.9 + 6.4 ms per 64 x 300 frames: eight waves at one block per CU run the contraction at 0.77 of the
// unfused 7-tap conv's rate) and 2 % slower for frame-rate pushes, so it stays behind tuning bit 64.
int q3_convgemm_fusable(int N, int dtype, int tuning) { return N <= (dtype == Q3_BF16 ? ((tuning & 64) ? 384 : 192) : 96); }

const char* q3_convgemm_variant(const ConvGemmArgs& a, int in_dtype, int out_dtype) {
    static const char* shapes[SH_COUNT] = {"256x96,k64B,4w", "256x128,k128B,8w", "128x128,k128B,4w", "128x128,k64B,4w", "256x32,k64B,4w",
                                           "resunit 256x192,k64B,8w", "resunit 256x96,k64B,4w", "resunit 256x32,k64B,4w",
                                           "256x128,k64B,8w", "resunit 128x192,k64B,4w", "resunit 128x384,k64B,8w", "256x64,k64B,8w", "128x32,k128B,4w", "128x64,k128B,4w", "resunit 768x96,k64B,12w"};
    static const char* types[4] = {"f32,f32", "bf16,bf16", "f32,bf16", "bf16,f32"};
    static char names[4][SH_COUNT][64];
    const int t = (in_dtype == Q3_F32 && out_dtype == Q3_F32) ? 0 : (in_dtype == Q3_BF16 ? (out_dtype == Q3_BF16 ? 1 : 3) : 2);
    const int sh = pick_shape(a, in_dtype == Q3_BF16 ? 2 : 4);
    if (sh < 0) return "convgemm<unsupported>";
    if (!names[t][sh][0]) snprintf(names[t][sh], sizeof(names[t][sh]), "convgemm<%s,%s>", types[t], shapes[sh]);
    return names[t][sh];
}

hipError_t q3_launch_convgemm(const ConvGemmArgs& a_in, int in_dtype, int out_dtype, hipStream_t s) {
    ConvGemmArgs a = a_in;
    a.debug = a.tuning >> 8;
    if (a.M <= 0 || a.B <= 0) return hipSuccess;
    const int epc_in = in_dtype == Q3_BF16 ? 8 : 4;
    if (a.K % epc_in || a.ldx % epc_in || a.x_batch_stride % epc_in || a.taps < 1 || a.off_step < 0)
        return hipErrorInvalidValue;
    if (((uintptr_t)a.X | (uintptr_t)a.W | (uintptr_t)a.W2) & 15) return hipErrorInvalidValue;
    if (a.W2 && (a.N % epc_in)) return hipErrorInvalidValue;
    if (a.seg_len > 0 && (a.B != 1 || a.seg_hist < 0 || a.seg_hist >= a.seg_len || a.M % a.seg_len)) return hipErrorInvalidValue;
    if (!a.Y2) a.y2_batch_stride = 0;
    if (in_dtype == Q3_F32 && out_dtype == Q3_F32) return q3_convgemm_f32_f32(a, s);
    if (in_dtype == Q3_BF16 && out_dtype == Q3_BF16) return q3_convgemm_bf16_bf16(a, s);
    if (in_dtype == Q3_F32 && out_dtype == Q3_BF16) return q3_convgemm_f32_bf16(a, s);
    if (in_dtype == Q3_BF16 && out_dtype == Q3_F32) return q3_convgemm_bf16_f32(a, s);
    return hipErrorInvalidValue;
}
