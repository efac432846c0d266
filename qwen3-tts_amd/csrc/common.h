// Shared device helpers and the internal launcher interface of the gfx950 decode library.
// Activations are channels-last [batch, time, channel]; weights are [tap][out][in] with `in` contiguous.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define Q3_WAVE 64

enum Q3Act : int {
    Q3_ACT_NONE = 0,
    Q3_ACT_GELU = 1,    // exact erf GELU
    Q3_ACT_SNAKE = 2,   // x + invb * sin(a x)^2 with per-output-channel (a, invb)
    Q3_ACT_SWIGLU = 3,  // adjacent 32-column tiles hold (gate, up); out = silu(gate) * up, N_out = N / 2
    Q3_ACT_CLAMP = 4,   // clamp(x, -1, 1)
    Q3_ACT_ELU = 5,     // x > 0 ? x : exp(x) - 1
};

// h = scale[n] * act( bias[n] + sum_tap sum_k X[b, m + off0 + tap*off_step, k] * W[tap][n][k] )
// with W2:  X <- pre(X) first (per-input-channel SnakeBeta: the fused residual unit reads the RAW stream), then
//           h <- bias2[n] + sum_j h[., j] * W2[n][j]     (fused 1x1 second contraction; N == K2, scale unused)
// y = R[b, m, n] + h;   Y <- post(y), or with Y2:  Y <- y and Y2 <- post(y)   (post: SnakeBeta with post_a / post_invb, or ELU with post_elu)
// rows of X outside [row_lo, L_in) read as zero.
struct ConvGemmArgs {
    const void* X;  long long x_batch_stride; int ldx; int L_in;
    const void* W;                       // [taps][N][K]
    const float* bias;                   // [N] or null
    void* Y;        long long y_batch_stride; int ldy;
    const void* R;  long long r_batch_stride; int ldr;   // residual, element type = output type, or null
    int B, M, N, K;
    int taps, off0, off_step;
    int act;
    const float* epi_a; const float* epi_invb;           // [N] for Q3_ACT_SNAKE
    const float* scale;                                  // [N_out] or null
    const float* post_a; const float* post_invb;         // [N_out] snake after the residual add, or null
    int post_elu;                                        // 1: post is ELU (no parameters; not with W2)
    void* Y2; long long y2_batch_stride;                 // optional second output (row stride ldy)
    const int* row_start;                                // [B] or null: output rows below row_start[b] are stored as zeros
    int seg_len, seg_hist;                               // > 0: B == 1 and the M rows are segments of seg_len rows whose first
                                                         // seg_hist rows are carried context (read, never written); segment j,
                                                         // row r >= seg_hist goes to batch item j, row r - seg_hist of Y / Y2 / R
    const void* W2; const float* bias2;                  // [N][N] second contraction, or null
    const float* pre_a; const float* pre_invb;           // [K] with W2: SnakeBeta applied to the staged INPUT tile in LDS
    const void* W3; float* P; long long p_batch_stride;  // with W2, bf16, one column tile per wave: the output is NOT stored;
                                                         // instead P[b, m, t] = sum_n post(y)[b, m, n] * W3[t][n] for the 8 taps t of
                                                         // a following one-channel conv (W3: [16][N], rows >= its tap count zero;
                                                         // P rows of 8 floats), which q3_launch_tap_sum then sums along the diagonal
    int nxb;                                             // input buffers in LDS (2 or 3): set by the launcher
    int tile_order;                                      // 0: N tile slowest, 1: N tile fastest, G >= 2: N tiles in groups of G (set by the launcher)
    int tuning;                                          // the context's measurement switches (0 = defaults; q3c_debug_set_tuning): shape choices read
                                                         // bits 0-7 and 16-20 on the host, bits 8.. reach the kernel as `debug`
    int debug;                                           // measurement aid: bit 4 = plain (not XCD-grouped) tile order (set by the launcher)
    const void* zeros;                                   // >= 16 bytes of zeros in device memory (source of out-of-sequence rows)
    int row_lo;                                          // 0, or negative: X points |row_lo| rows INTO a sequence whose earlier rows exist
                                                         // and are read as the halo (a launch that computes the sequence's tail only)
    int shape_rows;                                      // > 0: choose the tile shape as for this many rows per item instead of M
    long long* timeline; int timeline_slots;             // measurement aid (null normally): every 64th workgroup stores the
                                                         // clock at its phase boundaries, 8 values per slot
};

enum Q3Dtype : int { Q3_F32 = 0, Q3_BF16 = 1 };

// ---- launchers (each returns hipError_t of the launch) -------------------------------------------
hipError_t q3_launch_convgemm(const ConvGemmArgs& a, int in_dtype, int out_dtype, hipStream_t s);
const char* q3_convgemm_variant(const ConvGemmArgs& a, int in_dtype, int out_dtype);   // name of the instantiation that would run
int q3_convgemm_fusable(int N, int dtype, int tuning = 0);     // can a residual unit of N channels run as one fused launch?

hipError_t q3_launch_rvq_gather_sum(const long long* codes, int B, int Q, int T, int bins, int D,
                                    const float* const* tables_dev /*[Q] device array of table pointers*/,
                                    const int* group_start_dev /*[G+1] first codebook of each group*/, int G,
                                    float* out /*[B,T,G,D]*/, int* err_flag, hipStream_t s);
// the encode direction: embeddings [B][T][in_dim] fp32 -> codes [B][Q][T]; in_proj [G][D][in_dim] or null (then in_dim == D)
// tables_t_dev: per code book a dimension-major copy [D/4][bins] of float4, followed by the rows' squared norms [bins]
hipError_t q3_launch_rvq_encode(const float* emb, int B, int T, int in_dim, int Q, int bins, int D, const float* const* tables_dev,
                                const void* const* tables_t_dev, const int* group_start_dev, int G, const float* in_proj_dev,
                                long long* codes, hipStream_t s, int tuning = 0 /* bit 20: the vector-ALU search */);
hipError_t q3_launch_mean_embedding(const long long* codes, int B, int Q, int T, int bins, int H,
                                    const float* table, void* out, int out_dtype, int* err_flag, hipStream_t s);
// y[b, m] = act(bias + sum_t P[b, m - (taps - 1) + t, t]) over the rows m - (taps - 1) + t >= 0: the one-channel causal conv
// whose per-tap products a fused unit left in P ([B][M][8] fp32); act: Q3_ACT_NONE or Q3_ACT_CLAMP
// m_first: rows below it are not written (their P rows may not exist; the rows m_first - (taps - 1) .. do)
// spans (device, [B][2] = destination offset into y in samples, count; or null): item b's samples m_first .. m_first + count - 1 go
// to y + offset and nothing else is written
hipError_t q3_launch_tap_sum(const float* P, const float* bias, float* y, int B, int M, int taps, int act, hipStream_t s, int m_first = 0,
                             const long long* spans = nullptr);
hipError_t q3_launch_span_copy(const float* src /*[B][M]*/, int B, int M, int m_first, const long long* spans, float* out, hipStream_t s);
// ---- the encode side's small kernels: first conv (one audio channel -> F, k taps, causal; optionally also its ELU),
// LayerNorm with bias, replicate padding ----
// y2 (or null) receives ELU(y), items y2_bstride elements apart
hipError_t q3_launch_audio_conv(const float* audio, const float* w /*[F][k]*/, const float* bias, void* y, void* y2, long long y2_bstride, int B, int L, int F, int k, int dtype, hipStream_t s);
hipError_t q3_launch_layernorm(const void* x, const float* w, const float* bias, void* y, long long rows, int H, float eps, int dtype, hipStream_t s);
hipError_t q3_launch_replicate_pad(const void* x, void* y, int B, int L, int C, int left, int Lout, int dtype, hipStream_t s);
hipError_t q3_launch_rmsnorm(const void* x, const float* w, void* y, long long rows, int H, float eps, int dtype, hipStream_t s);
// `hist` carried rows sit in front of each item's L rows of x (0: rows before the sequence are zero padding)
hipError_t q3_launch_dwconv_ln(const void* x, const float* w /*[7][C], tap-major*/, const float* b, const float* lnw, const float* lnb,
                               void* y, int B, int L, int C, int k, float eps, int dtype, hipStream_t s, int hist = 0);
// Keys / values of the frames before a push, one ring buffer per (slot, layer) inside the slot's state record: frame f at
// ring row f mod hist, a row = [K (nkv*d) | V (nkv*d)] in the context's element type.
struct Q3KvRing { const char* state; long long slot_stride, state_off; const int* slots; int capacity; };
// qkv holds T rows per item (the queries and their keys / values); with hist > 0 the hist frames before them are read from
// `ring`, and pos0[b] is the frame number of item b's query 0 (null: 0)
hipError_t q3_launch_swa_attention(const void* qkv, void* out, const float* cos_t, const float* sin_t, int B, int T,
                                   int nh, int nkv, int d, int window, int dtype, hipStream_t s, int hist = 0,
                                   const int* pos0 = nullptr, int max_pos = 1 << 30, const Q3KvRing* ring = nullptr,
                                   int tuning = 0 /* bit 8: the vector-ALU kernel for every case */);

// one carried tensor of the streaming decode; offsets and strides in bytes, all multiples of 16
struct Q3CarryDesc {
    long long work_off, work_bstride;   // item b's [carried | new] rows start at work + work_off + b * work_bstride
    long long state_off;                // offset of the carried rows inside a slot's state record
    int work_ld, col_off, row_bytes;    // row stride of the work tensor; first byte and width of the carried columns
    int rows, new_rows;                 // carried rows; rows appended by one push
    int ring;                           // 1: the state side is a ring buffer of `rows` rows (frame f at row f mod rows) that is
                                        // only appended to: the work tensor holds the new rows alone
};
struct Q3StageStarts { int count; int rows[16]; };
hipError_t q3_launch_stream_carry(const Q3CarryDesc* descs_dev, int ndesc, void* work, void* state, long long slot_stride,
                                  const int* slots, int S, int capacity, int save, const int* pos0, hipStream_t s);
hipError_t q3_launch_stream_prep(const int* slots, int S, int capacity, int* pos, int n, int max_pos, int* pos0, int* row_start,
                                 const Q3StageStarts& st, int* err_flag, hipStream_t s);
hipError_t q3_launch_stream_reset(const int* slots, int count, int capacity, void* state, long long slot_stride, int* pos,
                                  int* err_flag, hipStream_t s);
hipError_t q3_launch_cast_to_f32(const void* x, float* y, long long n, int dtype, hipStream_t s);
hipError_t q3_launch_cast_from_f32(const float* x, void* y, long long n, int dtype, hipStream_t s);
hipError_t q3_launch_pcm16(const float* x, short* y, long long n, hipStream_t s);
hipError_t q3_launch_snake(const void* x, void* y, const float* a, const float* invb, long long rows, int C, int dtype, hipStream_t s);

// ---- device helpers -------------------------------------------------------------------------------
#ifdef __HIPCC__
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static constexpr int EPC = 4; };   // elements per 16-byte chunk
template <> struct ElemTraits<bf16>  { static constexpr int EPC = 8; };

__device__ __forceinline__ float q3_to_f32(float v) { return v; }
__device__ __forceinline__ float q3_to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T q3_from_f32(float v);
template <> __device__ __forceinline__ float q3_from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 q3_from_f32<bf16>(float v) { return (bf16)v; }

// Accurate sine for the fp32 path (parity bar 1e-3 after ~40 layers); the hardware v_sin for the bf16 path,
// whose storage rounding (2^-9 relative) is far coarser than v_sin's error.
// fp32 sine, branch-free: x = k*pi + r by a three-term Cody-Waite reduction, odd degree-11 polynomial on
// |r| <= pi/2, sign from the parity of k.  Absolute error <= 2.2e-7 for |x| <= 300 (the libm sinf inlined 16 times
// per accumulator tile makes the kernel too large to unroll, which pushes the accumulators to scratch).
__device__ __forceinline__ float q3_sin_f32(float x) {
    const float k = rintf(x * 0.318309886183790671538f);
    float r = fmaf(k, -3.140625f, x);
    r = fmaf(k, -9.67502593994140625e-4f, r);
    r = fmaf(k, -1.509957990978376432e-7f, r);
    const float r2 = r * r;
    float p = -2.50521083854417188e-8f;
    p = fmaf(p, r2, 2.75573192239858907e-6f);
    p = fmaf(p, r2, -1.98412698412698413e-4f);
    p = fmaf(p, r2, 8.33333333333333333e-3f);
    p = fmaf(p, r2, -1.66666666666666667e-1f);
    p = fmaf(r * r2, p, r);
    return ((int)k & 1) ? -p : p;
}
template <typename T> __device__ __forceinline__ float q3_sin(float x);
template <> __device__ __forceinline__ float q3_sin<float>(float x) { return q3_sin_f32(x); }
// bf16 path: the hardware sine takes its argument in REVOLUTIONS; the host folds 1/(2 pi) into the per-channel
// frequency it uploads for bf16 contexts, so q3_snake<bf16> is mul, v_fract, v_sin, mul, fma.  v_sin_f32 is only
// specified for |revolutions| <= 256 (beyond that it returns 0); sine has period one revolution, so the fractional
// part (v_fract_f32: x - floor(x), exact) is taken first and trained alpha * x of any size keeps its sin^2 term.
template <> __device__ __forceinline__ float q3_sin<bf16>(float x) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x)); }

template <typename T> __device__ __forceinline__ float q3_snake(float x, float a, float invb) {
    float s = q3_sin<T>(x * a);
    return x + invb * (s * s);
}
__device__ __forceinline__ float q3_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// ELU (alpha = 1) without the library expm1f (dozens of instructions in an epilogue): exp(x) - 1 on the hardware exponential,
// and the series x + x^2/2 + x^3/6 above -0.05 where that difference would cancel (truncation < 3e-7 absolute there)
__device__ __forceinline__ float q3_elu(float x) {
    const float ser = x * fmaf(x, fmaf(x, 0.16666667f, 0.5f), 1.0f);
    const float ex = __expf(x) - 1.0f;
    return x > 0.f ? x : (x > -0.05f ? ser : ex);
}
__device__ __forceinline__ float q3_silu(float x) { return x / (1.0f + expf(-x)); }
#endif
