// The decode's bandwidth-bound kernels: RVQ gather-and-sum, mean embedding, RMSNorm, depthwise conv + LayerNorm,
// sliding-window attention with fused RoPE, the stream-state copies, and small utilities.
// All tensors are channels-last [batch, time, channel].
#include "common.h"
#include <algorithm>


namespace {

// ------------------------------------------------------------------------------------------------
// RVQ gather-and-sum, fp32, bit-exact.  One thread owns 4 consecutive dims of one frame and walks the
// codebooks of each group in order:  acc = +0.0; acc = acc + row_q  (one IEEE add per codebook, no FMA,
// no reassociation) - the order of `quantized_out = quantized_out + quantized` in the Mimi RVQ decode.
// Tables (16 x 2048 x 256 x 4 B = 32 MiB at full size) exceed LDS; they stay L2 / Infinity-Cache resident
// and each frame's row is read as whole 16-byte pieces by consecutive lanes.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rvq_gather_sum_kernel(const long long* __restrict__ codes, int B, int Q, int T, int bins,
                                                             int D, const float* const* __restrict__ tables,
                                                             const int* __restrict__ group_start /*[G+1]*/, int G,
                                                             float* __restrict__ out, int* err_flag) {
    const int vpr = D >> 2;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long frame = idx / vpr;
    int vec = (int)(idx - frame * vpr);
    if (frame >= (long long)B * T) return;
    int b = (int)(frame / T), t = (int)(frame - (long long)b * T);
    for (int g = 0; g < G; ++g) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = group_start[g]; q < group_start[g + 1]; ++q) {
            long long code = codes[((long long)b * Q + q) * T + t];
            if (code < 0 || code >= bins) { atomicOr(err_flag, 1); code = 0; }
            const float4 row = *(const float4*)(tables[q] + (size_t)code * D + vec * 4);
            acc.x = __fadd_rn(acc.x, row.x); acc.y = __fadd_rn(acc.y, row.y);
            acc.z = __fadd_rn(acc.z, row.z); acc.w = __fadd_rn(acc.w, row.w);
        }
        *(float4*)(out + ((size_t)frame * G + g) * D + vec * 4) = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// RVQ encode (nearest code-book row per layer, residual passed on): the encode direction of the same code books.
// One block takes RVQ_FT frames through every layer of every group: the frames' residuals live in LDS, thread t scores
// code rows t, t + 256, ... against all of them (true squared distances, one fma chain per (frame, code) in dimension
// order; the rows come from a dimension-major copy of the table, so a wave's loads are contiguous), the block reduces to the smallest distance with the lowest index
// on ties (what argmin returns), and the winners' rows are subtracted from the residuals.
// ------------------------------------------------------------------------------------------------
constexpr int RVQ_FT = 8;

__global__ void __launch_bounds__(256) rvq_encode_kernel(const float* __restrict__ emb, long long frames, int T, int in_dim, int Q,
                                                         int bins, int D, const float* const* __restrict__ tables,
                                                         const float4* const* __restrict__ tables_t /*[Q] x [D/4][bins] float4*/,
                                                         const int* __restrict__ group_start, int G,
                                                         const float* __restrict__ in_proj /*[G][D][in_dim] or null*/,
                                                         long long* __restrict__ codes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rvq[];
    float* E = (float*)smem_rvq;                           // [RVQ_FT][in_dim]  the frames' embeddings
    float* R = E + (size_t)RVQ_FT * in_dim;                // [RVQ_FT][D]       residuals of the current group
    float* bd = R + (size_t)RVQ_FT * D;                    // [4 waves][RVQ_FT] per-wave best distance
    int* bi = (int*)(bd + 4 * RVQ_FT);                     // [4 waves][RVQ_FT] ... and its index
    int* win = bi + 4 * RVQ_FT;                            // [RVQ_FT]          the block's winner
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long f0 = (long long)blockIdx.x * RVQ_FT;
    const int nf = (int)min((long long)RVQ_FT, frames - f0);
    for (int i = tid; i < RVQ_FT * in_dim; i += 256) {
        const int f = i / in_dim, e = i - f * in_dim;
        E[i] = f < nf ? emb[(f0 + f) * in_dim + e] : 0.f;
    }
    __syncthreads();
    for (int g = 0; g < G; ++g) {
        // residual <- input_proj_g(embedding), or the embedding itself
        for (int i = tid; i < RVQ_FT * D; i += 256) {
            const int f = i / D, j = i - f * D;
            float a;
            if (in_proj) {
                const float* wr = in_proj + ((size_t)g * D + j) * in_dim;
                a = 0.f;
                for (int e = 0; e < in_dim; ++e) a = fmaf(wr[e], E[f * in_dim + e], a);
            } else {
                a = E[f * in_dim + j];
            }
            R[i] = a;
        }
        __syncthreads();
        for (int q = group_start[g]; q < group_start[g + 1]; ++q) {
            const float* tab = tables[q];
            float best[RVQ_FT];
            int besti[RVQ_FT];
#pragma unroll
            for (int f = 0; f < RVQ_FT; ++f) { best[f] = INFINITY; besti[f] = 0; }      // NaN / Inf frames score no row below INFINITY: row 0, as argmin
            const float4* tt = tables_t[q];                // dimension-major copy: consecutive codes are consecutive float4s
            for (int c = tid; c < bins; c += 256) {
                float acc[RVQ_FT];
#pragma unroll
                for (int f = 0; f < RVQ_FT; ++f) acc[f] = 0.f;
                for (int d0 = 0; d0 < D; d0 += 4) {
                    const float4 e4 = tt[(size_t)(d0 >> 2) * bins + c];
#pragma unroll
                    for (int f = 0; f < RVQ_FT; ++f) {
                        const float4 r4 = *(const float4*)(R + f * D + d0);
                        const float x = r4.x - e4.x, y = r4.y - e4.y, z = r4.z - e4.z, w = r4.w - e4.w;
                        acc[f] = fmaf(w, w, fmaf(z, z, fmaf(y, y, fmaf(x, x, acc[f]))));
                    }
                }
#pragma unroll
                for (int f = 0; f < RVQ_FT; ++f)
                    if (acc[f] < best[f]) { best[f] = acc[f]; besti[f] = c; }       // codes ascend: the first minimum stays
            }
#pragma unroll
            for (int f = 0; f < RVQ_FT; ++f) {
                float v = best[f];
                int ix = besti[f];
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(v, off);
                    const int oi = __shfl_xor(ix, off);
                    if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
                }
                if (lane == 0) { bd[wave * RVQ_FT + f] = v; bi[wave * RVQ_FT + f] = ix; }
            }
            __syncthreads();
            if (tid < RVQ_FT) {
                float v = bd[tid];
                int ix = bi[tid];
                for (int w2 = 1; w2 < 4; ++w2) {
                    const float ov = bd[w2 * RVQ_FT + tid];
                    const int oi = bi[w2 * RVQ_FT + tid];
                    if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
                }
                ix = min(max(ix, 0), bins - 1);            // the residual update indexes the table with it: never out of range
                win[tid] = ix;
                if (tid < nf) {
                    const long long fr = f0 + tid, b = fr / T, t = fr - b * T;
                    codes[(b * Q + q) * T + t] = ix;
                }
            }
            __syncthreads();
            for (int i = tid; i < RVQ_FT * D; i += 256) {
                const int f = i / D, j = i - f * D;
                R[i] = R[i] - tab[(size_t)win[f] * D + j];
            }
            __syncthreads();
        }
    }
}

// The same search on the fp32 matrix op, for code books of D dimensions and a multiple of 128 rows (the full-size 2048 x 256).
// The nearest row minimises |c|^2 - 2 c.r (|r|^2 is common to all rows), and c.r for 32 rows x 32 frames is a 32 x 32 tile of
// TABLE . RESIDUAL^T: v_mfma_f32_32x32x2_f32 with A = table[row][k] (from the dimension-major copy: one 16-byte load feeds two
// k-steps), B = residual[frame][k] (held in registers for the whole layer, D / 2 per lane), |c|^2 from a table built with the
// dimension-major copy.  Rows are accumulator rows, so a lane keeps the running minimum of ONE frame: compare in registers,
// one exchange with lane ^ 32, then across the four waves, which split the rows.  Lowest index on ties, as argmin.
constexpr int RVQM_FT = 32;

template <int D>
__global__ void __launch_bounds__(256) rvq_encode_mfma_kernel(const float* __restrict__ emb, long long frames, int T, int in_dim, int Q,
                                                              int bins, const float* const* __restrict__ tables,
                                                              const float4* const* __restrict__ tables_t /*[Q] x ([D/4][bins] float4, then [bins] |c|^2)*/,
                                                              const int* __restrict__ group_start, int G,
                                                              const float* __restrict__ in_proj, long long* __restrict__ codes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rvq[];
    constexpr int RLD = D + 1;                             // residual row stride: a wave's 32 frames on 32 banks
    const int ELD = in_dim + 4;                            // embedding row stride (16-byte rows, spread over the banks)
    float* E = (float*)smem_rvq;                           // [32][in_dim + 4]
    float* R = E + (size_t)RVQM_FT * ELD;                  // [32][D + 1]
    float* bd = R + (size_t)RVQM_FT * RLD;                 // [4 waves][32]
    int* bi = (int*)(bd + 4 * RVQM_FT);
    int* win = bi + 4 * RVQM_FT;                           // [32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr_l = lane & 31, hh = lane >> 5;
    const long long f0 = (long long)blockIdx.x * RVQM_FT;
    const int nf = (int)min((long long)RVQM_FT, frames - f0);
    for (int i = tid; i < RVQM_FT * in_dim; i += 256) {
        const int f = i / in_dim, e = i - f * in_dim;
        E[f * ELD + e] = f < nf ? emb[(f0 + f) * in_dim + e] : 0.f;
    }
    __syncthreads();
    const int bins_w = bins >> 2;                          // rows per wave
    for (int g = 0; g < G; ++g) {
        // residual <- input_proj_g(embedding) as W_g . E^T on the matrix op (rows = projection outputs, columns = frames; k-step
        // (q, c) of lane half hh is input dimension 8 q + 4 hh + c for both operands), or the embedding itself
        if (in_proj) {
#pragma unroll 1
            for (int t = 0; t < D / 128; ++t) {
                const int j0 = wave * (D / 4) + 32 * t;
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                const float* wrow = in_proj + ((size_t)g * D + j0 + fr_l) * in_dim + 4 * hh;
                const float* erow = E + fr_l * ELD + 4 * hh;
#pragma unroll 4
                for (int q8 = 0; q8 < (in_dim >> 3); ++q8) {
                    const float4 a4 = *(const float4*)(wrow + 8 * q8);
                    const float4 b4 = *(const float4*)(erow + 8 * q8);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) R[fr_l * RLD + j0 + (i & 3) + 8 * (i >> 2) + 4 * hh] = acc[i];
            }
        } else {
            for (int i = tid; i < RVQM_FT * D; i += 256) {
                const int f = i / D, j = i - f * D;
                R[f * RLD + j] = E[f * ELD + j];
            }
        }
        __syncthreads();
        for (int q = group_start[g]; q < group_start[g + 1]; ++q) {
            const float* tab = tables[q];
            const float4* tt = tables_t[q];
            const float* norms = (const float*)(tt + (size_t)(D >> 2) * bins);
            // k-step s of lane half hh is dimension 8 (s / 4) + 4 hh + (s % 4): a lane then feeds four consecutive k-steps from
            // ONE 16-byte piece of the dimension-major table and the two halves load different pieces
            float breg[D / 2];                             // this lane's frame at those dimensions
#pragma unroll
            for (int s2 = 0; s2 < D / 2; ++s2) breg[s2] = R[fr_l * RLD + 8 * (s2 >> 2) + 4 * hh + (s2 & 3)];
            float best = INFINITY;
            int besti = 0;                                 // a NaN / Inf frame scores no row below INFINITY and keeps row 0 (what argmin returns)
            // The table pieces of a tile are fetched eight at a time (32 k-steps), one group ahead of the group being multiplied,
            // also across tiles; every index below is a compile-time constant so that breg stays in registers.
            constexpr int NG = D / 64;                     // groups of eight 16-byte pieces (32 k-steps) per tile: D / 2 k-steps in all
            float4 abuf[2][8];
            const float4* ta0 = tt + (size_t)hh * bins + wave * bins_w + fr_l;      // piece 2 q8 + hh of row (tile base) + (lane & 31)
            auto fetch = [&](float4* dst, const float4* ta, int grp) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[j] = ta[(size_t)(2 * (8 * grp + j)) * bins];
            };
            fetch(abuf[0], ta0, 0);
            for (int bt = 0; bt < bins_w; bt += 32) {
                const int bin0 = wave * bins_w + bt;
                const float4* ta = ta0 + bt;
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int grp = 0; grp < NG; ++grp) {
                    if (grp + 1 < NG) fetch(abuf[(grp + 1) & 1], ta, grp + 1);
                    else if (bt + 32 < bins_w) fetch(abuf[(grp + 1) & 1], ta + 32, 0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4 a4 = abuf[grp & 1][j];
                        const int s0 = 4 * (8 * grp + j);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, breg[s0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, breg[s0 + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, breg[s0 + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, breg[s0 + 3], acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {           // accumulator i is row (i & 3) + 8 (i >> 2) + 4 hh of the tile
                    const float4 n4 = *(const float4*)(norms + bin0 + 8 * g4 + 4 * hh);
                    const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = fmaf(-2.0f, acc[4 * g4 + j], nn[j]);
                        const int idx = bin0 + 8 * g4 + 4 * hh + j;
                        if (v < best || (v == best && idx < besti)) { best = v; besti = idx; }
                    }
                }
            }
            {
                const float ov = __shfl_xor(best, 32);
                const int oi = __shfl_xor(besti, 32);
                if (ov < best || (ov == best && oi < besti)) { best = ov; besti = oi; }
            }
            if (lane < 32) { bd[wave * RVQM_FT + lane] = best; bi[wave * RVQM_FT + lane] = besti; }
            __syncthreads();
            if (tid < RVQM_FT) {
                float v = bd[tid];
                int ix = bi[tid];
                for (int w2 = 1; w2 < 4; ++w2) {
                    const float ov = bd[w2 * RVQM_FT + tid];
                    const int oi = bi[w2 * RVQM_FT + tid];
                    if (ov < v || (ov == v && oi < ix)) { v = ov; ix = oi; }
                }
                ix = min(max(ix, 0), bins - 1);            // the residual update indexes the table with it: never out of range
                win[tid] = ix;
                if (tid < nf) {
                    const long long fr = f0 + tid, b = fr / T, t = fr - b * T;
                    codes[(b * Q + q) * T + t] = ix;
                }
            }
            __syncthreads();
            for (int i = tid; i < RVQM_FT * D; i += 256) {
                const int f = i / D, j = i - f * D;
                R[f * RLD + j] = R[f * RLD + j] - tab[(size_t)win[f] * D + j];
            }
            __syncthreads();
        }
    }
}

// Embedding(Q*bins, H)(codes + q*bins).mean(1): ordered sum then divide by Q.
template <typename T>
__global__ void __launch_bounds__(256) mean_embedding_kernel(const long long* __restrict__ codes, int B, int Q, int Tn, int bins,
                                                             int H, const float* __restrict__ table, T* __restrict__ out,
                                                             int* err_flag) {
    const int vpr = H >> 2;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long frame = idx / vpr;
    int vec = (int)(idx - frame * vpr);
    if (frame >= (long long)B * Tn) return;
    int b = (int)(frame / Tn), t = (int)(frame - (long long)b * Tn);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < Q; ++q) {
        long long code = codes[((long long)b * Q + q) * Tn + t];
        if (code < 0 || code >= bins) { atomicOr(err_flag, 1); code = 0; }
        const float4 row = *(const float4*)(table + ((size_t)q * bins + code) * H + vec * 4);
        acc.x += row.x; acc.y += row.y; acc.z += row.z; acc.w += row.w;
    }
    const float fq = (float)Q;
    T* o = out + (size_t)frame * H + vec * 4;
    o[0] = q3_from_f32<T>(acc.x / fq); o[1] = q3_from_f32<T>(acc.y / fq);
    o[2] = q3_from_f32<T>(acc.z / fq); o[3] = q3_from_f32<T>(acc.w / fq);
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// RMSNorm: y = w * (x * rsqrt(mean(x^2) + eps)); one wave per row, fp32 math.
template <typename T>
__global__ void __launch_bounds__(256) rmsnorm_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                      long long rows, int H, float eps) {
    long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const T* xr = x + row * H;
    T* yr = y + row * H;
    constexpr int EPV = 16 / (int)sizeof(T);
    // rows of up to 64 x EPV x 4 elements in 16-byte pieces: the row is read ONCE, all of a lane's pieces requested before the
    // first is used (a frame-rate push runs 17 of these on 256 rows each: what they cost is their chain of memory round trips)
    if (H % EPV == 0 && H <= 64 * EPV * 4) {
        const int np = H / EPV;                                 // pieces per row
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pc = lane + 64 * k;
            v[k] = pc < np ? *(const uint4*)(xr + pc * EPV) : make_uint4(0, 0, 0, 0);
        }
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < EPV; ++e) { const float f = q3_to_f32(((const T*)&v[k])[e]); ss += f * f; }
        ss = wave_sum(ss);
        const float r = rsqrtf(ss / (float)H + eps);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pc = lane + 64 * k;
            if (pc < np) {
                uint4 o;
#pragma unroll
                for (int e = 0; e < EPV; e += 4) {
                    const float4 w4 = *(const float4*)(w + pc * EPV + e);
                    ((T*)&o)[e] = q3_from_f32<T>(w4.x * (q3_to_f32(((const T*)&v[k])[e]) * r));
                    ((T*)&o)[e + 1] = q3_from_f32<T>(w4.y * (q3_to_f32(((const T*)&v[k])[e + 1]) * r));
                    ((T*)&o)[e + 2] = q3_from_f32<T>(w4.z * (q3_to_f32(((const T*)&v[k])[e + 2]) * r));
                    ((T*)&o)[e + 3] = q3_from_f32<T>(w4.w * (q3_to_f32(((const T*)&v[k])[e + 3]) * r));
                }
                *(uint4*)(yr + pc * EPV) = o;
            }
        }
        return;
    }
    float ss = 0.f;
    for (int i = lane; i < H; i += 64) { float v = q3_to_f32(xr[i]); ss += v * v; }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)H + eps);
    for (int i = lane; i < H; i += 64) yr[i] = q3_from_f32<T>(w[i] * (q3_to_f32(xr[i]) * r));
}

// Depthwise causal conv (kernel k, weights [k][C]) followed by LayerNorm over C.  One block per (b, t) row.
template <typename T, int CPT>
__global__ void __launch_bounds__(256) dwconv_ln_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ lnw,
                                                        const float* __restrict__ lnb, T* __restrict__ y, int L, int C, int k,
                                                        float eps, int hist) {
    __shared__ float red[8];
    const long long row = blockIdx.x;            // b * L + t
    const int t = (int)(row % L);
    const int tid = threadIdx.x;
    // x holds `hist` carried rows in front of each item's L rows (hist = 0: rows before the sequence are zero padding)
    const T* xb = x + ((row / L) * (long long)(hist + L) + hist) * C;
    float v[CPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        int c = tid + i * 256;
        float a = 0.f;
        if (c < C) {
            a = bias[c];
            for (int j = 0; j < k; ++j) {
                int tt = t - (k - 1) + j;
                if (tt >= -hist) a += w[j * C + c] * q3_to_f32(xb[(long long)tt * C + c]);
            }
            s += a;
        }
        v[i] = a;
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        int c = tid + i * 256;
        if (c < C) { float d = v[i] - mean; q += d * d; }
    }
    q = wave_sum(q);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        int c = tid + i * 256;
        if (c < C) y[row * C + c] = q3_from_f32<T>((v[i] - mean) * rstd * lnw[c] + lnb[c]);
    }
}

// ------------------------------------------------------------------------------------------------
// Sliding-window causal attention with rotary embedding fused into the Q/K staging.
// qkv rows are [q (nh*d) | k (nkv*d) | v (nkv*d)].  One block = (32 queries, one head, one batch item);
// keys [max(0, q0 - W + 1), q0 + 31] and their values sit in LDS as fp32; one wave walks 8 queries:
// lane = key for the scores (softmax by wave reductions, fp32), lane = dim for P.V.
// ------------------------------------------------------------------------------------------------
constexpr int ATT_QT = 32;    // queries per block; the launcher halves it (down to 8) when a long window's keys would not fit the LDS
constexpr int ATT_MAXC = 4;   // window <= 64 * ATT_MAXC

template <typename T>
__global__ void __launch_bounds__(256) swa_attention_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                            const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                            int Tn, int nh, int nkv, int d, int window, float scale, int hist,
                                                            const int* __restrict__ pos0, int max_pos, Q3KvRing ring, int qt) {
    // An item has `hist` carried rows then its new rows (Tn rows in all); only the new rows are queries.  Row r is frame
    // pos0[b] - hist + r of its stream; rows of negative frame number do not exist yet.  The new rows are rows of `qkv`
    // (Tn - hist per item); the carried rows' keys and values are read where they live, in the slot's ring buffer inside
    // its state record: frame f sits at ring row f mod hist, as [K (nkv*d) | V (nkv*d)].
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_att[];
    const int q0 = hist + blockIdx.x * qt;
    const int h = blockIdx.y, b = blockIdx.z;
    const int shift = pos0 ? pos0[b] - hist : 0;
    const int r_min = max(0, -shift);
    const int hk = h / (nh / nkv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q_hi = min(q0 + qt, Tn) - 1;
    const int k_lo = max(r_min, q0 - window + 1);
    const int nk = q_hi - k_lo + 1;
    const int nkmax = qt + window - 1;
    const int ds = d + 4;                          // padded row stride: rows stay 16-byte aligned, and the 16 lanes a ds_read_b128
                                                   // serves together (lane = key) cover all 64 banks
    const int half = d >> 1;
    float* Ks = (float*)smem_att;
    float* Vs = Ks + (size_t)nkmax * ds;
    float* Qs = Vs + (size_t)nkmax * ds;
    float* Ps = Qs + (size_t)qt * d + (size_t)wave * 4 * 64 * ATT_MAXC;     // four probability rows per wave
    const int rs = (nh + 2 * nkv) * d;             // qkv row stride
    const T* base = qkv + ((long long)b * (Tn - hist) - hist) * (long long)rs;       // row r >= hist of the item is qkv row r - hist
    const T* ring_base = nullptr;
    if (hist > 0) {
        const int slot = min(max(ring.slots[b], 0), ring.capacity - 1);
        ring_base = (const T*)(ring.state + (long long)slot * ring.slot_stride + ring.state_off);
    }
    // keys of row `pos`: a new row, or a carried one in the ring
    auto k_row = [&](int pos) -> const T* {
        if (pos >= hist) return base + (size_t)pos * rs + (nh + hk) * d;
        return ring_base + (size_t)((pos + shift) % hist) * (2 * nkv * d) + hk * d;
    };

    // Staging: one thread takes EPV consecutive dimensions of both rotation halves of a row (16-byte global loads) when
    // the head size allows, else one dimension pair.  Either way an element is rotated by the same two expressions.
    constexpr int EPV = 16 / (int)sizeof(T);
    const int nq = q_hi - q0 + 1;
    auto rot_lo = [](float x1, float x2, float c, float s) { return x1 * c - x2 * s; };
    auto rot_hi = [](float x1, float x2, float c, float s) { return x2 * c + x1 * s; };
    if (half % EPV == 0) {
        const int gpr = half / EPV;                        // groups per half row
        for (int i = tid; i < nk * gpr; i += 256) {
            const int r = i / gpr, e = (i - r * gpr) * EPV;
            const int pos = k_lo + r;
            const T* kr = k_row(pos);
            const T* vr = kr + nkv * d;
            const int ap = min(pos + shift, max_pos - 1);
            const uint4 klo = *(const uint4*)(kr + e), khi = *(const uint4*)(kr + e + half);
            const uint4 vlo = *(const uint4*)(vr + e), vhi = *(const uint4*)(vr + e + half);
            const float* cr = cos_t + (size_t)ap * half + e;
            const float* sr = sin_t + (size_t)ap * half + e;
#pragma unroll
            for (int u = 0; u < EPV; ++u) {
                const float x1 = q3_to_f32(((const T*)&klo)[u]), x2 = q3_to_f32(((const T*)&khi)[u]);
                Ks[r * ds + e + u] = rot_lo(x1, x2, cr[u], sr[u]);
                Ks[r * ds + e + u + half] = rot_hi(x1, x2, cr[u], sr[u]);
                Vs[r * ds + e + u] = q3_to_f32(((const T*)&vlo)[u]);
                Vs[r * ds + e + u + half] = q3_to_f32(((const T*)&vhi)[u]);
            }
        }
        for (int i = tid; i < nq * gpr; i += 256) {
            const int r = i / gpr, e = (i - r * gpr) * EPV;
            const int pos = q0 + r;
            const T* qr = base + (size_t)pos * rs + h * d;
            const int ap = min(pos + shift, max_pos - 1);
            const uint4 qlo = *(const uint4*)(qr + e), qhi = *(const uint4*)(qr + e + half);
            const float* cr = cos_t + (size_t)ap * half + e;
            const float* sr = sin_t + (size_t)ap * half + e;
#pragma unroll
            for (int u = 0; u < EPV; ++u) {
                const float x1 = q3_to_f32(((const T*)&qlo)[u]), x2 = q3_to_f32(((const T*)&qhi)[u]);
                Qs[r * d + e + u] = rot_lo(x1, x2, cr[u], sr[u]);
                Qs[r * d + e + u + half] = rot_hi(x1, x2, cr[u], sr[u]);
            }
        }
    } else {
        for (int i = tid; i < nk * half; i += 256) {
            int r = i / half, e = i - r * half;
            int pos = k_lo + r;
            const T* kr = k_row(pos);
            const T* vr = kr + nkv * d;
            const int ap = min(pos + shift, max_pos - 1);
            float c = cos_t[(size_t)ap * half + e], s = sin_t[(size_t)ap * half + e];
            float x1 = q3_to_f32(kr[e]), x2 = q3_to_f32(kr[e + half]);
            Ks[r * ds + e] = rot_lo(x1, x2, c, s);
            Ks[r * ds + e + half] = rot_hi(x1, x2, c, s);
            Vs[r * ds + e] = q3_to_f32(vr[e]);
            Vs[r * ds + e + half] = q3_to_f32(vr[e + half]);
        }
        for (int i = tid; i < nq * half; i += 256) {
            int r = i / half, e = i - r * half;
            int pos = q0 + r;
            const T* qr = base + (size_t)pos * rs + h * d;
            const int ap = min(pos + shift, max_pos - 1);
            float c = cos_t[(size_t)ap * half + e], s = sin_t[(size_t)ap * half + e];
            float x1 = q3_to_f32(qr[e]), x2 = q3_to_f32(qr[e + half]);
            Qs[r * d + e] = rot_lo(x1, x2, c, s);
            Qs[r * d + e + half] = rot_hi(x1, x2, c, s);
        }
    }
    __syncthreads();

    // A wave walks its queries four at a time: one key (lane) is scored against four consecutive queries, and one
    // value element is weighted into four outputs, so every K / V element read from LDS feeds four FMAs.
    constexpr int QG = 4;
    const int nchunk = (window + QG - 1 + 63) >> 6;            // the four windows together span window + 3 keys
    for (int g0 = wave * QG; g0 < nq; g0 += 4 * QG) {
        const int i0 = q0 + g0;                                // first query row of the group
        const int jlo = max(r_min, i0 - window + 1);           // first key any of the four may see
        float sc[QG][ATT_MAXC];
        float mx[QG];
#pragma unroll
        for (int q = 0; q < QG; ++q) mx[q] = -INFINITY;
#pragma unroll
        for (int c = 0; c < ATT_MAXC; ++c) {
#pragma unroll
            for (int q = 0; q < QG; ++q) sc[q][c] = -INFINITY;
            if (c < nchunk) {
                const int j = jlo + lane + 64 * c;
                if (j <= min(i0 + QG - 1, q_hi)) {
                    const float* kr = Ks + (size_t)(j - k_lo) * ds;
                    float a[QG];
#pragma unroll
                    for (int q = 0; q < QG; ++q) a[q] = 0.f;
                    // four dimensions per LDS read (one b128 of the key row, one broadcast b128 per query); one explicit fma chain in
                    // dimension order, so the result does not depend on how the compiler would contract products and sums
                    for (int e = 0; e < d; e += 4) {
                        const float4 kv = *(const float4*)(kr + e);
#pragma unroll
                        for (int q = 0; q < QG; ++q) {
                            const float4 qv = *(const float4*)(Qs + min(g0 + q, nq - 1) * d + e);
                            a[q] = fmaf(qv.w, kv.w, fmaf(qv.z, kv.z, fmaf(qv.y, kv.y, fmaf(qv.x, kv.x, a[q]))));
                        }
                    }
#pragma unroll
                    for (int q = 0; q < QG; ++q) {
                        const int i = i0 + q;
                        if (g0 + q < nq && j <= i && j >= max(r_min, i - window + 1)) sc[q][c] = a[q] * scale;
                    }
                }
#pragma unroll
                for (int q = 0; q < QG; ++q) mx[q] = fmaxf(mx[q], sc[q][c]);
            }
        }
        float inv[QG];
#pragma unroll
        for (int q = 0; q < QG; ++q) {
            const float m = wave_max(mx[q]);
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < ATT_MAXC; ++c) {
                float pv = 0.f;
                if (c < nchunk && sc[q][c] != -INFINITY) pv = expf(sc[q][c] - m);
                sc[q][c] = pv;
                sum += pv;
            }
            sum = wave_sum(sum);
            inv[q] = sum > 0.f ? 1.0f / sum : 0.f;             // a query past the end of the tile has no keys
        }
#pragma unroll
        for (int q = 0; q < QG; ++q)
#pragma unroll
            for (int c = 0; c < ATT_MAXC; ++c)
                if (c < nchunk) Ps[q * 64 * ATT_MAXC + lane + 64 * c] = sc[q][c] * inv[q];
        // Ps is private to this wave; LDS ops of one wave complete in order, so no workgroup barrier is needed.
        __builtin_amdgcn_wave_barrier();
        const int cnt = min(i0 + QG - 1, q_hi) - jlo + 1;
        for (int e = lane; e < d; e += 64) {
            float a[QG];
#pragma unroll
            for (int q = 0; q < QG; ++q) a[q] = 0.f;
            const float* vr = Vs + (size_t)(jlo - k_lo) * ds + e;
            int j = 0;
            for (; j + 4 <= cnt; j += 4) {                     // four keys per broadcast read of the probabilities, summed in order
                const float v0 = vr[(size_t)j * ds], v1 = vr[(size_t)(j + 1) * ds], v2 = vr[(size_t)(j + 2) * ds], v3 = vr[(size_t)(j + 3) * ds];
#pragma unroll
                for (int q = 0; q < QG; ++q) {
                    const float4 pq = *(const float4*)(Ps + q * 64 * ATT_MAXC + j);
                    a[q] = fmaf(pq.w, v3, fmaf(pq.z, v2, fmaf(pq.y, v1, fmaf(pq.x, v0, a[q]))));
                }
            }
            for (; j < cnt; ++j) {
                const float vv = vr[(size_t)j * ds];
#pragma unroll
                for (int q = 0; q < QG; ++q) a[q] = fmaf(Ps[q * 64 * ATT_MAXC + j], vv, a[q]);
            }
#pragma unroll
            for (int q = 0; q < QG; ++q)
                if (g0 + q < nq) out[((size_t)b * (Tn - hist) + (i0 + q - hist)) * (nh * d) + h * d + e] = q3_from_f32<T>(a[q]);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the encode side's small kernels ---------------------------------------------------------------
// First conv of the audio encoder: one input channel, k taps, causal zero padding.  A thread produces eight consecutive
// channels of one sample (a 16-byte store in bf16), and optionally their ELU into a second tensor (the input of the first
// residual block's conv, whose skip path reads the raw one); the F x k weights and the bias sit in LDS.
template <typename T>
__global__ void __launch_bounds__(256) audio_conv_kernel(const float* __restrict__ audio, const float* __restrict__ w, const float* __restrict__ bias,
                                                         T* __restrict__ y, T* __restrict__ y2 /*ELU(y), or null*/, long long y2_bstride, int L, int F, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_enc[];
    float* ws = (float*)smem_enc;                    // [F][k] then [F]
    for (int i = threadIdx.x; i < F * k; i += 256) ws[i] = w[i];
    for (int i = threadIdx.x; i < F; i += 256) ws[F * k + i] = bias[i];
    __syncthreads();
    const long long b = blockIdx.y;
    const int g8 = F >> 3;                           // channel groups per sample
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)L * g8) return;
    const int t = (int)(i / g8), f0 = (int)(i - (long long)t * g8) * 8;
    const float* a = audio + b * L;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = ws[F * k + f0 + u];
    for (int j = 0; j < k; ++j) {
        const int tt = t - (k - 1) + j;
        if (tt < 0) continue;
        const float av = a[tt];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = fmaf(ws[(f0 + u) * k + j], av, acc[u]);
    }
    constexpr int V = (int)sizeof(T) * 8 / 16;       // 16-byte stores per eight channels
    union { T e[8]; uint4 v[V]; } o, o2;
#pragma unroll
    for (int u = 0; u < 8; ++u) { o.e[u] = q3_from_f32<T>(acc[u]); o2.e[u] = q3_from_f32<T>(q3_elu(acc[u])); }
    uint4* dst = (uint4*)(y + (b * L + t) * F + f0);
#pragma unroll
    for (int v = 0; v < V; ++v) dst[v] = o.v[v];
    if (y2) {
        uint4* dst2 = (uint4*)(y2 + b * y2_bstride + (long long)t * F + f0);
#pragma unroll
        for (int v = 0; v < V; ++v) dst2[v] = o2.v[v];
    }
}

// LayerNorm with bias over rows of H elements: one wave per row, statistics in fp32 (mean, then the centred second moment).
template <typename T>
__global__ void __launch_bounds__(256) layernorm_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ y, long long rows, int H, float eps) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const T* xr = x + row * H;
    float s = 0.f;
    for (int c = lane; c < H; c += 64) s += q3_to_f32(xr[c]);
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
    for (int c = lane; c < H; c += 64) { const float d = q3_to_f32(xr[c]) - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
    for (int c = lane; c < H; c += 64) y[row * H + c] = q3_from_f32<T>((q3_to_f32(xr[c]) - mean) * rstd * w[c] + bias[c]);
}

// y[b][i] = x[b][clamp(i - left, 0, L - 1)] for i < Lout: replicate padding of a [B][L][C] tensor on both sides
template <typename T>
__global__ void __launch_bounds__(256) replicate_pad_kernel(const T* __restrict__ x, T* __restrict__ y, int L, int C, int left, int Lout) {
    const long long b = blockIdx.y;
    const long long n = (long long)Lout * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / C), c = (int)(i - (long long)r * C);
        const int src = min(max(r - left, 0), L - 1);
        y[b * n + i] = x[(b * L + src) * C + c];
    }
}

// The one-channel causal conv whose per-tap products P[b][m][t] = w_t . z[b][m] a fused unit left behind (8 floats per row):
// y[b][m] = act(bias + sum_t P[b][m - (taps - 1) + t][t]).  A block stages its 256 rows and the 8 before them in LDS (rows of
// 9 floats: the diagonal reads of consecutive lanes then fall on distinct banks).
// spans (or null): item b's samples m_first .. m_first + spans[2b + 1] - 1 go to y + spans[2b] and nothing else is written
__global__ void __launch_bounds__(256) tap_sum_kernel(const float* __restrict__ P, const float* __restrict__ bias, float* __restrict__ y,
                                                      int M, int taps, int act, int m_first, const long long* __restrict__ spans) {
    __shared__ float sm[(256 + 8) * 9];
    const long long b = blockIdx.y;
    const int m0 = m_first + blockIdx.x * 256, tid = threadIdx.x;
    const float4* Pb = (const float4*)(P + b * (long long)M * 8);
    for (int i = tid; i < (256 + 8) * 2; i += 256) {
        const int lr = i >> 1, row = m0 - 8 + lr;
        const float4 v = (row >= 0 && row < M) ? Pb[(long long)row * 2 + (i & 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = sm + lr * 9 + 4 * (i & 1);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int m = m0 + tid;
    if (m >= M) return;
    float a = bias ? bias[0] : 0.f;
    for (int t = 0; t < taps; ++t) a += sm[(tid + 8 - (taps - 1) + t) * 9 + t];
    if (act == Q3_ACT_CLAMP) a = fminf(1.0f, fmaxf(-1.0f, a));
    if (spans) {
        if (m - m_first < spans[2 * b + 1]) y[spans[2 * b] + (m - m_first)] = a;
    } else {
        y[b * (long long)M + m] = a;
    }
}

// out[spans[2b] + i] = src[b * M + m_first + i] for i < spans[2b + 1]: a forward's kept samples to their places
__global__ void __launch_bounds__(256) span_copy_kernel(const float* __restrict__ src, int M, int m_first, const long long* __restrict__ spans,
                                                        float* __restrict__ out) {
    const long long b = blockIdx.y;
    const long long n = spans[2 * b + 1], dst = spans[2 * b];
    const float* s = src + b * (long long)M + m_first;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n && i < M - m_first; i += (long long)gridDim.x * 256) out[dst + i] = s[i];
}

// The bulk decode's attention in bf16 on the matrix cores (head size 64, window <= 97, no carried rows).  One block = 128
// consecutive queries of one (item, head), one wave per 32 of them.  The block stages the 224 key rows its queries can see
// (rotated, bf16) and their value rows in LDS once.  A wave then computes S^T = K.Q^T for its four 32-key tiles: with keys
// as accumulator ROWS a lane holds one query's scores, so mask, max, exp and sum are per-lane loops plus one exchange with
// lane ^ 32, and the bf16 probabilities are, register for register, the B operand of O^T = V^T.P^T (no lane movement, no
// LDS).  Its A operand, V^T, is read from the row-major value image by transposed LDS reads (four keys x one dimension per
// lane and read).  Rotated queries and keys and the probabilities are rounded to bf16 (operands of the matrix cores): the
// same roundings a bf16 run of the reference's attention makes.
constexpr int AM_QB = 128, AM_D = 64;     // NKT key tiles of 32 per wave: windows up to 32 (NKT - 1) + 1 (4: the decoder's 72; 9: the encoder's 250)
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned am_pack2(float a, float b) {
    const bf16 x = (bf16)a, y = (bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
// rotary embedding of eight (e, e + half) pairs: lo holds dimensions e .. e+7, hi their partners
__device__ __forceinline__ void am_rope8(const uint4& lo, const uint4& hi, const float* __restrict__ cr, const float* __restrict__ sr,
                                         uint4& olo, uint4& ohi) {
    const float4 c0 = *(const float4*)cr, c1 = *(const float4*)(cr + 4), s0 = *(const float4*)sr, s1 = *(const float4*)(sr + 4);
    const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    float l[8], hgh[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float x1 = (float)((const bf16*)&lo)[u], x2 = (float)((const bf16*)&hi)[u];
        l[u] = x1 * c[u] - x2 * sn[u];
        hgh[u] = x2 * c[u] + x1 * sn[u];
    }
    olo = make_uint4(am_pack2(l[0], l[1]), am_pack2(l[2], l[3]), am_pack2(l[4], l[5]), am_pack2(l[6], l[7]));
    ohi = make_uint4(am_pack2(hgh[0], hgh[1]), am_pack2(hgh[2], hgh[3]), am_pack2(hgh[4], hgh[5]), am_pack2(hgh[6], hgh[7]));
}
// LDS images, 128-byte rows of eight 16-byte chunks.  Keys are read by rows (16 bytes per lane, 32 consecutive rows per
// half wave): chunk ^ ((row >> 1) & 7).  Values are read transposed (a half wave takes 4 rows x 64 bytes): chunk ^ 4 on rows
// 2, 3 of every four puts the 16 pieces on distinct banks.
__device__ __forceinline__ int am_off_k(int row, int ch) { return 128 * row + 16 * (ch ^ ((row >> 1) & 7)); }
__device__ __forceinline__ int am_off_v(int row, int ch) { return 128 * row + 16 * (ch ^ (((row >> 1) & 1) << 2)); }

template <int NKT>
__global__ void __launch_bounds__(256) swa_attention_mfma_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                                 const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                                 int Tn, int nh, int nkv, int window, float scale, int max_pos) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_att[];
    unsigned char* Ki = smem_att;
    constexpr int AM_KR = AM_QB + 32 * (NKT - 1);       // key rows a block stages
    unsigned char* Vi = smem_att + AM_KR * 128;
    const int q0 = blockIdx.x * AM_QB, h = blockIdx.y, b = blockIdx.z;
    const int hk = h / (nh / nkv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rs = (nh + 2 * nkv) * AM_D;
    const bf16* base = qkv + (size_t)b * Tn * rs;
    const int k_lo = q0 - 32 * (NKT - 1);                       // key of image row 0; wave w reads rows 32 w .. 32 (w + NKT) - 1
    const int waves_on = min(4, (Tn - q0 + 31) >> 5);
    const int rows = 32 * (NKT - 1) + 32 * waves_on;
    for (int i = tid; i < rows * 4; i += 256) {                 // keys: one (row, 8-dimension group and its partner) per thread
        const int r = i >> 2, g = i & 3, pos = k_lo + r;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (pos >= 0 && pos < Tn) {
            const bf16* kr = base + (size_t)pos * rs + (nh + hk) * AM_D + 8 * g;
            const int ap = min(pos, max_pos - 1);
            am_rope8(*(const uint4*)kr, *(const uint4*)(kr + 32), cos_t + (size_t)ap * 32 + 8 * g, sin_t + (size_t)ap * 32 + 8 * g, lo, hi);
        }
        *(uint4*)(Ki + am_off_k(r, g)) = lo;
        *(uint4*)(Ki + am_off_k(r, 4 + g)) = hi;
    }
    for (int i = tid; i < rows * 8; i += 256) {                 // values: rows past the ends are zeros (a probability of 0 times
        const int r = i >> 3, ch = i & 7, pos = k_lo + r;       // whatever the LDS held would still be NaN for NaN bits)
        uint4 v = make_uint4(0, 0, 0, 0);
        if (pos >= 0 && pos < Tn) v = *(const uint4*)(base + (size_t)pos * rs + (nh + nkv + hk) * AM_D + 8 * ch);
        *(uint4*)(Vi + am_off_v(r, ch)) = v;
    }
    __syncthreads();
    const int qw0 = q0 + 32 * wave;
    if (qw0 >= Tn) return;                                      // whole waves only: the transposed reads need every lane
    const int r = lane & 31, hh = lane >> 5;
    const int qi = qw0 + r, qc = min(qi, Tn - 1);
    // the query fragments straight from memory: k-step s of lane (r, hh) is dimensions 16 s + 8 hh .. + 7 of query r, and
    // steps s, s + 2 hold each other's rotation partners
    bf16x8 qf[4];
    {
        const bf16* qr = base + (size_t)qc * rs + h * AM_D + 8 * hh;
        const int ap = min(qc, max_pos - 1);
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            uint4 lo, hi;
            am_rope8(*(const uint4*)(qr + 16 * sp), *(const uint4*)(qr + 16 * sp + 32), cos_t + (size_t)ap * 32 + 16 * sp + 8 * hh,
                     sin_t + (size_t)ap * 32 + 16 * sp + 8 * hh, lo, hi);
            qf[sp] = __builtin_bit_cast(bf16x8, lo);
            qf[sp + 2] = __builtin_bit_cast(bf16x8, hi);
        }
    }
    const int krow0 = 32 * wave;
    f32x16 sacc[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[t][i] = 0.f;
        const int row = krow0 + 32 * t + r;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 a = *(const bf16x8*)(Ki + am_off_k(row, 2 * s + hh));
            sacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], sacc[t], 0, 0, 0);
        }
    }
    // register i of tile t is key k_lo + krow0 + 32 t + 8 (i >> 2) + 4 hh + (i & 3)
    const int key_min = max(0, qi - window + 1);
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kpos = k_lo + krow0 + 32 * t + 8 * (i >> 2) + 4 * hh + (i & 3);
            const float x = (kpos >= key_min && kpos <= qi) ? sacc[t][i] * scale : -INFINITY;
            sacc[t][i] = x;
            m = fmaxf(m, x);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
    bf16x8 pf[NKT][2];
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float pv = __expf(sacc[t][i] - m);
            sum += pv;
            pf[t][i >> 3][i & 7] = (bf16)pv;
        }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    f32x16 oacc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[n][i] = 0.f;
    const int q4 = (lane >> 2) & 3, p4 = lane & 3, c16 = (lane >> 4) & 1;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bf16x8 a;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int row = krow0 + 32 * t + 16 * s2 + 8 * jj + 4 * hh + q4;
                    const unsigned char* ad = Vi + am_off_v(row, 4 * n + 2 * c16 + (p4 >> 1)) + 8 * (p4 & 1);
                    const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)ad);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[4 * jj + e] = v[e];
                }
                oacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[t][s2], oacc[n], 0, 0, 0);
            }
    if (qi < Tn) {
        bf16* orow = out + ((size_t)b * Tn + qi) * ((size_t)nh * AM_D) + h * AM_D + 4 * hh;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = am_pack2(oacc[n][4 * g] * inv, oacc[n][4 * g + 1] * inv);
                w.y = am_pack2(oacc[n][4 * g + 2] * inv, oacc[n][4 * g + 3] * inv);
                *(uint2*)(orow + 32 * n + 8 * g) = w;
            }
    }
}

// The single new query of a frame-rate push (one frame per stream and launch sequence): one wave per (stream, head), no
// staging and no workgroup barrier.  Scores: lane = key, each lane reads its key's row where it lives (the slot's ring
// buffer, or the new row for the query's own frame), rotates it by its frame number and dots it with the rotated query
// held in LDS; softmax by wave reductions in fp32; P.V: lane = dimension, the value rows read 128 bytes at a time.
// Sums are explicit fma chains, but not in the tiled kernel's order: in fp32 the two agree to rounding (a few 1e-5 at the
// waveform after the decoder's amplification), each within the fp32 bar of the reference decode.
// D > 0: the head size as a compile-time constant (the loops over a row's pieces unroll, so a key's eight 16-byte pieces and
// its rotation's table rows are all requested before the first product); D = 0: any head size.
template <typename T, int D>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) swa_attention_step_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                                 const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                                 int nh, int nkv, int d, int window, float scale, int hist,
                                                                 const int* __restrict__ pos0, int max_pos, Q3KvRing ring) {
    if (D > 0) d = D;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_att[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x * 4 + wave, b = blockIdx.y;
    const bool live = h < nh;                              // (every wave reaches the one barrier below)
    const int hk = (live ? h : nh - 1) / (nh / nkv), half = d >> 1;
    const int p = pos0[b];                                 // the query's frame number
    const int rs = (nh + 2 * nkv) * d;
    const T* qrow = qkv + (size_t)b * rs;
    const int slot = min(max(ring.slots[b], 0), ring.capacity - 1);
    const T* ring_base = (const T*)(ring.state + (long long)slot * ring.slot_stride + ring.state_off);
    float* Qs = (float*)smem_att + (size_t)wave * (d + 64 * ATT_MAXC);
    float* Ps = Qs + d;
    const int jlo = max(max(0, p - hist), p - window + 1);      // oldest frame the ring still holds and the window allows
    const int cnt = p - jlo + 1;
    // The rotation tables' rows of frames jlo .. p, staged once for the block's four heads (row stride half + 1: lane = key
    // reads hit distinct banks).  Read per lane from global memory they were twice the keys' traffic, 16 bytes of every
    // line a request touches.
    const int tld = half + 1;
    float* Cs = (float*)smem_att + (size_t)4 * (d + 64 * ATT_MAXC);
    float* Ss = Cs + (size_t)window * tld;
    constexpr int EPV = 16 / (int)sizeof(T);
    // Scores, known head size whose half row is a power-of-two number of 16-byte pieces (64 in bf16: 4, in fp32: 8): LPK lanes
    // share a key - lane (g, piece) reads piece `piece` of the key's lower and upper half (the rotation pairs element e with
    // e + half), so one load instruction covers 64 / LPK keys in whole 64- or 128-byte runs instead of 16 bytes of 64 different
    // rows.  Five rounds of loads (40 registers) are in flight at a time: 80 keys in bf16, one pass for the decoder's 72-frame
    // window - and the first pass is requested HERE, before the tables are staged, with the query's own elements: neither
    // depends on the tables, so the keys' round trip to memory runs under the staging and its barrier.
    constexpr int NPH = D > 0 ? (D / 2) / EPV : 0;             // 16-byte pieces per half row
    constexpr bool GROUPED = D > 0 && D <= 128 && (D / 2) % EPV == 0 && NPH >= 1 && NPH <= 32 && (NPH & (NPH - 1)) == 0;
    constexpr int LPK = GROUPED ? NPH : 1, KPI = 64 / LPK, BATCH = 5;
    const int kg = lane / LPK, kpiece = lane - kg * LPK, ke0 = kpiece * EPV;
    uint4 klo[BATCH], khi[BATCH];
    auto load_keys = [&](int j0) {
#pragma unroll
        for (int it = 0; it < BATCH; ++it) {
            const int jj = j0 + it * KPI + kg, j = jlo + jj;
            if (j0 + it * KPI < cnt) {                           // wave-uniform
                const T* kr = (jj >= cnt || j == p) ? qrow + (nh + hk) * d : ring_base + (size_t)(j % hist) * (2 * nkv * d) + hk * d;
                klo[it] = *(const uint4*)(kr + ke0);
                khi[it] = *(const uint4*)(kr + ke0 + half);
            }
        }
    };
    float qx1 = 0.f, qx2 = 0.f;
    if constexpr (GROUPED) {
        load_keys(0);
        if (live && lane < half) { qx1 = q3_to_f32(qrow[h * d + lane]); qx2 = q3_to_f32(qrow[h * d + lane + half]); }
    }
    if (half % 4 == 0) {
        const int q4 = half >> 2;
        for (int i = tid; i < cnt * q4; i += 256) {
            const int r = i / q4, e = (i - r * q4) * 4;
            const size_t src = (size_t)min(jlo + r, max_pos - 1) * half + e;
            const float4 c4 = *(const float4*)(cos_t + src), s4 = *(const float4*)(sin_t + src);
            float* cd = Cs + r * tld + e;
            float* sd = Ss + r * tld + e;
            cd[0] = c4.x; cd[1] = c4.y; cd[2] = c4.z; cd[3] = c4.w;
            sd[0] = s4.x; sd[1] = s4.y; sd[2] = s4.z; sd[3] = s4.w;
        }
    } else {
        for (int i = tid; i < cnt * half; i += 256) {
            const int r = i / half, e = i - r * half;
            const size_t src = (size_t)min(jlo + r, max_pos - 1) * half + e;
            Cs[r * tld + e] = cos_t[src];
            Ss[r * tld + e] = sin_t[src];
        }
    }
    __syncthreads();
    if (!live) return;
    {
        const float* cq = Cs + (cnt - 1) * tld;
        const float* sq = Ss + (cnt - 1) * tld;
        if constexpr (GROUPED) {                               // half <= 64: one element pair per lane, loaded above
            if (lane < half) {
                const float c = cq[lane], sn = sq[lane];
                Qs[lane] = qx1 * c - qx2 * sn;
                Qs[lane + half] = qx2 * c + qx1 * sn;
            }
        } else {
            for (int e = lane; e < half; e += 64) {
                const float x1 = q3_to_f32(qrow[h * d + e]), x2 = q3_to_f32(qrow[h * d + e + half]);
                const float c = cq[e], sn = sq[e];
                Qs[e] = x1 * c - x2 * sn;
                Qs[e + half] = x2 * c + x1 * sn;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    float sc[ATT_MAXC];
    float mx = -INFINITY;
    if constexpr (GROUPED) {
        float qlo[EPV], qhi[EPV];
#pragma unroll
        for (int u = 0; u < EPV; ++u) { qlo[u] = Qs[ke0 + u]; qhi[u] = Qs[ke0 + u + half]; }
        for (int j0 = 0; j0 < cnt; j0 += BATCH * KPI) {
            if (j0 > 0) load_keys(j0);
#pragma unroll
            for (int it = 0; it < BATCH; ++it) {
                if (j0 + it * KPI < cnt) {
                    const int jj = j0 + it * KPI + kg;
                    const bool ok = jj < cnt;
                    const float* cr = Cs + (ok ? jj : 0) * tld + ke0;
                    const float* sr = Ss + (ok ? jj : 0) * tld + ke0;
                    float alo = 0.f, ahi = 0.f;
#pragma unroll
                    for (int u = 0; u < EPV; ++u) {
                        const float x1 = q3_to_f32(((const T*)&klo[it])[u]), x2 = q3_to_f32(((const T*)&khi[it])[u]);
                        const float cc = cr[u], ss = sr[u];
                        alo = fmaf(qlo[u], x1 * cc - x2 * ss, alo);
                        ahi = fmaf(qhi[u], x2 * cc + x1 * ss, ahi);
                    }
                    float part = alo + ahi;
#pragma unroll
                    for (int off = 1; off < LPK; off <<= 1) part += __shfl_xor(part, off, 64);
                    if (kpiece == 0 && ok) Ps[jj] = part * scale;          // raw score; the softmax below reads it back with lane = key
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < ATT_MAXC; ++c) {
            sc[c] = (lane + 64 * c < cnt) ? Ps[lane + 64 * c] : -INFINITY;
            mx = fmaxf(mx, sc[c]);
        }
    } else {

#pragma unroll
    for (int c = 0; c < ATT_MAXC; ++c) {
        sc[c] = -INFINITY;
        const int j = jlo + lane + 64 * c;
        if (64 * c < cnt && j <= p) {
            const T* kr = (j == p) ? qrow + (nh + hk) * d : ring_base + (size_t)(j % hist) * (2 * nkv * d) + hk * d;
            const float* cr = Cs + (j - jlo) * tld;
            const float* sr = Ss + (j - jlo) * tld;
            float alo = 0.f, ahi = 0.f;
            if (half % EPV == 0) {
#pragma unroll
                for (int e = 0; e < half; e += EPV) {
                    const uint4 klo = *(const uint4*)(kr + e), khi = *(const uint4*)(kr + e + half);
#pragma unroll
                    for (int u = 0; u < EPV; ++u) {
                        const float x1 = q3_to_f32(((const T*)&klo)[u]), x2 = q3_to_f32(((const T*)&khi)[u]);
                        const float cc = cr[e + u], ss = sr[e + u];
                        alo = fmaf(Qs[e + u], x1 * cc - x2 * ss, alo);
                        ahi = fmaf(Qs[e + u + half], x2 * cc + x1 * ss, ahi);
                    }
                }
            } else {
                for (int e = 0; e < half; ++e) {
                    const float x1 = q3_to_f32(kr[e]), x2 = q3_to_f32(kr[e + half]);
                    alo = fmaf(Qs[e], x1 * cr[e] - x2 * sr[e], alo);
                    ahi = fmaf(Qs[e + half], x2 * cr[e] + x1 * sr[e], ahi);
                }
            }
            sc[c] = (alo + ahi) * scale;
        }
        mx = fmaxf(mx, sc[c]);
    }
    }
    // P.V, known head size: the first eight rounds of value loads do not depend on the probabilities - requested before the
    // softmax, their round trip runs under it
    const size_t vcol = (size_t)nkv * d + hk * d;
    constexpr int VU = 8, VNP = GROUPED ? D / EPV : 1, VNG = 64 / VNP;
    const int vg = lane / VNP, vpiece = lane - vg * VNP;
    const int ring0 = jlo % hist;
    uint4 vv[VU];
    auto load_vals = [&](int j0) {
        const T* vnew = qrow + (size_t)(nh + nkv + hk) * d + vpiece * EPV;
        const T* vring = ring_base + vcol + vpiece * EPV;
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int jj = j0 + u * VNG + vg;
            const bool ok = jj < cnt;
            int idx = ring0 + (ok ? jj : 0);
            idx = idx >= hist ? idx - hist : idx;
            vv[u] = *(const uint4*)((!ok || jj == cnt - 1) ? vnew : vring + (size_t)idx * (2 * nkv * d));
        }
    };
    if constexpr (GROUPED) load_vals(0);
    const float m = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < ATT_MAXC; ++c) {
        const float pv = sc[c] != -INFINITY ? expf(sc[c] - m) : 0.f;
        sc[c] = pv;
        sum += pv;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;                          // the query's own frame is always among its keys
#pragma unroll
    for (int c = 0; c < ATT_MAXC; ++c)
        if (64 * c < cnt) Ps[lane + 64 * c] = sc[c] * inv;
    __builtin_amdgcn_wave_barrier();
    if constexpr (GROUPED) {
        float acc[EPV];
#pragma unroll
        for (int u = 0; u < EPV; ++u) acc[u] = 0.f;
        for (int j0 = 0; j0 < cnt; j0 += VU * VNG) {
            if (j0 > 0) load_vals(j0);
            float pj[VU];
#pragma unroll
            for (int u = 0; u < VU; ++u) { const int jj = j0 + u * VNG + vg; pj[u] = jj < cnt ? Ps[jj] : 0.f; }
#pragma unroll
            for (int u = 0; u < VU; ++u)
#pragma unroll
                for (int q = 0; q < EPV; ++q) acc[q] = fmaf(pj[u], q3_to_f32(((const T*)&vv[u])[q]), acc[q]);
        }
        for (int off = VNP; off < 64; off <<= 1)
#pragma unroll
            for (int q = 0; q < EPV; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
        if (vg == 0) {
            uint4 o;
#pragma unroll
            for (int q = 0; q < EPV; ++q) ((T*)&o)[q] = q3_from_f32<T>(acc[q]);
            *(uint4*)(out + (size_t)b * (nh * d) + h * d + vpiece * EPV) = o;
        }
        return;
    }
    // P.V.  A value row is NP = d / EPV 16-byte pieces; when NP is a power of two <= 64 the wave's lanes form 64 / NP key groups
    // of NP lanes: lane (g, piece) walks keys g, g + NG, ... with ONE 16-byte load per key, eight keys' loads in flight before
    // their products are summed (a serial walk over the keys pays a memory round trip per key: 36 of them for a 72-frame
    // window), and the groups' partial sums meet through lane exchanges.
    const int NP = d / EPV;
    if (d % EPV == 0 && NP <= 64 && (NP & (NP - 1)) == 0) {
        const int NG = 64 / NP, g = lane / NP, piece = lane - g * NP;
        float acc[EPV];
#pragma unroll
        for (int u = 0; u < EPV; ++u) acc[u] = 0.f;
        const T* vnew = qrow + (size_t)(nh + nkv + hk) * d + piece * EPV;
        const T* vring = ring_base + vcol + piece * EPV;
        constexpr int U = 8;
        for (int j0 = 0; j0 < cnt; j0 += U * NG) {
            uint4 v[U];
            float pj[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = j0 + u * NG + g;
                const bool ok = jj < cnt;
                int idx = ring0 + (ok ? jj : 0);
                idx = idx >= hist ? idx - hist : idx;
                const T* src = (!ok || jj == cnt - 1) ? vnew : vring + (size_t)idx * (2 * nkv * d);
                v[u] = *(const uint4*)src;
                pj[u] = ok ? Ps[jj] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < EPV; ++q) acc[q] = fmaf(pj[u], q3_to_f32(((const T*)&v[u])[q]), acc[q]);
        }
        for (int off = NP; off < 64; off <<= 1)
#pragma unroll
            for (int q = 0; q < EPV; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
        if (g == 0) {
            uint4 o;
#pragma unroll
            for (int q = 0; q < EPV; ++q) ((T*)&o)[q] = q3_from_f32<T>(acc[q]);
            *(uint4*)(out + (size_t)b * (nh * d) + h * d + piece * EPV) = o;
        }
        return;
    }
    for (int e = lane; e < d; e += 64) {
        float a0 = 0.f, a1 = 0.f;
        int idx = jlo % hist;
        int jj = 0;
        for (; jj + 2 <= cnt - 1; jj += 2) {               // carried frames, two at a time; the last key is the new row
            const int i0 = idx, i1 = idx + 1 == hist ? 0 : idx + 1;
            idx = i1 + 1 == hist ? 0 : i1 + 1;
            const float v0 = q3_to_f32(ring_base[(size_t)i0 * (2 * nkv * d) + vcol + e]);
            const float v1 = q3_to_f32(ring_base[(size_t)i1 * (2 * nkv * d) + vcol + e]);
            a0 = fmaf(Ps[jj], v0, a0);
            a1 = fmaf(Ps[jj + 1], v1, a1);
        }
        for (; jj < cnt - 1; ++jj) {
            a0 = fmaf(Ps[jj], q3_to_f32(ring_base[(size_t)idx * (2 * nkv * d) + vcol + e]), a0);
            idx = idx + 1 == hist ? 0 : idx + 1;
        }
        a0 = fmaf(Ps[cnt - 1], q3_to_f32(qrow[(nh + nkv + hk) * d + e]), a0);
        out[(size_t)b * (nh * d) + h * d + e] = q3_from_f32<T>(a0 + a1);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) cast_to_f32_kernel(const T* __restrict__ x, float* __restrict__ y, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n; i += stride) y[i] = q3_to_f32(x[i]);
}

template <typename T>
__global__ void __launch_bounds__(256) cast_from_f32_kernel(const float* __restrict__ x, T* __restrict__ y, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n; i += stride) y[i] = q3_from_f32<T>(x[i]);
}

template <typename T>
__global__ void __launch_bounds__(256) snake_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ a,
                                                    const float* __restrict__ invb, long long n, int C) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n; i += stride) {
        int c = (int)(i % C);
        y[i] = q3_from_f32<T>(q3_snake<T>(q3_to_f32(x[i]), a[c], invb[c]));
    }
}

// ------------------------------------------------------------------------------------------------
// Stream state.  A stream's carried context (the last rows of every tensor a causal kernel looks back into) lives
// in one contiguous record per stream slot; a push copies the records of the participating slots into the front
// rows of the work buffers, runs the layers over [carried | new] rows, and copies the last rows back.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) stream_carry_kernel(const Q3CarryDesc* __restrict__ descs, char* __restrict__ work,
                                                           char* __restrict__ state, long long slot_stride,
                                                           const int* __restrict__ slots, int capacity, int save,
                                                           const int* __restrict__ pos0) {
    const Q3CarryDesc d = descs[blockIdx.x];
    const int b = blockIdx.y;
    const int slot = min(max(slots[b], 0), capacity - 1);
    const int ppr = d.row_bytes >> 4;
    if (d.ring) {
        // a ring buffer (keys / values of one layer): nothing to load - the attention kernel reads the ring itself; after
        // the push the new rows (the last `rows` of them) are appended at their frame number modulo the ring size
        if (!save) return;
        const int first = max(0, d.new_rows - d.rows), p0 = pos0[b];
        const char* w0 = work + d.work_off + (long long)b * d.work_bstride + d.col_off;
        char* s0 = state + (long long)slot * slot_stride + d.state_off;
        for (int id = threadIdx.x; id < (d.new_rows - first) * ppr; id += 256) {
            const int row = first + id / ppr, ch = id % ppr;
            *(uint4*)(s0 + (long long)((p0 + row) % d.rows) * d.row_bytes + ch * 16) = *(const uint4*)(w0 + (long long)row * d.work_ld + ch * 16);
        }
        return;
    }
    char* w0 = work + d.work_off + (long long)b * d.work_bstride + d.col_off + (save ? (long long)d.new_rows * d.work_ld : 0);
    char* s0 = state + (long long)slot * slot_stride + d.state_off;
    for (int id = threadIdx.x; id < d.rows * ppr; id += 256) {
        const int row = id / ppr, ch = id - row * ppr;
        uint4* wp = (uint4*)(w0 + (long long)row * d.work_ld + ch * 16);
        uint4* sp = (uint4*)(s0 + (long long)row * d.row_bytes + ch * 16);
        if (save) *sp = *wp; else *wp = *sp;
    }
}

// pos0[b] = frames the stream has consumed so far; row_start[st][b] = rows of stage st that precede the stream's
// first sample (only on a stream's first push); then the slot's frame counter advances by n.
__global__ void stream_prep_kernel(const int* __restrict__ slots, int S, int capacity, int* __restrict__ pos, int n, int max_pos,
                                   int* __restrict__ pos0, int* __restrict__ row_start, Q3StageStarts st, int* err_flag) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S) return;
    int slot = slots[b];
    if (slot < 0 || slot >= capacity) { atomicOr(err_flag, 2); slot = min(max(slot, 0), capacity - 1); }
    int p = pos[slot];
    if (p + n > max_pos) { atomicOr(err_flag, 4); p = max(0, max_pos - n); }
    pos0[b] = p;
    for (int i = 0; i < st.count; ++i) row_start[i * S + b] = p == 0 ? st.rows[i] : 0;
    pos[slot] = p + n;
}

__global__ void __launch_bounds__(256) stream_reset_kernel(const int* __restrict__ slots, int capacity, char* __restrict__ state,
                                                           long long slot_stride, int* __restrict__ pos, int* err_flag) {
    int slot = slots[blockIdx.y];
    if (slot < 0 || slot >= capacity) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(err_flag, 2); return; }
    uint4* sp = (uint4*)(state + (long long)slot * slot_stride);
    const long long n16 = slot_stride >> 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) sp[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) pos[slot] = 0;
}

inline unsigned grid_for(long long n, int per_block) {
    long long g = (n + per_block - 1) / per_block;
    return (unsigned)g;
}

}  // namespace

hipError_t q3_launch_rvq_gather_sum(const long long* codes, int B, int Q, int T, int bins, int D,
                                    const float* const* tables_dev, const int* group_start_dev, int G,
                                    float* out, int* err_flag, hipStream_t s) {
    if (D % 4) return hipErrorInvalidValue;
    long long n = (long long)B * T * (D / 4);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rvq_gather_sum_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, codes, B, Q, T, bins, D, tables_dev,
                       group_start_dev, G, out, err_flag);
    return hipGetLastError();
}

hipError_t q3_launch_rvq_encode(const float* emb, int B, int T, int in_dim, int Q, int bins, int D, const float* const* tables_dev,
                                const void* const* tables_t_dev, const int* group_start_dev, int G, const float* in_proj_dev,
                                long long* codes, hipStream_t s, int tuning) {
    const long long frames = (long long)B * T;
    if (frames == 0) return hipSuccess;
    if ((D & 3) || D < 4 || in_dim < 1 || (!in_proj_dev && in_dim != D)) return hipErrorInvalidValue;
    if (D == 256 && bins % 128 == 0 && !(tuning & (1 << 20))) {       // full-size code books: the search on the fp32 matrix op
        const size_t ldsm = ((size_t)RVQM_FT * (in_dim + 4 + D + 1) + 4 * RVQM_FT) * sizeof(float) + (4 * RVQM_FT + RVQM_FT) * sizeof(int);
        if (ldsm <= 160 * 1024 && in_dim % 8 == 0) {
            auto km = rvq_encode_mfma_kernel<256>;
            hipError_t em = hipFuncSetAttribute((const void*)km, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm);
            if (em != hipSuccess) return em;
            const long long blocksm = (frames + RVQM_FT - 1) / RVQM_FT;
            if (blocksm > 0x7fffffffLL) return hipErrorInvalidValue;
            hipLaunchKernelGGL(km, dim3((unsigned)blocksm), dim3(256), ldsm, s, emb, frames, T, in_dim, Q, bins, tables_dev,
                               (const float4* const*)tables_t_dev, group_start_dev, G, in_proj_dev, codes);
            return hipGetLastError();
        }
    }
    const size_t lds = ((size_t)RVQ_FT * (in_dim + D) + 4 * RVQ_FT) * sizeof(float) + (4 * RVQ_FT + RVQ_FT) * sizeof(int);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipError_t e;
    if (lds > 48 * 1024 && (e = hipFuncSetAttribute((const void*)rvq_encode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    const long long blocks = (frames + RVQ_FT - 1) / RVQ_FT;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rvq_encode_kernel, dim3((unsigned)blocks), dim3(256), lds, s, emb, frames, T, in_dim, Q, bins, D, tables_dev,
                       (const float4* const*)tables_t_dev, group_start_dev, G, in_proj_dev, codes);
    return hipGetLastError();
}

hipError_t q3_launch_mean_embedding(const long long* codes, int B, int Q, int T, int bins, int H, const float* table, void* out,
                                    int out_dtype, int* err_flag, hipStream_t s) {
    if (H % 4) return hipErrorInvalidValue;
    long long n = (long long)B * T * (H / 4);
    if (n == 0) return hipSuccess;
    if (out_dtype == Q3_BF16)
        hipLaunchKernelGGL(mean_embedding_kernel<bf16>, dim3(grid_for(n, 256)), dim3(256), 0, s, codes, B, Q, T, bins, H, table,
                           (bf16*)out, err_flag);
    else
        hipLaunchKernelGGL(mean_embedding_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, s, codes, B, Q, T, bins, H, table,
                           (float*)out, err_flag);
    return hipGetLastError();
}

hipError_t q3_launch_rmsnorm(const void* x, const float* w, void* y, long long rows, int H, float eps, int dtype, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    dim3 grid(grid_for(rows, 4));
    if (dtype == Q3_BF16)
        hipLaunchKernelGGL(rmsnorm_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, w, (bf16*)y, rows, H, eps);
    else
        hipLaunchKernelGGL(rmsnorm_kernel<float>, grid, dim3(256), 0, s, (const float*)x, w, (float*)y, rows, H, eps);
    return hipGetLastError();
}

// The same operation with the input read about once: a block walks RT consecutive rows of one item and keeps the K - 1 rows a
// depthwise tap looks back into in registers, so a row is fetched by one block (plus K - 1 rows of overlap per RT) instead of by
// the K blocks whose windows contain it - whose workgroups, dealt round-robin over the eight XCDs, each pulled it into their
// own L2.  Same arithmetic per element and the same reduction order as dwconv_ln_kernel: bit-identical results.
template <typename T, int CPT, int K, int RT>
__global__ void __launch_bounds__(256) dwconv_ln_rows_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ lnw,
                                                             const float* __restrict__ lnb, T* __restrict__ y, int L, int C,
                                                             float eps, int hist) {
    __shared__ float red[2][8];
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * RT;
    const T* xb = x + ((long long)b * (hist + L) + hist) * C;
    T* yb = y + (long long)b * L * C;
    float wk[K][CPT], bs[CPT], gw[CPT], gb[CPT], win[K][CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = tid + i * 256;
        const bool ok = c < C;
        bs[i] = ok ? bias[c] : 0.f; gw[i] = ok ? lnw[c] : 0.f; gb[i] = ok ? lnb[c] : 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) wk[j][i] = ok ? w[j * C + c] : 0.f;
#pragma unroll
        for (int j = 0; j < K - 1; ++j) {        // rows t0 - (K - 1) + j: carried rows, or zeros before the sequence
            const int tt = t0 - (K - 1) + j;
            win[j + 1][i] = (ok && tt >= -hist) ? q3_to_f32(xb[(long long)tt * C + c]) : 0.f;
        }
    }
    const int rows = min(RT, L - t0);
    for (int r = 0; r < rows; ++r) {
        const int t = t0 + r, p = r & 1;
        float v[CPT], s = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * 256;
#pragma unroll
            for (int j = 0; j < K - 1; ++j) win[j][i] = win[j + 1][i];
            win[K - 1][i] = c < C ? q3_to_f32(xb[(long long)t * C + c]) : 0.f;
            float a = 0.f;
            if (c < C) {
                a = bs[i];
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if (t - (K - 1) + j >= -hist) a += wk[j][i] * win[j][i];
                s += a;
            }
            v[i] = a;
        }
        s = wave_sum(s);
        if ((tid & 63) == 0) red[p][tid >> 6] = s;
        __syncthreads();
        const float mean = (red[p][0] + red[p][1] + red[p][2] + red[p][3]) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * 256;
            if (c < C) { const float d = v[i] - mean; q += d * d; }
        }
        q = wave_sum(q);
        if ((tid & 63) == 0) red[p][4 + (tid >> 6)] = q;
        __syncthreads();
        const float rstd = rsqrtf((red[p][4] + red[p][5] + red[p][6] + red[p][7]) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * 256;
            if (c < C) yb[(long long)t * C + c] = q3_from_f32<T>((v[i] - mean) * rstd * gw[i] + gb[i]);
        }
    }
}

hipError_t q3_launch_dwconv_ln(const void* x, const float* w, const float* b, const float* lnw, const float* lnb, void* y, int B,
                               int L, int C, int k, float eps, int dtype, hipStream_t s, int hist) {
    if (hist < 0) return hipErrorInvalidValue;
    long long rows = (long long)B * L;
    if (rows == 0) return hipSuccess;
    if (C > 256 * 8) return hipErrorInvalidValue;
    // the usual case (kernel 7, up to 1024 channels, rows enough to fill the chip in 32-row walks): the input is read ~once
    if (k == 7 && C <= 1024 && B <= 65535 && rows >= 32 * 512) {
        constexpr int RT = 32;
        dim3 g2((unsigned)((L + RT - 1) / RT), (unsigned)B);
        if (dtype == Q3_BF16)
            hipLaunchKernelGGL((dwconv_ln_rows_kernel<bf16, 4, 7, RT>), g2, dim3(256), 0, s, (const bf16*)x, w, b, lnw, lnb, (bf16*)y, L, C, eps, hist);
        else
            hipLaunchKernelGGL((dwconv_ln_rows_kernel<float, 4, 7, RT>), g2, dim3(256), 0, s, (const float*)x, w, b, lnw, lnb, (float*)y, L, C, eps, hist);
        return hipGetLastError();
    }
    dim3 grid((unsigned)rows);
#define Q3_DW(TT, CPT) hipLaunchKernelGGL((dwconv_ln_kernel<TT, CPT>), grid, dim3(256), 0, s, (const TT*)x, w, b, lnw, lnb, (TT*)y, L, C, k, eps, hist)
    const int cpt = (C + 255) / 256;
    if (dtype == Q3_BF16) {
        if (cpt <= 1) Q3_DW(bf16, 1); else if (cpt <= 4) Q3_DW(bf16, 4); else Q3_DW(bf16, 8);
    } else {
        if (cpt <= 1) Q3_DW(float, 1); else if (cpt <= 4) Q3_DW(float, 4); else Q3_DW(float, 8);
    }
#undef Q3_DW
    return hipGetLastError();
}

hipError_t q3_launch_swa_attention(const void* qkv, void* out, const float* cos_t, const float* sin_t, int B, int T, int nh,
                                   int nkv, int d, int window, int dtype, hipStream_t s, int hist, const int* pos0, int max_pos,
                                   const Q3KvRing* ring_in, int tuning) {
    if (B == 0 || T == 0) return hipSuccess;
    if (hist < 0 || max_pos < 1) return hipErrorInvalidValue;
    if (hist > 0 && (!ring_in || !ring_in->state || !ring_in->slots || !pos0)) return hipErrorInvalidValue;
    Q3KvRing ring{};
    if (ring_in) ring = *ring_in;
    if (T == 1 && hist > 0) {          // a frame-rate push: one wave per (stream, head)
        const size_t lds1 = ((size_t)4 * (d + 64 * ATT_MAXC) + (size_t)2 * window * (d / 2 + 1)) * sizeof(float);   // + the staged table rows
        if (lds1 > 160 * 1024) return hipErrorInvalidValue;
        dim3 grid1((nh + 3) / 4, B);
        const float scale1 = 1.0f / sqrtf((float)d);
        auto step = [&](auto kern, auto* qp, auto* op) -> hipError_t {
            if (lds1 > 48 * 1024) {
                hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
                if (ea != hipSuccess) return ea;
            }
            hipLaunchKernelGGL(kern, grid1, dim3(256), lds1, s, qp, op, cos_t, sin_t, nh, nkv, d, window, scale1, hist, pos0, max_pos, ring);
            return hipGetLastError();
        };
        if (dtype == Q3_BF16)
            return d == 64 ? step(swa_attention_step_kernel<bf16, 64>, (const bf16*)qkv, (bf16*)out) : step(swa_attention_step_kernel<bf16, 0>, (const bf16*)qkv, (bf16*)out);
        return d == 64 ? step(swa_attention_step_kernel<float, 64>, (const float*)qkv, (float*)out) : step(swa_attention_step_kernel<float, 0>, (const float*)qkv, (float*)out);
    }
    if (dtype == Q3_BF16 && d == AM_D && hist == 0 && !pos0 && window >= 1 && window <= 257 && nkv > 0 && nh % nkv == 0 && !(tuning & 256)) {
        const bool wide = window > 97;             // nine key tiles per wave instead of four
        const size_t ldsm = (size_t)2 * (AM_QB + 32 * (wide ? 8 : 3)) * 128;
        auto km = wide ? swa_attention_mfma_kernel<9> : swa_attention_mfma_kernel<4>;
        hipError_t em = hipFuncSetAttribute((const void*)km, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm);
        if (em != hipSuccess) return em;
        dim3 gridm((T + AM_QB - 1) / AM_QB, nh, B);
        hipLaunchKernelGGL(km, gridm, dim3(256), ldsm, s, (const bf16*)qkv, (bf16*)out, cos_t, sin_t, T, nh, nkv, window, 1.0f / sqrtf((float)d), max_pos);
        return hipGetLastError();
    }
    if (window < 1 || window + 3 > 64 * ATT_MAXC || (d & 3) || d > 256) return hipErrorInvalidValue;    // rows are read 16 bytes at a time
    int qt = ATT_QT;
    auto lds_for = [&](int q) { return ((size_t)2 * (q + window - 1) * (d + 4) + (size_t)q * d + 4 * 4 * 64 * ATT_MAXC) * sizeof(float); };
    while (qt > 8 && lds_for(qt) > 160 * 1024) qt >>= 1;      // a long window (the encoder's 250 at head size 64): fewer queries per block
    const size_t lds = lds_for(qt);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid((T + qt - 1) / qt, nh, B);
    const float scale = 1.0f / sqrtf((float)d);
    hipError_t e;
    if (dtype == Q3_BF16) {
        auto k = swa_attention_kernel<bf16>;
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, (const bf16*)qkv, (bf16*)out, cos_t, sin_t, hist + T, nh, nkv, d, window, scale, hist, pos0, max_pos, ring, qt);
    } else {
        auto k = swa_attention_kernel<float>;
        if (lds > 48 * 1024 && (e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, (const float*)qkv, (float*)out, cos_t, sin_t, hist + T, nh, nkv, d, window, scale, hist, pos0, max_pos, ring, qt);
    }
    return hipGetLastError();
}

hipError_t q3_launch_audio_conv(const float* audio, const float* w, const float* bias, void* y, void* y2, long long y2_bstride, int B, int L, int F,
                                int k, int dtype, hipStream_t s) {
    if (B == 0 || L == 0) return hipSuccess;
    if (B > 65535 || F < 8 || (F & 7) || k < 1 || (size_t)F * (k + 1) * 4 > 48 * 1024) return hipErrorInvalidValue;
    if (((uintptr_t)y & 15) || ((uintptr_t)y2 & 15) || (y2 && (y2_bstride & 7))) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(((long long)L * (F / 8) + 255) / 256), B);
    const size_t lds = (size_t)F * (k + 1) * 4;
    if (dtype == Q3_BF16) hipLaunchKernelGGL(audio_conv_kernel<bf16>, grid, dim3(256), lds, s, audio, w, bias, (bf16*)y, (bf16*)y2, y2_bstride, L, F, k);
    else hipLaunchKernelGGL(audio_conv_kernel<float>, grid, dim3(256), lds, s, audio, w, bias, (float*)y, (float*)y2, y2_bstride, L, F, k);
    return hipGetLastError();
}

hipError_t q3_launch_layernorm(const void* x, const float* w, const float* bias, void* y, long long rows, int H, float eps, int dtype, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == Q3_BF16) hipLaunchKernelGGL(layernorm_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, w, bias, (bf16*)y, rows, H, eps);
    else hipLaunchKernelGGL(layernorm_kernel<float>, grid, dim3(256), 0, s, (const float*)x, w, bias, (float*)y, rows, H, eps);
    return hipGetLastError();
}

hipError_t q3_launch_replicate_pad(const void* x, void* y, int B, int L, int C, int left, int Lout, int dtype, hipStream_t s) {
    if (B == 0 || Lout == 0) return hipSuccess;
    if (B > 65535 || L < 1) return hipErrorInvalidValue;
    const long long blocks = ((long long)Lout * C + 255) / 256;
    const dim3 grid((unsigned)(blocks > 8192 ? 8192 : blocks), B);
    if (dtype == Q3_BF16) hipLaunchKernelGGL(replicate_pad_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)x, (bf16*)y, L, C, left, Lout);
    else hipLaunchKernelGGL(replicate_pad_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (float*)y, L, C, left, Lout);
    return hipGetLastError();
}

hipError_t q3_launch_tap_sum(const float* P, const float* bias, float* y, int B, int M, int taps, int act, hipStream_t s, int m_first,
                             const long long* spans) {
    if (B == 0 || M == 0) return hipSuccess;
    if (taps < 1 || taps > 8 || B > 65535 || m_first < 0 || m_first >= M) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tap_sum_kernel, dim3((M - m_first + 255) / 256, B), dim3(256), 0, s, P, bias, y, M, taps, act, m_first, spans);
    return hipGetLastError();
}

hipError_t q3_launch_span_copy(const float* src, int B, int M, int m_first, const long long* spans, float* out, hipStream_t s) {
    if (B == 0 || M == 0) return hipSuccess;
    if (B > 65535 || m_first < 0 || m_first >= M || !spans) return hipErrorInvalidValue;
    const int gx = std::min((M - m_first + 255) / 256, 4096);
    hipLaunchKernelGGL(span_copy_kernel, dim3(gx, B), dim3(256), 0, s, src, M, m_first, spans, out);
    return hipGetLastError();
}

hipError_t q3_launch_cast_to_f32(const void* x, float* y, long long n, int dtype, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned g = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (dtype == Q3_BF16)
        hipLaunchKernelGGL(cast_to_f32_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)x, y, n);
    else
        hipLaunchKernelGGL(cast_to_f32_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, y, n);
    return hipGetLastError();
}

hipError_t q3_launch_cast_from_f32(const float* x, void* y, long long n, int dtype, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned g = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (dtype == Q3_BF16)
        hipLaunchKernelGGL(cast_from_f32_kernel<bf16>, dim3(g), dim3(256), 0, s, x, (bf16*)y, n);
    else
        hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(g), dim3(256), 0, s, x, (float*)y, n);
    return hipGetLastError();
}

// float samples -> 16-bit PCM the way libsndfile writes a float array to a PCM_16 file: lrintf(x * 32767), round half to
// even (the reference's sf.write(buffer, wav, sr, format="WAV"), server.py:108-113); saturated outside [-1, 1]
__global__ void __launch_bounds__(256) pcm16_kernel(const float* __restrict__ x, short* __restrict__ y, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n; i += stride) {
        const float v = fminf(32767.0f, fmaxf(-32768.0f, __fmul_rn(x[i], 32767.0f)));
        y[i] = (short)__float2int_rn(v);
    }
}

hipError_t q3_launch_pcm16(const float* x, short* y, long long n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned g = (unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(pcm16_kernel, dim3(g), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}

hipError_t q3_launch_snake(const void* x, void* y, const float* a, const float* invb, long long rows, int C, int dtype, hipStream_t s) {
    long long n = rows * C;
    if (n == 0) return hipSuccess;
    unsigned g = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (dtype == Q3_BF16)
        hipLaunchKernelGGL(snake_kernel<bf16>, dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)y, a, invb, n, C);
    else
        hipLaunchKernelGGL(snake_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, (float*)y, a, invb, n, C);
    return hipGetLastError();
}

hipError_t q3_launch_stream_carry(const Q3CarryDesc* descs_dev, int ndesc, void* work, void* state, long long slot_stride,
                                  const int* slots, int S, int capacity, int save, const int* pos0, hipStream_t s) {
    if (ndesc == 0 || S == 0) return hipSuccess;
    if (slot_stride & 15 || !pos0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stream_carry_kernel, dim3(ndesc, S), dim3(256), 0, s, descs_dev, (char*)work, (char*)state, slot_stride, slots,
                       capacity, save, pos0);
    return hipGetLastError();
}

hipError_t q3_launch_stream_prep(const int* slots, int S, int capacity, int* pos, int n, int max_pos, int* pos0, int* row_start,
                                 const Q3StageStarts& st, int* err_flag, hipStream_t s) {
    if (S == 0) return hipSuccess;
    hipLaunchKernelGGL(stream_prep_kernel, dim3((S + 63) / 64), dim3(64), 0, s, slots, S, capacity, pos, n, max_pos, pos0, row_start, st,
                       err_flag);
    return hipGetLastError();
}

hipError_t q3_launch_stream_reset(const int* slots, int count, int capacity, void* state, long long slot_stride, int* pos,
                                  int* err_flag, hipStream_t s) {
    if (count == 0) return hipSuccess;
    if (slot_stride & 15) return hipErrorInvalidValue;
    const unsigned gx = (unsigned)std::min<long long>(64, (slot_stride / 16 + 255) / 256 + 1);
    hipLaunchKernelGGL(stream_reset_kernel, dim3(gx, count), dim3(256), 0, s, slots, capacity, (char*)state, slot_stride, pos, err_flag);
    return hipGetLastError();
}
