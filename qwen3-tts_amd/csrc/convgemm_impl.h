#pragma once
// Multi-tap row-shifted GEMM on the gfx950 matrix cores: the one dense-contraction kernel of the decode.
//
//   h[b, m, n]   = act( bias[n] + sum_tap sum_k X[b, m + off0 + tap*off_step, k] * W[tap][n][k] ) * scale[n]
//   FUSE2 only:    h <- bias2[n] + sum_j h[b, m, j] * W2[n][j]                (a second, 1x1 contraction in the same launch)
//   y            = R[b, m, n] + h
//   Y <- y (or post(y) when Y2 is null);   Y2 <- post(y)                      (post = SnakeBeta of the consumer)
//
// With channels-last activations every layer of the decoder is this form:
//   causal Conv1d(k, dilation d)      taps = k, off0 = -(k-1)d, off_step = d
//   ConvTranspose1d(k = 2s, stride s) taps = 2, off0 = 0, off_step = 1, N = s*Cout  (row q of [L-1, s*Cout] IS rows q*s .. q*s+s-1 of [.., Cout])
//   ConvTranspose1d(k = s,  stride s) taps = 1, N = s*Cout
//   Linear                            taps = 1
//   residual unit                     FUSE2: snake -> conv k7 -> snake -> conv k1 -> + x in ONE launch.  It reads the RAW
//                                     residual stream: the unit's first snake is applied to the staged input tile in LDS
//                                     (each lane transforms the 16-byte pieces its own DMAs brought in), and the residual
//                                     rows are re-read from the same tensor, so the unit moves two streams, not four
//
// A block owns a BM x BN output tile.  Per K-chunk the (BM + halo) input rows are staged ONCE in LDS and every tap
// reads them at a row offset, so the im2col matrix never exists.  Weight tiles are staged per (K-chunk, tap) step,
// two steps ahead in three buffers; the next chunk's input tile arrives in slices spread over the current chunk's
// steps.  Both go global -> LDS directly (LDS-DMA: `global_load_lds` / `buffer_load ... lds`, 16 bytes per lane), rows
// outside the sequence zero-filled by the buffer descriptor's range check; each wave waits with a counted vmcnt.
// LDS rows are BKB bytes; 16-byte chunk c of row r sits at slot c ^ f(r), so the 16-lane groups of ds_read_b128 touch
// 16 distinct slots; with LDS-DMA the destination is lane-linear and the XOR is applied to the SOURCE address.
//
// Element types: bf16 operands -> v_mfma_f32_32x32x16_bf16; fp32 operands -> v_mfma_f32_32x32x2_f32 (bit-exact fp32
// FMA chains, used by the parity path).  Accumulation is fp32 in both.
#include "common.h"

#include <algorithm>
#include <cstdio>
#include <type_traits>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_cvoid;

template <int CPR> __device__ __forceinline__ int swz(int row, int c) {
    if constexpr (CPR == 8) return c ^ ((row >> 1) & 7);
    else return c ^ ((row >> 2) & 3);
}

template <typename TA> struct Mma;
template <> struct Mma<bf16> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // lane half h holds k = 4h .. 4h+3 of an 8-deep group for BOTH operands; MFMA j contracts k in {j, 4 + j}
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), c, 0, 0, 0);
    }
};

__host__ __device__ constexpr size_t round_kb(size_t v) { return (v + 1023) / 1024 * 1024; }

// s_waitcnt vmcnt(n) for a wave-uniform run-time n: all but the n youngest vector-memory operations are done.
// n above 7 waits for vmcnt(7) (always safe: it only waits longer).
__device__ __forceinline__ void wait_vmcnt(int n) {
    if (n >= 4) {
        if (n >= 6) { if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }
        else { if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
    } else {
        if (n >= 2) { if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
        else { if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
}
template <int N> __device__ __forceinline__ void wait_vmcnt_c() {
    static_assert(N >= 0 && N <= 15, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
// One LDS-DMA instruction: 16 bytes per lane from global memory to LDS at lds_addr + lane * 16 (lds_addr wave-uniform).
// Issued through inline asm on purpose: the compiler then neither counts it nor guards later ds_reads with vmcnt(0), so
// tiles can stay in flight across steps; completion is tracked by wait_vmcnt below.  M0 (the LDS base of an LDS-DMA) is
// written in the same statement.  M0 is a reserved register of the AMDGPU backend: it is never allocated to a value, and the
// compiler's own uses of it write it immediately before the instruction that reads it, so nothing can be live in M0 across
// one of these statements (naming it in the clobber list is rejected as "reserved register" for that reason).
//   dma16    : every lane supplies a full 64-bit address
//   dma16_so : wave-uniform 64-bit base (SGPR pair) + unsigned 32-bit per-lane byte offset
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_addr) {
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);      // wave-uniform by construction; pins it to an SGPR
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma16_so(const void* sbase, unsigned voff, unsigned lds_addr) {
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_address(const void* generic_ptr_into_lds) {
    return (unsigned)(unsigned long long)(lds_void*)generic_ptr_into_lds;
}
__device__ __forceinline__ void block_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Waves per SIMD the register allocation must leave room for: the 8-wave 64-byte-row shape runs two blocks per CU (four
// waves per SIMD, 128 registers each); every other shape two waves per SIMD.
template <typename TA, typename TO, int NW, int WM, int WN, int BKB, bool FUSE2> constexpr int waves_per_simd() {
    // fp32 operands or outputs (the parity path, taps) need more than 128 registers in that shape: one block per CU there
    // the twelve-wave fused unit (768 rows, one block per CU): three waves per SIMD, 168 registers each
    if (FUSE2 && NW == 12) return 3;
    return (sizeof(TA) == 2 && sizeof(TO) == 2 && NW == 8 && !FUSE2 && BKB == 64 && WM * WN <= 4) ? 4 : 2;
}

template <typename TA, typename TO, int WAVES_M, int WAVES_N, int WM, int WN, int BKB, bool FUSE2, bool SEG>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N)
__attribute__((amdgpu_waves_per_eu(waves_per_simd<TA, TO, WAVES_M * WAVES_N, WM, WN, BKB, FUSE2>())))
convgemm_kernel(const ConvGemmArgs p) {
    constexpr int NW = WAVES_M * WAVES_N;            // waves per block
    constexpr int NTHR = 64 * NW;
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    constexpr int CPR = BKB / 16;                    // 16-byte chunks per LDS row
    constexpr int EPC = ElemTraits<TA>::EPC;         // elements per chunk
    constexpr int BK = CPR * EPC;                    // K elements per chunk row
    constexpr int KG = BKB / 32;                     // fragment groups per step
    constexpr int WPIECES = BN * CPR;
    constexpr int WITER = (WPIECES + NTHR - 1) / NTHR;
    constexpr size_t WBUF = round_kb((size_t)BN * BKB);
    static_assert(NW == 4 || NW == 8 || NW == 12, "4, 8 or 12 waves per block");
    static_assert(NTHR % CPR == 0, "column chunk of a thread is fixed");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    int tid = threadIdx.x, lane = tid & 63;          // (re-derived after the main loop by the lean-register shapes)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // 1-D tile index with the N tiles fastest.  Workgroups are dealt round-robin over the 8 XCDs (each with its own
    // L2), so workgroup ids congruent mod 8 share an L2: give each such group a CONTIGUOUS run of tiles, so the blocks
    // that read the same input rows (the N tiles of one M tile) and walk the same weights run side by side on one L2.
    const int n_tiles = (p.N + BN - 1) / BN;
    int b = blockIdx.y;
    int tile = blockIdx.x;
    int n0, m0;
    if (!(p.debug & 16)) {
        const int wg = blockIdx.y * gridDim.x + blockIdx.x;        // dispatch order: x fastest
        const int nwg = gridDim.x * gridDim.y, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
        // Which operand is worth keeping in the XCD's L2 decides the order inside its run of tiles (the launcher picks).
        // Layers whose weights exceed what an L2 holds beside the streams re-stream taps x BN x K per block: N tile
        // slowest over the WHOLE grid (tile_order 0), so the run walks down the rows of one or two N tiles.  Layers with
        // small weights: N tile fastest (1), so the blocks sharing an input tile run side by side and it comes from
        // HBM once.
        if (p.tile_order >= 2 && !(p.debug & 32)) {
            // N tiles in GROUPS of G = tile_order: the run walks down the rows with the G tiles of a row side by side, so
            // the weights of G N tiles stay in L2 (as one does with order 0) while each input row tile comes from HBM once
            // per group, not once per tile; the blocks of an XCD that run together form a (64 / G) x G patch of tiles
            const int G = p.tile_order;
            const int m_tiles = gridDim.x / n_tiles, rows_all = m_tiles * gridDim.y;
            const int full = n_tiles / G, in_full = full * G * rows_all;
            int nt, rest;
            if (id < in_full) { const int gr = id / (G * rows_all), r = id - gr * G * rows_all; rest = r / G; nt = gr * G + (r - rest * G); }
            else { const int rem = n_tiles - full * G, r = id - in_full; rest = r / rem; nt = full * G + (r - rest * rem); }
            b = rest / m_tiles;
            n0 = nt * BN;
            m0 = (rest - b * m_tiles) * BM;
        } else if ((p.tile_order == 0) != ((p.debug & 32) != 0)) {
            const int m_tiles = gridDim.x / n_tiles, per_n = m_tiles * gridDim.y;
            const int nt = id / per_n, rest = id - nt * per_n;
            b = rest / m_tiles;
            n0 = nt * BN;
            m0 = (rest - b * m_tiles) * BM;
        } else {
            b = id / (int)gridDim.x;
            tile = id - b * (int)gridDim.x;
            n0 = (tile % n_tiles) * BN;
            m0 = (tile / n_tiles) * BM;
        }
    } else {
        n0 = (tile % n_tiles) * BN;
        m0 = (tile / n_tiles) * BM;
    }

    // measurement aid: clock at phase boundaries (0 start, 1 first tiles landed, 2 main loop done, 3 h tile written,
    // 4 second contraction done, 5 epilogue done), every 64th workgroup, wave 0
    const int tl_wg = blockIdx.y * gridDim.x + blockIdx.x;
    const bool tl_on = p.timeline != nullptr && (tl_wg & 63) == 0 && (tl_wg >> 6) < p.timeline_slots && wave == 0;
    auto mark = [&](int k) {
        if (tl_on) {
            const long long t = clock64();
            if (lane == 0) p.timeline[(size_t)(tl_wg >> 6) * 8 + k] = t;
            // slots 6 / 7: the constant 100 MHz counter at start and end, which gives the shader clock the tile ran at
            if (k == 0 || k == 5) {
                const long long w = wall_clock64();
                if (lane == 0) p.timeline[(size_t)(tl_wg >> 6) * 8 + (k == 0 ? 6 : 7)] = w;
            }
        }
    };
    mark(0);
    const int halo = (p.taps - 1) * p.off_step;
    const int HR = BM + halo;                        // staged input rows
    const int xpieces = HR * CPR;
    const int xi_total = (xpieces + NTHR - 1) / NTHR;
    const int xps = (xi_total + p.taps - 1) / p.taps;   // pieces per thread per step (<= MAX_XPS, host-checked)
    // An input buffer holds the KiB-sized wave DMAs that carry staged rows; the last slice is usually partial (a two-tap
    // layer's third slice carries ONE row), and waves whose KiB of it would lie past the staged rows aim at the spare KiB.
    const int x_kb = (xpieces + 63) / 64;
    const size_t XBUF = (size_t)x_kb * 1024;
    const bool last_slice_real = (xi_total - 1) * NW + wave < x_kb;       // this wave's part of the last slice holds staged rows
    // the fused unit's h tile: shared (tile-sized) when several waves split the columns, wave-private 32-row pieces otherwise
    const size_t hs_bytes = WAVES_N == 1 ? (size_t)NW * 32 * (BN * sizeof(TA) + 16) : (size_t)BM * (BN * sizeof(TA) + 16);
    // single-tap layers (k = 1 convs, linears) use a fresh input tile every step: three buffers, fetched two steps
    // ahead like the weights; multi-tap layers keep a tile for `taps` steps: two buffers, the next one filled in slices
    // layers of two or three taps also get three (when they fit): their chunks are too short to hide an HBM fetch
    const int nxb = p.nxb;
    // LDS map: three weight buffers, then the input buffers.  The fused unit's h tile starts at the THIRD weight buffer
    // (its second contraction cycles through the first two only) and runs over the input buffers, idle by then.
    const size_t main_bytes = 3 * WBUF + round_kb(nxb * XBUF);
    const size_t fuse_bytes = (WAVES_N == 1 ? 3 : 2) * WBUF + round_kb(hs_bytes);

    unsigned char* Ws0 = smem;                       // three weight buffers
    unsigned char* Xs0 = smem + 3 * WBUF;            // the input buffers

    const TA* __restrict__ Xb = (const TA*)p.X + (long long)b * p.x_batch_stride;
    const TA* __restrict__ Wg = (const TA*)p.W;
    const int nkc = (p.K + BK - 1) / BK;
    const int nsteps = nkc * p.taps;
    const int my_c = tid % CPR;                      // column chunk of every piece this thread stages
    const bool dma = (p.K % BK) == 0;                // whole chunks only: LDS-DMA cannot zero-fill a K tail

    const int g0 = m0 + p.off0;                      // sequence row of staged row 0

    // ---- input tile, register path (K tail only) ----
    auto x_load = [&](int id, int kc) -> uint4 {
        int row = id / CPR;
        int grow = g0 + row;
        int kel = kc * BK + my_c * EPC;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (id < xpieces && grow >= p.row_lo && grow < p.L_in && kel < p.K) {
            v = *(const uint4*)(Xb + (long long)grow * p.ldx + kel);
            if constexpr (FUSE2) {                   // the unit's first SnakeBeta, here in registers
                TA* e = (TA*)&v;
#pragma unroll
                for (int i = 0; i < EPC; ++i)
                    if (kel + i < p.K) e[i] = q3_from_f32<TA>(q3_snake<TA>(q3_to_f32(e[i]), p.pre_a[kel + i], p.pre_invb[kel + i]));
            }
        }
        return v;
    };
    auto x_store = [&](unsigned char* xs, int id, uint4 v) {
        if (id < xpieces) {
            int row = id / CPR;
            *(uint4*)(xs + (size_t)row * BKB + swz<CPR>(row, my_c) * 16) = v;
        }
    };
    // ---- weight tile ----
    auto w_load = [&](const TA* W, int ldw, int id, int kc, int tap) -> uint4 {
        int row = id / CPR, c = id % CPR;
        int n = n0 + row;
        int kel = kc * BK + c * EPC;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (id < WPIECES && n < p.N && kel < ldw)
            v = *(const uint4*)(W + ((long long)tap * p.N + n) * ldw + kel);
        return v;
    };
    auto w_store = [&](unsigned char* ws, int id, uint4 v) {
        if (id < WPIECES) {
            int row = id / CPR, c = id % CPR;
            *(uint4*)(ws + (size_t)row * BKB + swz<CPR>(row, c) * 16) = v;
        }
    };
    // ---- LDS-DMA staging.  LDS destinations are lane-linear (1 KiB per wave instruction), so the chunk XOR is applied
    //      to the SOURCE address.  Every call issues a fixed number of DMA instructions per wave (waves with nothing to
    //      fetch aim a padding DMA at a spare KiB) so that the vmcnt bookkeeping needs no per-wave counts.
    const unsigned lds0 = lds_address(smem);
    const unsigned ws_lds = lds0, xs_lds = lds0 + 3 * (unsigned)WBUF;
    const unsigned spare_lds = lds0 + (unsigned)((FUSE2 && fuse_bytes > main_bytes) ? fuse_bytes : main_bytes);
    const unsigned char* zero_page = (const unsigned char*)p.zeros;
    constexpr int XROWS_PER_IT = NTHR / CPR;                     // rows one iteration of all waves covers (a multiple of 16)
    const int xrow0 = tid / CPR;
    // a tile whose staged rows (padding lanes included) all lie inside the sequence needs no per-lane checks
    const bool interior = g0 >= p.row_lo && (g0 + xi_total * XROWS_PER_IT) <= p.L_in;
    const TA* xtile = Xb + (long long)g0 * p.ldx;                                      // wave-uniform
    const unsigned xvoff0 = (unsigned)((xrow0 * p.ldx + swz<CPR>(xrow0, my_c) * EPC) * (int)sizeof(TA));
    // input rows [it*XROWS_PER_IT ...) of chunk kc -> buffer at xs; rows outside the sequence come from the zero page
    auto x_dma_issue = [&](unsigned xs, int it, int kc) {
        const int slot = it * NW + wave;
        if (it >= xi_total || slot >= x_kb) { dma16_so(zero_page, 0u, spare_lds); return; }     // nothing (left) to stage: padding DMA
        const TA* sbase = xtile + ((long long)it * XROWS_PER_IT * p.ldx + kc * BK);
        if (interior) {
            dma16_so(sbase, xvoff0, xs + (unsigned)slot * 1024u);
        } else {
            const int grow = g0 + xrow0 + it * XROWS_PER_IT;
            const unsigned long long real = (unsigned long long)sbase + xvoff0;
            const unsigned long long src = (grow >= p.row_lo && grow < p.L_in) ? real : (unsigned long long)zero_page;
            dma16((const void*)src, xs + (unsigned)slot * 1024u);
        }
    };
    // ---- FUSE2: the unit's first SnakeBeta on the staged input tile.  A lane transforms, in place, exactly the pieces
    //      its own DMAs wrote (lane-linear destinations), after its own vmcnt wait and before the barrier that publishes
    //      the tile - no other synchronisation.  Its channels within a K chunk are the same for every slice (the slices
    //      are a multiple of 16 rows apart, so the source swizzle repeats); rows outside the sequence hold zeros, which
    //      SnakeBeta maps to zero.  The per-channel parameters sit in LDS behind the spare KiB: [K] frequencies, [K] gains.
    float* pre_lds = (float*)(smem + (spare_lds - lds0) + 1024);
    const int pre_c0 = swz<CPR>(xrow0, my_c) * EPC;
    auto x_snake = [&](unsigned char* xs, int it, int kc) {
        if constexpr (FUSE2) {
            const int slot = it * NW + wave;
            if (it >= xi_total || slot >= x_kb) return;                  // a padding DMA: nothing staged
            uint4* pp = (uint4*)(xs + (size_t)slot * 1024 + lane * 16);
            uint4 v = *pp;
            const float* pa = pre_lds + kc * BK + pre_c0;
            const float* pb = pa + p.K;
            float fa[EPC], fb[EPC];
#pragma unroll
            for (int i = 0; i < EPC; i += 4) {
                const float4 t = *(const float4*)(pa + i), u = *(const float4*)(pb + i);
                fa[i] = t.x; fa[i + 1] = t.y; fa[i + 2] = t.z; fa[i + 3] = t.w;
                fb[i] = u.x; fb[i + 1] = u.y; fb[i + 2] = u.z; fb[i + 3] = u.w;
            }
            TA* e = (TA*)&v;
#pragma unroll
            for (int i = 0; i < EPC; ++i) e[i] = q3_from_f32<TA>(q3_snake<TA>(q3_to_f32(e[i]), fa[i], fb[i]));
            *pp = v;
        }
    };
    int late0 = 0;                       // see after the main loop
    unsigned wvoff[WITER], w2voff[WITER];
#pragma unroll
    for (int q = 0; q < WITER; ++q) {
        const int id = (q * NW + wave) * 64 + lane;
        const int row = id / CPR, cs = id % CPR;
        int n = n0 + row;
        n = n < p.N ? n : p.N - 1;                               // columns past N are never stored: any finite row will do
        wvoff[q] = (unsigned)((n * p.K + swz<CPR>(row, cs) * EPC) * (int)sizeof(TA));
        w2voff[q] = (unsigned)((n * p.N + swz<CPR>(row, cs) * EPC) * (int)sizeof(TA));
    }
    auto w2voff_late = [&]() {           // the same offsets recomputed where they are used (lean-register shapes: not carried through the main loop)
#pragma unroll
        for (int q = 0; q < WITER; ++q) {
            const int id = (q * NW + wave) * 64 + lane;
            const int row = id / CPR, cs = id % CPR;
            int n = n0 + row;
            n = n < p.N ? n : p.N - 1;
            w2voff[q] = (unsigned)((n * p.N + swz<CPR>(row, cs) * EPC) * (int)sizeof(TA)) + (unsigned)late0;
        }
    };
    auto w_dma_one = [&](unsigned ws, const TA* sbase, const unsigned* voff, int q) {
        if (WITER * NW * 64 == WPIECES || (q * NW + wave) * 64 < WPIECES) dma16_so(sbase, voff[q], ws + (unsigned)(q * NW + wave) * 1024u);
        else dma16_so(zero_page, 0u, spare_lds);
    };
    auto w_dma_issue = [&](unsigned ws, const TA* sbase, const unsigned* voff) {
#pragma unroll
        for (int q = 0; q < WITER; ++q) w_dma_one(ws, sbase, voff, q);
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // MFMAs of one step: every wave multiplies its rows of the staged input (shifted by the tap) with the weight tile.
    // The fragments of group kg + 1 are read from LDS before the MFMAs of group kg issue, so a wave's LDS latency hides
    // under its own matrix work.  `between(kg)` runs ahead of group kg: the pipeline issues its LDS-DMAs there, spread
    // over the groups, so the address unit works through them while the matrix cores run.
    auto compute = [&](const unsigned char* xs, const unsigned char* ws, int row_shift, auto&& between) {
        const int arow0 = wm * 32 * WM + (lane & 31) + row_shift;
        const int brow0 = wn * 32 * WN + (lane & 31);
        uint4 af[2][WM], bfr[2][WN];
        auto load = [&](int buf, int kg) {
            const int c = kg * 2 + (lane >> 5);
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                int row = arow0 + i * 32;
                af[buf][i] = *(const uint4*)(xs + (size_t)row * BKB + swz<CPR>(row, c) * 16);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                int row = brow0 + j * 32;
                bfr[buf][j] = *(const uint4*)(ws + (size_t)row * BKB + swz<CPR>(row, c) * 16);
            }
        };
        load(0, 0);
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            between(kg);
            if (kg + 1 < KG) load((kg + 1) & 1, kg + 1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) Mma<TA>::run(af[kg & 1][i], bfr[kg & 1][j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        }
    };

    // The fused unit's inner bias and SnakeBeta parameters are fetched HERE and pinned right after the main loop (where
    // the DMA queue is empty): fetched after the main loop, the compiler's wait for them is a vmcnt(0) that also waits
    // for the residual rows and the W2 tile requested just before - a round trip to HBM in front of the inner activation.
    // Shapes that run three waves per SIMD have 168 registers per lane.  NOTHING may spill in this kernel: the rows and
    // parameters requested through inline asm are in flight without the compiler knowing, and a register it spills and hands to
    // another value while such a load is pending gets overwritten when the load lands.  So these shapes keep less in
    // registers: the inner activation's parameters wait in LDS ([BN] frequencies, [BN] gains, [BN] biases behind the input
    // SnakeBeta's, zeros past N; an LDS read involves no vmcnt), residual rows are requested one pass later, and the folded
    // final conv reads its taps per pass.
    constexpr bool LEAN_REGS = FUSE2 && waves_per_simd<TA, TO, NW, WM, WN, BKB, FUSE2>() == 3;
    constexpr bool INNER_LDS = FUSE2 && WAVES_N == 1 && LEAN_REGS;
    float* inn_lds = pre_lds + 2 * p.K;
    float ia[WN], iib[WN], ibias[WN];                           // the fused unit's inner bias and SnakeBeta
    auto inner_params = [&]() {                                 // called behind the prologue's requests: the first tiles go first
        if constexpr (FUSE2 && !INNER_LDS) {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = wn * 32 * WN + j * 32 + (lane & 31);
                ia[j] = n < p.N ? p.epi_a[n] : 0.f;
                iib[j] = n < p.N ? p.epi_invb[n] : 0.f;
                ibias[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
            }
        }
    };
    if (dma) {
        // ---- LDS-DMA pipeline: weight tiles two steps ahead in three buffers, the next chunk's input tile in slices
        //      spread over this chunk's steps; each step waits with vmcnt(n) for all but the DMAs that may stay in
        //      flight, then one raw barrier.  The slice count per step is a compile-time constant of the loop (one
        //      instantiation per value) so that the steady-state waits are immediates and the slice loop unrolls.
        const long long tap_stride = (long long)p.N * p.K;       // elements between taps of W
        const long long slice_stride = (long long)XROWS_PER_IT * p.ldx;   // elements between input slices
        // single-tap steady state needs at least one step with a tile two steps on
        const bool lean1 = p.taps == 1 && nxb == 3 && interior && nsteps >= 3;
        const bool lean2 = p.taps > 1 && nxb == 3 && interior && nkc >= 3;       // multi-tap, input tiles two chunks ahead
        const int x_ahead = lean2 ? 2 : 1;
        // seven-tap layers (every k = 7 convolution of the decoder) on interior tiles run the unrolled loop below
        constexpr int TAPS_S = 7;
        constexpr int XI_S = ((BM + 54) * CPR + NTHR - 1) / NTHR;          // input slices of a 7-tap tile with dilation <= 9
        const bool unrolled7 = XI_S <= TAPS_S - 2 && p.taps == TAPS_S && nxb == 2 && interior && xi_total == XI_S;
        // one- and two-tap layers with three input buffers (run_static below)
        constexpr int XI_1 = (BM * CPR + NTHR - 1) / NTHR, XI_2 = ((BM + 1) * CPR + NTHR - 1) / NTHR;
        const bool static1 = lean1 && xi_total == XI_1;
        const bool static2 = lean2 && p.taps == 2 && p.off_step == 1 && xi_total == XI_2;
        // FUSE2: the input SnakeBeta's parameters (2K floats) requested FIRST, through inline asm like the residual rows:
        // loads retire in order, so every prologue wait below (at most the DMAs issued after this) covers them
        u32x4 ppre;
        const int npre4 = p.K / 2;                                       // float4 pieces: K / 4 of frequencies, then K / 4 of gains
        if constexpr (FUSE2) {
            const int q = tid < npre4 ? tid : 0;
            const float* src = q < (npre4 >> 1) ? p.pre_a + 4 * q : p.pre_invb + 4 * (q - (npre4 >> 1));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ppre) : "v"(src) : "memory");
        }
        u32x4 pinn;
        constexpr int ninn4 = 3 * BN / 4;                                // float4 pieces of [BN a | BN invb | BN bias]
        if constexpr (INNER_LDS) {      // every lane loads (a piece past N or of a missing bias: the zero page): no branch around the request
            const int q = tid < ninn4 ? tid : 0, arr = q / (BN / 4), n4 = 4 * (q - arr * (BN / 4));
            const float* base = arr == 0 ? p.epi_a : (arr == 1 ? p.epi_invb : p.bias);
            const float* src = (base != nullptr && n4 < p.N) ? base + n4 : (const float*)p.zeros;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pinn) : "v"(src) : "memory");
        }
        for (int it = 0; it < xi_total; ++it) x_dma_issue(xs_lds, it, 0);
        w_dma_issue(ws_lds, Wg, wvoff);
        inner_params();
        if (nsteps > 1) {
            w_dma_issue(ws_lds + (unsigned)WBUF, Wg + (p.taps > 1 ? tap_stride : (long long)BK), wvoff);
            if (lean1 || lean2) for (int it = 0; it < xi_total; ++it) x_dma_issue(xs_lds + (unsigned)XBUF, it, 1);
            if (unrolled7) wait_vmcnt(0);
            else wait_vmcnt(WITER + ((lean1 || lean2) ? xi_total : 0));
        } else {
            wait_vmcnt(0);
        }
        if constexpr (FUSE2) {      // parameters -> LDS, published by a barrier of their own, then chunk 0 of the input tile
            asm volatile("" : "+v"(ppre));
            if (tid < npre4) *(u32x4*)(pre_lds + 4 * tid) = ppre;
            if constexpr (INNER_LDS) {
                asm volatile("" : "+v"(pinn));
                if (tid < ninn4) *(u32x4*)(inn_lds + 4 * tid) = pinn;
            }
            block_barrier();
            for (int it = 0; it < xi_total; ++it) x_snake(Xs0, it, 0);
        }
        block_barrier();
        mark(1);
        // pipeline state, shared by the loops below
        int step = 0, kc = 0, tap = 0, wb = 0, xb = 0;           // wb / xb: weight / input buffer of the current step
        int x_issued = (lean1 || lean2) ? 1 : 0;                 // highest chunk whose input tile has been requested
        int ltap = p.taps > 2 ? 2 : 0, lkc = p.taps > 2 ? 0 : (p.taps == 2 ? 1 : 2);   // (tap, chunk) of step + 2
        const TA* wsrc = Wg + (ltap * tap_stride + (long long)lkc * BK);               // weight tile of step + 2
        const TA* xsrc = xtile + (long long)x_ahead * BK;        // slice `xit` of the chunk being fetched (interior tiles)
        int xit = 0;
        auto advance = [&](bool boundary, int xps_c) {
            if (boundary) { tap = 0; ++kc; xit = 0; xsrc = xtile + (long long)(kc + x_ahead) * BK; xb = xb + 1 == nxb ? 0 : xb + 1; }
            else { ++tap; xit += xps_c; xsrc += xps_c * slice_stride; }
            wsrc += tap_stride;
            if (++ltap == p.taps) { ltap = 0; ++lkc; wsrc += (long long)BK - p.taps * tap_stride; }
            wb = wb == 2 ? 0 : wb + 1;
        };
        // Steady state of an interior tile (every staged row inside the sequence, a next chunk and a weight tile two
        // steps on still to fetch): no per-lane or per-step conditions, immediates for the waits.  Issue slots are the
        // scarce resource next to the matrix pipe - every instruction a wave issues here adds to the step's time.
        auto lean = [&](auto xps_tag, int lean_steps) {
            constexpr int XPS = decltype(xps_tag)::value;
            const unsigned xdst_w = xs_lds + (unsigned)wave * 1024u, wdst_w = ws_lds + (unsigned)wave * 1024u;
            // steps between the request of a chunk's last real slice and the chunk's first use (two chunks ahead: plenty)
            const int slack = lean2 ? 2 : p.taps - (xi_total + XPS - 1) / XPS;
            for (; step < lean_steps; ++step) {
                const unsigned xsn = xdst_w + (unsigned)((lean2 ? (xb == 0 ? 2 : xb - 1) : (xb ^ 1)) * XBUF);
                const unsigned wsn = wdst_w + (unsigned)((wb == 0 ? 2 : wb - 1) * WBUF);
                // this step's DMAs, oldest first: the weight tile of step + 2 (groups 0..), then the next chunk's input slices
                // (last group).  Loads retire in order, so a weight tile can only be waited for together with everything
                // issued before it: the slices, which come from HBM and take longest, go last and get a whole extra step.
                compute(Xs0 + (size_t)xb * XBUF, Ws0 + (size_t)wb * WBUF, tap * p.off_step, [&](int kg) {
#pragma unroll
                    for (int q = 0; q < WITER; ++q)
                        if ((q < KG ? q : KG - 1) == kg) {
                            if (WITER * NW * 64 == WPIECES || (q * NW + wave) * 64 < WPIECES) dma16_so(wsrc, wvoff[q], wsn + (unsigned)(q * NW) * 1024u);
                            else dma16_so(zero_page, 0u, spare_lds);
                        }
                    if (kg == KG - 1) {
#pragma unroll
                        for (int i = 0; i < XPS; ++i) {
                            if (xit + i < xi_total - 1 || (xit + i == xi_total - 1 && last_slice_real))
                                dma16_so(xsrc + i * slice_stride, xvoff0, xsn + (unsigned)((xit + i) * NW) * 1024u);
                            else dma16_so(zero_page, 0u, spare_lds);
                        }
                    }
                });
                // The next step reads weight tile step + 1, issued a step ago and followed since by that step's slices,
                // this step's weight tile and this step's slices: those may stay in flight.  A step that closes a chunk
                // also needs the chunk's last real slice; when the slices ran out two or more steps ago that is implied.
                const bool boundary = (tap + 1) == p.taps;
                if (step == 0) wait_vmcnt(WITER + XPS);
                else if (boundary && slack < 2) wait_vmcnt(slack == 1 ? WITER + XPS : 0);
                else wait_vmcnt_c<WITER + 2 * XPS>();
                if constexpr (FUSE2) {      // the next chunk's tile has landed (this wave's part): its SnakeBeta, then publish
                    if (boundary)
                        for (int it = 0; it < xi_total; ++it) x_snake(Xs0 + (size_t)(xb ^ 1) * XBUF, it, kc + 1);
                }
                block_barrier();
                advance(boundary, XPS);
            }
            x_issued = kc + x_ahead - 1;    // the general loop requests chunk kc + 1 itself when it is not among these
        };
        // Single-tap steady state: input tile AND weight tile of step + 2 requested every step, three buffers each.
        auto lean_single = [&](auto xps_tag, int lean_steps) {
            constexpr int XPS = decltype(xps_tag)::value;            // == xi_total: the whole tile every step
            const unsigned xdst_w = xs_lds + (unsigned)wave * 1024u, wdst_w = ws_lds + (unsigned)wave * 1024u;
            const TA* xsrc2 = xtile + 2 * BK;                        // input tile of step + 2
            for (; step < lean_steps; ++step) {
                const unsigned xsn = xdst_w + (unsigned)((xb == 0 ? 2 : xb - 1) * XBUF);
                const unsigned wsn = wdst_w + (unsigned)((wb == 0 ? 2 : wb - 1) * WBUF);
                compute(Xs0 + (size_t)xb * XBUF, Ws0 + (size_t)wb * WBUF, 0, [&](int kg) {
#pragma unroll
                    for (int q = 0; q < WITER; ++q)
                        if ((q < KG ? q : KG - 1) == kg) {
                            if (WITER * NW * 64 == WPIECES || (q * NW + wave) * 64 < WPIECES) dma16_so(wsrc, wvoff[q], wsn + (unsigned)(q * NW) * 1024u);
                            else dma16_so(zero_page, 0u, spare_lds);
                        }
                    // the tile's slices spread over the later groups
#pragma unroll
                    for (int i = 0; i < XPS; ++i)
                        if ((KG - 1 - (XPS - 1 - i) * (KG - 1) / (XPS > 1 ? XPS - 1 : 1)) == kg)
                            dma16_so(xsrc2 + i * slice_stride, xvoff0, xsn + (unsigned)(i * NW) * 1024u);
                });
                wait_vmcnt_c<WITER + XPS>();       // all but this step's requests: tiles of step + 1 are in
                block_barrier();
                xsrc2 += BK;
                advance(true, XPS);
            }
            x_issued = nkc - 1;
        };
        // Seven taps, interior tile: the seven steps of a chunk unrolled.  Tap, slice and weight-tile bookkeeping are then
        // compile-time, the buffer rotation is register renaming and every wait is an immediate; the last chunk (no input
        // tile left to request, weight requests ending two steps early) is a second unrolled copy.  Requests in a step:
        // the weight tile of step + 2 (spread over the fragment groups), then slice `tap` of the next chunk's input tile.
        // The wait closing a step lets this step's requests and the previous step's slice stay in flight; the slices end
        // at tap XI - 1 <= 4, so the chunk's last step finds the whole next tile landed.
        auto run_unrolled7 = [&]() {
            constexpr int TAPS = TAPS_S, XI = XI_S < TAPS_S - 2 ? XI_S : TAPS_S - 2;
            const unsigned xdst_w = xs_lds + (unsigned)wave * 1024u, wdst_w = ws_lds + (unsigned)wave * 1024u;
            unsigned wo0 = 0, wo1 = (unsigned)WBUF, wo2 = 2 * (unsigned)WBUF;      // weight buffers of step, step + 1, step + 2
            const TA* wnext = Wg + 2 * tap_stride;                                  // weight tile of step + 2
            int off_step = p.off_step;
            // FUSE2: slice s of the next tile is requested in step s and has landed, for this wave, by the wait closing
            // step s + 2; its SnakeBeta runs at the head of step s + 3, under that step's MFMAs, and for the slices
            // with s + 3 past the chunk's last step right after the last step's wait.
            auto one = [&](auto tap_tag, auto last_tag, const unsigned char* xs, unsigned xsn, const TA* xnext, unsigned char* xs_next, int kc_next) {
                constexpr int tap = decltype(tap_tag)::value;
                constexpr bool last = decltype(last_tag)::value;
                constexpr bool issue_w = !last || tap + 2 < TAPS;
                constexpr bool issue_x = !last && tap < XI;
                const unsigned wsn = wdst_w + wo2;
                compute(xs, Ws0 + wo0, tap * off_step, [&](int kg) {
                    if constexpr (FUSE2 && !last && tap >= 3 && tap - 3 < XI) {
                        if (kg == 0) x_snake(xs_next, tap - 3, kc_next);
                    }
                    if (issue_w) {
#pragma unroll
                        for (int q = 0; q < WITER; ++q)
                            if ((q < KG ? q : KG - 1) == kg) {
                                if (WITER * NW * 64 == WPIECES || (q * NW + wave) * 64 < WPIECES) dma16_so(wnext, wvoff[q], wsn + (unsigned)(q * NW) * 1024u);
                                else dma16_so(zero_page, 0u, spare_lds);
                            }
                    }
                    if (issue_x && kg == KG - 1) {
                        if (tap < XI - 1 || last_slice_real) dma16_so(xnext + tap * slice_stride, xvoff0, xsn + (unsigned)(tap * NW) * 1024u);
                        else dma16_so(zero_page, 0u, spare_lds);
                    }
                });
                constexpr int in_flight = last ? (issue_w ? WITER : 0)
                                               : WITER + (tap < XI ? 1 : 0) + ((tap >= 1 && tap - 1 < XI) ? 1 : 0);
                wait_vmcnt_c<in_flight>();
                if constexpr (FUSE2 && !last && tap == TAPS - 1) {
#pragma unroll
                    for (int sl = TAPS - 3; sl < XI; ++sl) x_snake(xs_next, sl, kc_next);
                }
                block_barrier();
                if (issue_w) wnext += ((tap + 3) % TAPS == 0) ? ((long long)BK - (TAPS - 1) * tap_stride) : tap_stride;
                const unsigned t = wo0; wo0 = wo1; wo1 = wo2; wo2 = t;
            };
            auto chunk = [&](auto last_tag, const unsigned char* xs, unsigned xsn, const TA* xnext, unsigned char* xs_next, int kc_next) {
                // With 128 registers per wave the seven taps' fragment addresses must not be kept across chunks (they
                // spill, and a reload's wait drains the DMA queue): hide the tap offset's loop invariance from the compiler.
                if constexpr (waves_per_simd<TA, TO, NW, WM, WN, BKB, FUSE2>() >= 3) asm volatile("" : "+s"(off_step));
                one(std::integral_constant<int, 0>{}, last_tag, xs, xsn, xnext, xs_next, kc_next);
                one(std::integral_constant<int, 1>{}, last_tag, xs, xsn, xnext, xs_next, kc_next);
                one(std::integral_constant<int, 2>{}, last_tag, xs, xsn, xnext, xs_next, kc_next);
                one(std::integral_constant<int, 3>{}, last_tag, xs, xsn, xnext, xs_next, kc_next);
                one(std::integral_constant<int, 4>{}, last_tag, xs, xsn, xnext, xs_next, kc_next);
                one(std::integral_constant<int, 5>{}, last_tag, xs, xsn, xnext, xs_next, kc_next);
                one(std::integral_constant<int, 6>{}, last_tag, xs, xsn, xnext, xs_next, kc_next);
            };
            int xbs = 0;
            for (int c = 0; c + 1 < nkc; ++c) {
                chunk(std::false_type{}, Xs0 + (size_t)xbs * XBUF, xdst_w + (unsigned)((xbs ^ 1) * XBUF), xtile + (long long)(c + 1) * BK,
                      Xs0 + (size_t)(xbs ^ 1) * XBUF, c + 1);
                xbs ^= 1;
            }
            chunk(std::true_type{}, Xs0 + (size_t)xbs * XBUF, 0u, xtile, Xs0, 0);
        };
        // One- and two-tap layers (linears, k = 1 convs, transposed convs) on interior tiles: the same idea with three input
        // buffers.  Chunk kc + 2's input tile is requested during chunk kc (all of it in a single-tap step, split over the
        // two steps otherwise), the weight tile of step + 2 during every step; the wait closing a step lets exactly one
        // step's worth of requests stay in flight (WITER weight + XI input DMAs, whichever two steps they came from).
        auto run_static = [&](auto taps_tag) {
            constexpr int TAPS = decltype(taps_tag)::value;
            constexpr int XI = TAPS == 1 ? XI_1 : XI_2;
            constexpr int X0N = (XI + TAPS - 1) / TAPS;                            // slices requested in tap 0
            const unsigned xdst_w = xs_lds + (unsigned)wave * 1024u, wdst_w = ws_lds + (unsigned)wave * 1024u;
            unsigned wo0 = 0, wo1 = (unsigned)WBUF, wo2 = 2 * (unsigned)WBUF;      // weight buffers of step, step + 1, step + 2
            unsigned xo0 = 0, xo1 = (unsigned)XBUF, xo2 = 2 * (unsigned)XBUF;      // input buffers of chunk kc, kc + 1, kc + 2
            // weight tiles of step + 2: single tap -> chunk kc + 2; two taps -> the same tap of chunk kc + 1
            const TA* wn0 = TAPS == 1 ? Wg + 2 * BK : Wg + BK;
            const TA* wn1 = wn0 + tap_stride;
            const TA* xnext = xtile + 2 * BK;                                      // chunk kc + 2, slice 0
            const int off_step = p.off_step;
            for (int c = 0; c < nkc; ++c) {
                const bool more_x = c + 2 < nkc;
                auto one = [&](auto tap_tag) {
                    constexpr int tap = decltype(tap_tag)::value;
                    const bool more_w = c * TAPS + tap + 2 < nsteps;
                    const TA* wnext = tap == 0 ? wn0 : wn1;
                    const unsigned wsn = wdst_w + wo2, xsn = xdst_w + xo2;
                    compute(Xs0 + xo0, Ws0 + wo0, tap * off_step, [&](int kg) {
                        if (more_w) {
#pragma unroll
                            for (int q = 0; q < WITER; ++q)
                                if ((q < KG ? q : KG - 1) == kg) {
                                    if (WITER * NW * 64 == WPIECES || (q * NW + wave) * 64 < WPIECES) dma16_so(wnext, wvoff[q], wsn + (unsigned)(q * NW) * 1024u);
                                    else dma16_so(zero_page, 0u, spare_lds);
                                }
                        }
                        if (more_x && kg == KG - 1) {
#pragma unroll
                            for (int i = (tap == 0 ? 0 : X0N); i < (tap == 0 ? X0N : XI); ++i) {
                                if (i < XI - 1 || last_slice_real) dma16_so(xnext + i * slice_stride, xvoff0, xsn + (unsigned)(i * NW) * 1024u);
                                else dma16_so(zero_page, 0u, spare_lds);
                            }
                        }
                    });
                    if (more_w && more_x) wait_vmcnt_c<WITER + XI>();
                    else if (more_w) wait_vmcnt_c<WITER>();
                    else wait_vmcnt_c<0>();
                    block_barrier();
                    const unsigned t = wo0; wo0 = wo1; wo1 = wo2; wo2 = t;
                };
                one(std::integral_constant<int, 0>{});
                if constexpr (TAPS == 2) one(std::integral_constant<int, 1>{});
                wn0 += BK; wn1 += BK; xnext += BK;
                const unsigned t = xo0; xo0 = xo1; xo1 = xo2; xo2 = t;
            }
        };
        auto pipeline = [&](auto xps_tag) {
            constexpr int XPS = decltype(xps_tag)::value;
            for (; step < nsteps; ++step) {
                const int wb2 = wb == 0 ? 2 : wb - 1;
                const bool more_x = (kc + 1) < nkc && (kc + 1) > x_issued, more_w = (step + 2) < nsteps;
                const unsigned xsn = xs_lds + (unsigned)((xb + 1 == nxb ? 0 : xb + 1) * XBUF);
                const unsigned wsn = ws_lds + (unsigned)(wb2 * WBUF);
                // same issue order as the steady-state loop; the waits let only THIS step's DMAs stay in flight, which is
                // right whatever the steps before left outstanding
                compute(Xs0 + (size_t)xb * XBUF, Ws0 + (size_t)wb * WBUF, tap * p.off_step, [&](int kg) {
                    if (more_w) {
#pragma unroll
                        for (int q = 0; q < WITER; ++q)
                            if ((q < KG ? q : KG - 1) == kg) w_dma_one(wsn, wsrc, wvoff, q);
                    }
                    if (kg == KG - 1 && more_x) {
#pragma unroll 1
                        for (int i = 0; i < XPS; ++i) x_dma_issue(xsn, xit + i, kc + 1);
                    }
                });
                const bool boundary = (tap + 1) == p.taps;
                if (boundary) wait_vmcnt(more_x ? 0 : (more_w ? WITER : 0));
                else wait_vmcnt((more_w ? WITER : 0) + (more_x ? XPS : 0));
                if constexpr (FUSE2) {
                    if (boundary && kc + 1 < nkc)
                        for (int it = 0; it < xi_total; ++it) x_snake(Xs0 + (size_t)(xb + 1 == nxb ? 0 : xb + 1) * XBUF, it, kc + 1);
                }
                block_barrier();
                advance(boundary, XPS);
            }
        };
        const int lean_steps = lean1 ? nsteps - 2 : lean2 ? (nkc - 2) * p.taps
                             : (interior && nkc >= 2 && p.taps >= 2 && nxb == 2) ? (nkc - 1) * p.taps : 0;
        auto run = [&](auto tag) {
            if (lean1) lean_single(tag, lean_steps); else lean(tag, lean_steps);
            pipeline(tag);
        };
        if (unrolled7) run_unrolled7();
        else if (static1) run_static(std::integral_constant<int, 1>{});
        else if (static2) run_static(std::integral_constant<int, 2>{});
        else switch (xps) {
            case 1: run(std::integral_constant<int, 1>{}); break;
            case 2: run(std::integral_constant<int, 2>{}); break;
            case 3: run(std::integral_constant<int, 3>{}); break;
            default: run(std::integral_constant<int, 4>{}); break;
        }
    } else {
        // ---- K tail (narrow test shapes): synchronous register staging, zero-filled past K ----
        inner_params();
        if constexpr (INNER_LDS) {          // published by the loop's barriers (there is at least one step)
            for (int idx = tid; idx < 3 * BN; idx += NTHR) {
                const int arr = idx / BN, n = idx - arr * BN;
                const float* base = arr == 0 ? p.epi_a : (arr == 1 ? p.epi_invb : p.bias);
                inn_lds[idx] = (base != nullptr && n < p.N) ? base[n] : 0.f;
            }
        }
        int kc = 0, tap = 0;
        for (int step = 0; step < nsteps; ++step) {
            if (tap == 0)
                for (int it = 0; it < xi_total; ++it) x_store(Xs0, tid + it * NTHR, x_load(tid + it * NTHR, kc));
#pragma unroll
            for (int it = 0; it < WITER; ++it) w_store(Ws0, tid + it * NTHR, w_load(Wg, p.K, tid + it * NTHR, kc, tap));
            __syncthreads();
            compute(Xs0, Ws0, tap * p.off_step, [](int) {});
            __syncthreads();
            if (++tap == p.taps) { tap = 0; ++kc; }
        }
    }

    if constexpr (LEAN_REGS) {
        // the lane number from the execution mask counters instead of a register carried through the main loop
        lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        tid = wave * 64 + lane;
    }
    const int half = lane >> 5, l31 = lane & 31;
    mark(2);
    // lean-register shapes: the epilogue's per-lane row offsets must not be computed before the main loop and carried through
    // it (they are loop-invariant, so the compiler would): an opaque zero defined HERE goes into every one of them
    if constexpr (LEAN_REGS) asm volatile("" : "+v"(late0));
    if constexpr (FUSE2 && !INNER_LDS) {      // the compiler's wait for the parameters lands here, where the DMA queue is empty
#pragma unroll
        for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(ia[j]), "+v"(iib[j]), "+v"(ibias[j]));
    }

    // ---- output addressing (used by the fused unit's residual prefetch and by the epilogue) ----------------------
    TO* __restrict__ Yb = (TO*)p.Y + (long long)b * p.y_batch_stride;
    TO* __restrict__ Y2b = p.Y2 ? (TO*)p.Y2 + (long long)b * p.y2_batch_stride : nullptr;
    const TO* __restrict__ Rb = p.R ? (const TO*)p.R + (long long)b * p.r_batch_stride : nullptr;
    constexpr int EPO = 16 / (int)sizeof(TO);            // output elements per 16-byte piece
    constexpr int JG = WN <= 3 ? WN : 3;                  // column tiles per slab pass
    constexpr int SCOLS = 32 * JG;
    constexpr int SLD = SCOLS + 4;                        // slab row stride in floats (keeps rows 16-byte aligned)
    constexpr int SROWS = 16;
    static_assert(WN % JG == 0, "slab passes must tile the wave's columns");
    float* slabA = (float*)smem + wave * (2 * SROWS * SLD);
    float* slabB = slabA + SROWS * SLD;
    // The fused residual unit always has a residual, no SwiGLU and 16-byte-aligned rows (the launcher checks): with those
    // known at compile time its epilogue loops have constant trip counts; the consumer's snake (only the last unit of a
    // block applies one) is a wave-uniform choice per group of eight elements.
    const bool swiglu = !FUSE2 && p.act == Q3_ACT_SWIGLU;
    const int n_out = swiglu ? p.N / 2 : p.N;
    const bool post_elu = !FUSE2 && p.post_elu;           // the post activation is ELU, not SnakeBeta
    const bool has_post = p.post_a != nullptr || post_elu;
    const bool has_res = FUSE2 || Rb != nullptr;
    const bool vec_ok = FUSE2 || ((p.ldy % EPO == 0) && (!Rb || p.ldr % EPO == 0) && (n_out % EPO == 0));
    const float* obias = FUSE2 ? p.bias2 : p.bias;
    const int first_act = FUSE2 ? Q3_ACT_NONE : p.act;    // FUSE2 consumed the activation before its second contraction
    // Segmented rows (streams stacked in one row space, each segment = carried rows + new rows): GEMM row m belongs
    // to segment m / seg_len and is output row m % seg_len - seg_hist of it; carried rows produce no output.
    // SEG instantiations serve the streaming decode only; the bulk path compiles to plain row addressing.
    auto locate = [&](int m, int& bb, int& r) -> bool {
        bb = 0; r = m;
        if (m >= p.M) return false;
        if constexpr (SEG) { bb = m / p.seg_len; r = m - bb * p.seg_len - p.seg_hist; }
        return r >= 0;
    };

    // Per-column epilogue parameters of this lane's WN columns, fetched once (inside the pass loops the stores in between
    // would force a reload per pass, each with its latency exposed).  SwiGLU pairs columns differently and loads its own.
    // The fused unit fetches them after its inner activation (registers are scarce there), under its second contraction.
    float cbias[WN], csc[WN], cea[WN], ceib[WN], cqa[WN], cqib[WN];
    auto column_params = [&]() {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + wn * 32 * WN + j * 32 + (lane & 31);
            const bool nv = n < n_out && !swiglu;
            cbias[j] = (obias && nv) ? obias[n] : 0.f;
            csc[j] = (p.scale && nv && !FUSE2) ? p.scale[n] : 1.f;
            cea[j] = ceib[j] = cqa[j] = cqib[j] = 0.f;
            if (first_act == Q3_ACT_SNAKE && nv) { cea[j] = p.epi_a[n]; ceib[j] = p.epi_invb[n]; }
            if (p.post_a && nv) { cqa[j] = p.post_a[n]; cqib[j] = p.post_invb[n]; }
        }
    };
    if constexpr (!FUSE2) column_params();
    // The fused unit fetches the residual rows (the raw stream, rows of this block only) of each epilogue pass one
    // pass ahead, into registers: the first pass's NOW, so its HBM latency hides under the second contraction, the
    // others' under the pass before them (requested as soon as the registers have been emptied into the slab).
    constexpr int CPRW_F = SCOLS / EPO;                         // 16-byte pieces per slab row (no SwiGLU here)
    constexpr int NPIECE = (SROWS * CPRW_F + 63) / 64;          // pieces per lane and pass
    constexpr int NPASS = WM * (WN / JG) * 2;
    // A tile with every row inside the sequence and every column inside N (and 16-byte rows) takes the branch-free
    // epilogue below: its loads and stores are unconditional and counted at compile time, so waiting for the next
    // pass's residual rows does not also wait for this pass's stores (a skipped store would force vmcnt(0)).
    constexpr bool FAST_OK = !SEG && (SROWS * CPRW_F) % 64 == 0;
    const bool full = FAST_OK && vec_ok && !swiglu && (m0 + BM <= p.M) && (n0 + BN <= n_out);
    // The rows are requested through inline asm: the compiler would otherwise sink the loads to their first use, behind
    // the stores, and wait for everything.  It therefore does not know that they are pending; an explicit s_waitcnt vmcnt(N) is the wait,
    // N = the vector-memory operations issued since (the pass's stores, unconditional in this path).
    u32x4 rpre[NPIECE];
    // second set of rows in flight, for the fused unit's late residual add (below): there the rows of pass p are consumed at
    // the END of pass p, so the request for pass p + 1 needs registers of its own to be a whole pass ahead
    u32x4 rpre2[(FUSE2 && !LEAN_REGS) ? NPIECE : 1];
    auto r_fetch_into = [&](int pass, u32x4* dst) {
        const int rh = pass & 1, q = pass >> 1, jgi = q % (WN / JG), i = q / (WN / JG);
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) {
            const int id = lane + 64 * k;
            const int row = id / CPRW_F, ch = id - row * CPRW_F;
            const int m = m0 + wm * 32 * WM + i * 32 + rh * 16 + row + late0;
            const int n = n0 + wn * 32 * WN + jgi * JG * 32 + ch * EPO;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[k]) : "v"(Rb + (long long)m * p.ldr + n) : "memory");
        }
    };
    auto r_fetch_full = [&](int pass) { r_fetch_into(pass, rpre); };
    auto r_pin_buf = [&](u32x4* buf) {                           // after the wait: uses of the rows stay behind it
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) asm volatile("" : "+v"(buf[k]));
    };
    auto r_pin = [&]() { r_pin_buf(rpre); };
    auto r_fetch = [&](int pass, u32x4* dst) {                  // pass = (i * (WN / JG) + jg / JG) * 2 + rh
        const int rh = pass & 1, q = pass >> 1, jgi = q % (WN / JG), i = q / (WN / JG);
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) {
            const int id = lane + 64 * k;
            const int row = id / CPRW_F, ch = id - row * CPRW_F;
            const int m = m0 + wm * 32 * WM + i * 32 + rh * 16 + row;
            const int n = n0 + wn * 32 * WN + jgi * JG * 32 + ch * EPO;
            uint4 v = make_uint4(0, 0, 0, 0);
            int bb, r;
            if (id < SROWS * CPRW_F && locate(m, bb, r) && n < n_out)
                v = *(const uint4*)(Rb + (SEG ? (long long)bb * p.r_batch_stride : 0LL) + (long long)r * p.ldr + n);
            dst[k] = u32x4{v.x, v.y, v.z, v.w};
        }
    };
    if constexpr (FUSE2 && !LEAN_REGS) { if (full) r_fetch_full(0); else r_fetch(0, rpre); }

    // ---- fused second contraction (k = 1): h = snake(acc + bias) -> LDS -> acc = h . W2^T -------------------------
    if constexpr (FUSE2 && WAVES_N == 1) {
        // One column of waves: a wave's second contraction consumes exactly the h rows that wave produced, so the h tile is
        // WAVE-PRIVATE - 32 rows at a time (one row tile i of the wave), written, read back as the A operand and overwritten by
        // the next row tile without any workgroup barrier - and 6.5 KB per wave instead of a tile-sized overlay.  The W2 chunk
        // tiles go into the three weight buffers together (all of W2 when it has at most three chunks: one barrier, before the
        // first MFMA; wider K in groups of three).
        constexpr int HLD = BN * (int)sizeof(TA) + 16;              // padded row stride (compile-time: row offsets are immediates)
        unsigned char* Hw = smem + 3 * WBUF + (size_t)wave * (32 * HLD);   // over the input buffers, idle now
        const TA* __restrict__ W2 = (const TA*)p.W2;
        const bool w2_dma = (p.N % BK) == 0;
        const int nkc2 = (p.N + BK - 1) / BK;
        if constexpr (LEAN_REGS) w2voff_late();
        auto w2_group = [&](int g0) {               // chunks g0 .. g0 + 2 -> weight buffers 0 .. 2 (the caller has made the ring free)
            for (int q = 0; q < 3 && g0 + q < nkc2; ++q) {
                if (w2_dma) w_dma_issue(ws_lds + (unsigned)(q * WBUF), W2 + (long long)(g0 + q) * BK, w2voff);
                else {
#pragma unroll
                    for (int it = 0; it < WITER; ++it) w_store(Ws0 + (size_t)q * WBUF, tid + it * NTHR, w_load(W2, p.N, tid + it * NTHR, g0 + q, 0));
                }
            }
        };
        w2_group(0);                                // in flight while the first h rows are produced
        bool first = true;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = j * 32 + l31;
                const bool nv = n < p.N;
                // the unit's inner activation is always SnakeBeta
                const float ea = INNER_LDS ? inn_lds[n] : ia[j], eib = INNER_LDS ? inn_lds[BN + n] : iib[j], ib = INNER_LDS ? inn_lds[2 * BN + n] : ibias[j];
                unsigned char* hcol = Hw + (size_t)(4 * half) * HLD + n * sizeof(TA);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = q3_snake<TA>(acc[i][j][r] + ib, ea, eib);
                    if (nv) *(TA*)(hcol + ((r & 3) + 8 * (r >> 2)) * HLD) = q3_from_f32<TA>(v);
                    acc[i][j][r] = 0.f;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's h rows are in LDS (its own reads follow)
            __builtin_amdgcn_wave_barrier();
            for (int g0 = 0; g0 < nkc2; g0 += 3) {
                if (!(first || nkc2 <= 3)) {            // a wider K: the ring is reloaded per group, for every row tile
                    __syncthreads();
                    w2_group(g0);
                }
                if (first || nkc2 > 3) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (first) { mark(3); column_params(); }
                    first = false;
                }
                for (int q = 0; q < 3 && g0 + q < nkc2; ++q) {
                    const unsigned char* ws = Ws0 + (size_t)q * WBUF;
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg) {
                        const int c = kg * 2 + half;
                        const int kbyte = (g0 + q) * BKB + c * 16;
                        // columns past N (a K tail of the last chunk) multiply zero-filled weight columns; read them as zero too
                        const uint4 af = (w2_dma || kbyte < p.N * (int)sizeof(TA)) ? *(const uint4*)(Hw + (size_t)l31 * HLD + kbyte)
                                                                                   : make_uint4(0, 0, 0, 0);
                        uint4 bfr[WN];
#pragma unroll
                        for (int j = 0; j < WN; ++j) {
                            const int row = j * 32 + l31;
                            bfr[j] = *(const uint4*)(ws + (size_t)row * BKB + swz<CPR>(row, c) * 16);
                        }
#pragma unroll
                        for (int j = 0; j < WN; ++j) Mma<TA>::run(af, bfr[j], acc[i][j]);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the A fragments are in registers before the next row tile overwrites them
            __builtin_amdgcn_wave_barrier();
        }
        // the epilogue's slabs overlay the weight buffers: every wave must be done reading W2
        __syncthreads();
    } else if (FUSE2) {
        constexpr int HLD = BN * (int)sizeof(TA) + 16;              // padded row stride of the h tile (compile-time: row offsets are immediates)
        unsigned char* Hs = smem + 2 * WBUF;                        // overlays the third weight buffer and the input buffers, idle now
        const TA* __restrict__ W2 = (const TA*)p.W2;
        const bool w2_dma = (p.N % BK) == 0;      // the h tile is consumed in whole chunks too
        // first W2 tile in flight while h is produced
        if (w2_dma) w_dma_issue(ws_lds, W2, w2voff);
        else {
#pragma unroll
            for (int it = 0; it < WITER; ++it) w_store(Ws0, tid + it * NTHR, w_load(W2, p.N, tid + it * NTHR, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = wn * 32 * WN + j * 32 + l31;      // the h tile is shared through LDS: any wave grid works
                const bool nv = n < p.N;
                const float ea = ia[j], eib = iib[j];           // the unit's inner activation is always SnakeBeta
                unsigned char* hcol = Hs + (size_t)(wm * 32 * WM + 4 * half) * HLD + n * sizeof(TA);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = q3_snake<TA>(acc[i][j][r] + ibias[j], ea, eib);
                    if (nv) *(TA*)(hcol + (i * 32 + (r & 3) + 8 * (r >> 2)) * HLD) = q3_from_f32<TA>(v);
                    acc[i][j][r] = 0.f;
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        mark(3);
        column_params();
        const int nkc2 = (p.N + BK - 1) / BK;
        for (int s2 = 0; s2 < nkc2; ++s2) {
            const bool has_next = (s2 + 1) < nkc2;
            unsigned char* wsn = Ws0 + (size_t)((s2 + 1) & 1) * WBUF;
            uint4 wreg[WITER];
            if (has_next) {
                if (w2_dma) w_dma_issue(ws_lds + (unsigned)(((s2 + 1) & 1) * WBUF), W2 + (long long)(s2 + 1) * BK, w2voff);
                else {
#pragma unroll
                    for (int it = 0; it < WITER; ++it) wreg[it] = w_load(W2, p.N, tid + it * NTHR, s2 + 1, 0);
                }
            }
            const unsigned char* ws = Ws0 + (size_t)(s2 & 1) * WBUF;
            const int arow0 = wm * 32 * WM + l31;
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const int c = kg * 2 + half;
                const int kbyte = s2 * BKB + c * 16;
                uint4 af[WM], bfr[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    // columns past N (a K tail of the last chunk) multiply zero-filled weight columns; read them as zero too
                    af[i] = (w2_dma || kbyte < p.N * (int)sizeof(TA)) ? *(const uint4*)(Hs + (size_t)(arow0 + i * 32) * HLD + kbyte)
                                                                      : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    int row = wn * 32 * WN + j * 32 + l31;
                    bfr[j] = *(const uint4*)(ws + (size_t)row * BKB + swz<CPR>(row, c) * 16);
                }
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) Mma<TA>::run(af[i], bfr[j], acc[i][j]);
            }
            if (has_next && !w2_dma) {
#pragma unroll
                for (int it = 0; it < WITER; ++it) w_store(wsn, tid + it * NTHR, wreg[it]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    mark(4);
    // ---- epilogue -----------------------------------------------------------------------------------
    // The accumulator layout gives each lane ONE column of 16 rows, while memory wants 16 contiguous bytes per lane.
    // Each wave therefore passes its tile through private fp32 LDS slabs, 16 rows x <= 96 columns at a time:
    //   phase 0  residual rows -> slab A            (row-wise, 16-byte loads)
    //   phase 1  accumulator + bias, activation, scale, + slab A  -> slab A (raw), snake of the consumer -> slab B
    //            (column-wise: the per-channel parameters are two registers per lane)
    //   phase 2  slab A -> Y, slab B -> Y2           (row-wise, 16-byte stores)
    // Branch-free form for full tiles (see `full` above): same three phases, every load and store unconditional.
    // Fold of a following one-channel conv (p.W3): a pass's 16 rows x SCOLS channels of post(y) sit in slab A as fp32; instead of
    // storing them, the wave multiplies them by the conv's taps: P^T[tap][row] = W3[tap][:] . z[row][:] on the 16x16x32 matrix
    // op (A = W3, 16 taps x 32 channels per step; B = the slab rows, read 32 bytes per lane and rounded to bf16 exactly as the
    // stored tensor would have been).  Lane (row = lane & 15, g = lane >> 4) ends with taps 4g .. 4g+3 of its row: lanes 0-31
    // store the row's 8 floats.
    constexpr bool FOLD_OK = FUSE2 && sizeof(TA) == 2 && WAVES_N == 1 && WN == JG && !SEG;
    constexpr int FOLD_KS = SCOLS / 32;
    const bool fold = FOLD_OK && p.W3 != nullptr;
    float* __restrict__ Pb = (FOLD_OK && p.P) ? p.P + (long long)b * p.p_batch_stride : nullptr;
    bf16x8 w3f[(FOLD_OK && !LEAN_REGS) ? FOLD_KS : 1];
    if constexpr (FOLD_OK && !LEAN_REGS) {
        if (fold) {
#pragma unroll
            for (int ks = 0; ks < FOLD_KS; ++ks)
                w3f[ks] = *(const bf16x8*)((const bf16*)p.W3 + (size_t)(lane & 15) * p.N + 32 * ks + 8 * (lane >> 4));
        }
    }
    auto fold_pass = [&](int mrow0, bool row_ok) {
        if constexpr (FOLD_OK) {
            f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < FOLD_KS; ++ks) {
                const float* sp = slabA + (lane & 15) * SLD + 32 * ks + 8 * (lane >> 4);
                const float4 t0 = *(const float4*)sp, t1 = *(const float4*)(sp + 4);
                bf16x8 z;
                z[0] = (bf16)t0.x; z[1] = (bf16)t0.y; z[2] = (bf16)t0.z; z[3] = (bf16)t0.w;
                z[4] = (bf16)t1.x; z[5] = (bf16)t1.y; z[6] = (bf16)t1.z; z[7] = (bf16)t1.w;
                bf16x8 taps_ks;
                if constexpr (LEAN_REGS) taps_ks = *(const bf16x8*)((const bf16*)p.W3 + (size_t)(lane & 15) * p.N + 32 * ks + 8 * (lane >> 4));   // 3 KB, L1 / L2 resident
                else taps_ks = w3f[ks];
                pacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(taps_ks, z, pacc, 0, 0, 0);
            }
            if (lane < 32 && row_ok) *(f32x4*)(Pb + (long long)(mrow0 + (lane & 15)) * 8 + 4 * (lane >> 4)) = pacc;
        }
    };
    // LATE (fused unit, nothing non-linear after the residual add, one output): the residual rows skip the slab.  Phase 1
    // leaves acc + bias in the slab and phase 2, which reads it row-wise anyway, adds the rows straight from the registers
    // they were loaded into (the same fp32 add on the same operands: bit-identical results): no residual -> slab writes,
    // no per-element residual reads.  Rows of pass p + 1 are requested at the start of pass p into the other register set.
    auto fast_epilogue = [&](auto dual_tag, auto fold_tag, auto late_tag) {
        constexpr bool DUAL = decltype(dual_tag)::value;
        constexpr bool FOLD = decltype(fold_tag)::value;
        constexpr bool LATE = decltype(late_tag)::value;
        // (the lean-register shapes request nothing before the epilogue: their first rows are asked for here)
        if ((!FUSE2 || LEAN_REGS) && has_res && !LATE) r_fetch_full(0);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int jg = 0; jg < WN; jg += JG) {
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    const int pidx = (i * (WN / JG) + jg / JG) * 2 + rh;
                    const int mrow0 = m0 + wm * 32 * WM + i * 32 + rh * 16, ncol0 = n0 + wn * 32 * WN + jg * 32;
                    u32x4* const rcur = (LATE && !LEAN_REGS && (pidx & 1)) ? rpre2 : rpre;
                    if constexpr (LATE) {
                        // one register set (lean-register shapes): the rows of pass p are requested at its start and used at
                        // its end, under its own column-wise phase; two sets: a whole pass ahead
                        if constexpr (LEAN_REGS) r_fetch_into(pidx, rpre);
                        else if (pidx + 1 < NPASS) r_fetch_into(pidx + 1, (pidx & 1) ? rpre : rpre2);
                    }
                    if (has_res && !LATE) {
                        // since the request: this tile's earlier work and waits (first pass) or one pass's stores
                        if (pidx == 0) wait_vmcnt_c<0>();
                        else wait_vmcnt_c<FOLD ? 1 : NPIECE * (DUAL ? 2 : 1)>();
                        r_pin();
#pragma unroll
                        for (int k = 0; k < NPIECE; ++k) {
                            const int id = lane + 64 * k;
                            const int row = id / CPRW_F, ch = id - row * CPRW_F;
                            const u32x4 raw = rpre[k];
                            const TO* rr = (const TO*)&raw;
                            float* sp = slabA + row * SLD + ch * EPO;
#pragma unroll
                            for (int e = 0; e < EPO; e += 4)
                                *(float4*)(sp + e) = make_float4(q3_to_f32(rr[e]), q3_to_f32(rr[e + 1]), q3_to_f32(rr[e + 2]), q3_to_f32(rr[e + 3]));
                        }
                        if (pidx + 1 < NPASS) r_fetch_full(pidx + 1);       // next pass's rows: in flight during this pass
                        __builtin_amdgcn_wave_barrier();
                    }
#pragma unroll
                    for (int j = 0; j < JG; ++j) {
                        const int cl = j * 32 + l31;
                        const float bias = cbias[jg + j], sc = csc[jg + j];
                        const float ea = cea[jg + j], eib = ceib[jg + j], qa = cqa[jg + j], qib = cqib[jg + j];
                        float* sa = slabA + (4 * half) * SLD + cl;
                        float* sb = slabB + (4 * half) * SLD + cl;
                        float vv[8], rr[8];
                        if (has_res && !LATE) {
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) rr[r8] = sa[((r8 & 3) + 8 * (r8 >> 2)) * SLD];
                        }
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            float v = acc[i][jg + j][rh * 8 + r8] + bias;
                            if (first_act == Q3_ACT_GELU) v = q3_gelu(v);
                            else if (first_act == Q3_ACT_SNAKE) v = q3_snake<TO>(v, ea, eib);
                            else if (first_act == Q3_ACT_CLAMP) v = fminf(1.0f, fmaxf(-1.0f, v));
                            else if (first_act == Q3_ACT_ELU) v = q3_elu(v);
                            if (!FUSE2) v *= sc;
                            if (has_res && !LATE) v += rr[r8];
                            vv[r8] = v;
                        }
                        if (DUAL) {
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) {
                                sa[((r8 & 3) + 8 * (r8 >> 2)) * SLD] = vv[r8];
                                sb[((r8 & 3) + 8 * (r8 >> 2)) * SLD] = (post_elu ? q3_elu(vv[r8]) : q3_snake<TO>(vv[r8], qa, qib));
                            }
                        } else if (has_post) {
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) sa[((r8 & 3) + 8 * (r8 >> 2)) * SLD] = (post_elu ? q3_elu(vv[r8]) : q3_snake<TO>(vv[r8], qa, qib));
                        } else {
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) sa[((r8 & 3) + 8 * (r8 >> 2)) * SLD] = vv[r8];
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    if constexpr (FOLD) fold_pass(mrow0, true);
                    if constexpr (LATE) {
                        // rows of this pass: requested a pass ago; issued since: the last pass's stores and the next pass's request
                        if constexpr (LEAN_REGS) wait_vmcnt_c<0>();          // nothing was issued since this pass's request
                        else if (pidx == 0) wait_vmcnt_c<(NPASS > 1 ? NPIECE : 0)>();
                        else if (pidx + 1 < NPASS) wait_vmcnt_c<2 * NPIECE>();
                        else wait_vmcnt_c<NPIECE>();
                        r_pin_buf(rcur);
                    }
#pragma unroll
                    for (int k = 0; k < (FOLD ? 0 : NPIECE); ++k) {
                        const int id = lane + 64 * k;
                        const int row = id / CPRW_F, ch = id - row * CPRW_F;
                        const long long yo = (long long)(mrow0 + row + late0) * p.ldy + ncol0 + ch * EPO;
#pragma unroll
                        for (int which = 0; which < (DUAL ? 2 : 1); ++which) {
                            const float* sp = (which ? slabB : slabA) + row * SLD + ch * EPO;
                            uint4 outv;
                            TO* oo = (TO*)&outv;
                            const u32x4 rraw = rcur[k];
                            const TO* rres = (const TO*)&rraw;
#pragma unroll
                            for (int e = 0; e < EPO; e += 4) {
                                float4 t4 = *(const float4*)(sp + e);
                                if constexpr (LATE) {
                                    t4.x += q3_to_f32(rres[e]); t4.y += q3_to_f32(rres[e + 1]);
                                    t4.z += q3_to_f32(rres[e + 2]); t4.w += q3_to_f32(rres[e + 3]);
                                }
                                oo[e] = q3_from_f32<TO>(t4.x); oo[e + 1] = q3_from_f32<TO>(t4.y);
                                oo[e + 2] = q3_from_f32<TO>(t4.z); oo[e + 3] = q3_from_f32<TO>(t4.w);
                            }
                            *(uint4*)((which ? Y2b : Yb) + yo) = outv;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    };
    if (full) {
        if constexpr (FOLD_OK) {
            if (fold) { fast_epilogue(std::false_type{}, std::true_type{}, std::false_type{}); mark(5); return; }
        }
        if constexpr (FUSE2) {
            if (!Y2b && !has_post && !(p.debug & 128)) { fast_epilogue(std::false_type{}, std::false_type{}, std::true_type{}); mark(5); return; }
        }
        // (the lean-register fused shapes are launched for those two forms only - one raw output, or the folded final conv:
        // the launcher refuses anything else - so they carry no other branch-free epilogue)
        if constexpr (!(FUSE2 && LEAN_REGS)) {
            if (Y2b) fast_epilogue(std::true_type{}, std::false_type{}, std::false_type{});
            else fast_epilogue(std::false_type{}, std::false_type{}, std::false_type{});
            mark(5);
            return;
        }
    }
    if constexpr (FUSE2 && LEAN_REGS) r_fetch(0, rpre);     // partial tiles: the first pass's rows (the other shapes asked for them earlier)
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int jg = 0; jg < WN; jg += JG) {
            const int pcols = swiglu ? SCOLS / 2 : SCOLS;              // output columns of this pass
            const int ncol0 = swiglu ? ((n0 >> 1) + wn * (16 * WN) + (jg >> 1) * 32) : (n0 + wn * 32 * WN + jg * 32);
            const int cprw = pcols / EPO;                               // 16-byte pieces per slab row
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {                            // rows 0-15, then 16-31 of the 32-row tile
                const int mrow0 = m0 + wm * 32 * WM + i * 32 + rh * 16;
                // phase 0: residual -> slab A
                if constexpr (FUSE2) {
                    const int pidx = (i * (WN / JG) + jg / JG) * 2 + rh;              // compile-time after unrolling
#pragma unroll
                    for (int k = 0; k < NPIECE; ++k) {
                        const int id = lane + 64 * k;
                        if (id < SROWS * CPRW_F) {
                            const int row = id / CPRW_F, ch = id - row * CPRW_F;
                            const u32x4 raw = rpre[k];
                            const TO* rr = (const TO*)&raw;
                            float* sp = slabA + row * SLD + ch * EPO;
#pragma unroll
                            for (int e = 0; e < EPO; e += 4)
                                *(float4*)(sp + e) = make_float4(q3_to_f32(rr[e]), q3_to_f32(rr[e + 1]), q3_to_f32(rr[e + 2]), q3_to_f32(rr[e + 3]));
                        }
                    }
                    if (pidx + 1 < NPASS) r_fetch(pidx + 1, rpre);      // next pass's rows: in flight during this pass
                    __builtin_amdgcn_wave_barrier();
                } else if (has_res) {
                    for (int id = lane; id < SROWS * cprw; id += 64) {
                        const int row = id / cprw, ch = id - row * cprw;
                        const int m = mrow0 + row, n = ncol0 + ch * EPO;
                        float v[EPO];
#pragma unroll
                        for (int e = 0; e < EPO; ++e) v[e] = 0.f;
                        int bb, r;
                        if (locate(m, bb, r) && n < n_out) {
                            const TO* Rr = Rb + (SEG ? (long long)bb * p.r_batch_stride : 0LL) + (long long)r * p.ldr;
                            if (vec_ok) {
                                const uint4 raw = *(const uint4*)(Rr + n);
                                const TO* rr = (const TO*)&raw;
#pragma unroll
                                for (int e = 0; e < EPO; ++e) v[e] = q3_to_f32(rr[e]);
                            } else {
#pragma unroll
                                for (int e = 0; e < EPO; ++e)
                                    if (n + e < n_out) v[e] = q3_to_f32(Rr[n + e]);
                            }
                        }
                        float* sp = slabA + row * SLD + ch * EPO;
#pragma unroll
                        for (int e = 0; e < EPO; e += 4) *(float4*)(sp + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                // phase 1: column-wise math
                if (swiglu) {
                    if constexpr (JG % 2 == 0) {
#pragma unroll
                        for (int j = 0; j < JG; j += 2) {
                            const int cl = (j >> 1) * 32 + l31;
                            const int n = ncol0 + cl;
                            const float sc = (p.scale && n < n_out) ? p.scale[n] : 1.f;
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) {
                                const int r = rh * 8 + r8;
                                const int row = (r8 & 3) + 8 * (r8 >> 2) + 4 * half;
                                float v = q3_silu(acc[i][jg + j][r]) * acc[i][jg + j + 1][r] * sc;
                                if (has_res) v += slabA[row * SLD + cl];
                                slabA[row * SLD + cl] = v;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < JG; ++j) {
                        const int cl = j * 32 + l31;
                        const float bias = cbias[jg + j], sc = csc[jg + j];
                        const float ea = cea[jg + j], eib = ceib[jg + j], qa = cqa[jg + j], qib = cqib[jg + j];
                        // eight elements at a time: residuals read first, then the math, then the stores, so that the
                        // eight chains interleave and the run-time choices are made once per group, not per element
                        float* sa = slabA + (4 * half) * SLD + cl;
                        float* sb = slabB + (4 * half) * SLD + cl;
                        float vv[8], rr[8];
                        if (has_res) {
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) rr[r8] = sa[((r8 & 3) + 8 * (r8 >> 2)) * SLD];
                        }
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            float v = acc[i][jg + j][rh * 8 + r8] + bias;
                            if (first_act == Q3_ACT_GELU) v = q3_gelu(v);
                            else if (first_act == Q3_ACT_SNAKE) v = q3_snake<TO>(v, ea, eib);
                            else if (first_act == Q3_ACT_CLAMP) v = fminf(1.0f, fmaxf(-1.0f, v));
                            else if (first_act == Q3_ACT_ELU) v = q3_elu(v);
                            if (!FUSE2) v *= sc;
                            if (has_res) v += rr[r8];
                            vv[r8] = v;
                        }
                        if (Y2b) {
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) {
                                sa[((r8 & 3) + 8 * (r8 >> 2)) * SLD] = vv[r8];
                                sb[((r8 & 3) + 8 * (r8 >> 2)) * SLD] = (post_elu ? q3_elu(vv[r8]) : q3_snake<TO>(vv[r8], qa, qib));
                            }
                        } else if (has_post) {
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) sa[((r8 & 3) + 8 * (r8 >> 2)) * SLD] = (post_elu ? q3_elu(vv[r8]) : q3_snake<TO>(vv[r8], qa, qib));
                        } else {
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) sa[((r8 & 3) + 8 * (r8 >> 2)) * SLD] = vv[r8];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if constexpr (FOLD_OK) {
                    if (fold) {      // partial tiles: the same product, rows past the sequence are not stored
                        fold_pass(mrow0, mrow0 + (lane & 15) < p.M);
                        __builtin_amdgcn_wave_barrier();
                        continue;
                    }
                }
                // phase 2: slabs -> memory, 16 bytes per lane
                for (int id = lane; id < SROWS * cprw; id += 64) {
                    const int row = id / cprw, ch = id - row * cprw;
                    const int m = mrow0 + row, n = ncol0 + ch * EPO;
                    int bb, r;
                    if (!locate(m, bb, r) || n >= n_out) continue;
                    const long long yo = (long long)r * p.ldy + n;
                    const bool blank = SEG && p.row_start && r < p.row_start[bb];   // rows before a stream's first sample
#pragma unroll
                    for (int which = 0; which < 2; ++which) {
                        if (which == 1 && !Y2b) break;
                        const float* sp = (which ? slabB : slabA) + row * SLD + ch * EPO;
                        TO* dst = (which ? Y2b : Yb) + yo;
                        if constexpr (SEG) dst += (long long)bb * (which ? p.y2_batch_stride : p.y_batch_stride);
                        float v[EPO];
#pragma unroll
                        for (int e = 0; e < EPO; e += 4) {
                            const float4 t4 = blank ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(sp + e);
                            v[e] = t4.x; v[e + 1] = t4.y; v[e + 2] = t4.z; v[e + 3] = t4.w;
                        }
                        if (vec_ok) {
                            uint4 outv;
                            TO* oo = (TO*)&outv;
#pragma unroll
                            for (int e = 0; e < EPO; ++e) oo[e] = q3_from_f32<TO>(v[e]);
                            *(uint4*)dst = outv;
                        } else {
#pragma unroll
                            for (int e = 0; e < EPO; ++e)
                                if (n + e < n_out) dst[e] = q3_from_f32<TO>(v[e]);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    mark(5);
}

// ---- host side ----------------------------------------------------------------------------------------
// A/B switches for measurements (ConvGemmArgs::tuning, the context's word): bit 0 flips the 128- / 256-row choice for
// layers of few rows and keeps 64-byte K rows for layers of at most one tile per CU, bit 5 = no narrow tiles for frame-rate pushes,
// bit 7 = no 64-column tiles, bit 1 = one block per CU (128-byte K rows) for large layers, bit 2 = the fused 192-channel unit on
// 256-row tiles, bit 3 = exact 96-column tiling for the 288-column transposed conv; bits 8.. go to the kernel as
// ConvGemmArgs::debug


template <typename TA, typename TO, int WAVES_M, int WAVES_N, int WM, int WN, int BKB, bool FUSE2>
hipError_t launch_cfg(const ConvGemmArgs& a_in, hipStream_t s) {
    constexpr int NW = WAVES_M * WAVES_N, NTHR = 64 * NW;
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    constexpr int CPR = BKB / 16;
    constexpr int JG = WN <= 3 ? WN : 3;
    ConvGemmArgs a = a_in;
    const int halo = (a.taps - 1) * a.off_step;
    const int HR = BM + halo;
    const int xi_total = (HR * CPR + NTHR - 1) / NTHR;
    const int xps = (xi_total + a.taps - 1) / a.taps;
    if (xps > (BM * CPR + NTHR - 1) / NTHR || xps > 4) return hipErrorInvalidValue;      // halo too large for the staged-slice scheme
    // the twelve-wave shape: one raw output (no consumer's SnakeBeta), or the folded final conv
    if (FUSE2 && NW == 12 && (a.Y2 || (a.post_a && !a.W3))) return hipErrorInvalidValue;
    if (a.W3 && !(FUSE2 && sizeof(TA) == 2 && WAVES_N == 1 && WN == JG && a.P && a.post_a && !a.Y2 && a.seg_len == 0 && a.N % 32 == 0 && a.N == BN))
        return hipErrorInvalidValue;
    if (FUSE2 && (a.post_elu || a.N > BN || a.K != a.N || !a.W2 || a.act != Q3_ACT_SNAKE || !a.epi_a || !a.epi_invb || !a.pre_a || !a.pre_invb || !a.R ||
                  a.ldy % 8 || a.ldr % 8 || a.N % 8))
        return hipErrorInvalidValue;
    const size_t xbuf = (size_t)((HR * CPR + 63) / 64) * 1024, wbuf = round_kb((size_t)BN * BKB), wring = 3 * wbuf + 1024;   // + a spare KiB for padding DMAs
    {
        const long long wbytes = (long long)a.taps * a.N * a.K * (long long)sizeof(TA);
        // weights that fit an XCD's L2 beside the streams (<= 2.5 MB): N tile fastest, the input comes from HBM once; larger
        // ones: N tiles in groups (their weights resident, the input once per group; pairs measured against N tile slowest:
        // transposed conv 768 -> 5 x 384 +8.5 %, k = 7 conv of 768 channels +0.8 %).  Tuning bit 16 = N tile slowest.
        if (wbytes <= (5LL << 19)) a.tile_order = 1;
        else if (a.tuning & 16) a.tile_order = 0;
        else {
            // group size: as many N tiles as keep their weight slices within ~3 MB of the XCD's 4 MB L2 (at least a pair),
            // evened out over the groups (transposed conv 768 -> 5 x 384: groups of 5 against pairs +6 %)
            const long long slice = (long long)a.taps * BN * a.K * sizeof(TA);
            const int n_tiles = (a.N + BN - 1) / BN;
            int gmax = (int)((3LL << 20) / slice);
            if ((a.tuning >> 16) & 15) gmax = (a.tuning >> 16) & 15;
            gmax = gmax < 2 ? 2 : gmax;
            const int groups = (n_tiles + gmax - 1) / gmax;
            const int g = (n_tiles + groups - 1) / groups;
            a.tile_order = g < 2 ? 2 : g;
        }
    }
    a.nxb = (a.taps <= 3 && round_kb(3 * xbuf) + wring <= 160 * 1024) ? 3 : 2;
    // a third input buffer is not worth a CU's second block
    if (a.nxb == 3 && a.taps > 1 && round_kb(3 * xbuf) + wring > 80 * 1024 && round_kb(2 * xbuf) + wring <= 80 * 1024) a.nxb = 2;
    size_t lds = wring + round_kb(a.nxb * xbuf);
    // the fused unit's h tile: over the third weight buffer and the input buffers; wave-private 32-row pieces behind the whole
    // ring when one column of waves covers every channel
    if (FUSE2) lds = std::max(lds, WAVES_N == 1 ? 3 * wbuf + round_kb((size_t)NW * 32 * (BN * sizeof(TA) + 16)) + 1024
                                                : 2 * wbuf + round_kb((size_t)BM * (BN * sizeof(TA) + 16)) + 1024);
    // the input SnakeBeta's parameters, behind the spare KiB; the twelve-wave shape also keeps the inner activation's [3][BN] there
    if (FUSE2) lds += round_kb((size_t)(2 * a.K + (NW == 12 ? 3 * BN : 0)) * sizeof(float));
    lds = std::max(lds, (size_t)NW * 2 * 16 * (32 * JG + 4) * sizeof(float));    // two wave-private 16-row epilogue slabs per wave
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const bool seg = a.seg_len > 0;
    if (a.row_start && !seg) return hipErrorInvalidValue;
    constexpr bool SEG_OK = NW != 12;        // the twelve-wave shape serves the bulk decode only (pick_shape never chooses it for segmented rows)
    if (seg && !SEG_OK) return hipErrorInvalidValue;
    auto kern = (seg && SEG_OK) ? convgemm_kernel<TA, TO, WAVES_M, WAVES_N, WM, WN, BKB, FUSE2, SEG_OK>
                                : convgemm_kernel<TA, TO, WAVES_M, WAVES_N, WM, WN, BKB, FUSE2, false>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const long long tiles = (long long)((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM);
    if (tiles > 0x7fffffffLL || a.B > 65535) return hipErrorInvalidValue;
    dim3 grid((unsigned)tiles, a.B, 1);
    hipLaunchKernelGGL(kern, grid, dim3(NTHR), lds, s, a);
    return hipGetLastError();
}

// tile shapes (rows x columns, K-row bytes, waves)
enum { SH_256x96 = 0, SH_256x128_K128, SH_128x128_K128, SH_128x128_K64, SH_256x32, SH_F_256x192, SH_F_256x96, SH_F_256x32,
       SH_256x128_K64, SH_F_128x192, SH_F_128x384, SH_256x64, SH_128x32_K128, SH_128x64_K128, SH_F_768x96, SH_COUNT };

int pick_shape(const ConvGemmArgs& a, size_t elem) {
    // rows that decide the shape: a launch that computes only the tail of its rows keeps the whole tensor's shape (the shape
    // fixes the order in which K chunks and taps are summed, and the tail must come out bit for bit as in the whole launch)
    const long long rows_all = (long long)a.B * (a.shape_rows > 0 ? a.shape_rows : a.M);
    if (a.W2) {                                   // fused residual unit: one N tile must hold every channel
        if (a.N <= 32) return SH_F_256x32;
        // 33-96 channels in bf16 over at least four rounds of tiles: 768-row tiles on TWELVE waves (64 rows x 96 columns each, as in
        // the four-wave shape, so the sums come out bit for bit the same), 150 KB of LDS, one block per CU = three waves per SIMD
        // at 168 registers: 3-4 % faster than two four-wave blocks per CU (tuning bit 1 << 22 keeps the four-wave shape).
        // Measured and dropped on the way: 384-row tiles on six waves (two blocks per CU would fit by LDS and registers, but a
        // six-wave block takes two wave slots on every SIMD when it is placed, so only ONE block is ever resident: 550 against
        // 665 TFLOP/s) and 128-row tiles on four waves, three blocks per CU (650 against 680).
        if (a.N <= 96 && a.N > 32 && elem == 2 && a.seg_len == 0 && a.N % 32 == 0 && rows_all >= 768LL * 1024 && !a.Y2 && (!a.post_a || a.W3) &&
            !(a.tuning & (1 << 22))) return SH_F_768x96;
        if (a.N <= 96) return SH_F_256x96;       // (128-row tiles, three blocks per CU: measured 9 % slower)
        // 128-row tiles on four waves keep the unit's LDS under 80 KB: two independent blocks per CU, one's h tile and
        // epilogue under the other's MFMAs, against one 8-wave block in lockstep
        if (a.N <= 192) return (elem == 2 && !(a.tuning & 4)) ? SH_F_128x192 : SH_F_256x192;
        // 384 channels (bf16): 128 rows x 384 columns on eight waves, one block per CU; the whole first-contraction weight
        // (2 MB) is re-streamed from L2 per 128 rows, which the matrix pipe's time per tile still covers
        if (a.N <= 384 && elem == 2) return SH_F_128x384;
        return -1;
    }
    const bool wide_k = ((size_t)a.K * elem) % 128 == 0;
    if (a.N <= 32 && a.act != Q3_ACT_SWIGLU) return SH_256x32;
    // 33-64 columns over many rows (the encode side's 24 kHz and 12 kHz convs): exact 64-column tiles on eight waves of 32 rows
    // x 64 columns, two blocks per CU.  Such layers are all prologue and epilogue (K = 32-128): in a 128-column tile half the
    // columns are padding and the tile takes the general (bounds-checked) epilogue.
    if (a.N > 32 && a.N <= 64 && a.act != Q3_ACT_SWIGLU && rows_all >= 8192 && !(a.tuning & 128)) return SH_256x64;
    const bool use_96 = (a.act != Q3_ACT_SWIGLU) && (a.N % 128 != 0) && (a.N % 96 == 0) && a.N < 512;
    if (use_96 && (a.N <= 96 || rows_all < 8192 || (a.tuning & 8))) return SH_256x96;     // larger N: more waves per CU beat exact tiling (memory-bound layers)
    if (!wide_k) return SH_128x128_K64;
    // Layers with enough rows to fill the chip: 256x128 tiles with 64-byte K rows need 65-73 KB of LDS, so TWO blocks
    // (16 waves) share a CU and one block's prologue, barriers and epilogue hide under the other's MFMAs (measured against one 129-KB block with
    // 128-byte rows: k = 7 convs +6..14 %, k = 1 convs +36..49 %, transposed convs +6..19 %).
    if (!(a.tuning & 2) && rows_all >= 8192) {
        // At most one tile per CU (a frame-rate push's first block): a second block per CU has nothing to overlap with, and the
        // launch takes what one tile's chain of steps takes - 128-byte K rows halve the steps (k = 7 conv of 768 channels over
        // 9 700 rows: 128 -> 100 us).
        const long long t256 = ((rows_all + 255) / 256) * ((a.N + 127) / 128);
        if (!(a.tuning & 1) && t256 <= 256) return SH_256x128_K128;
        return SH_256x128_K64;
    }
    // Few rows (a streaming push, a short utterance): 128-row tiles on four waves double the number of blocks when
    // 256-row tiles would leave most of the 256 CUs idle (256 streams x 1 frame: push 5.75 -> 5.46 ms)
    // A frame-rate push (a few hundred rows): with 128-column tiles a 1024-column layer is 16 workgroups, each streaming its
    // 128 weight rows from HBM through ONE CU's DMA path (12-17 B per cycle): the layer takes what that path takes.  32-column
    // tiles put four times as many CUs on the weights; the rows they all re-read (256 x K) come from L2.
    const long long tiles128 = ((rows_all + 127) / 128) * ((a.N + 127) / 128);
    // (SwiGLU pairs adjacent 32-column tiles inside a wave: 64-column tiles there)
    if (rows_all <= 1024 && a.N >= 256 && a.N % 64 == 0 && !(a.tuning & 32)) {
        if (a.act != Q3_ACT_SWIGLU && tiles128 < 128) return SH_128x32_K128;
        if (a.act == Q3_ACT_SWIGLU && tiles128 < 256) return SH_128x64_K128;
    }
    const long long tiles256 = ((rows_all + 255) / 256) * ((a.N + 127) / 128);
    const bool small = tiles256 < 512;
    return (small != ((a.tuning & 1) != 0)) ? SH_128x128_K128 : SH_256x128_K128;
}

template <typename TA, typename TO, bool ALLOW_FUSE>
hipError_t launch_typed(const ConvGemmArgs& a, hipStream_t s) {
    switch (pick_shape(a, sizeof(TA))) {
        case SH_256x96: return launch_cfg<TA, TO, 4, 1, 2, 3, 64, false>(a, s);
        case SH_256x128_K128: return launch_cfg<TA, TO, 4, 2, 2, 2, 128, false>(a, s);
        case SH_128x128_K128: return launch_cfg<TA, TO, 2, 2, 2, 2, 128, false>(a, s);
        case SH_128x128_K64: return launch_cfg<TA, TO, 2, 2, 2, 2, 64, false>(a, s);
        case SH_256x32: return launch_cfg<TA, TO, 4, 1, 2, 1, 64, false>(a, s);
        case SH_256x128_K64: return launch_cfg<TA, TO, 4, 2, 2, 2, 64, false>(a, s);
        case SH_256x64: return launch_cfg<TA, TO, 8, 1, 1, 2, 64, false>(a, s);
        case SH_128x32_K128: return launch_cfg<TA, TO, 4, 1, 1, 1, 128, false>(a, s);
        case SH_128x64_K128: return launch_cfg<TA, TO, 4, 1, 1, 2, 128, false>(a, s);
        default: break;
    }
    if constexpr (ALLOW_FUSE) {      // the fused form re-reads its input's element type as the residual: same-type pairs only
        switch (pick_shape(a, sizeof(TA))) {
            case SH_F_256x192: return launch_cfg<TA, TO, 4, 2, 2, 3, 64, true>(a, s);    // 2x3 tiles per wave: 5 LDS reads per 6 MFMAs (1x6: 7)
            case SH_F_128x192: return launch_cfg<TA, TO, 2, 2, 2, 3, 64, true>(a, s);
            case SH_F_128x384:
                if constexpr (sizeof(TA) == 2) return launch_cfg<TA, TO, 2, 4, 2, 3, 64, true>(a, s);
                break;
            case SH_F_768x96:
                if constexpr (sizeof(TA) == 2 && sizeof(TO) == 2) return launch_cfg<TA, TO, 12, 1, 2, 3, 64, true>(a, s);
                else return launch_cfg<TA, TO, 4, 1, 2, 3, 64, true>(a, s);
            case SH_F_256x96: return launch_cfg<TA, TO, 4, 1, 2, 3, 64, true>(a, s);
            case SH_F_256x32: return launch_cfg<TA, TO, 4, 1, 2, 1, 64, true>(a, s);
            default: break;
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace
