// Step-structured probe of the contraction kernel's main loop on gfx950 (one 8-wave block per CU, ring of 3 buffers,
// DMAs issued two steps ahead, counted vmcnt + one barrier per step), with every per-step cost switchable at compile
// time: DMAs per wave, MFMAs, LDS fragment reads, the lgkmcnt drain before the barrier, L2-resident or streamed source.
// Build: hipcc --offload-arch=gfx950 -O3 tools/dma_probe3.hip -o tools/bin/dma_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void dma16_so(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// DQ DMAs (1 KiB each) per wave per step; MF: 16 MFMAs per wave per step; DS: 16 ds_read_b128 per wave per step;
// LGKM: drain lgkmcnt before the barrier; STREAM: every block walks its own part of a large buffer
template <int DQ, bool MF, bool DS, bool LGKM, bool STREAM, int DUMMY>
__global__ void __launch_bounds__(512) probe(const char* src, long long region, int steps, float* sink, int ntap, long long tap_bytes, int nmod) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void*)smem;
    constexpr unsigned SLOT = DQ * NW * 1024;                         // bytes per ring slot
    const unsigned voff = (unsigned)((lane >> 3) * 1536 + (lane & 7) * 16);
    // wave's piece q of a step: 8 rows of 1536 B starting at row (q * NW + wave) * 8; steps walk the 12 chunks of a row
    const long long blockoff = STREAM ? ((long long)blockIdx.x * (DQ * NW * 8 * 1536LL)) % (region - (DQ * NW * 8 + 8) * 1536LL) : 0;
    const char* base = src + blockoff + (long long)wave * 8 * 1536 + (long long)(blockIdx.x % nmod) * 196608;
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    const unsigned char* rd = smem + lane * 16;
    uint4 x = make_uint4(0, 0, 0, 0);

    auto issue = [&](int slot, int chunk) {
#pragma unroll
        for (int q = 0; q < DQ; ++q) dma16_so(base + (long long)q * NW * 8 * 1536 + chunk * 128, voff, lds0 + slot * SLOT + (q * NW + wave) * 1024);
    };
    if (DQ > 0) { issue(0, 0); issue(1, 1); wait_vm<DQ>(); }
    int tap = 0;
    __builtin_amdgcn_s_barrier();
    int slot = 0, chunk = 2;
    for (int s = 0; s < steps; ++s) {
        const int slot2 = slot >= 1 ? slot - 1 : 2;       // slot of step s + 2 == slot of step s - 1
        if (DQ > 0) issue(slot2, chunk);
        const unsigned char* buf = rd + (DS ? slot * (SLOT > 16384 ? 16384 : SLOT) : 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x8 a0 = {1, 1, 1, 1, 1, 1, 1, 1}, a1 = a0, b0 = a0, b1 = a0;
            if (DS) {
                const uint4 v0 = *(const uint4*)(buf + g * 4096), v1 = *(const uint4*)(buf + g * 4096 + 1024);
                const uint4 v2 = *(const uint4*)(buf + g * 4096 + 2048), v3 = *(const uint4*)(buf + g * 4096 + 3072);
                if (MF) { a0 = __builtin_bit_cast(bf16x8, v0); a1 = __builtin_bit_cast(bf16x8, v1); b0 = __builtin_bit_cast(bf16x8, v2); b1 = __builtin_bit_cast(bf16x8, v3); }
                else { x.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x; }
            }
            if (MF) {
                __builtin_amdgcn_s_setprio(1);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc3, 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        if (DUMMY > 0) {        // extra issue slots: DUMMY scalar + DUMMY vector instructions per wave per step
            int sd = s;
            unsigned d0 = lane, d1 = lane, d2 = lane, d3 = lane;
#pragma unroll
            for (int i = 0; i < DUMMY / 4; ++i) {
                asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1" : "+s"(sd));
                asm volatile("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\tv_add_u32 %2, %2, 1\n\tv_add_u32 %3, %3, 1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            }
            x.y ^= d0 ^ d1 ^ d2 ^ d3 ^ sd;
        }
        if (DQ > 0) wait_vm<DQ>();
        if (LGKM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = slot == 2 ? 0 : slot + 1;
        // like the conv: consecutive steps walk the taps of one chunk, each tap's tile tap_bytes further on
        if (ntap > 1) { base += tap_bytes; if (++tap == ntap) { tap = 0; base -= ntap * tap_bytes; chunk = chunk == 11 ? 0 : chunk + 1; } }
        else chunk = chunk == 11 ? 0 : chunk + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = (x.x == 0x12345u || x.y == 0x54321u) ? 1.f : 0.f;
    for (int r = 0; r < 16; ++r) t += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    if (t == 12345.f) sink[blockIdx.x] = t;
}

template <int DQ, bool MF, bool DS, bool LGKM, bool STREAM, int DUMMY = 0>
void run(const char* name, const char* src, long long region, float* sink, int ntap = 1, long long tap_bytes = 0, int nmod = 1) {
    const int steps = 96, blocks = 256 * 8;
    size_t lds = (size_t)3 * DQ * 8 * 1024;
    if (lds < 64 * 1024) lds = 64 * 1024;
    auto k = probe<DQ, MF, DS, LGKM, STREAM, DUMMY>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, 0, src, region, steps, sink, ntap, tap_bytes, nmod);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double us = ms * 1e3 / (blocks / 256.0 * steps);
    printf("%-64s %7.3f us/step %6.0f cyc  %6.2f B/clk/CU\n", name, us, us * 2400, DQ * 8 * 1024 / (us * 2400));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
    const long long big = 4LL << 30;
    char* buf; float* sink;
    CHECK(hipMalloc(&buf, big)); CHECK(hipMalloc(&sink, 1 << 20));
    CHECK(hipMemset(buf, 0, big));
    //   DQ MF     DS     LGKM   STREAM
    run<2, false, false, true, false>("16K/step L2, dma only, lgkm drain", buf, big, sink);
    run<2, false, false, false, false>("16K/step L2, dma only, no lgkm drain", buf, big, sink);
    run<0, true, false, true, false>("mfma only", buf, big, sink);
    run<0, true, true, true, false>("mfma + ds_read", buf, big, sink);
    run<0, false, true, true, false>("ds_read only", buf, big, sink);
    run<2, true, false, true, false>("16K/step L2 + mfma, lgkm drain", buf, big, sink);
    run<2, true, true, true, false>("16K/step L2 + mfma + ds_read, lgkm drain", buf, big, sink);
    run<2, true, true, false, false>("16K/step L2 + mfma + ds_read, no lgkm drain", buf, big, sink);
    run<3, true, true, true, false>("24K/step L2 + mfma + ds_read", buf, big, sink);
    run<6, true, true, true, false>("48K/step L2 + mfma + ds_read", buf, big, sink);
    run<6, false, false, true, false>("48K/step L2, dma only", buf, big, sink);
    run<1, true, true, true, false>("8K/step L2 + mfma + ds_read", buf, big, sink);
    run<2, false, false, true, true>("16K/step streamed, dma only", buf, big, sink);
    run<2, true, true, true, true>("16K/step streamed + mfma + ds_read", buf, big, sink);
    run<6, true, true, true, true>("48K/step streamed + mfma + ds_read", buf, big, sink);
    run<2, true, true, true, false, 100>("16K/step L2 + mfma + ds_read + 100 salu + 100 valu per wave", buf, big, sink);
    run<2, true, true, true, false, 200>("16K/step L2 + mfma + ds_read + 200 salu + 200 valu per wave", buf, big, sink);
    run<0, true, true, true, false, 200>("mfma + ds_read + 200 salu + 200 valu per wave", buf, big, sink);
    run<0, false, false, true, false, 200>("200 salu + 200 valu per wave only", buf, big, sink);
    // weight working sets: 7 taps, each tap's [N][K] slab `tap_bytes` apart; the 128 rows a block reads span 196 KB per tap
    run<2, true, true, true, false>("16K/step, 7 taps x 196 KB apart (1.4 MB set) + compute", buf, big, sink, 7, 196608);
    run<2, true, true, true, false>("16K/step, 7 taps x 1.18 MB apart (one N tile of 8.3 MB) + compute", buf, big, sink, 7, 1179648);
    run<2, false, false, true, false>("16K/step, 7 taps x 1.18 MB apart, dma only", buf, big, sink, 7, 1179648);
    run<2, true, true, true, false>("16K/step, 7 taps x 1.18 MB, 6 N tiles by block%6 (8.3 MB set) + compute", buf, big, sink, 7, 1179648, 6);
    run<2, false, false, true, false>("16K/step, 7 taps x 1.18 MB, 6 N tiles by block%6, dma only", buf, big, sink, 7, 1179648, 6);
    run<2, true, true, true, false>("16K/step, 7 taps x 1.18 MB, 3 N tiles by block%3 (4.1 MB set) + compute", buf, big, sink, 7, 1179648, 3);
    run<2, true, true, true, false>("16K/step, 7 taps x 1.18 MB, 2 N tiles by block%2 (2.8 MB set) + compute", buf, big, sink, 7, 1179648, 2);
    return 0;
}
