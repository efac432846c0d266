// Raw LDS-DMA rate on gfx950: waves stream 1-KiB pieces global -> LDS with a sliding window of outstanding DMAs, no
// barriers, no compute.  Variants: waves per block, window, address form, source region (L2-resident / streamed).
// Build: hipcc --offload-arch=gfx950 -O3 tools/dma_probe2.hip -o tools/bin/dma_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void dma16_so(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma16_va(const void* gsrc, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma4_so(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

// FORM 0: SGPR base + lane offset, 16 B/lane; 1: per-lane 64-bit address, 16 B/lane; 2: SGPR base, 4 B/lane;
// 3: plain global_load_dwordx4 into registers (no LDS) for reference
template <int FORM, int WINDOW>
__global__ void __launch_bounds__(1024) stream(const char* src, long long region, long long per_block, int pieces, int row_stride,
                                               float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_void*)smem;
    const unsigned voff = FORM == 2 ? (unsigned)((lane >> 5) * row_stride + (lane & 31) * 4)
                                    : (unsigned)((lane >> 3) * row_stride + (lane & 7) * 16);
    const long long rows_per_piece = FORM == 2 ? 2 : 8;
    uint4 accv = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < pieces; ++i) {
        long long off = ((long long)blockIdx.x * per_block + ((long long)i * nw + wave) * rows_per_piece * row_stride) % (region - 16LL * row_stride);
        off &= ~127LL;
        const unsigned dst = lds0 + (unsigned)(((i % WINDOW) * nw + wave) * 1024);
        if (FORM == 0) dma16_so(src + off, voff, dst);
        else if (FORM == 1) dma16_va(src + off + voff, dst);
        else if (FORM == 2) dma4_so(src + off, voff, dst);
        else { uint4 v = *(const uint4*)(src + off + voff); accv.x ^= v.x; accv.y ^= v.y; accv.z ^= v.z; accv.w ^= v.w; }
        if (FORM != 3 && i >= WINDOW - 1) {
            if (WINDOW == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (WINDOW == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (WINDOW == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (accv.x == 0x12345678u && accv.y == 1) sink[blockIdx.x] = 1.f;
}

template <int FORM, int WINDOW>
void run(const char* name, const char* src, long long region, int waves, int blocks_per_cu, int row_stride, bool shared_region, float* sink) {
    const int pieces = 512;
    const int blocks = 256 * blocks_per_cu * 4;
    const size_t lds = (size_t)WINDOW * waves * 1024;
    if (lds * blocks_per_cu > 160 * 1024) { printf("%-58s skipped (LDS)\n", name); return; }
    auto k = stream<FORM, WINDOW>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long long per_block = shared_region ? 0 : (long long)pieces * waves * 8 * row_stride;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), lds, 0, src, region, per_block, pieces, row_stride, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double bytes = (double)blocks * pieces * waves * (FORM == 2 ? 256 : 1024);
    printf("%-58s %8.3f ms %8.2f TB/s %7.2f B/clk/CU\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
    const long long big = 4LL << 30;
    char* buf; float* sink;
    CHECK(hipMalloc(&buf, big)); CHECK(hipMalloc(&sink, 1 << 20));
    CHECK(hipMemset(buf, 0, big));
    const long long L2R = 1 << 20;     // 1 MiB: resident in every XCD's L2
    run<0, 8>("L2 1MiB, sgpr-base x4, 8 waves, window 8, stride 1536", buf, L2R, 8, 1, 1536, true, sink);
    run<0, 16>("L2 1MiB, sgpr-base x4, 8 waves, window 16, stride 1536", buf, L2R, 8, 1, 1536, true, sink);
    run<0, 4>("L2 1MiB, sgpr-base x4, 8 waves, window 4, stride 1536", buf, L2R, 8, 1, 1536, true, sink);
    run<0, 8>("L2 1MiB, sgpr-base x4, 4 waves, window 8, stride 1536", buf, L2R, 4, 1, 1536, true, sink);
    run<0, 8>("L2 1MiB, sgpr-base x4, 4 waves x 2 blocks/CU, window 8", buf, L2R, 4, 2, 1536, true, sink);
    run<0, 8>("L2 1MiB, sgpr-base x4, 16 waves, window 8, stride 1536", buf, L2R, 16, 1, 1536, true, sink);
    run<0, 8>("L2 1MiB, sgpr-base x4, 8 waves, window 8, stride 128 (contig)", buf, L2R, 8, 1, 128, true, sink);
    run<1, 8>("L2 1MiB, per-lane addr x4, 8 waves, window 8, stride 1536", buf, L2R, 8, 1, 1536, true, sink);
    run<2, 8>("L2 1MiB, sgpr-base dword, 8 waves, window 8, stride 1536", buf, L2R, 8, 1, 1536, true, sink);
    run<3, 8>("L2 1MiB, plain loads to registers, 8 waves, stride 1536", buf, L2R, 8, 1, 1536, true, sink);
    run<3, 8>("L2 1MiB, plain loads to registers, 16 waves, stride 1536", buf, L2R, 16, 1, 1536, true, sink);
    run<0, 8>("32MiB shared region (MALL), sgpr-base x4, 8 waves, w8", buf, 32LL << 20, 8, 1, 1536, true, sink);
    run<0, 8>("streamed 4GiB (HBM), sgpr-base x4, 8 waves, window 8", buf, big, 8, 1, 1536, false, sink);
    run<0, 16>("streamed 4GiB (HBM), sgpr-base x4, 8 waves, window 16", buf, big, 8, 1, 1536, false, sink);
    run<0, 8>("streamed 4GiB (HBM), sgpr-base x4, 16 waves, window 8", buf, big, 16, 1, 1536, false, sink);
    run<3, 8>("streamed 4GiB (HBM), plain loads to registers, 16 waves", buf, big, 16, 1, 1536, false, sink);
    return 0;
}
