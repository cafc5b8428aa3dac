// L2 -> LDS streaming rate per CU (tools/micro, not part of the library): every workgroup streams an L2-resident window
// into LDS with global_load_lds_dwordx4 (1 KiB per wave instruction), DEPTH instructions in flight per wave, no compute.
//   ./l2_to_lds            -> table over waves per workgroup, workgroups per CU, depth, window size
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DEPTH, bool VGPR>
__global__ __launch_bounds__(512) void k(const unsigned char* src, size_t window, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    // each wave walks the window in 1-KiB pieces, waves interleaved; workgroups start at different offsets
    size_t off = ((size_t)blockIdx.x * 7919 % (window >> 10)) << 10;
    unsigned char* dst = smem + wave * (DEPTH * 1024);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const unsigned char* p = src + ((off + (size_t)(s * nw + wave) * 1024) & (window - 1)) + lane * 16;
            if (VGPR) {
                const uint4 v = *reinterpret_cast<const uint4*>(p);
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            } else {
                __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(dst + s * 1024), 16, 0, 0);
            }
        }
        off += (size_t)DEPTH * nw * 1024;
        if (!VGPR) wait_vmcnt<DEPTH / 2>();          // keep half a round in flight across iterations
    }
    if (!VGPR) wait_vmcnt<0>();
    if (acc.x == 0x12345678u && acc.y == 77u) sink[0] = (float)acc.z + (float)acc.w + smem[threadIdx.x];
}

template <int DEPTH, bool VGPR>
void run(const unsigned char* src, float* sink, int nwaves, int wg_per_cu, size_t window) {
    const int iters = 400;
    const int grid = 256 * wg_per_cu;
    const size_t lds = VGPR ? 1024 : (size_t)nwaves * DEPTH * 1024;
    hipFuncSetAttribute((const void*)k<DEPTH, VGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<DEPTH, VGPR><<<grid, nwaves * 64, lds>>>(src, window, 20, sink);
    hipEventRecord(a);
    k<DEPTH, VGPR><<<grid, nwaves * 64, lds>>>(src, window, iters, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * nwaves * DEPTH * 1024.0 * iters;
    printf("%s waves/WG %d  WG/CU %d  depth %2d  window %6.1f MB : %6.2f TB/s  %5.1f B/clk/CU (2.3 GHz)\n", VGPR ? "vgpr" : "lds ",
           nwaves, wg_per_cu, DEPTH, window / 1048576.0, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.3e9);
}

int main() {
    unsigned char* src; float* sink;
    hipMalloc(&src, 512u << 20); hipMalloc(&sink, 64);
    hipMemset(src, 1, 512u << 20);
    for (size_t window : {(size_t)1 << 20, (size_t)16 << 20, (size_t)512 << 20}) {
        run<4, false>(src, sink, 4, 1, window);
        run<8, false>(src, sink, 4, 1, window);
        run<16, false>(src, sink, 4, 1, window);
        run<8, false>(src, sink, 4, 2, window);
        run<8, false>(src, sink, 8, 1, window);
        run<16, false>(src, sink, 8, 1, window);
        run<8, true>(src, sink, 4, 1, window);
        run<16, true>(src, sink, 4, 2, window);
    }
    return 0;
}
