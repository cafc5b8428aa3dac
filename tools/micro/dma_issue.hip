// What does ISSUING an LDS-DMA instruction cost a wave?  (tools/micro, not part of the library; r06)
// Every wave of a workgroup issues NP pieces (global_load_lds_dwordx4, 1 KiB each) back to back from an L2-resident panel and stamps
// s_memtime before the first and after the last ISSUE (not the landing), then waits for them; REPS rounds, the minimum and the mean per
// round are reported per form:   0 = 64-bit per-lane address (what __builtin_amdgcn_global_load_lds compiles to in the library's loops),
//                                1 = scalar base + 32-bit per-lane offset (saddr form, inline asm).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/dma_issue tools/micro/dma_issue.hip && tools/micro/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int FORM, int NP>
__global__ __launch_bounds__(512) void k(const unsigned char* src, int reps, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    unsigned char* dst = smem + wave * (NP * 1024);
    const unsigned lds = (unsigned)(size_t)(lptr_t)dst;
    unsigned best = ~0u, sum = 0;
    for (int r = 0; r < reps; ++r) {
        const size_t base = (size_t)((r * 31 + blockIdx.x * 7) & 255) * 65536 + (size_t)wave * NP * 1024;
        __builtin_amdgcn_s_barrier();
        const unsigned t0 = (unsigned)__builtin_readcyclecounter();
        if (FORM == 0) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const unsigned char* q = src + base + p * 1024 + lane * 16;
                __builtin_amdgcn_global_load_lds((gptr_t)q, (lptr_t)(dst + p * 1024), 16, 0, 0);
            }
        } else {
            const unsigned char* sb = src + base;          // wave-uniform
            const unsigned vo = lane * 16;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(lds + p * 1024), "v"(vo), "s"(sb), "n"(0) : "memory");
                sb += 1024;
            }
        }
        const unsigned t1 = (unsigned)__builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned dt = t1 - t0;
        best = dt < best ? dt : best;
        sum += dt;
    }
    if (lane == 0) { out[(blockIdx.x * nw + wave) * 2] = best; out[(blockIdx.x * nw + wave) * 2 + 1] = sum / reps; }
    if (smem[threadIdx.x] == 123 && reps < 0) out[0] = 1;
}

template <int FORM, int NP>
void run(const unsigned char* src, unsigned* out, int nwaves) {
    const int grid = 256, reps = 200;
    hipFuncSetAttribute((const void*)k<FORM, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<FORM, NP><<<grid, nwaves * 64, nwaves * NP * 1024>>>(src, reps, out);
    hipDeviceSynchronize();
    std::vector<unsigned> h(grid * nwaves * 2);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    double mb = 0, mm = 0;
    for (int i = 0; i < grid * nwaves; ++i) { mb += h[2 * i]; mm += h[2 * i + 1]; }
    mb /= grid * nwaves; mm /= grid * nwaves;
    printf("form %d (%s)  waves/WG %d  pieces %d : issue of all pieces  min %.0f  mean %.0f cycles  = %.0f cycles per piece (mean)\n", FORM,
           FORM ? "scalar base + 32-bit offset" : "64-bit per-lane address", nwaves, NP, mb, mm, mm / NP);
}

int main() {
    unsigned char* src; unsigned* out;
    hipMalloc(&src, 32u << 20); hipMalloc(&out, 256 * 8 * 2 * 4);
    hipMemset(src, 1, 32u << 20);
    for (int nwv : {1, 4, 8}) {
        if (nwv == 1) { run<0, 3>(src, out, 1); run<1, 3>(src, out, 1); run<0, 6>(src, out, 1); run<1, 6>(src, out, 1); }
        if (nwv == 4) { run<0, 3>(src, out, 4); run<1, 3>(src, out, 4); run<0, 6>(src, out, 4); run<1, 6>(src, out, 4); }
        if (nwv == 8) { run<0, 3>(src, out, 8); run<1, 3>(src, out, 8); run<0, 6>(src, out, 8); run<1, 6>(src, out, 8); }
    }
    return 0;
}
