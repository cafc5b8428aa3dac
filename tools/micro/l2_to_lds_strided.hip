// L2 -> LDS streaming rate per CU by ACCESS PATTERN of one global_load_lds_dwordx4 wave instruction (1 KiB):
//   rows = 1: 1 KiB contiguous;  rows = 8: 8 rows x 128 B;  rows = 16: 16 rows x 64 B (row stride S bytes) -- what a GEMM
//   operand panel [rows][K] gives the ring kernel at BK = 64 / BK = 32.  All workgroups walk the same few row bands (L2-resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DEPTH, int ROWS>
__global__ __launch_bounds__(512) void k(const unsigned char* src, int stride, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int SEG = 1024 / ROWS;                         // contiguous bytes per row
    constexpr int LPR = SEG / 16;                            // lanes per row
    const int row = lane / LPR, col = (lane % LPR) * 16;
    // band of ROWS rows per wave (shared by every workgroup); walk along the row in SEG-byte steps
    const unsigned char* base = src + (size_t)(wave * 16 + row) * stride + col;
    unsigned char* dst = smem + wave * (DEPTH * 1024);
    int koff = (blockIdx.x * 37 % 64) * SEG;
    const int kmax = stride / SEG * SEG;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            __builtin_amdgcn_global_load_lds((gptr_t)(base + koff), (lptr_t)(dst + s * 1024), 16, 0, 0);
            koff += SEG;
            if (koff >= kmax) koff = 0;
        }
        wait_vmcnt<DEPTH / 2>();
    }
    wait_vmcnt<0>();
}

template <int DEPTH, int ROWS>
void run(const unsigned char* src, int nwaves, int wg_per_cu, int stride) {
    const int iters = 400, grid = 256 * wg_per_cu;
    const size_t lds = (size_t)nwaves * DEPTH * 1024;
    hipFuncSetAttribute((const void*)k<DEPTH, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<DEPTH, ROWS><<<grid, nwaves * 64, lds>>>(src, stride, 20);
    hipEventRecord(a);
    k<DEPTH, ROWS><<<grid, nwaves * 64, lds>>>(src, stride, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * nwaves * DEPTH * 1024.0 * iters;
    printf("rows x seg %2d x %4d B  stride %6d  waves/WG %d  WG/CU %d  depth %2d : %6.2f TB/s  %5.1f B/clk/CU\n", ROWS, 1024 / ROWS, stride,
           nwaves, wg_per_cu, DEPTH, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.3e9);
}

int main() {
    unsigned char* src;
    hipMalloc(&src, 64u << 20);
    hipMemset(src, 1, 64u << 20);
    for (int stride : {17280, 1152, 384}) {
        run<8, 1>(src, 4, 2, stride);
        run<8, 8>(src, 4, 2, stride);
        run<8, 16>(src, 4, 2, stride);
        run<8, 8>(src, 8, 1, stride);
        run<8, 16>(src, 8, 1, stride);
    }
    return 0;
}
