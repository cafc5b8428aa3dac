// Store-pattern microbenchmark for the GEMM epilogue (tools/micro, not part of the library).
// 256 workgroups x 4 waves; every wave writes a 128 x 96 bf16 sub-tile of a [65536][192] output.
//   mode 0: row-major, 12 lanes x 16 B per row (what the LDS-transposed epilogue issues)
//   mode 1: 16 rows x 64 B per instruction (lane l: row l & 15, 16 B at column 8 * (l >> 4) of a 32-column block)
//   mode 2: 16 rows x 32 B per instruction with 8-byte stores (raw MFMA C^T layout)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint16_t* out, int ld, uint32_t seed) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = blockIdx.x * 256 + (wave >> 1) * 128, n0 = (wave & 1) * 96;
    uint4 v = make_uint4(seed + lane, seed * 3 + lane, seed * 5, seed * 7);
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int r = p * 5 + lane / 12, c = (lane % 12) * 8;
                if (lane < 60 && r < 16) *reinterpret_cast<uint4*>(out + (int64_t)(m0 + i * 16 + r) * ld + n0 + c) = v;
                v.x += 1;
            }
    } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int J = 0; J < 3; ++J) {
                *reinterpret_cast<uint4*>(out + (int64_t)(m0 + i * 16 + (lane & 15)) * ld + n0 + J * 32 + (lane >> 4) * 8) = v;
                v.x += 1;
            }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                *reinterpret_cast<uint2*>(out + (int64_t)(m0 + i * 16 + (lane & 15)) * ld + n0 + j * 16 + (lane >> 4) * 4) = make_uint2(v.x, v.y);
                v.x += 1;
            }
    }
}
int main() {
    const int M = 65536, N = 192;
    uint16_t* out;
    hipMalloc(&out, (size_t)M * N * 2 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            auto launch = [&](int it) {
                uint16_t* o = out + (size_t)(it & 3) * M * N;
                if (mode == 0) k<0><<<256, 256>>>(o, N, it);
                else if (mode == 1) k<1><<<256, 256>>>(o, N, it);
                else k<2><<<256, 256>>>(o, N, it);
            };
            for (int it = 0; it < 20; ++it) launch(it);
            hipEventRecord(a);
            for (int it = 0; it < 200; ++it) launch(it);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("mode %d: %.2f us per 25 MB tile set (%.2f TB/s)\n", mode, ms * 5.f, 25.17e6 / (ms * 5e-6) / 1e12);
        }
    return 0;
}
