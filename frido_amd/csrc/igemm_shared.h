// Device helpers shared by the translation units of the implicit-GEMM family (igemm.hip: ring + bf16 patch kernels; convgn.hip: the
// fused GroupNorm + conv kernel): LDS / waitcnt primitives, the XCD-aware work-item mapping and the common tile epilogue.
#pragma once
#include "common.h"
#include <type_traits>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ bf16x8 lds_read128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// dispatch-order id of this workgroup within its batch slice -> work item (see igemm_kernel): XCD x = id & 7 takes items
// [start(x), start(x) + count(x)) of the nb * gridDim.z items
__device__ __forceinline__ int xcd_item(int nb) {
    const int total = nb * (int)gridDim.z;
    const int id = (int)blockIdx.z * (int)gridDim.x + (int)blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = id & 7, loc = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// r05 experiment for round 6 (-DFRIDO_STAGGER_RT=1 builds only; the shipped library is built without and contains none of it): the
// ONE-workgroup-per-CU kernels (8-wave igemm tiles, the fused GroupNorm + conv kernel) run a launch's prologue loads, its k-loop and its
// epilogue stores chip-wide at the same time.  FridoGemm.flags bit 26 lets the odd XCDs (dispatch id & 1) start bits 8..15 quarter
// microseconds late, so that one half's HBM phases fall under the other half's MFMA phase.  Results unchanged.
#ifndef FRIDO_STAGGER_RT
#define FRIDO_STAGGER_RT 0
#endif
// (r06) First weight slab of a plain-loop k-step (j == 0): the MFMAs of pixel slab i used to issue as soon as ITS fragments were back --
// three MFMAs on ONE accumulator in a row (hi*lo, lo*hi, hi*hi), the first separated from the second by the s_waitcnt of the lo plane: a
// dependent MFMA chain with an issue slot in it (MI355X_MICROARCH.md constants table: +43 cycles for the first extra state between two MFMAs
// on the same accumulator; a plain dependent pair waits for the pass pipeline too).  1: slabs are issued in PAIRS, pass-major (acc[i] and
// acc[i + 1] alternate: every dependent MFMA has an independent one in front of it); 2: all TM slabs pass-major after ONE wait, like the
// later weight slabs, which the compiler already orders that way; 0: the r03 order.  The per-accumulator order of the three products is
// unchanged in every form: bit-identical results.
#ifndef FRIDO_SLAB0
#define FRIDO_SLAB0 1
#endif
// (r06) 1: the plain loop of the two-slot two-plane tiles keeps its per-k-tile barrier INSIDE the k-step (igemm.hip "MID-STEP BARRIER"); 0: at its top
#ifndef FRIDO_MIDBAR
#define FRIDO_MIDBAR 0
#endif
__device__ __forceinline__ void stagger_one_per_cu(int flags) {
#if FRIDO_STAGGER_RT
    const int ticks = ((flags >> 8) & 255) * 25;           // 100 MHz
    const int id = ((int)blockIdx.y * (int)gridDim.z + (int)blockIdx.z) * (int)gridDim.x + (int)blockIdx.x;
    if (((flags >> 26) & 1) && ticks && (id & 1) && id < 256) {
        const uint64_t t0 = wall_clock64();
        while (wall_clock64() - t0 < (uint64_t)ticks) __builtin_amdgcn_s_sleep(4);
    }
#else
    (void)flags;
#endif
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int OFF>
__device__ __forceinline__ float4 lds_read128f(unsigned addr) {
    float4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ uint4 lds_read128u(unsigned addr) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// ---- epilogue (shared by the ring kernel and the patch-staged 3x3 kernel) --------------------------------------
// Accumulator layout.  The main loops issue every MFMA with the WEIGHT fragment as the first operand and the pixel fragment
// as the second, so the 16x16 accumulator tile is C^T: lane l holds, for pixel (row of C) l & 15, the FOUR CONSECUTIVE
// channels 4 * (l >> 4) + e of the tile.  On top of that the weight rows of a tile are dealt to LDS rows PERMUTED
// (chan_of_pos below, applied to the DMA source address: free): n-tile pair (2J, 2J+1) of a lane covers channels
// 32 J + 8 (l >> 4) + {0..3 | 4..7}, i.e. EIGHT CONSECUTIVE channels of one pixel = one 16-byte bf16 store.  The hot
// epilogues therefore go straight from accumulators to global memory: no LDS transposition (r01/r02 spent 3-7 us of a
// 54 us conv launch writing slabs to LDS and reading them back), no lgkmcnt round trips, one 16-byte store per 8 values,
// 16 pixel rows x 64 B per wave instruction (store-pattern microbenchmark tools/micro/store_pattern.hip: 6.2 vs 5.5 us per
// 25 MB for full rows -- the stores cost 0.7 us more, the transposition they replace 3+).
// GEGLU launches keep the identity row order (value / gate tiles must stay adjacent fragments).
__device__ __forceinline__ int chan_of_pos(int p) {      // LDS row (position in the tile's B panel) -> channel of the tile
    return (p & ~31) | ((p & 12) << 1) | ((p & 16) >> 2) | (p & 3);
}

constexpr int SK_HDR = 16384;           // split-K workspace header: arrival tickets (uint32 each), then the partial sums

template <int OFF>
__device__ __forceinline__ void lds_write128(unsigned addr, f32x4 v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

template <int BM, int BN, int NS, int WM, bool RSTAGE>
__device__ __forceinline__ void tile_epilogue(const FridoGemm& d, f32x4 (&acc)[BM / WM / 16][BN / 2 / 16], unsigned char* smem,
                                              int m0, int n0, int wave, int lane, int zo, int zi, int kz, bool reduced = false) {
    // `reduced`: split-K launch whose slices were already added up in this workgroup's accumulators (FridoGemm.sk_mode 1): the
    // full epilogue runs as if there were no split
    constexpr int WN = 2, TM = BM / WM / 16, TN = BN / WN / 16, TJ = TN / 2;
    static_assert(TN % 2 == 0, "n-tile pairs");
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int WR = BM / WM, WC = BN / WN, EPS = WC + 4;          // +4 floats: conflict-free slab writes
    constexpr int LPR8 = WC / 8, RPP8 = 64 / LPR8;                    // slab read-back: lanes per row (8 columns each), rows per pass
    if (d.act == 99 || (d.flags & 4)) {      // profiling aid (flags bit 2: the same inside a captured graph) (tools/gemm_bench.py NOEPI=1): keep the accumulators live, store nothing
        float sink = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) sink += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sink == 1.2345e-30f) d.out_f32[0] = sink;
        return;
    }
    const int px_l = lane & 15, cg = lane >> 4;                        // this lane's pixel row in a 16-row slab, its channel group
    const bool perm = !d.geglu;
    int vstep = 0;
    if (d.rowvec && d.rowvec_step) vstep = *d.rowvec_step;
    const int nbase = n0 + wn * WC;
    // vector paths (8 columns per lane: 16-byte bf16 / 2 x 16-byte f32 accesses) need 8-element aligned rows and planes
    const bool partial = gridDim.z > 1 && !reduced;
    const bool vec_ok = partial ? (d.N & 3) == 0
                                      : ((d.ldo | d.ldr | d.ldoo | d.ldv | d.N | d.of_bs | d.of_bs2 | d.oo_bs | d.oo_bs2 | d.res_bs | d.oo_lo) & 7) == 0;
    const int64_t of_base = (int64_t)zo * d.of_bs + (int64_t)zi * d.of_bs2;
    const int64_t oo_base = (int64_t)zo * d.oo_bs + (int64_t)zi * d.oo_bs2;
    const int64_t rs_base = (int64_t)zo * d.res_bs;
    float* wsp = partial ? d.ws + SK_HDR + (int64_t)kz * d.M * d.N : nullptr;
    // 2x2 phase convolution of an upsample: GEMM row (img, y, x) -> output row (img, 2y + a, 2x + b); Ho, Wo powers of two
    const int up2 = d.up2_phase == 5 ? zo + 1 : d.up2_phase;
    const int lw = 31 - __builtin_clz((unsigned)(d.Wo > 0 ? d.Wo : 1)), lhw = lw + 31 - __builtin_clz((unsigned)(d.Ho > 0 ? d.Ho : 1));
    auto out_row = [&](int m) -> int64_t {
        if (!up2) return m;
        const int img = m >> lhw, rem = m & ((1 << lhw) - 1);
        const int y = rem >> lw, x = rem & ((1 << lw) - 1);
        const int a = (up2 - 1) >> 1, bq = (up2 - 1) & 1;
        return ((int64_t)(img * 2 * d.Ho + 2 * y + a) * (2 * d.Wo)) + 2 * x + bq;
    };
    const bool fast = vec_ok && (d.N & 7) == 0 && !d.out_u8;      // every lane's 8 columns are then all inside or all outside N (uint8 images: element-wise path)
    const bool rv_hoist = d.rowvec && d.rows_per_vec >= (1 << 29) && !(d.flags & 2);
    // ---- direct paths (accumulators -> global) ----
    const bool plain = fast && perm && d.act == FRIDO_ACT_NONE && !d.row_bias && (!d.rowvec || rv_hoist) && !(d.flags & 16);
    const float alpha = d.alpha;                                        // (split-K partials stay raw: the reduce kernel scales)
    const int ncol0 = nbase + 8 * cg;                                   // first of this lane's 8 channels of pair 0 (+32 per pair)
    if (plain && wsp) {                                                 // split-K: raw partial sums to the workspace
        static_for<0, TM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int m = m0 + wm * WR + i * 16 + px_l;
            float* w = wsp + (int64_t)m * d.N + ncol0;
            static_for<0, TJ>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                if (m < d.M && ncol0 + 32 * J < d.N) {
                    *reinterpret_cast<f32x4*>(w + 32 * J) = acc[i][2 * J];
                    *reinterpret_cast<f32x4*>(w + 32 * J + 4) = acc[i][2 * J + 1];
                }
            });
        });
        return;
    }
    bool direct = false;
    if constexpr (NS == 1) {
        const bool one_out = (d.out_f32 && d.out_bf16 && !d.out_op) || (d.out_op && !d.out_f32);
        direct = plain && !wsp && one_out && (!d.residual || (RSTAGE && d.res_bf16 && !(d.flags & 1)));
    } else {
        const bool one_out = (d.out_f32 && !d.out_bf16 && !d.out_op) || (d.out_op && !d.out_f32);
        direct = plain && !wsp && one_out && !(d.residual && d.res_bf16) && !up2;
    }
    if (direct) {
        // bias (+ the launch-wide timestep vector: rows_per_vec >= 2^29 means ONE vector serves every row, so it is a second bias)
        float bia[TJ][8];
        static_for<0, TJ>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            const int n = ncol0 + 32 * J;
#pragma unroll
            for (int e = 0; e < 8; ++e) bia[J][e] = 0.f;
            if (n < d.N) {
                if (d.bias) {
                    const float4 t0 = *reinterpret_cast<const float4*>(d.bias + n), t1 = *reinterpret_cast<const float4*>(d.bias + n + 4);
                    bia[J][0] = t0.x; bia[J][1] = t0.y; bia[J][2] = t0.z; bia[J][3] = t0.w;
                    bia[J][4] = t1.x; bia[J][5] = t1.y; bia[J][6] = t1.z; bia[J][7] = t1.w;
                }
                if (rv_hoist) {
                    const float* rp = d.rowvec + (int64_t)vstep * d.ldv + n;
                    const float4 t0 = *reinterpret_cast<const float4*>(rp), t1 = *reinterpret_cast<const float4*>(rp + 4);
                    bia[J][0] += t0.x; bia[J][1] += t0.y; bia[J][2] += t0.z; bia[J][3] += t0.w;
                    bia[J][4] += t1.x; bia[J][5] += t1.y; bia[J][6] += t1.z; bia[J][7] += t1.w;
                }
            }
        });
        if constexpr (NS == 1) {
            // bf16 stream / operand output.  A bf16 residual comes through LDS: the wave's [WR][WC] sub-tile is DMA'd into the
            // idle ring (L2 -> LDS, no VGPRs, nothing waits until it is needed) and read back 16 bytes per (pixel, pair).
            const bool has_res = d.residual != nullptr;
            unsigned char* rstage = smem + wave * (WR * WC * 2);
            if (has_res) {
                if constexpr (RSTAGE) {
                    __builtin_amdgcn_s_barrier();                          // every wave is done reading the ring
                    constexpr int NI = WR * WC / 512;                      // 1-KiB pieces of the sub-tile
                    const frido_bf16* rbase = reinterpret_cast<const frido_bf16*>(d.residual) + rs_base;
#pragma unroll
                    for (int k = 0; k < NI; ++k) {
                        const int L = k * 64 + lane, row = L / LPR8, c8 = L - row * LPR8;
                        int m = m0 + wm * WR + row, n = nbase + c8 * 8;
                        m = m < d.M ? m : d.M - 1;
                        n = n + 8 <= d.N ? n : 0;                          // columns past N: any valid address (masked later)
                        __builtin_amdgcn_global_load_lds((gptr_t)(rbase + (int64_t)m * d.ldr + n), (lptr_t)(rstage + k * 1024), 16, 0, 0);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's own DMA: no barrier needed
                }
            }
            frido_bf16* obase = d.out_f32 ? reinterpret_cast<frido_bf16*>(d.out_f32) + of_base : d.out_op + oo_base;
            const int64_t ldout = d.out_f32 ? d.ldo : d.ldoo;
            const unsigned rsa = (unsigned)(size_t)(lptr_t)rstage + (unsigned)(px_l * LPR8 + cg) * 16u;
            uint4 rs[2][TJ];
            if (has_res) {
                static_for<0, TJ>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    rs[0][J] = lds_read128u<(4 * J) * 16>(rsa);
                });
            }
            static_for<0, TM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (has_res) {
                    if constexpr (i + 1 < TM) {
                        static_for<0, TJ>([&](auto jc) {
                            constexpr int J = decltype(jc)::value;
                            rs[(i + 1) & 1][J] = lds_read128u<(((i + 1) * 16) * LPR8 + 4 * J) * 16>(rsa);
                        });
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TJ) : "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int m = m0 + wm * WR + i * 16 + px_l;
                frido_bf16* orow = obase + out_row(m) * ldout + ncol0;     // out_row: upsample phase interleave
                static_for<0, TJ>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    float v[8] = {fmaf(acc[i][2 * J][0], alpha, bia[J][0]), fmaf(acc[i][2 * J][1], alpha, bia[J][1]),
                                  fmaf(acc[i][2 * J][2], alpha, bia[J][2]), fmaf(acc[i][2 * J][3], alpha, bia[J][3]),
                                  fmaf(acc[i][2 * J + 1][0], alpha, bia[J][4]), fmaf(acc[i][2 * J + 1][1], alpha, bia[J][5]),
                                  fmaf(acc[i][2 * J + 1][2], alpha, bia[J][6]), fmaf(acc[i][2 * J + 1][3], alpha, bia[J][7])};
                    if (has_res) {
                        const uint4 u = rs[i & 1][J];
                        v[0] += __uint_as_float(u.x << 16); v[1] += __uint_as_float(u.x & 0xffff0000u);
                        v[2] += __uint_as_float(u.y << 16); v[3] += __uint_as_float(u.y & 0xffff0000u);
                        v[4] += __uint_as_float(u.z << 16); v[5] += __uint_as_float(u.z & 0xffff0000u);
                        v[6] += __uint_as_float(u.w << 16); v[7] += __uint_as_float(u.w & 0xffff0000u);
                    }
                    if (m < d.M && ncol0 + 32 * J < d.N)
                        *reinterpret_cast<uint4*>(orow + 32 * J) =
                            make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
                });
            });
        } else {
            // bf16x3 (parity) mode: f32 stream or hi/lo operand output; an f32 residual row segment is fetched one slab ahead
            const bool has_res = d.residual != nullptr;
            const float* rbase = reinterpret_cast<const float*>(d.residual) + rs_base + ncol0;
            float4 r0[2][TJ], r1[2][TJ];
            auto fetch = [&](auto ic, auto slot) {
                constexpr int i = decltype(ic)::value, sl = decltype(slot)::value;
                int m = m0 + wm * WR + i * 16 + px_l;
                m = m < d.M ? m : d.M - 1;
                static_for<0, TJ>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    r0[sl][J] = make_float4(0.f, 0.f, 0.f, 0.f);
                    r1[sl][J] = r0[sl][J];
                    if (ncol0 + 32 * J < d.N) {
                        r0[sl][J] = *reinterpret_cast<const float4*>(rbase + (int64_t)m * d.ldr + 32 * J);
                        r1[sl][J] = *reinterpret_cast<const float4*>(rbase + (int64_t)m * d.ldr + 32 * J + 4);
                    }
                });
            };
            if (has_res) fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            // GroupNorm statistics for free (r03): the values a lane stores are 8 consecutive channels of one pixel, so per-channel
            // {sum, sum of squares} over a 32-row block = two slabs in registers + a 4-step reduction over the 16 pixel lanes;
            // frido_gn_stats sums these partials (3 % of the tensor's bytes) instead of re-reading the tensor
            float* const gnp = d.gn_part;
            float gs[TJ][8], gq[TJ][8];
            static_for<0, TM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i + 1 < TM) {
                    if (has_res) fetch(std::integral_constant<int, i + 1>{}, std::integral_constant<int, (i + 1) & 1>{});
                }
                const int m = m0 + wm * WR + i * 16 + px_l;
                static_for<0, TJ>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    float v[8] = {fmaf(acc[i][2 * J][0], alpha, bia[J][0]), fmaf(acc[i][2 * J][1], alpha, bia[J][1]),
                                  fmaf(acc[i][2 * J][2], alpha, bia[J][2]), fmaf(acc[i][2 * J][3], alpha, bia[J][3]),
                                  fmaf(acc[i][2 * J + 1][0], alpha, bia[J][4]), fmaf(acc[i][2 * J + 1][1], alpha, bia[J][5]),
                                  fmaf(acc[i][2 * J + 1][2], alpha, bia[J][6]), fmaf(acc[i][2 * J + 1][3], alpha, bia[J][7])};
                    if (has_res) {
                        const float4 a = r0[i & 1][J], b = r1[i & 1][J];
                        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
                        v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
                    }
                    if (gnp) {
                        const bool on = m < d.M && ncol0 + 32 * J < d.N;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float x = on ? v[e] : 0.f;
                            if constexpr ((i & 1) == 0) { gs[J][e] = x; gq[J][e] = x * x; }
                            else { gs[J][e] += x; gq[J][e] = fmaf(x, x, gq[J][e]); }
                        }
                        if constexpr ((i & 1) == 1) {       // a 32-row block is complete: reduce over the 16 pixel lanes, lane px_l == 0 stores
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
#pragma unroll
                                for (int o = 1; o < 16; o <<= 1) {
                                    gs[J][e] += __shfl_xor(gs[J][e], o, 64);
                                    gq[J][e] += __shfl_xor(gq[J][e], o, 64);
                                }
                            }
                            const int blk = (m0 + wm * WR + (i - 1) * 16) >> 5;
                            if (px_l == 0 && ncol0 + 32 * J < d.N && (blk << 5) < d.M) {
                                float* o = gnp + ((int64_t)blk * d.N + ncol0 + 32 * J) * 2;
#pragma unroll
                                for (int e = 0; e < 8; e += 2)
                                    *reinterpret_cast<float4*>(o + 2 * e) = make_float4(gs[J][e], gq[J][e], gs[J][e + 1], gq[J][e + 1]);
                            }
                        }
                    }
                    if (m < d.M && ncol0 + 32 * J < d.N) {
                        if (d.out_f32) {
                            float* o = d.out_f32 + of_base + (int64_t)m * d.ldo + ncol0 + 32 * J;
                            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        } else {
                            uint32_t h[8], l[8];
#pragma unroll
                            for (int e = 0; e < 8; e += 2) { split_op2(v[e], v[e + 1], NS, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
                            if (op_sat8(v)) status_raise(true);      // (rare branch: no flag register kept live across the tile)
                            frido_bf16* op = d.out_op + oo_base + (int64_t)m * d.ldoo + ncol0 + 32 * J;
                            *reinterpret_cast<uint4*>(op) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                            *reinterpret_cast<uint4*>(op + d.oo_lo) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
                        }
                    }
                });
            });
        }
        return;
    }

    // ---- paths through LDS: each 16-row slab of the wave's sub-tile is written to the (now idle) ring as [pixel][channel] f32
    //      and read back so that every lane owns 8 consecutive columns of one row (activations, GEGLU, two outputs, row
    //      vectors, ragged shapes).  A lane's four accumulator values of a tile are four consecutive channels: one
    //      ds_write_b128 per tile.
    __builtin_amdgcn_s_barrier();                                      // every wave is done reading the ring
    if (d.act == 96) { if (acc[0][0][0] == 1.2345e-30f) d.out_f32[0] = acc[0][0][1]; return; }
    float* ep = reinterpret_cast<float*>(smem) + wave * (16 * EPS);
    const int er8 = lane / LPR8, ec8 = (lane - er8 * LPR8) * 8;        // this lane's (row, first column) in a read-back pass
    const bool lane_on8 = lane < RPP8 * LPR8;
    constexpr int NP = (16 + RPP8 - 1) / RPP8;                         // passes per 16-row slab
    const int ncol = nbase + ec8;
    // slab column of tile j's four values of this lane
    auto slab_col = [&](int j) { return perm ? 32 * (j >> 1) + 8 * cg + 4 * (j & 1) : 16 * j + 4 * cg; };
    float bia[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bia[e] = 0.f;
    if (fast && d.bias && !wsp && !d.geglu && lane_on8 && ncol < d.N) {
        const float4 t0 = *reinterpret_cast<const float4*>(d.bias + ncol), t1 = *reinterpret_cast<const float4*>(d.bias + ncol + 4);
        bia[0] = t0.x; bia[1] = t0.y; bia[2] = t0.z; bia[3] = t0.w; bia[4] = t1.x; bia[5] = t1.y; bia[6] = t1.z; bia[7] = t1.w;
    }
    if (fast && rv_hoist && !wsp && !d.geglu && lane_on8 && ncol < d.N) {
        const float* rp = d.rowvec + (int64_t)vstep * d.ldv + ncol;
        const float4 t0 = *reinterpret_cast<const float4*>(rp), t1 = *reinterpret_cast<const float4*>(rp + 4);
        bia[0] += t0.x; bia[1] += t0.y; bia[2] += t0.z; bia[3] += t0.w; bia[4] += t1.x; bia[5] += t1.y; bia[6] += t1.z; bia[7] += t1.w;
    }
    // GEGLU (attention.py:42-44): the projection's rows are packed so that 16-row blocks alternate [a | gate] and the launch keeps
    // the identity row order: a value and its gate are then the SAME element of adjacent accumulator fragments, so
    // a * gelu(gate) is formed in registers and only the WC/2 outputs go through the LDS transposition.
    float gba[TJ][4], gbg[TJ][4];
    if (d.geglu) {
#pragma unroll
        for (int jo = 0; jo < TJ; ++jo)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = nbase + jo * 32 + 4 * cg + e;
                gba[jo][e] = d.bias && n + 16 < d.N ? d.bias[n] : 0.f;
                gbg[jo][e] = d.bias && n + 16 < d.N ? d.bias[n + 16] : 0.f;
            }
    }
    {
        // the fused GEGLU projection of the sampler (r04: both arithmetic modes -- the two-plane mode ran the rolled generic loop
        // below, whose epilogue cost twice its 12 k-steps of MFMAs on the 32^2 planes): every LDS access in inline asm, one
        // hand-counted lgkmcnt(0) per slab (hipcc fences C++ LDS reads of a kernel that issues LDS-DMA with vmcnt(0), which on
        // CDNA4 also waits for STORES).  (x + b) == fma(x, 1, b): the values are the generic loop's, bit for bit.
        if (d.geglu && d.out_op && d.alpha == 1.0f && !(d.flags & 80)) {
            constexpr int OC = WC / 2, LPRG = OC / 8, RPPG = 64 / LPRG < 16 ? 64 / LPRG : 16, NPG = (16 + RPPG - 1) / RPPG;
            const int gr = lane / LPRG, oc = (lane - gr * LPRG) * 8;          // row, first output column of this lane
            const int no = (nbase >> 1) + oc;
            const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
            const unsigned wa = lds0 + (unsigned)((wave * 16 + px_l) * EPS + 4 * cg) * 4u;
            const unsigned ra = lds0 + (unsigned)((wave * 16 + gr) * EPS + oc) * 4u;
            const bool col_ok = lane < RPPG * LPRG && no + 7 < (d.N >> 1);
            static_for<0, TM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                static_for<0, TJ>([&](auto jc) {
                    constexpr int jo = decltype(jc)::value;
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (acc[i][2 * jo][e] + gba[jo][e]) * gelu_f(acc[i][2 * jo + 1][e] + gbg[jo][e]);
                    lds_write128<jo * 16 * 4>(wa, o);
                });
                float4 lo[NPG], hi[NPG];
                static_for<0, NPG>([&](auto pc) {
                    constexpr int pp = decltype(pc)::value;
                    lo[pp] = lds_read128f<pp * RPPG * EPS * 4>(ra);
                    hi[pp] = lds_read128f<pp * RPPG * EPS * 4 + 16>(ra);
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, NPG>([&](auto pc) {
                    constexpr int pp = decltype(pc)::value;
                    const int r = pp * RPPG + gr;
                    const int m = m0 + wm * WR + i * 16 + r;
                    if (col_ok && r < 16 && m < d.M) {
                        const float4 a = lo[pp], b = hi[pp];
                        if constexpr (NS == 1) {
                            *reinterpret_cast<uint4*>(d.out_op + (int64_t)m * d.ldoo + no) =
                                make_uint4(pack2_bf16(a.x, a.y), pack2_bf16(a.z, a.w), pack2_bf16(b.x, b.y), pack2_bf16(b.z, b.w));
                        } else {
                            const float ov[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                            uint32_t h[8], l[8];
#pragma unroll
                            for (int e = 0; e < 8; e += 2) { split_op2(ov[e], ov[e + 1], NS, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
                            if (op_sat8(ov)) status_raise(true);
                            frido_bf16* op = d.out_op + (int64_t)m * d.ldoo + no;
                            *reinterpret_cast<uint4*>(op) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                            *reinterpret_cast<uint4*>(op + d.oo_lo) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
                        }
                    }
                });
            });
            return;
        }
    }
    for (int i = 0; i < TM; ++i) {
        // static accumulator indices only: if the compiler keeps this (large) loop rolled, acc[i] with a dynamic i would
        // move the whole accumulator tile to scratch
        f32x4 sel[TN];                   // this slab's accumulators, picked with static indices (see above)
#pragma unroll
        for (int ii = 0; ii < TM; ++ii)
            if (ii == i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) sel[j] = acc[ii][j];
            }
        if (d.geglu) {
#pragma unroll
            for (int jo = 0; jo < TJ; ++jo) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(sel[2 * jo][e], d.alpha, gba[jo][e]) * gelu_f(fmaf(sel[2 * jo + 1][e], d.alpha, gbg[jo][e]));
                *reinterpret_cast<f32x4*>(ep + px_l * EPS + jo * 16 + 4 * cg) = o;
            }
            constexpr int OC = WC / 2;                                        // output columns of this wave
            constexpr int LPRG = OC / 8, RPPG = 64 / LPRG < 16 ? 64 / LPRG : 16;
            const int gr = lane / LPRG, oc = (lane - gr * LPRG) * 8;          // row, first output column of this lane
            const int no = (nbase >> 1) + oc;
#pragma unroll 1
            for (int ps = 0; ps < 16; ps += RPPG) {
                const int r = ps + gr;
                const int m = m0 + wm * WR + i * 16 + r;
                if (lane >= RPPG * LPRG || r >= 16 || m >= d.M || no + 7 >= (d.N >> 1)) continue;
                const float* sa = ep + r * EPS + oc;
                const float4 o0 = *reinterpret_cast<const float4*>(sa), o1 = *reinterpret_cast<const float4*>(sa + 4);
                const float ov[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                uint32_t h[8], l[8];
#pragma unroll
                for (int e = 0; e < 8; e += 2) { split_op2(ov[e], ov[e + 1], NS, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
                if (NS == 2 && op_sat8(ov)) status_raise(true);
                frido_bf16* op = d.out_op + (int64_t)m * d.ldoo + no;
                *reinterpret_cast<uint4*>(op) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                if (d.nsplit == 2)
                    *reinterpret_cast<uint4*>(op + d.oo_lo) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(ep + px_l * EPS + slab_col(j)) = sel[j];
        if (!fast) {
            // generic element-wise path (ragged N or unaligned strides: the 3-channel output conv, odd test shapes).  Rolled
            // and scalar on purpose: unrolled per-element fallbacks inside the vector path tripled the kernel's code size.
#pragma unroll 1
            for (int idx = lane; idx < 16 * WC; idx += 64) {
                const int r = idx / WC, c = idx - r * WC;
                const int m = m0 + wm * WR + i * 16 + r, n = nbase + c;
                if (m >= d.M || n >= d.N) continue;
                float x = ep[r * EPS + c];
                if (wsp) { wsp[(int64_t)m * d.N + n] = x; continue; }
                x = x * d.alpha + (d.bias ? d.bias[n] : 0.f) + (d.row_bias ? d.row_bias[m] : 0.f);
                if (d.rowvec) x += d.rowvec[(int64_t)(m / d.rows_per_vec + vstep) * d.ldv + n];
                if (d.act == FRIDO_ACT_RELU) x = fmaxf(x, 0.f);
                else if (d.act == FRIDO_ACT_SILU) x = silu_f(x);
                else if (d.act == FRIDO_ACT_GELU) x = gelu_f(x);
                else if (d.act == FRIDO_ACT_QUICKGELU) x = quickgelu_f(x);
                if (d.residual) x += load_act1(d.residual, rs_base + (int64_t)m * d.ldr + n, d.res_bf16);
                const int64_t mo = out_row(m);
                if (d.out_f32) store_act1(d.out_f32, of_base + mo * d.ldo + n, d.out_bf16, x);
                if (d.out_op) store_op1(d.out_op + oo_base, d.oo_lo, d.nsplit, mo * d.ldoo + n, x);
                if (NS == 2 && d.out_op && op_sat(x)) status_raise(true);
                if (d.out_u8) {      // sample_diffusion.py:103-121: every step its own fp32 rounding (no contraction), then truncation
                    float u;
                    if (d.u8_mode == 2) u = __fmul_rn(255.0f, __fmul_rn(__fadd_rn(fminf(fmaxf(x, -1.0f), 1.0f), 1.0f), 0.5f));
                    else u = fminf(fmaxf(__fmul_rn(__fadd_rn(x, 1.0f), 127.5f), 0.0f), 255.0f);
                    d.out_u8[mo * d.ldu8 + n] = (uint8_t)u;
                }
            }
            continue;
        }
#pragma unroll 1
        for (int p = 0; p < NP; ++p) {
            const int r = p * RPP8 + er8;
            const int m = m0 + wm * WR + i * 16 + r, n = ncol;
            if (!lane_on8 || r >= 16 || m >= d.M || n >= d.N) continue;
            float v[8];
            {
                const float4 a4 = *reinterpret_cast<const float4*>(ep + r * EPS + ec8);
                const float4 b4 = *reinterpret_cast<const float4*>(ep + r * EPS + ec8 + 4);
                v[0] = a4.x; v[1] = a4.y; v[2] = a4.z; v[3] = a4.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
            }
            if (wsp) {                       // split-K: raw partial sums; splitk_reduce_kernel applies the epilogue
                float* w = wsp + (int64_t)m * d.N + n;
                *reinterpret_cast<float4*>(w) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(w + 4) = make_float4(v[4], v[5], v[6], v[7]);
                continue;
            }
            float rb = d.row_bias ? d.row_bias[m] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * d.alpha + bia[e] + rb;
            if (d.rowvec && !rv_hoist) {
                const float* rp = d.rowvec + (int64_t)(m / d.rows_per_vec + vstep) * d.ldv + n;
                const float4 t0 = *reinterpret_cast<const float4*>(rp), t1 = *reinterpret_cast<const float4*>(rp + 4);
                v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
            }
            // the activation switch stays OUTSIDE the element loops (inside, hipcc if-converts it and evaluates expf and the
            // erff polynomial for every element of every GEMM)
            if (d.act == FRIDO_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (d.act == FRIDO_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
            } else if (d.act == FRIDO_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
            } else if (d.act == FRIDO_ACT_QUICKGELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = quickgelu_f(v[e]);
            }
            if (d.residual) {
                const int64_t ro = rs_base + (int64_t)m * d.ldr + n;
                if (d.res_bf16) {
                    const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const frido_bf16*>(d.residual) + ro);
                    v[0] += __uint_as_float(u.x << 16); v[1] += __uint_as_float(u.x & 0xffff0000u);
                    v[2] += __uint_as_float(u.y << 16); v[3] += __uint_as_float(u.y & 0xffff0000u);
                    v[4] += __uint_as_float(u.z << 16); v[5] += __uint_as_float(u.z & 0xffff0000u);
                    v[6] += __uint_as_float(u.w << 16); v[7] += __uint_as_float(u.w & 0xffff0000u);
                } else {
                    const float* rp = reinterpret_cast<const float*>(d.residual) + ro;
                    const float4 t0 = *reinterpret_cast<const float4*>(rp), t1 = *reinterpret_cast<const float4*>(rp + 4);
                    v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
                }
            }
            if (d.act == 98 || (d.flags & 8)) { if (v[0] == 1.2345e-30f) d.out_f32[0] = v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7]; continue; }
            if (d.out_f32) {
                const int64_t o = of_base + out_row(m) * d.ldo + n;
                if (d.out_bf16) {
                    *reinterpret_cast<uint4*>(reinterpret_cast<frido_bf16*>(d.out_f32) + o) =
                        make_uint4(f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16), f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16),
                                   f32_to_bf16_bits(v[4]) | (f32_to_bf16_bits(v[5]) << 16), f32_to_bf16_bits(v[6]) | (f32_to_bf16_bits(v[7]) << 16));
                } else {
                    *reinterpret_cast<float4*>(d.out_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(d.out_f32 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
            if (d.out_op) {
                uint32_t h[8], l[8];
#pragma unroll
                for (int e = 0; e < 8; e += 2) { split_op2(v[e], v[e + 1], NS, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
                if (NS == 2 && op_sat8(v)) status_raise(true);
                frido_bf16* op = d.out_op + oo_base + out_row(m) * d.ldoo + n;
                *reinterpret_cast<uint4*>(op) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                if (d.nsplit == 2)
                    *reinterpret_cast<uint4*>(op + d.oo_lo) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
            }
        }
    }
}

}  // namespace
