// Fused attention core for short key sequences (gfx950): O = softmax(alpha * Q K^T) V with Nk <= 128 keys, one launch,
// the score matrix living in registers / LDS only.  Replaces the QK^T GEMM -> row softmax -> PV GEMM chain of the
// U-Net's cross-attention (Nk = 26 / 92 / 1 context tokens) and of self-attention on its 8x8 plane
// (frido/modules/attention.py:170-193).  Those launches move a few hundred KB each and were pure launch latency.
//
// One workgroup = 16 query rows x NW waves (4, 8 or 16).
//   phase 1  the d/32 k-steps of S = Q K^T are dealt round-robin to the waves; every lane feeds the MFMA straight from
//            global memory (16-byte operand loads, K rows clamped to the last valid key); partial scores go to LDS.
//   phase 2  the 16 rows are dealt to the waves: each sums the partials (fixed order), scales, masks keys >= Nk, softmax with wave reductions, and
//            leaves P as bf16 (hi / lo planes in bf16x3 mode) in LDS in MFMA A-operand order.
//   phase 3  the dv/16 output column fragments are split into NW contiguous ranges, one per wave: O = P V^T-rows with V^T
//            fragments loaded straight from global memory; groups of 2-4 fragments are transposed through LDS so that the
//            operand leaves as 16-byte stores.
#include "common.h"
#include <atomic>

namespace {

__device__ __forceinline__ bf16x8 ldg8(const frido_bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

template <int NS>
__device__ __forceinline__ f32x4 mma(const bf16x8 (&a)[2], const bf16x8 (&b)[2], f32x4 acc) {
    if (NS == 2) {
        acc = mfma_op<NS>(a[1], b[0], acc);
        acc = mfma_op<NS>(a[0], b[1], acc);
    }
    return mfma_op<NS>(a[0], b[0], acc);
}

// float4 slots per wave of the LayerNorm form's lane-private parking lot: the wave's share of the dv / 16 output fragments (at most
// ceil(ntall / NW)), G fragments = G float4 per lane per pass of its loop.  One slot = 64 lanes x 16 bytes.
__host__ __device__ constexpr int attn_park_slots(int nw, int g, int ntall) { return ((ntall + nw - 1) / nw + g - 1) / g * g; }
constexpr int attn_lanes_g(int nw, int nf) { return (nw == 16 || (nw == 4 && nf == 2)) ? 2 : 4; }

// NW waves per workgroup (4 when the grid alone fills the chip, 8 / 16 on the small planes where only more waves per
// workgroup shorten the serial load -> MFMA chain), NF = compile-time bound on the 16-key score fragments (Nk <= 16 NF).
template <int NS, int NW, int NF>
__global__ __launch_bounds__(NW * 64) void attn_small_kernel(const FridoAttnSmall d) {
    // output fragments per LDS transpose group.  r04: 2 as well for the four-wave short-key form (<= 32 keys, the cross-attention of the
    // 32 x 32 plane): its static LDS drops from 20 to 12 KB, so with the 25-KB row buffer of the fused LayerNorm FOUR workgroups fit a
    // CU instead of three and the launch's 1024 workgroups are all resident at once
    constexpr int G = attn_lanes_g(NW, NF);
    // (r05) transpose slabs.  G = 4: rows of 68 floats, the 16-byte read-back (lane -> row lane/4, 16 columns) is conflict-free as it is.
    // G = 2 (32 columns per row, 8 per lane): with 36-float rows the ds_read_b128 lane groups {rows 0,3,5,6} / {1,2,4,7} met on the
    // same banks two ways (PMC r04: 25 % of the kernel's LDS cycles were conflict cycles); rows of 48 floats with the 4-float chunk
    // index XORed by (row & 7) give every lane group 16 distinct 4-bank spans, and the b32 slab stores stay conflict-free
    constexpr int SP_LD = NF * 16 + 4, SO_LD = G == 2 ? 48 : G * 16 + 4, P_LD = NF * 16 + 8;   // P rows: 16-byte aligned fragment reads
    constexpr bool SO_SWZ = G == 2;
    constexpr int SLD = SP_LD > SO_LD ? SP_LD : SO_LD;
    __shared__ __attribute__((aligned(16))) float s_part[NW * 16 * SLD];          // phase 1/2 partial scores; phase 3 slabs
    __shared__ __attribute__((aligned(16))) frido_bf16 s_p[NS * 16 * P_LD];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * 16;
    const int z = row0 / d.Nq;
    const int nf = (d.Nk + 15) >> 4;

    // ---- phase 1: partial scores over this wave's k-steps (two k-steps of loads in flight per iteration) ----
    {
        f32x4 s[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const frido_bf16* qrow = d.Q + (int64_t)(row0 + r) * d.ldq + g * 8;
        const frido_bf16* kbase = d.K + (int64_t)z * d.k_bs + g * 8;
        int64_t koff[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            int key = j * 16 + r;
            key = key < d.Nk ? key : d.Nk - 1;
            koff[j] = (int64_t)key * d.ldk;
        }
        const int nks = d.d >> 5;
        for (int ks = 2 * wave; ks < nks; ks += 2 * NW) {      // adjacent k-step pairs: whole 128-byte lines per row
            const bool two = ks + 1 < nks;
            bf16x8 a[2][2], b[2][NF][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
                const int kk = (ks + u) * 32;
                a[u][0] = ldg8(qrow + kk);
                if (NS == 2) a[u][1] = ldg8(qrow + d.q_lo + kk);
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    if (j < nf) {
                        b[u][j][0] = ldg8(kbase + koff[j] + kk);
                        if (NS == 2) b[u][j][1] = ldg8(kbase + d.k_lo + koff[j] + kk);
                    }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    if (j < nf) s[j] = mma<NS>(a[u], b[u][j], s[j]);
            }
        }
        float* sp = s_part + wave * 16 * SP_LD;
#pragma unroll
        for (int j = 0; j < NF; ++j)
            if (j < nf) {
#pragma unroll
                for (int e = 0; e < 4; ++e) sp[(g * 4 + e) * SP_LD + j * 16 + r] = s[j][e];
            }
    }
    __syncthreads();

    // ---- phase 2: softmax of this wave's rows (partials summed in a fixed order: results are run-to-run identical) ----
    const int npad = d.ldvt;
    constexpr int RPW = 16 / NW > 0 ? 16 / NW : 1;
    constexpr int NH = NF > 4 ? 2 : 1;                 // 64-column halves per row
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int row = wave * RPW + rr;
        if (row >= 16) break;
        float v[NH], mx = -3.0e38f;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int c = lane + h * 64;
            v[h] = -3.0e38f;
            if (c < d.Nk) {
                const float* sp = s_part + row * SP_LD + c;
                float acc = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) acc += sp[w * 16 * SP_LD];
                v[h] = d.alpha * acc;
            }
            mx = fmaxf(mx, v[h]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int c = lane + h * 64;
            v[h] = c < d.Nk ? __expf(v[h] - mx) : 0.f;
            sum += v[h];
        }
        const float inv = 1.0f / wave_sum(sum);
        status_raise(false, lane == 0 && stat_bad(mx, inv));
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int c = lane + h * 64;
            if (c < npad) {
                uint32_t hi, lo;
                split_op(v[h] * inv, NS, hi, lo);
                s_p[row * P_LD + c] = (frido_bf16)hi;
                if (NS == 2) s_p[16 * P_LD + row * P_LD + c] = (frido_bf16)lo;
            }
        }
    }
    __syncthreads();

    // ---- phase 3: O = P V over this wave's output column fragments ----
    constexpr int KP = NF / 2;                          // 32-key k-steps
    const int nkp = npad >> 5;
    bf16x8 pa[KP][2];
#pragma unroll
    for (int ks = 0; ks < KP; ++ks)
        if (ks < nkp) {
            pa[ks][0] = *reinterpret_cast<const bf16x8*>(s_p + r * P_LD + ks * 32 + g * 8);
            if (NS == 2) pa[ks][1] = *reinterpret_cast<const bf16x8*>(s_p + 16 * P_LD + r * P_LD + ks * 32 + g * 8);
        }
    // the column fragments are first split over gridDim.y workgroups (small planes: S is recomputed per split, which
    // costs nothing next to the idle CUs), then over the waves
    const int ntall = d.dv >> 4;
    const int c0 = ntall * blockIdx.y / gridDim.y, nt = ntall * (blockIdx.y + 1) / gridDim.y - c0;
    const int t0 = c0 + nt * wave / NW, t1 = c0 + nt * (wave + 1) / NW;
    const frido_bf16* vbase = d.VT + (int64_t)z * d.vt_bs + g * 8;
    float* so = s_part + wave * 16 * SO_LD;         // this wave's transpose slab (the partial scores are dead by now)
    constexpr int CPL = G * 4;                      // columns per lane on the read-back: 16 rows x 4 lanes
    const int orow = lane >> 2, oc = (lane & 3) * CPL;
    bool sat = false;                              // an operand value beyond the fp16 planes' range (common.h status word)
    float ln_s = 0.f, ln_q = 0.f;                  // LayerNorm of the stream rows (ln_op): this lane's share of its row's sum / sum of squares
    // ... and the values themselves, parked in (dynamic) LDS: every lane reads back only what it wrote itself, so the second pass needs
    // no barrier -- and no round trip through the stores it has just issued (re-reading out_act cost ~15 us a launch).
    // (r05) the parking lot is LANE-PRIVATE: slot k of this wave holds the k-th float4 of each of its 64 lanes at [k][lane] -- a
    // ds_write_b128 / ds_read_b128 of consecutive lanes on consecutive 16-byte slots is conflict-free for every lane grouping of the
    // hardware.  The r03 form [row][dv + 4] had the read-back's lane groups (rows {0, 3, 5, 6}: row stride 388 = 4 mod 64 banks)
    // meet on the same banks two ways, in both LayerNorm passes (PMC: 23 % of the kernel's LDS cycles after the slab fix).
    extern __shared__ __attribute__((aligned(16))) float s_rows[];
    float* park = s_rows + ((size_t)wave * attn_park_slots(NW, G, d.dv >> 4) * 64 + lane) * 4;
    for (int tb = t0; tb < t1; tb += G) {
        const int ng = t1 - tb < G ? t1 - tb : G;
        f32x4 acc[G];
#pragma unroll
        for (int q = 0; q < G; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (r06) the stream form's residual / bias values of this lane's read-back slot go out WITH the V^T fragments: fetched where they are
        // consumed (after the MFMAs and the LDS transpose) they put a second global round trip on every pass of this serial loop
        float4 pre_r[CPL / 4], pre_b[CPL / 4];
        const bool pre = d.out_act && oc < ng * 16;
        if (pre) {
            const int colp = tb * 16 + oc;
            const int64_t rop = (int64_t)(row0 + orow) * d.ldr + colp;
#pragma unroll
            for (int i = 0; i < CPL / 4; ++i) {
                if (d.bias) pre_b[i] = *reinterpret_cast<const float4*>(d.bias + colp + i * 4);
                if (d.residual) pre_r[i] = load_act4(d.residual, rop + i * 4, d.act_bf16);
            }
        }
#pragma unroll
        for (int ks = 0; ks < KP; ++ks)
            if (ks < nkp) {
                bf16x8 b[G][2];
#pragma unroll
                for (int q = 0; q < G; ++q)
                    if (q < ng) {
                        const frido_bf16* vr = vbase + (int64_t)((tb + q) * 16 + r) * d.ldvt + ks * 32;
                        b[q][0] = ldg8(vr);
                        if (NS == 2) b[q][1] = ldg8(vr + d.vt_lo);
                    }
#pragma unroll
                for (int q = 0; q < G; ++q)
                    if (q < ng) acc[q] = mma<NS>(pa[ks], b[q], acc[q]);
            }
#pragma unroll
        for (int q = 0; q < G; ++q)
            if (q < ng) {
#pragma unroll
                for (int e = 0; e < 4; ++e) so[(g * 4 + e) * SO_LD + ((q * 16 + r) ^ (SO_SWZ ? ((g * 4 + e) & 7) << 2 : 0))] = acc[q][e];
            }
        // transposed read-back: lane -> row lane/4, CPL consecutive columns, 16-byte accesses
        if (oc < ng * 16) {
            const float* srow = so + orow * SO_LD;
            const int sxor = SO_SWZ ? (orow & 7) << 2 : 0;      // 4-float chunk c of this row lives at chunk c ^ (row & 7)
            const int col = tb * 16 + oc;
            if (d.out_act) {                    // residual-stream form: O + bias + residual
                const int64_t oo = (int64_t)(row0 + orow) * d.ld_act + col;
                float keep[CPL];                // (r03) the same values once more as an operand, when out_op is given as well
#pragma unroll
                for (int i = 0; i < CPL / 4; ++i) {
                    float4 v = *reinterpret_cast<const float4*>(srow + ((oc + i * 4) ^ sxor));
                    if (d.bias) {
                        const float4 bb = pre_b[i];
                        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                    }
                    if (d.residual) {
                        const float4 rr = pre_r[i];
                        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                    }
                    if (d.skip_act_store) {
                        // (r05) nobody reads the f32 stream rows: the operand copy and / or the fused LayerNorm are the only consumers
                    } else if (d.act_bf16) {
                        *reinterpret_cast<uint2*>(reinterpret_cast<frido_bf16*>(d.out_act) + oo + i * 4) =
                            make_uint2(f32_to_bf16_bits(v.x) | (f32_to_bf16_bits(v.y) << 16), f32_to_bf16_bits(v.z) | (f32_to_bf16_bits(v.w) << 16));
                    } else {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.out_act) + oo + i * 4) = v;
                    }
                    keep[i * 4] = v.x; keep[i * 4 + 1] = v.y; keep[i * 4 + 2] = v.z; keep[i * 4 + 3] = v.w;
                    if (d.ln_op) {
                        ln_s += (v.x + v.y) + (v.z + v.w);
                        *reinterpret_cast<float4*>(park + (((tb - t0) / G) * (CPL / 4) + i) * 256) = v;
                    }
                }
                if (d.out_op) {                 // operand copy of the stream values (A2 of the chained FF2 + proj_out GEMM)
                    frido_bf16* dst = d.out_op + (int64_t)(row0 + orow) * d.ldo + col;
#pragma unroll
                    for (int i = 0; i < CPL / 8; ++i) {
                        uint32_t h[8], l[8];
#pragma unroll
                        for (int e = 0; e < 8; e += 2) { split_op2(keep[i * 8 + e], keep[i * 8 + e + 1], NS, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
                        if (NS == 2) sat |= op_sat8(keep + i * 8);
                        *reinterpret_cast<uint4*>(dst + i * 8) =
                            make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                        if (NS == 2)
                            *reinterpret_cast<uint4*>(dst + d.out_lo + i * 8) =
                                make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
                    }
                }
            } else {
                frido_bf16* dst = d.out_op + (int64_t)(row0 + orow) * d.ldo + col;
#pragma unroll
                for (int i = 0; i < CPL / 8; ++i) {
                    uint32_t h[8], l[8];
                    const float4 s0 = *reinterpret_cast<const float4*>(srow + ((oc + i * 8) ^ sxor)), s1 = *reinterpret_cast<const float4*>(srow + ((oc + i * 8 + 4) ^ sxor));
                    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                    for (int e = 0; e < 8; e += 2) { split_op2(sv[e], sv[e + 1], NS, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
                    if (NS == 2) sat |= op_sat8(sv);
                    *reinterpret_cast<uint4*>(dst + i * 8) =
                        make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                    if (NS == 2)
                        *reinterpret_cast<uint4*>(dst + d.out_lo + i * 8) =
                            make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
                }
            }
        }
    }
    // ---- (r03) LayerNorm of the 16 stream rows this workgroup just wrote, as an operand: norm2 / norm3 of the transformer block
    //      (attention.py:225-226) without their own launch.  gridDim.y == 1 (validated): the workgroup owns the whole rows.
    if (d.ln_op) {
        // two-pass statistics like layernorm_kernel (mean first, then the centred sum of squares: a row with |mean| >> sigma would
        // lose its variance in E[x^2] - mean^2); the parked rows make the second pass an LDS read of this lane's own values
        ln_s += __shfl_xor(ln_s, 1, 64); ln_s += __shfl_xor(ln_s, 2, 64);      // the four lanes of a row
        __syncthreads();                            // every wave is done with its transpose slab: s_part is free
        if ((lane & 3) == 0) s_part[(wave * 16 + orow) * 2] = ln_s;
        __syncthreads();
        float S = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) S += s_part[(w * 16 + orow) * 2];          // fixed order
        const float mean = S / d.dv;
        ln_q = 0.f;
        for (int tb = t0; tb < t1; tb += G) {
            const int ng = t1 - tb < G ? t1 - tb : G;
            if (oc >= ng * 16) continue;
            const float* xq = park + ((tb - t0) / G) * (CPL / 4) * 256;
#pragma unroll
            for (int i = 0; i < CPL / 4; ++i) {
                const float4 x = *reinterpret_cast<const float4*>(xq + i * 256);
                const float a0 = x.x - mean, a1 = x.y - mean, a2 = x.z - mean, a3 = x.w - mean;
                ln_q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        ln_q += __shfl_xor(ln_q, 1, 64); ln_q += __shfl_xor(ln_q, 2, 64);
        if ((lane & 3) == 0) s_part[(wave * 16 + orow) * 2 + 1] = ln_q;
        __syncthreads();
        float Q = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) Q += s_part[(w * 16 + orow) * 2 + 1];
        const float rstd = 1.0f / sqrtf(Q / d.dv + d.ln_eps);
        status_raise(false, wave == 0 && (lane & 3) == 0 && stat_bad(mean, rstd));
        // second pass over the values this lane parked in LDS
        frido_bf16* dst = d.ln_op + (int64_t)(row0 + orow) * d.ld_ln;
        // (r06) the affine parameters of pass tb + G are fetched before pass tb's stores go out: the compiler cannot move a load above the
        // stores of the previous pass, so every pass of this loop waited out its own L2 round trip
        float4 nw[CPL / 8][2], nb[CPL / 8][2];
        auto fetch_affine = [&](int tb) {
            const int ng = t1 - tb < G ? t1 - tb : G;
            if (tb >= t1 || oc >= ng * 16) return;
#pragma unroll
            for (int i = 0; i < CPL / 8; ++i) {
                const int c = tb * 16 + oc + i * 8;
                nw[i][0] = *reinterpret_cast<const float4*>(d.ln_w + c); nw[i][1] = *reinterpret_cast<const float4*>(d.ln_w + c + 4);
                nb[i][0] = *reinterpret_cast<const float4*>(d.ln_b + c); nb[i][1] = *reinterpret_cast<const float4*>(d.ln_b + c + 4);
            }
        };
        fetch_affine(t0);
        for (int tb = t0; tb < t1; tb += G) {
            const int ng = t1 - tb < G ? t1 - tb : G;
            float4 cw[CPL / 8][2], cb[CPL / 8][2];
#pragma unroll
            for (int i = 0; i < CPL / 8; ++i) { cw[i][0] = nw[i][0]; cw[i][1] = nw[i][1]; cb[i][0] = nb[i][0]; cb[i][1] = nb[i][1]; }
            fetch_affine(tb + G);
            if (oc >= ng * 16) continue;
            const int col = tb * 16 + oc;
#pragma unroll
            for (int i = 0; i < CPL / 8; ++i) {
                const int c = col + i * 8;
                const float* xr = park + (((tb - t0) / G) * (CPL / 4) + i * 2) * 256;
                const float4 x0 = *reinterpret_cast<const float4*>(xr), x1 = *reinterpret_cast<const float4*>(xr + 256);
                const float4 w0 = cw[i][0], w1 = cw[i][1];
                const float4 b0 = cb[i][0], b1 = cb[i][1];
                const float y[8] = {(x0.x - mean) * rstd * w0.x + b0.x, (x0.y - mean) * rstd * w0.y + b0.y, (x0.z - mean) * rstd * w0.z + b0.z,
                                    (x0.w - mean) * rstd * w0.w + b0.w, (x1.x - mean) * rstd * w1.x + b1.x, (x1.y - mean) * rstd * w1.y + b1.y,
                                    (x1.z - mean) * rstd * w1.z + b1.z, (x1.w - mean) * rstd * w1.w + b1.w};
                uint32_t h[8], l[8];
#pragma unroll
                for (int e = 0; e < 8; e += 2) { split_op2(y[e], y[e + 1], NS, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
                sat |= op_sat8(y);
                *reinterpret_cast<uint4*>(dst + c) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                *reinterpret_cast<uint4*>(dst + d.ln_lo + c) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
            }
        }
    }
    status_raise(sat);
}

// static + dynamic LDS of the LayerNorm form can pass 64 KiB (eight score fragments, d >= 384): per-device opt-in, set once
template <int NS, int NW, int NF>
bool attn_lds_optin() {
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_small_kernel<NS, NW, NF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                16 * (1024 + 4) * (int)sizeof(float)) != hipSuccess)
            return false;
        done.fetch_or(bit, std::memory_order_release);
    }
    return true;
}

template <int NS>
int launch_attn(const FridoAttnSmall& d, hipStream_t s) {
    const int blocks = d.B * (d.Nq / 16);
    const bool many = blocks >= 512;                    // >= 2 workgroups per CU: the grid hides the latency chain
    const int cs = blocks >= 256 ? 1 : (blocks >= 128 ? 2 : 4);
    // the values parked for their LayerNorm (lane-private slots of 1 KiB per wave: <= 64 KiB at dv = 1024 in every form)
#define ATTN_LAUNCH(NW, NF)                                                                                               \
    do {                                                                                                                  \
        const size_t dyn = d.ln_op ? (size_t)NW * attn_park_slots(NW, attn_lanes_g(NW, NF), d.dv >> 4) * 1024 : 0;        \
        if (dyn && !attn_lds_optin<NS, NW, NF>()) {                                                                       \
            frido_set_error("attn_small: cannot opt in to %zu bytes of dynamic LDS", dyn);                                \
            return FRIDO_EHIP;                  /* no launch with an under-provisioned row buffer */                      \
        }                                                                                                                 \
        hipLaunchKernelGGL((attn_small_kernel<NS, NW, NF>), dim3(blocks, cs), dim3(NW * 64), dyn, s, d);                  \
    } while (0)
    if (d.Nk <= 32) {
        if (many) ATTN_LAUNCH(4, 2); else ATTN_LAUNCH(16, 2);
    } else if (d.Nk <= 64) {
        if (many) ATTN_LAUNCH(4, 4); else ATTN_LAUNCH(8, 4);
    } else {
        ATTN_LAUNCH(4, 8);
    }
#undef ATTN_LAUNCH
    return FRIDO_OK;
}

}  // namespace

extern "C" int frido_attn_small(const FridoAttnSmall* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->Q && d->K && d->VT && (d->out_op || d->out_act), "null pointer");
    FRIDO_REQUIRE(!d->out_act || ((d->ld_act & 3) == 0 && (d->ldr & 3) == 0), "stream strides must be multiples of 4");
    FRIDO_REQUIRE(d->B > 0 && d->Nq > 0 && (d->Nq & 15) == 0, "Nq must be a positive multiple of 16");
    FRIDO_REQUIRE(d->Nk > 0 && d->Nk <= 128 && d->ldvt >= d->Nk && d->ldvt <= 128 && (d->ldvt & 31) == 0, "Nk must be in [1, 128], ldvt a multiple of 32");
    FRIDO_REQUIRE(d->d > 0 && (d->d & 31) == 0 && d->dv > 0 && (d->dv & 15) == 0, "d must be a multiple of 32, dv of 16");
    FRIDO_REQUIRE((d->ldq & 7) == 0 && (d->ldk & 7) == 0 && (d->ldo & 7) == 0 && (d->q_lo & 7) == 0 && (d->k_lo & 7) == 0 &&
                      (d->vt_lo & 7) == 0 && (d->out_lo & 7) == 0 && (d->k_bs & 7) == 0 && (d->vt_bs & 7) == 0,
                  "strides and plane offsets must keep 16-byte alignment");
    FRIDO_REQUIRE(d->nsplit == 1 || d->nsplit == 2, "nsplit must be 1 or 2");
    FRIDO_REQUIRE(!d->skip_act_store || (d->out_act && (d->out_op || d->ln_op)), "skip_act_store: stream form with an operand copy and / or the fused LayerNorm as its consumers");
    FRIDO_REQUIRE(!d->ln_op || (d->out_act && !d->act_bf16 && d->nsplit == 2 && d->B * (d->Nq / 16) >= 256 && d->dv <= 1024 && d->ln_w && d->ln_b &&
                                (d->ld_ln & 7) == 0 && (d->ln_lo & 7) == 0),
                  "ln_op: bf16x3 f32-stream output, B * Nq / 16 >= 256 (one workgroup per 16 rows owns them whole), weight and bias given");
    const int rc = d->nsplit == 2 ? launch_attn<2>(*d, (hipStream_t)s) : launch_attn<1>(*d, (hipStream_t)s);
    if (rc != FRIDO_OK) return rc;
    return frido_check_launch("attn_small");
}
