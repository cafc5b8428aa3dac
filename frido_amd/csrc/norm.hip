// HBM-bound normalisation / activation kernels: GroupNorm statistics + apply (with SPADE and SiLU),
// LayerNorm, row softmax, GEGLU.  All read NHWC f32 with 16-byte coalesced accesses and write bf16
// "operand" tensors (hi plane, plus the residual plane in bf16x3 mode) for the MFMA kernels.
// Reductions use wave64 shuffles; cross-wave / cross-workgroup combination is in a fixed order
// (no float atomics) so results are bit-reproducible run to run.
#include "common.h"
#include <cstdlib>

namespace {

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics: grid (nsplit_px, B).  Thread t owns float4 column c4 = t % TC of pixel lane
// t / TC and accumulates {sum, sumsq} over its pixels in registers; LDS holds [PL][C] partials which
// `groups` threads reduce in double.
constexpr int GN_MAXC = 4096;

__device__ __forceinline__ float4 load_cat4(const float* x1, int C1, const float* x2, int C2, int64_t pix, int c, int bf) {
    if (c < C1) return load_act4(x1, pix * C1 + c, bf);
    return load_act4(x2, pix * C2 + (c - C1), bf);
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const FridoGnStats d) {
    __shared__ float s_sum[GN_MAXC];
    __shared__ float s_sq[GN_MAXC];
    const int C = d.C1 + d.C2, C4 = C >> 2;
    const int TC = C4 < 256 ? C4 : 256;
    const int PL = 256 / TC;
    const int t = threadIdx.x, b = blockIdx.y, sp = blockIdx.x;
    const int ppx = (d.HW + d.nsplit_px - 1) / d.nsplit_px;
    const int p0 = sp * ppx, p1 = min(d.HW, p0 + ppx);
    const int pl = t / TC, tc = t - pl * TC;
    const int cpg = C / d.groups;
    double gsum = 0.0, gsq = 0.0;
    if (d.x_bf16 && ((d.C1 | d.C2) & 7) == 0 && (C >> 3) <= 256) {
        // bf16 stream fast path: thread owns 8 channels (16-B loads), PL8 pixel lanes per workgroup
        const int C8 = C >> 3, PL8 = 256 / C8;
        const int pl8 = t / C8, tc8 = t - pl8 * C8;
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (pl8 < PL8) {
            const int c = tc8 * 8;
            const frido_bf16* base = reinterpret_cast<const frido_bf16*>(c < d.C1 ? d.x1 : d.x2);
            const int cw = c < d.C1 ? d.C1 : d.C2, co = c < d.C1 ? c : c - d.C1;
            for (int p = p0 + pl8; p < p1; p += PL8) {
                const u32x4 xv = *reinterpret_cast<const u32x4*>(base + ((int64_t)b * d.HW + p) * cw + co);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(xv[e] << 16), hi = __uint_as_float(xv[e] & 0xffff0000u);
                    s8[2 * e] += lo; q8[2 * e] += lo * lo;
                    s8[2 * e + 1] += hi; q8[2 * e + 1] += hi * hi;
                }
            }
        }
        for (int l = 0; l < PL8; ++l) {       // fixed-order combine of the pixel lanes
            if (pl8 == l) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (l == 0) { s_sum[tc8 * 8 + e] = s8[e]; s_sq[tc8 * 8 + e] = q8[e]; }
                    else { s_sum[tc8 * 8 + e] += s8[e]; s_sq[tc8 * 8 + e] += q8[e]; }
                }
            }
            __syncthreads();
        }
    } else
    // one pass per 256-column chunk (a single pass whenever C <= 1024)
    for (int cbase = 0; cbase < C4; cbase += TC) {
        const int c4 = cbase + tc;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
        if (pl < PL && c4 < C4) {
            for (int p = p0 + pl; p < p1; p += PL) {
                const float4 v = load_cat4(d.x1, d.C1, d.x2, d.C2, (int64_t)b * d.HW + p, c4 * 4, d.x_bf16);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
            }
        }
        // combine the PL pixel lanes of each column in a fixed order through LDS
        for (int l = 0; l < PL; ++l) {
            if (pl == l && c4 < C4) {
                float* ps = s_sum + c4 * 4; float* pq = s_sq + c4 * 4;
                if (l == 0) { ps[0] = s.x; ps[1] = s.y; ps[2] = s.z; ps[3] = s.w; pq[0] = q.x; pq[1] = q.y; pq[2] = q.z; pq[3] = q.w; }
                else { ps[0] += s.x; ps[1] += s.y; ps[2] += s.z; ps[3] += s.w; pq[0] += q.x; pq[1] += q.y; pq[2] += q.z; pq[3] += q.w; }
            }
            __syncthreads();
        }
    }
    if (t < d.groups) {
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) { gsum += (double)s_sum[c]; gsq += (double)s_sq[c]; }
        double* out = d.partials + (((int64_t)b * d.nsplit_px + sp) * d.groups + t) * 2;
        out[0] = gsum;
        out[1] = gsq;
        status_raise(false, !(fabs(gsum) <= 1.0e300) || !(gsq <= 1.0e300));
    }
}

// GroupNorm statistics from the PRODUCERS' per-channel partial sums (FridoGemm.gn_part: {sum, sumsq} per 32-row block and
// channel): one workgroup per (sample, group) walks its (block, channel) items in a fixed order, doubles from the first add on.
__global__ __launch_bounds__(256) void gn_stats_parts_kernel(const FridoGnStats d) {
    __shared__ double s_red[4][2];
    const int C = d.C1 + d.C2, cpg = C / d.groups;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.x / d.groups, g = blockIdx.x - b * d.groups;      // one workgroup per (sample, group)
    const int nblk = d.HW >> 5, items = nblk * cpg;
    double s = 0.0, q = 0.0;
    // item idx = (block k, channel of the group): up to 4 independent loads in flight per thread, summed in index order
    for (int i0 = t; i0 < items; i0 += 1024) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = i0 + 256 * u;
            v[u] = make_float2(0.f, 0.f);
            if (idx < items) {
                const int k = idx / cpg, c = g * cpg + (idx - k * cpg);
                const int64_t blk = (int64_t)b * nblk + k;
                v[u] = c < d.C1 ? *reinterpret_cast<const float2*>(d.p1 + (blk * d.C1 + c) * 2)
                                : *reinterpret_cast<const float2*>(d.p2 + (blk * d.C2 + (c - d.C1)) * 2);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += (double)v[u].x; q += (double)v[u].y; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if (lane == 0) { s_red[wave][0] = s; s_red[wave][1] = q; }
    __syncthreads();
    if (t == 0) {
        double* out = d.partials + ((int64_t)b * d.groups + g) * 2;
        out[0] = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
        out[1] = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
        status_raise(false, !(fabs(out[0]) <= 1.0e300) || !(out[1] <= 1.0e300));
    }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm apply: grid (blocks_per_image, B); prologue turns the partials into {mean, rstd}.
__global__ __launch_bounds__(256) void gn_apply_kernel(const FridoGnApply d) {
    __shared__ float s_mean[64], s_rstd[64];
    __shared__ double s_part[8][32][2];
    __shared__ __attribute__((aligned(16))) float s_sc[GN_MAXC], s_sh[GN_MAXC];
    const int C = d.C1 + d.C2, C4 = C >> 2;
    const int t = threadIdx.x, b = blockIdx.y;
    const int cpg = C / d.groups;
    {   // combine the per-split partials: 8 lanes per group in parallel, then a fixed-order sum of the 8
        const int g = t & 31, l8 = t >> 5;
        double s = 0.0, q = 0.0;
        if (g < d.groups) {
            double2 pv[8];        // nsplit_px <= 64: issue all (independent) loads first, then add in a fixed order
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int sp = l8 + 8 * k;
                pv[k] = sp < d.nsplit_px
                            ? *reinterpret_cast<const double2*>(d.partials + (((int64_t)b * d.nsplit_px + sp) * d.groups + g) * 2)
                            : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { s += pv[k].x; q += pv[k].y; }
            s_part[l8][g][0] = s;
            s_part[l8][g][1] = q;
        }
        __syncthreads();
        if (t < d.groups) {
            s = 0.0; q = 0.0;
            for (int l = 0; l < 8; ++l) { s += s_part[l][t][0]; q += s_part[l][t][1]; }
            const double n = (double)d.HW * cpg;
            const double mean = s / n;
            double var = q / n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mean[t] = (float)mean;
            s_rstd[t] = (float)(1.0 / sqrt(var + (double)d.eps));
            if (blockIdx.x == 0) status_raise(false, stat_bad(s_mean[t], s_rstd[t]));
        }
    }
    __syncthreads();
    // per-channel scale / shift table in LDS: y = x * sc[c] + sh[c]
    for (int c = t; c < C; c += 256) {
        const int g = c / cpg;
        const float sc = s_rstd[g] * d.weight[c];
        s_sc[c] = sc;
        s_sh[c] = d.bias[c] - s_mean[g] * sc;
    }
    __syncthreads();
    // bf16 stream fast path: 8 channels (16 B) per lane per tensor
    if (d.x_bf16 && (!d.gamma || d.gb_bf16) && d.nsplit == 1 && !d.out_f32 && ((d.C1 | d.C2) & 7) == 0) {
        const unsigned C8 = (unsigned)C >> 3;
        const unsigned total8 = (unsigned)d.HW * C8;
        const frido_bf16* xb1 = reinterpret_cast<const frido_bf16*>(d.x1);
        const frido_bf16* xb2 = reinterpret_cast<const frido_bf16*>(d.x2);
        const frido_bf16* gb = reinterpret_cast<const frido_bf16*>(d.gamma);
        const frido_bf16* bb = reinterpret_cast<const frido_bf16*>(d.beta);
        // two independent vectors per iteration (the x, gamma, beta loads of both are issued before either is consumed)
        auto one = [&](unsigned i, u32x4& xv, u32x4& gv, u32x4& bv, int64_t& o, int& c) {
            const unsigned p = i / C8;
            c = (int)(i - p * C8) * 8;
            const int64_t pix = (int64_t)b * d.HW + p;
            o = pix * C + c;
            xv = c < d.C1 ? *reinterpret_cast<const u32x4*>(xb1 + pix * d.C1 + c)
                          : *reinterpret_cast<const u32x4*>(xb2 + pix * d.C2 + (c - d.C1));
            if (d.gamma) {
                gv = *reinterpret_cast<const u32x4*>(gb + o);
                bv = *reinterpret_cast<const u32x4*>(bb + o);
            }
        };
        auto fin = [&](const u32x4& xv, const u32x4& gv, const u32x4& bv, int64_t o, int c) {
            float y[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[2 * e] = fmaf(__uint_as_float(xv[e] << 16), s_sc[c + 2 * e], s_sh[c + 2 * e]);
                y[2 * e + 1] = fmaf(__uint_as_float(xv[e] & 0xffff0000u), s_sc[c + 2 * e + 1], s_sh[c + 2 * e + 1]);
            }
            if (d.gamma) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[2 * e] = fmaf(y[2 * e], 1.f + __uint_as_float(gv[e] << 16), __uint_as_float(bv[e] << 16));
                    y[2 * e + 1] = fmaf(y[2 * e + 1], 1.f + __uint_as_float(gv[e] & 0xffff0000u), __uint_as_float(bv[e] & 0xffff0000u));
                }
            }
            if (d.act == FRIDO_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = silu_f(y[e]);
            }
            u32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = f32_to_bf16_bits(y[2 * e]) | (f32_to_bf16_bits(y[2 * e + 1]) << 16);
            *reinterpret_cast<u32x4*>(d.out_op + o) = ov;
            if (d.raw_op) *reinterpret_cast<u32x4*>(d.raw_op + o) = xv;       // concatenated raw operand for the 1x1 skip conv
        };
        const unsigned stride = gridDim.x * 256u;
        unsigned i = blockIdx.x * 256u + t;
        constexpr int U = 2;
        for (; i + (U - 1) * stride < total8; i += U * stride) {
            u32x4 xs[U], gs[U], bs[U];
            int64_t os[U];
            int cs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) one(i + u * stride, xs[u], gs[u], bs[u], os[u], cs[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) fin(xs[u], gs[u], bs[u], os[u], cs[u]);
        }
        for (; i < total8; i += stride) {
            u32x4 x0, g0, b0;
            int64_t o0;
            int c0;
            one(i, x0, g0, b0, o0, c0);
            fin(x0, g0, b0, o0, c0);
        }
        return;
    }
    // bf16x3 stream fast path (r03): f32 rows in, hi / lo operand planes out.  8 channels per lane -- two 16-byte loads per
    // tensor and ONE 16-byte store per plane (the generic loop below moves 4 channels: 8-byte stores) -- and two independent
    // vectors in flight per lane.  Same per-element arithmetic as the generic loop.
    if (!d.x_bf16 && d.nsplit == 2 && d.out_op && !d.out_f32 && (!d.gamma || !d.gb_bf16) && ((d.C1 | d.C2) & 7) == 0) {
        const unsigned C8 = (unsigned)C >> 3;
        const unsigned total8 = (unsigned)d.HW * C8;
        const float* xf1 = reinterpret_cast<const float*>(d.x1);
        const float* xf2 = reinterpret_cast<const float*>(d.x2);
        const float* gf = reinterpret_cast<const float*>(d.gamma);
        const float* bf = reinterpret_cast<const float*>(d.beta);
        struct Vec { float4 x0, x1, g0, g1, b0, b1; int64_t o; int c; };
        bool sat = false;
        auto one = [&](unsigned i, Vec& v) {
            const unsigned p = i / C8;
            v.c = (int)(i - p * C8) * 8;
            const int64_t pix = (int64_t)b * d.HW + p;
            v.o = pix * C + v.c;
            const float* src = v.c < d.C1 ? xf1 + pix * d.C1 + v.c : xf2 + pix * d.C2 + (v.c - d.C1);
            v.x0 = *reinterpret_cast<const float4*>(src);
            v.x1 = *reinterpret_cast<const float4*>(src + 4);
            if (d.gamma) {
                v.g0 = *reinterpret_cast<const float4*>(gf + v.o);
                v.g1 = *reinterpret_cast<const float4*>(gf + v.o + 4);
                v.b0 = *reinterpret_cast<const float4*>(bf + v.o);
                v.b1 = *reinterpret_cast<const float4*>(bf + v.o + 4);
            }
        };
        auto fin = [&](const Vec& v) {
            const float xin[8] = {v.x0.x, v.x0.y, v.x0.z, v.x0.w, v.x1.x, v.x1.y, v.x1.z, v.x1.w};
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = fmaf(xin[e], s_sc[v.c + e], s_sh[v.c + e]);
            if (d.gamma) {
                const float ga[8] = {v.g0.x, v.g0.y, v.g0.z, v.g0.w, v.g1.x, v.g1.y, v.g1.z, v.g1.w};
                const float be[8] = {v.b0.x, v.b0.y, v.b0.z, v.b0.w, v.b1.x, v.b1.y, v.b1.z, v.b1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = fmaf(y[e], 1.f + ga[e], be[e]);
            }
            if (d.act == FRIDO_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = silu_f(y[e]);
            }
            uint32_t h[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; e += 2) { split_op2(y[e], y[e + 1], 2, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
            sat |= op_sat8(y);
            *reinterpret_cast<uint4*>(d.out_op + v.o) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            *reinterpret_cast<uint4*>(d.out_op + d.out_lo + v.o) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
            if (d.raw_op) {                 // concatenated raw operand for the 1x1 skip conv
                sat |= op_sat8(xin);
#pragma unroll
                for (int e = 0; e < 8; e += 2) { split_op2(xin[e], xin[e + 1], 2, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
                *reinterpret_cast<uint4*>(d.raw_op + v.o) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
                *reinterpret_cast<uint4*>(d.raw_op + d.raw_lo + v.o) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
            }
        };
        const unsigned stride = gridDim.x * 256u;
        unsigned i = blockIdx.x * 256u + t;
        for (; i + stride < total8; i += 2 * stride) {
            Vec v0, v1;
            one(i, v0);
            one(i + stride, v1);
            fin(v0);
            fin(v1);
        }
        if (i < total8) {
            Vec v0;
            one(i, v0);
            fin(v0);
        }
        status_raise(sat);
        return;
    }
    bool sat = false;
    const unsigned total = (unsigned)d.HW * (unsigned)C4;      // < 2^31 on every shape of this path
    for (unsigned i = blockIdx.x * 256u + t; i < total; i += gridDim.x * 256u) {
        const unsigned p = i / (unsigned)C4;
        const int c = (int)(i - p * (unsigned)C4) * 4;
        const int64_t pix = (int64_t)b * d.HW + p;
        const float4 v = load_cat4(d.x1, d.C1, d.x2, d.C2, pix, c, d.x_bf16);
        const float4 sc = *reinterpret_cast<const float4*>(s_sc + c);
        const float4 sh = *reinterpret_cast<const float4*>(s_sh + c);
        const float xin[4] = {v.x, v.y, v.z, v.w};
        float y[4] = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
        const int64_t o = pix * C + c;
        if (d.gamma) {
            const float4 ga = load_act4(d.gamma, o, d.gb_bf16);
            const float4 be = load_act4(d.beta, o, d.gb_bf16);
            y[0] = fmaf(y[0], 1.f + ga.x, be.x);
            y[1] = fmaf(y[1], 1.f + ga.y, be.y);
            y[2] = fmaf(y[2], 1.f + ga.z, be.z);
            y[3] = fmaf(y[3], 1.f + ga.w, be.w);
        }
        if (d.act == FRIDO_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = silu_f(y[e]);
        }
        if (d.out_op) store_op4(d.out_op, d.out_lo, d.nsplit, o, y);
        if (d.raw_op) store_op4(d.raw_op, d.raw_lo, d.nsplit, o, xin);
        if (d.nsplit == 2) sat |= (d.out_op && op_sat4(y)) || (d.raw_op && op_sat4(xin));
        if (d.out_f32) *reinterpret_cast<float4*>(d.out_f32 + o) = make_float4(y[0], y[1], y[2], y[3]);
    }
    status_raise(sat);
}

// ---------------------------------------------------------------------------------------------
// One-launch GroupNorm (bf16 stream): workgroup = one sample x one CHUNK of whole groups (Cc channels, a multiple of 8).
// The slice [HW][Cc] is read ONCE into registers (16-byte vectors; lane -> fixed 8 channels, so the channel -> group map is
// resolved once), reduced (registers -> wave shuffles -> LDS -> double, fixed order), then normalised / SPADE-modulated /
// SiLU'd from the registers and written as the operand.  Replaces gn_stats + gn_apply (two launches, a partials round trip
// and a per-workgroup table prologue) on the small planes (16 x 16 and below in the layout2i U-Net), where those launches are
// pure latency; on the large planes a chunk is only 24-48 bytes of every pixel row and the two coalesced kernels win
// (measured: 88 vs 38 us at 64 x 64 x 192, 18 vs 21 us at 16 x 16 x 576, 12 vs 16 us at 8 x 8 x 960).
constexpr int GNF_MAXV = 10;

template <int NT>
__global__ __launch_bounds__(NT) void gn_fused_kernel(const FridoGnApply d, int Cc) {
    __shared__ double s_red[NT / 64][8];
    __shared__ float s_mean[4], s_rstd[4];
    const int C = d.C1 + d.C2, cpg = C / d.groups;
    const int t = threadIdx.x, b = blockIdx.y, lane = t & 63, wave = t >> 6;
    const int c0 = blockIdx.x * Cc;                       // first channel of this chunk
    const int vpp = Cc >> 3;                              // 16-byte vectors per pixel
    const int ppi = NT / vpp;                             // pixels per sweep
    const int cv = t % vpp, pl = t / vpp;
    const bool live = pl < ppi;
    const int c = c0 + cv * 8;                            // this lane's 8 channels
    const frido_bf16* src;
    int ldx;
    if (c < d.C1) { src = reinterpret_cast<const frido_bf16*>(d.x1) + c; ldx = d.C1; }
    else { src = reinterpret_cast<const frido_bf16*>(d.x2) + (c - d.C1); ldx = d.C2; }
    src += (int64_t)b * d.HW * ldx;
    int gi[8];                                            // local group of each of the 8 channels
#pragma unroll
    for (int e = 0; e < 8; ++e) gi[e] = (cv * 8 + e) / cpg;

    u32x4 xv[GNF_MAXV];
    float cs[8], cq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = cq[e] = 0.f;
#pragma unroll
    for (int k = 0; k < GNF_MAXV; ++k) {
        const int p = pl + k * ppi;
        xv[k] = u32x4{0u, 0u, 0u, 0u};
        if (live && p < d.HW) xv[k] = *reinterpret_cast<const u32x4*>(src + (int64_t)p * ldx);
    }
#pragma unroll
    for (int k = 0; k < GNF_MAXV; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = __uint_as_float(xv[k][e] << 16), bb = __uint_as_float(xv[k][e] & 0xffff0000u);
            cs[2 * e] += a; cq[2 * e] = fmaf(a, a, cq[2 * e]);
            cs[2 * e + 1] += bb; cq[2 * e + 1] = fmaf(bb, bb, cq[2 * e + 1]);
        }
    }
    // SPADE gamma / beta do not depend on the statistics: fetch them now, under the reduction
    const frido_bf16* gb = reinterpret_cast<const frido_bf16*>(d.gamma);
    const frido_bf16* bb = reinterpret_cast<const frido_bf16*>(d.beta);
    u32x4 gv[GNF_MAXV], bv[GNF_MAXV];
    if (d.gamma) {
#pragma unroll
        for (int k = 0; k < GNF_MAXV; ++k) {
            const int p = pl + k * ppi;
            if (live && p < d.HW) {
                const int64_t o = ((int64_t)b * d.HW + p) * C + c;
                gv[k] = *reinterpret_cast<const u32x4*>(gb + o);
                bv[k] = *reinterpret_cast<const u32x4*>(bb + o);
            }
        }
    }
    // per-lane channel sums -> per-group sums (<= 4 groups per chunk) -> wave -> workgroup
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (gi[e] == g) { gs[g] += cs[e]; gq[g] += cq[e]; }
#pragma unroll
    for (int g = 0; g < 4; ++g) { gs[g] = wave_sum(gs[g]); gq[g] = wave_sum(gq[g]); }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { s_red[wave][g] = (double)gs[g]; s_red[wave][4 + g] = (double)gq[g]; }
    }
    __syncthreads();
    if (t < 4) {
        double sm = 0.0, sq = 0.0;
        for (int w = 0; w < NT / 64; ++w) { sm += s_red[w][t]; sq += s_red[w][4 + t]; }
        const double n = (double)d.HW * cpg;
        const double mean = sm / n;
        double var = sq / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        s_mean[t] = (float)mean;
        s_rstd[t] = (float)(1.0 / sqrt(var + (double)d.eps));
        if ((t + 1) * cpg <= Cc) status_raise(false, stat_bad(s_mean[t], s_rstd[t]));
    }
    __syncthreads();
    if (!live) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float r = s_rstd[gi[e]] * d.weight[c + e];
        sc[e] = r;
        sh[e] = d.bias[c + e] - s_mean[gi[e]] * r;
    }
#pragma unroll
    for (int k = 0; k < GNF_MAXV; ++k) {
        const int p = pl + k * ppi;
        if (p >= d.HW) break;
        const int64_t o = ((int64_t)b * d.HW + p) * C + c;
        float y[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[2 * e] = fmaf(__uint_as_float(xv[k][e] << 16), sc[2 * e], sh[2 * e]);
            y[2 * e + 1] = fmaf(__uint_as_float(xv[k][e] & 0xffff0000u), sc[2 * e + 1], sh[2 * e + 1]);
        }
        if (d.gamma) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[2 * e] = fmaf(y[2 * e], 1.f + __uint_as_float(gv[k][e] << 16), __uint_as_float(bv[k][e] << 16));
                y[2 * e + 1] = fmaf(y[2 * e + 1], 1.f + __uint_as_float(gv[k][e] & 0xffff0000u), __uint_as_float(bv[k][e] & 0xffff0000u));
            }
        }
        if (d.act == FRIDO_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = silu_f(y[e]);
        }
        u32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = f32_to_bf16_bits(y[2 * e]) | (f32_to_bf16_bits(y[2 * e + 1]) << 16);
        *reinterpret_cast<u32x4*>(d.out_op + o) = ov;
        if (d.raw_op) *reinterpret_cast<u32x4*>(d.raw_op + o) = xv[k];
    }
}

// One-launch GroupNorm for the F32 stream of the bf16x3 (parity) mode: the same decomposition (workgroup = one sample x one
// chunk of whole groups, the [HW][Cc] slice read ONCE into registers), f32 input / f32 SPADE maps, hi + lo operand planes
// out.  r03: in parity mode every GroupNorm was two launches (gn_stats + gn_apply) plus a partials round trip, 122 launches per
// denoiser forward; on the 16x16 and 8x8 planes those launches are pure latency (7-9 us each for a few hundred KB).
// f32 slices: x + gamma + beta of 4 vectors = 96 VGPRs (3 vectors in the 1024-thread form, whose waves get 128 registers); larger
// slices take more threads per workgroup instead of more registers per lane
// x1 from the raw split-K partial sums of its producer (FridoGnApply.sk_*): splitk_reduce8_kernel's arithmetic, expression for
// expression (slices added in ascending order from 0.f, then alpha * sum + (bias + rowvec), then the residual), so the value is
// bit-identical to what the reduce launch would have stored; INFL slices' loads in flight per lane (the 1024-thread form has 128 registers: 4)
template <int INFL, int VW>
__device__ __forceinline__ void sk_finish(const FridoGnApply& d, int64_t m, int c, int vstep, float (&v)[VW]) {
    constexpr int Q = VW / 4;      // float4 pieces per vector
    const int64_t plane = (int64_t)d.B * d.HW * d.C1;
    const float* w = d.sk_ws + m * d.C1 + c;
#pragma unroll
    for (int e = 0; e < VW; ++e) v[e] = 0.f;
    for (int z0 = 0; z0 < d.sk_n; z0 += INFL) {
        float4 a[INFL][Q];
#pragma unroll
        for (int u = 0; u < INFL; ++u)
            if (z0 + u < d.sk_n) {
#pragma unroll
                for (int q = 0; q < Q; ++q) a[u][q] = *reinterpret_cast<const float4*>(w + (z0 + u) * plane + 4 * q);
            }
#pragma unroll
        for (int u = 0; u < INFL; ++u)
            if (z0 + u < d.sk_n) {
#pragma unroll
                for (int q = 0; q < Q; ++q) { v[4 * q] += a[u][q].x; v[4 * q + 1] += a[u][q].y; v[4 * q + 2] += a[u][q].z; v[4 * q + 3] += a[u][q].w; }
            }
    }
    float add[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) add[e] = 0.f;
    if (d.sk_bias) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(d.sk_bias + c + 4 * q);
            add[4 * q] += a.x; add[4 * q + 1] += a.y; add[4 * q + 2] += a.z; add[4 * q + 3] += a.w;
        }
    }
    if (d.sk_rowvec) {
        const float* rp = d.sk_rowvec + (int64_t)((int)m / d.sk_rows_per_vec + vstep) * d.sk_ldv + c;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(rp + 4 * q);
            add[4 * q] += a.x; add[4 * q + 1] += a.y; add[4 * q + 2] += a.z; add[4 * q + 3] += a.w;
        }
    }
#pragma unroll
    for (int e = 0; e < VW; ++e) v[e] = v[e] * d.sk_alpha + add[e];
    if (d.sk_residual) {
        const float* rp = d.sk_residual + m * d.sk_ldr + c;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(rp + 4 * q);
            v[4 * q] += a.x; v[4 * q + 1] += a.y; v[4 * q + 2] += a.z; v[4 * q + 3] += a.w;
        }
    }
    if (d.sk_out) {
        float* o = d.sk_out + m * d.C1 + c;
#pragma unroll
        for (int q = 0; q < Q; ++q) *reinterpret_cast<float4*>(o + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}

// VW = channels per lane: 8 (two 16-byte loads, one 16-byte store per plane) or -- r05 -- 4: a chunk is then any run of whole groups
// that is a multiple of FOUR channels, i.e. TWO groups instead of four at C = 576 / 960, so the launches of the 16 x 16 and 8 x 8
// planes put 256 workgroups on the chip instead of 128 (a CU streams ~10 - 25 B / cycle: with half the CUs idle the slices' loads,
// not the arithmetic, set the launch time).
template <int VW>
__device__ __forceinline__ void store_planes(frido_bf16* op, int64_t lo_off, int nsplit, const float (&y)[VW]) {
    uint32_t h[VW], l[VW];
#pragma unroll
    for (int e = 0; e < VW; e += 2) { split_op2(y[e], y[e + 1], nsplit, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
    if constexpr (VW == 8) {
        *reinterpret_cast<u32x4*>(op) = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
        if (nsplit == 2)
            *reinterpret_cast<u32x4*>(op + lo_off) = u32x4{l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
    } else {
        *reinterpret_cast<uint2*>(op) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        if (nsplit == 2) *reinterpret_cast<uint2*>(op + lo_off) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    }
}
template <int VW>
__device__ __forceinline__ bool sat_vec(const float (&y)[VW]) {
    if constexpr (VW == 8) return op_sat8(y);
    else return op_sat4(y);
}

template <int NT, int GNF32_MAXV, int VW>
__global__ __launch_bounds__(NT) void gn_fused_f32_kernel(const FridoGnApply d, int Cc) {
    constexpr int Q = VW / 4;
    __shared__ double s_red[NT / 64][8];
    __shared__ float s_mean[4], s_rstd[4];
    const int C = d.C1 + d.C2, cpg = C / d.groups;
    const int t = threadIdx.x, b = blockIdx.y, lane = t & 63, wave = t >> 6;
    const int c0 = blockIdx.x * Cc;
    const int vpp = Cc / VW;                              // VW-channel vectors per pixel
    const int ppi = NT / vpp;                             // pixels per sweep
    const int cv = t % vpp, pl = t / vpp;
    const bool live = pl < ppi;
    const int c = c0 + cv * VW;
    const float* src;
    int ldx;
    if (c < d.C1) { src = d.x1 + c; ldx = d.C1; }
    else { src = d.x2 + (c - d.C1); ldx = d.C2; }
    src += (int64_t)b * d.HW * ldx;
    int gi[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) gi[e] = (cv * VW + e) / cpg;
    float xv[GNF32_MAXV][VW];
    float cs[VW], cq[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) cs[e] = cq[e] = 0.f;
    const bool from_sk = d.sk_ws != nullptr && c < d.C1;       // this lane's channels come from a split-K producer's partial sums
    int sk_vstep = 0;
    if (from_sk && d.sk_rowvec && d.sk_rowvec_step) sk_vstep = *d.sk_rowvec_step;
#pragma unroll
    for (int k = 0; k < GNF32_MAXV; ++k) {
        const int p = pl + k * ppi;
        if (from_sk) {
#pragma unroll
            for (int e = 0; e < VW; ++e) xv[k][e] = 0.f;
            if (live && p < d.HW) sk_finish<(NT == 1024 ? 4 : 8), VW>(d, (int64_t)b * d.HW + p, c, sk_vstep, xv[k]);
            continue;
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live && p < d.HW) a = *reinterpret_cast<const float4*>(src + (int64_t)p * ldx + 4 * q);
            xv[k][4 * q] = a.x; xv[k][4 * q + 1] = a.y; xv[k][4 * q + 2] = a.z; xv[k][4 * q + 3] = a.w;
        }
    }
#pragma unroll
    for (int k = 0; k < GNF32_MAXV; ++k)
#pragma unroll
        for (int e = 0; e < VW; ++e) { cs[e] += xv[k][e]; cq[e] = fmaf(xv[k][e], xv[k][e], cq[e]); }
    // SPADE gamma / beta do not depend on the statistics: fetch them now, under the reduction
    float4 gv[GNF32_MAXV][Q], bv[GNF32_MAXV][Q];
    if (d.gamma) {
#pragma unroll
        for (int k = 0; k < GNF32_MAXV; ++k) {
            const int p = pl + k * ppi;
            if (live && p < d.HW) {
                const int64_t o = ((int64_t)b * d.HW + p) * C + c;
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    gv[k][q] = *reinterpret_cast<const float4*>(d.gamma + o + 4 * q);
                    bv[k][q] = *reinterpret_cast<const float4*>(d.beta + o + 4 * q);
                }
            }
        }
    }
    // (r06) the affine parameters too: after the second barrier their loads sat on the launch's latency chain (a workgroup fence keeps the
    // compiler from hoisting them) -- one L2 round trip in every one of the 34 launches of a forward
    // (not in the 1024-thread x 8-channel form: 128 registers per lane, the 16 more would spill)
    constexpr bool EARLY_AFFINE = !(NT == 1024 && VW == 8);
    float wgt[VW], bia[VW];
    if constexpr (EARLY_AFFINE) {
#pragma unroll
        for (int e = 0; e < VW; ++e) { wgt[e] = live ? d.weight[c + e] : 0.f; bia[e] = live ? d.bias[c + e] : 0.f; }
    }
    float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < VW; ++e)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (gi[e] == g) { gs[g] += cs[e]; gq[g] += cq[e]; }
#pragma unroll
    for (int g = 0; g < 4; ++g) { gs[g] = wave_sum(gs[g]); gq[g] = wave_sum(gq[g]); }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { s_red[wave][g] = (double)gs[g]; s_red[wave][4 + g] = (double)gq[g]; }
    }
    __syncthreads();
    if (t < 4) {
        double sm = 0.0, sq = 0.0;
        for (int w = 0; w < NT / 64; ++w) { sm += s_red[w][t]; sq += s_red[w][4 + t]; }
        const double n = (double)d.HW * cpg;
        const double mean = sm / n;
        double var = sq / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        s_mean[t] = (float)mean;
        s_rstd[t] = (float)(1.0 / sqrt(var + (double)d.eps));
        if ((t + 1) * cpg <= Cc) status_raise(false, stat_bad(s_mean[t], s_rstd[t]));
    }
    __syncthreads();
    if (!live) return;
    float sc[VW], sh[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) {
        if constexpr (!EARLY_AFFINE) { wgt[e] = d.weight[c + e]; bia[e] = d.bias[c + e]; }
        const float r = s_rstd[gi[e]] * wgt[e];
        sc[e] = r;
        sh[e] = bia[e] - s_mean[gi[e]] * r;
    }
    bool sat = false;
#pragma unroll
    for (int k = 0; k < GNF32_MAXV; ++k) {
        const int p = pl + k * ppi;
        if (p >= d.HW) break;
        const int64_t o = ((int64_t)b * d.HW + p) * C + c;
        float y[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) y[e] = fmaf(xv[k][e], sc[e], sh[e]);
        if (d.gamma) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const float4 g4 = gv[k][q], b4 = bv[k][q];
                y[4 * q] = fmaf(y[4 * q], 1.f + g4.x, b4.x); y[4 * q + 1] = fmaf(y[4 * q + 1], 1.f + g4.y, b4.y);
                y[4 * q + 2] = fmaf(y[4 * q + 2], 1.f + g4.z, b4.z); y[4 * q + 3] = fmaf(y[4 * q + 3], 1.f + g4.w, b4.w);
            }
        }
        if (d.act == FRIDO_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < VW; ++e) y[e] = silu_f(y[e]);
        }
        if (d.nsplit == 2) sat |= sat_vec<VW>(y) || (d.raw_op && sat_vec<VW>(xv[k]));
        store_planes<VW>(d.out_op + o, d.out_lo, d.nsplit, y);
        if (d.raw_op) store_planes<VW>(d.raw_op + o, d.raw_lo, d.nsplit, xv[k]);
    }
    status_raise(sat);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row cached in registers (C <= 1024).
__global__ __launch_bounds__(256) void layernorm_kernel(const FridoLayerNorm d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.rows) return;
    if (d.x_bf16 && (d.C & 7) == 0 && d.nsplit == 1 && !d.out_f32) {
        // bf16 stream fast path: 16-byte loads / stores, up to two 8-channel chunks per lane (C <= 1024)
        const int C8 = d.C >> 3;
        const frido_bf16* xb = reinterpret_cast<const frido_bf16*>(d.x) + (int64_t)row * d.C;
        float x[2][8];
        float4 w4[2][2], b4[2][2];      // affine parameters: fetched now, under the two reductions
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c8 = lane + i * 64;
            if (c8 < C8) {
                const u32x4 xv = *reinterpret_cast<const u32x4*>(xb + c8 * 8);
                w4[i][0] = *reinterpret_cast<const float4*>(d.weight + c8 * 8); w4[i][1] = *reinterpret_cast<const float4*>(d.weight + c8 * 8 + 4);
                b4[i][0] = *reinterpret_cast<const float4*>(d.bias + c8 * 8); b4[i][1] = *reinterpret_cast<const float4*>(d.bias + c8 * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[i][2 * e] = __uint_as_float(xv[e] << 16); x[i][2 * e + 1] = __uint_as_float(xv[e] & 0xffff0000u); }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[i][e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[i][e];
        }
        const float mean = wave_sum(s) / d.C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (lane + i * 64 < C8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float a = x[i][e] - mean; q += a * a; }
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / d.C + d.eps);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c8 = lane + i * 64;
            if (c8 < C8) {
                u32x4 ov;
                const float wv[8] = {w4[i][0].x, w4[i][0].y, w4[i][0].z, w4[i][0].w, w4[i][1].x, w4[i][1].y, w4[i][1].z, w4[i][1].w};
                const float bv[8] = {b4[i][0].x, b4[i][0].y, b4[i][0].z, b4[i][0].w, b4[i][1].x, b4[i][1].y, b4[i][1].z, b4[i][1].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y0 = (x[i][2 * e] - mean) * rstd * wv[2 * e] + bv[2 * e];
                    const float y1 = (x[i][2 * e + 1] - mean) * rstd * wv[2 * e + 1] + bv[2 * e + 1];
                    ov[e] = f32_to_bf16_bits(y0) | (f32_to_bf16_bits(y1) << 16);
                }
                *reinterpret_cast<u32x4*>(d.out_op + (int64_t)row * d.C + c8 * 8) = ov;
            }
        }
        return;
    }
    const int C4 = d.C >> 2;
    float4 v[4], wv[4], bv[4];      // (r05) the affine parameters are fetched with the row, under the two reductions, not after them
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = lane + i * 64;
        v[i] = c4 < C4 ? load_act4(d.x, (int64_t)row * d.C + c4 * 4, d.x_bf16) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < C4) {
            wv[i] = *reinterpret_cast<const float4*>(d.weight + c4 * 4);
            bv[i] = *reinterpret_cast<const float4*>(d.bias + c4 * 4);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) / d.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (lane + i * 64 < C4) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, e = v[i].w - mean;
            q += (a * a + b * b) + (c * c + e * e);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / d.C + d.eps);
    bool sat = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < C4) {
            const float4 w = wv[i], bi = bv[i];
            const float y[4] = {(v[i].x - mean) * rstd * w.x + bi.x, (v[i].y - mean) * rstd * w.y + bi.y,
                                (v[i].z - mean) * rstd * w.z + bi.z, (v[i].w - mean) * rstd * w.w + bi.w};
            if (d.out_op) store_op4(d.out_op, d.out_lo, d.nsplit, (int64_t)row * d.C + c4 * 4, y);
            if (d.out_op && d.nsplit == 2) sat |= op_sat4(y);
            if (d.out_f32) *reinterpret_cast<float4*>(d.out_f32 + (int64_t)row * d.C + c4 * 4) = make_float4(y[0], y[1], y[2], y[3]);
        }
    }
    status_raise(sat, lane == 0 && stat_bad(mean, rstd));
}

// ---------------------------------------------------------------------------------------------
// Row softmax: one wave per row, N <= 4096 (64 values per lane in registers), zero-padded output.
__global__ __launch_bounds__(256) void softmax_kernel(const FridoSoftmax d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.rows) return;
    const float* x = d.x + (int64_t)row * d.ld;
    constexpr int MAXV = 64;
    float v[MAXV];
    const int nv = (d.N + 63) >> 6;
    const int nvis = d.causal_nq > 0 ? min(d.N, row % d.causal_nq + 1) : d.N;      // keys this row may see
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < nv) {
            const int c = lane + i * 64;
            v[i] = c < nvis ? x[c] : -3.0e38f;
            mx = fmaxf(mx, v[i]);
        }
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < nv) {
            const int c = lane + i * 64;
            v[i] = c < nvis ? __expf(v[i] - mx) : 0.f;
            s += v[i];
        }
    }
    const float inv = 1.0f / wave_sum(s);
    status_raise(false, lane == 0 && stat_bad(mx, inv));
    frido_bf16* o = d.out_op + (int64_t)row * d.Npad;
    const int npv = (d.Npad + 63) >> 6;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < npv) {
            const int c = lane + i * 64;
            if (c < d.Npad) store_op1(o, d.out_lo, d.nsplit, c, i < nv ? v[i] * inv : 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row L2 normalisation (one wave per row): the CLIP text embedding's z / ||z|| (encoders/modules.py:213-214).
__global__ __launch_bounds__(256) void l2norm_kernel(const FridoL2Norm d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= d.rows) return;
    const float* x = d.x + (int64_t)row * d.C;
    float s = 0.f;
    for (int c = lane; c < d.C; c += 64) s = fmaf(x[c], x[c], s);
    const float inv = 1.0f / sqrtf(wave_sum(s));
    for (int c = lane; c < d.C; c += 64) d.out[(int64_t)row * d.C + c] = x[c] * inv;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geglu_kernel(const FridoGeglu d) {
    const int H4 = d.H >> 2;
    const int64_t total = (int64_t)d.rows * H4;
    bool sat = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / H4;
        const int c = (int)(i - r * H4) * 4;
        const float4 a = *reinterpret_cast<const float4*>(d.x + r * 2 * d.H + c);
        const float4 g = *reinterpret_cast<const float4*>(d.x + r * 2 * d.H + d.H + c);
        const float aa[4] = {a.x, a.y, a.z, a.w}, gg[4] = {g.x, g.y, g.z, g.w};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = aa[e] * (0.5f * gg[e] * (1.0f + erff(gg[e] * 0.70710678118654752f)));
        store_op4(d.out_op, d.out_lo, d.nsplit, r * d.H + c, y);
        if (d.nsplit == 2) sat |= op_sat4(y);
    }
    status_raise(sat);
}

inline int grid_for(int64_t work_items, int cap = 4096) {
    int64_t b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int frido_gn_stats(const FridoGnStats* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->x1 && d->partials, "null pointer");
    const int C = d->C1 + d->C2;
    FRIDO_REQUIRE(d->groups > 0 && d->groups <= 64 && C % d->groups == 0, "channels not divisible by groups");
    FRIDO_REQUIRE((d->C1 & 3) == 0 && (d->C2 & 3) == 0 && C <= GN_MAXC, "channel counts must be multiples of 4, <= 4096");
    FRIDO_REQUIRE(d->B > 0 && d->HW > 0 && d->nsplit_px > 0, "empty");
    FRIDO_REQUIRE(d->C2 == 0 || d->x2, "x2 missing");
    if (d->p1) {
        FRIDO_REQUIRE(d->nsplit_px == 1 && (d->HW & 31) == 0 && d->groups <= 32 && (d->C2 == 0 || d->p2),
                      "statistics from partial sums: nsplit_px == 1, HW % 32 == 0, <= 32 groups, p2 for the second tensor");
        hipLaunchKernelGGL(gn_stats_parts_kernel, dim3(d->B * d->groups), dim3(256), 0, (hipStream_t)s, *d);
        return frido_check_launch("gn_stats(parts)");
    }
    hipLaunchKernelGGL(gn_stats_kernel, dim3(d->nsplit_px, d->B), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("gn_stats");
}

extern "C" int frido_gn_apply(const FridoGnApply* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->x1 && d->partials && d->weight && d->bias, "null pointer");
    FRIDO_REQUIRE(!d->sk_ws, "sk_ws (x1 from split-K partial sums) is served by frido_gn_fused only");
    const int C = d->C1 + d->C2;
    FRIDO_REQUIRE(d->groups > 0 && d->groups <= 32 && C % d->groups == 0, "channels not divisible by groups (<= 32 groups)");
    FRIDO_REQUIRE((d->C1 & 3) == 0 && (d->C2 & 3) == 0, "channel counts must be multiples of 4");
    FRIDO_REQUIRE(d->out_op || d->out_f32, "no output");
    FRIDO_REQUIRE((d->gamma == nullptr) == (d->beta == nullptr), "gamma/beta must come together");
    FRIDO_REQUIRE(C <= GN_MAXC && (int64_t)d->HW * (C >> 2) < (1ll << 31), "tensor too large for the 32-bit index path");
    const int64_t per_img = (int64_t)d->HW * (C >> 2);
    FRIDO_REQUIRE(d->nsplit_px <= 64, "at most 64 pixel splits");
    int gx = grid_for(per_img, 1536 / (d->B < 1536 ? d->B : 1536) + 8);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gx, d->B), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("gn_apply");
}

// (Cc, nt) of the f32 / bf16 form for a given vector width; 0 if no chunk of <= 4 whole groups is a multiple of `vw` channels
static int gn_fused_pick(const FridoGnApply* d, int vw, bool f32_form, int* nthreads) {
    const int C = d->C1 + d->C2, cpg = C / d->groups;
    int G = 1;
    while (G <= 4 && (G * cpg) % vw) ++G;
    if (G > 4 || d->groups % G) return 0;
    const int Cc = G * cpg, vpp = Cc / vw;
    for (int nt = 256; nt <= (f32_form ? 1024 : 256); nt *= 2) {
        if (vpp > nt) continue;
        const int ppi = nt / vpp;
        if ((d->HW + ppi - 1) / ppi <= (f32_form ? (nt == 1024 ? 3 : 4) : GNF_MAXV)) {
            if (nthreads) *nthreads = nt;
            return Cc;
        }
    }
    return 0;
}

// Chunk width of the one-launch GroupNorm: the smallest run of whole groups that is a multiple of 8 channels -- or (r05, f32 form) of 4
// channels, when that doubles the workgroups of a launch that would leave CUs idle; 0 if the op does not qualify (then the caller
// uses gn_stats + gn_apply).  Cc % 8 != 0 tells the launcher that the 4-channel form was chosen.
extern "C" int frido_gn_fused_chunk(const FridoGnApply* d, int* nthreads) {
    if (!d || d->out_f32 || !d->out_op) return 0;
    const bool bf16_form = d->x_bf16 && (!d->gamma || d->gb_bf16) && d->nsplit == 1;      // bf16 stream -> one bf16 plane
    const bool f32_form = !d->x_bf16 && (!d->gamma || !d->gb_bf16);                        // f32 stream -> hi (+ lo) planes
    if (!bf16_form && !f32_form) return 0;
    if (((d->C1 | d->C2) & 7) || d->groups <= 0) return 0;
    const int C = d->C1 + d->C2;
    if (C % d->groups) return 0;
    int nt8 = 0, nt4 = 0;
    int Cc8 = 0;      // (a lane's vector never straddles the two tensors of a virtual concat: C1 % 8 == 0)
    {
        const int cpg = C / d->groups;
        int G = 1;
        while (G <= 4 && (G * cpg) % 8) ++G;
        if (G <= 4 && d->groups % G == 0) {
            const int Cc = G * cpg, vpp = Cc / 8;
            for (int nt = 256; nt <= (f32_form ? 1024 : 256); nt *= 2) {
                if (vpp > nt) continue;
                const int ppi = nt / vpp;
                if ((d->HW + ppi - 1) / ppi <= (f32_form ? (nt == 1024 ? 3 : 4) : GNF_MAXV)) { nt8 = nt; Cc8 = Cc; break; }
            }
        }
    }
    static const bool v4_on = !(getenv("FRIDO_GN_FUSED_V4") && atoi(getenv("FRIDO_GN_FUSED_V4")) == 0);      // A/B switch
    if (f32_form && v4_on) {
        const int Cc4 = gn_fused_pick(d, 4, true, &nt4);
        // the 4-channel form where it at least doubles a grid that does not fill the chip (256 CUs)
        if (Cc4 > 0 && (Cc4 & 7) && (Cc8 == 0 || ((C / Cc8) * d->B < 256 && Cc4 * 2 <= Cc8))) {
            if (nthreads) *nthreads = nt4;
            return Cc4;
        }
    }
    if (Cc8 > 0 && nthreads) *nthreads = nt8;
    return Cc8;
}

extern "C" int frido_gn_fused(const FridoGnApply* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->x1 && d->out_op && d->weight && d->bias, "null pointer");
    int nt = 0;
    const int Cc = frido_gn_fused_chunk(d, &nt);
    FRIDO_REQUIRE(Cc > 0, "GroupNorm does not qualify for the one-launch kernel (use gn_stats + gn_apply)");
    FRIDO_REQUIRE(!d->sk_ws || (!d->x_bf16 && d->sk_n > 1 && (d->C1 & 7) == 0 && (d->sk_ldr & 3) == 0 && (d->sk_ldv & 3) == 0 &&
                                (!d->sk_rowvec || d->sk_rows_per_vec > 0) && (int64_t)d->B * d->HW < (1ll << 31)),
                  "sk_ws (x1 from split-K partial sums): f32 input, >= 2 slices, C1 % 8 == 0, 4-element aligned strides");
    const int C = d->C1 + d->C2;
    const int vw = (Cc & 7) ? 4 : 8;
    // (r05) an sk_ws launch takes the 1024-thread form whatever the plain launch would use: every lane then owns ONE vector of ONE
    // pixel and has all its slices' loads in flight at once (the 256-thread form walked 4 vectors x sk_n slices in 4 dependent rounds;
    // measured r04: -2 us per launch).  The statistics' wave-partial order differs from the plain launch's: deferred and undeferred
    // reductions agree to fp32 rounding (<= 1e-6 relative on the normalised output), no longer bit for bit.
    static const bool sk1024 = !(getenv("FRIDO_GN_FUSED_SK1024") && atoi(getenv("FRIDO_GN_FUSED_SK1024")) == 0);      // A/B switch
    if (sk1024 && d->sk_ws && !d->x_bf16 && nt < 1024 && Cc / vw <= 1024) nt = 1024;
    const dim3 grid(C / Cc, d->B);
    hipStream_t st = (hipStream_t)s;
    if (d->x_bf16) hipLaunchKernelGGL(gn_fused_kernel<256>, grid, dim3(256), 0, st, *d, Cc);
    else if (vw == 8) {
        if (nt == 256) hipLaunchKernelGGL((gn_fused_f32_kernel<256, 4, 8>), grid, dim3(256), 0, st, *d, Cc);
        else if (nt == 512) hipLaunchKernelGGL((gn_fused_f32_kernel<512, 4, 8>), grid, dim3(512), 0, st, *d, Cc);
        else hipLaunchKernelGGL((gn_fused_f32_kernel<1024, 3, 8>), grid, dim3(1024), 0, st, *d, Cc);
    } else {
        if (nt == 256) hipLaunchKernelGGL((gn_fused_f32_kernel<256, 4, 4>), grid, dim3(256), 0, st, *d, Cc);
        else if (nt == 512) hipLaunchKernelGGL((gn_fused_f32_kernel<512, 4, 4>), grid, dim3(512), 0, st, *d, Cc);
        else hipLaunchKernelGGL((gn_fused_f32_kernel<1024, 3, 4>), grid, dim3(1024), 0, st, *d, Cc);
    }
    return frido_check_launch("gn_fused");
}

extern "C" int frido_layernorm(const FridoLayerNorm* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->x && d->weight && d->bias && (d->out_op || d->out_f32), "null pointer");
    FRIDO_REQUIRE((d->C & 3) == 0 && d->C <= 1024 && d->rows > 0, "C must be a multiple of 4, <= 1024");
    hipLaunchKernelGGL(layernorm_kernel, dim3((d->rows + 3) / 4), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("layernorm");
}

extern "C" int frido_softmax(const FridoSoftmax* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->x && d->out_op, "null pointer");
    FRIDO_REQUIRE(d->N > 0 && d->N <= 4096 && d->Npad >= d->N && d->Npad <= 4096 && d->rows > 0, "N must be in [1, 4096]");
    hipLaunchKernelGGL(softmax_kernel, dim3((d->rows + 3) / 4), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("softmax");
}

extern "C" int frido_l2norm(const FridoL2Norm* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->x && d->out && d->rows > 0 && d->C > 0, "bad arguments");
    hipLaunchKernelGGL(l2norm_kernel, dim3((d->rows + 3) / 4), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("l2norm");
}

extern "C" int frido_geglu(const FridoGeglu* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->x && d->out_op, "null pointer");
    FRIDO_REQUIRE((d->H & 3) == 0 && d->rows > 0, "H must be a multiple of 4");
    hipLaunchKernelGGL(geglu_kernel, dim3(grid_for((int64_t)d->rows * (d->H >> 2))), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("geglu");
}
