// Small HBM-bound kernels of the sampling loop: operand packing, layout changes, the fused
// vector-quantiser lookup, the DDIM/PLMS state update (with CFG mix, Adams-Bashforth combine and a
// counter-based Philox RNG so results do not depend on how the batch is sharded over GPUs), the stage
// hand-off block mean, and a few utility fills.
#include "common.h"

namespace {

inline int grid_for(int64_t work_items, int cap = 8192) {
    int64_t b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---- f32 -> operand with channel slice + zero pad --------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(const FridoPack d) {
    const int P4 = d.Cpad >> 2;
    const int64_t total = (int64_t)d.B * d.HW * P4;
    bool sat = false;
    if (!d.nchw && d.nsplit == 2 && d.Cuse == d.Cpad && ((d.Cpad | d.Csrc | d.c0) & 7) == 0 && (d.out_lo & 7) == 0) {
        // (r05) the residual stream as an operand (Down / Upsample inputs, pyunet.py:110-156): 8 channels per lane -- two 16-byte
        // loads, one 16-byte store per plane (the element-wise loop below moved 4 scalar loads and two 8-byte stores per lane)
        const int P8 = d.Cpad >> 3;
        const int64_t total8 = (int64_t)d.B * d.HW * P8;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
            const int64_t pix = i / P8;
            const int c = (int)(i - pix * P8) * 8;
            const float* src = d.src + pix * d.Csrc + d.c0 + c;
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            const float v[8] = {a.x * d.scale, a.y * d.scale, a.z * d.scale, a.w * d.scale, b.x * d.scale, b.y * d.scale, b.z * d.scale, b.w * d.scale};
            uint32_t h[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; e += 2) { split_op2(v[e], v[e + 1], 2, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
            sat |= op_sat8(v);
            frido_bf16* o = d.out_op + pix * d.Cpad + c;
            *reinterpret_cast<uint4*>(o) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            *reinterpret_cast<uint4*>(o + d.out_lo) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
        }
        status_raise(sat);
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i / P4;
        const int c = (int)(i - pix * P4) * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cc = c + e;
            if (cc < d.Cuse) {
                if (d.nchw) {
                    const int64_t b = pix / d.HW, p = pix - b * d.HW;
                    v[e] = d.src[(b * d.Csrc + d.c0 + cc) * d.HW + p] * d.scale;
                } else {
                    v[e] = d.src[pix * d.Csrc + d.c0 + cc] * d.scale;
                }
            } else {
                v[e] = 0.f;
            }
        }
        store_op4(d.out_op, d.out_lo, d.nsplit, pix * d.Cpad + c, v);
        if (d.nsplit == 2) sat |= op_sat4(v);
    }
    status_raise(sat);
}

__global__ __launch_bounds__(256) void relayout_kernel(const FridoRelayout d) {
    const int64_t total = (int64_t)d.B * d.Cuse * d.HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % d.HW;
        const int64_t bc = i / d.HW;
        const int c = (int)(bc % d.Cuse);
        const int64_t b = bc / d.Cuse;
        if (d.to_nchw == 2)      // NHWC -> NHWC column block copy
            d.dst[(b * d.HW + p) * d.Cdst + d.d0 + c] = d.src[(b * d.HW + p) * d.Csrc + d.c0 + c];
        else if (d.to_nchw)
            d.dst[(b * d.Cdst + d.d0 + c) * d.HW + p] = d.src[(b * d.HW + p) * d.Csrc + d.c0 + c];
        else
            d.dst[(b * d.HW + p) * d.Cdst + d.d0 + c] = d.src[(b * d.Csrc + d.c0 + c) * d.HW + p];
    }
}

// ---- vector quantiser ---------------------------------------------------------------------------
constexpr int VQ_CHUNK = 2048, VQ_MAXE = 8;

__global__ __launch_bounds__(256) void vq_kernel(const FridoVq d) {
    __shared__ float s_code[VQ_CHUNK * VQ_MAXE];
    __shared__ float s_norm[VQ_CHUNK];
    const int t = threadIdx.x;
    const int64_t pix = (int64_t)blockIdx.x * 256 + t;
    const bool ok = pix < d.npix;
    float z[VQ_MAXE];
    float zz = 0.f;
#pragma unroll
    for (int k = 0; k < VQ_MAXE; ++k) {
        z[k] = (ok && k < d.e) ? d.x[pix * d.Cx + d.c0 + k] * d.inv_scale : 0.f;
        if (k < d.e) zz = __fadd_rn(zz, __fmul_rn(z[k], z[k]));
    }
    float best = 3.0e38f;
    int best_j = 0;
    const bool forced = d.force_idx != nullptr;         // uniform: every thread skips the search (and its barriers)
    if (forced && ok) {
        const int64_t f = d.force_idx[pix];
        best_j = (int)(f < 0 ? 0 : (f >= d.n_codes ? d.n_codes - 1 : f));
    }
    for (int j0 = 0; j0 < (forced ? 0 : d.n_codes); j0 += VQ_CHUNK) {
        const int nj = min(VQ_CHUNK, d.n_codes - j0);
        __syncthreads();
        for (int j = t; j < nj; j += 256) {
            float ee = 0.f;
            for (int k = 0; k < d.e; ++k) {
                const float c = d.codebook[(int64_t)(j0 + j) * d.e + k];
                s_code[j * VQ_MAXE + k] = c;
                ee = __fadd_rn(ee, __fmul_rn(c, c));
            }
            s_norm[j] = ee;
        }
        __syncthreads();
        for (int j = 0; j < nj; ++j) {
            float ze = 0.f;
#pragma unroll
            for (int k = 0; k < VQ_MAXE; ++k)
                if (k < d.e) ze = fmaf(z[k], s_code[j * VQ_MAXE + k], ze);
            const float dist = __fsub_rn(__fadd_rn(zz, s_norm[j]), __fmul_rn(2.0f, ze));
            if (dist < best) { best = dist; best_j = j0 + j; }
        }
    }
    if (!ok) return;
    for (int k = 0; k < d.e; ++k) {
        const float c = d.codebook[(int64_t)best_j * d.e + k];
        d.zq[pix * d.Cq + d.q0 + k] = __fadd_rn(z[k], __fsub_rn(c, z[k]));   // z + (z_q - z), quantize.py:294
    }
    if (d.idx) d.idx[pix] = best_j;
}

// ---- Philox4x32-10 + Box-Muller ---------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
// 4 standard normals for group `grp` of draw `draw` of global sample `sample`
__device__ __forceinline__ void randn4(uint64_t seed, int64_t sample, uint32_t draw, uint32_t stream, uint32_t grp, float out[4]) {
    uint32_t c[4] = {grp, draw, (uint32_t)sample, (uint32_t)((uint64_t)sample >> 32) ^ (stream << 20)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u0 = ((float)c[0] + 1.0f) * 2.3283064365386963e-10f;
    const float u1 = (float)c[1] * 2.3283064365386963e-10f;
    const float u2 = ((float)c[2] + 1.0f) * 2.3283064365386963e-10f;
    const float u3 = (float)c[3] * 2.3283064365386963e-10f;
    const float r0 = sqrtf(-2.0f * logf(fminf(u0, 1.0f))), r1 = sqrtf(-2.0f * logf(fminf(u2, 1.0f)));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u1, &s0, &c0);
    sincosf(6.283185307179586f * u3, &s1, &c1);
    out[0] = r0 * c0; out[1] = r0 * s0; out[2] = r1 * c1; out[3] = r1 * s1;
}

__global__ __launch_bounds__(256) void randn_kernel(const FridoRandn d) {
    const int64_t groups = (d.n + 3) >> 2;
    const int64_t gps = (d.per_sample + 3) >> 2;   // groups per sample
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
        // per_sample is a multiple of 4 on every caller, so groups never straddle samples
        const int64_t smp = g / gps;
        float r[4];
        randn4(d.seed, d.sample0 + smp, 0u, (uint32_t)d.rng_stream, (uint32_t)(g - smp * gps), r);
        for (int e = 0; e < 4; ++e)
            if (g * 4 + e < d.n) d.dst[g * 4 + e] = r[e];
    }
}

// ---- DDIM / PLMS update ---------------------------------------------------------------------------
constexpr int COEF_ROW = 12;   // a_t, a_prev, sigma, sqrt(1-a_t), ab0, ab1, ab2, ab3, den, pad x3

__global__ __launch_bounds__(256) void sampler_step_kernel(const FridoSamplerStep d) {
    const int64_t npix = (int64_t)d.B * d.HW;
    const int step = d.step ? *d.step : 0;
    const float* cf = d.coef + (int64_t)(step + d.coef_row_offset) * COEF_ROW;
    const float a_t = cf[0], a_prev = cf[1], sigma = cf[2], sqrt1m = cf[3];
    const float ab0 = cf[4], ab1 = cf[5], ab2 = cf[6], ab3 = cf[7], den = cf[8];
    const float sqrt_at = sqrtf(a_t), sqrt_ap = sqrtf(a_prev);
    const float dir_c = sqrtf(__fsub_rn(__fsub_rn(1.0f, a_prev), __fmul_rn(sigma, sigma)));
    const uint64_t seed = d.rng_dev ? (uint64_t)d.rng_dev[0] : d.seed;
    const int64_t sample0 = d.rng_dev ? d.rng_dev[1] : d.sample0;
    const float cfg = d.cfg_dev ? *d.cfg_dev : d.cfg_scale;
    // PLMS history: explicit pointers, or the 4-slot ring indexed by the device step counter (graph replay)
    const float* h1 = d.hist1;
    const float* h2 = d.hist2;
    const float* h3 = d.hist3;
    float* eout = d.eps_out;
    if (d.hist_ring && d.hist_mode) {
        auto slot = [&](int i) { return d.hist_ring + (int64_t)(i & 3) * d.hist_stride; };
        if (d.hist_mode == 3) {                 // second half of the Heun-style first step: combine with this step's own e_t
            h1 = slot(step); h2 = nullptr; h3 = nullptr; eout = nullptr;
        } else {
            eout = slot(step);
            h1 = step >= 1 ? slot(step - 1) : nullptr;
            h2 = step >= 2 ? slot(step - 2) : nullptr;
            h3 = step >= 3 ? slot(step - 3) : nullptr;
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / d.HW, p = i - b * d.HW;
        float nz[12];
        if (sigma != 0.f && d.write_x) {
            if (d.noise) {
                const float* np_ = d.noise + (int64_t)step * d.noise_stride + i * d.noise_C + d.noise_c0;
                for (int c = 0; c < d.nch; ++c) nz[c] = np_[c];
            } else {
                const int ngrp = (d.nch + 3) >> 2;
                for (int g = 0; g < ngrp; ++g)
                    randn4(seed, sample0 + b, (uint32_t)(step + d.coef_row_offset) + 1u, (uint32_t)d.rng_stream,
                           (uint32_t)(p * ngrp + g), nz + g * 4);
            }
        } else {
            for (int c = 0; c < d.nch; ++c) nz[c] = 0.f;
        }
        for (int c = 0; c < d.start; ++c) {   // frozen channels: x0 = x, x' = x0 (ddim.py:245-246,265-266)
            const float xv = d.x[i * d.Cx + c];
            if (d.write_x) d.x_out[i * d.Cx + c] = xv;
            if (d.pred_x0) d.pred_x0[i * d.Cx + c] = xv;
        }
        for (int c = 0; c < d.nch; ++c) {
            const int64_t ei = i * d.nch + c;
            float e = d.eps_cond[ei];
            if (d.eps_uncond) {
                const float eu = d.eps_uncond[ei];
                e = __fadd_rn(eu, __fmul_rn(cfg, __fsub_rn(e, eu)));
            }
            if (eout) eout[ei] = e;
            if (h1) {
                float acc = __fmul_rn(ab0, e);
                acc = __fadd_rn(acc, __fmul_rn(ab1, h1[ei]));
                if (h2) acc = __fadd_rn(acc, __fmul_rn(ab2, h2[ei]));
                if (h3) acc = __fadd_rn(acc, __fmul_rn(ab3, h3[ei]));
                e = __fdiv_rn(acc, den);
            }
            if (!d.write_x && !d.pred_x0) continue;
            const float xv = d.x[i * d.Cx + d.start + c];
            const float x0 = __fdiv_rn(__fsub_rn(xv, __fmul_rn(sqrt1m, e)), sqrt_at);
            if (d.pred_x0) d.pred_x0[i * d.Cx + d.start + c] = x0;
            if (d.write_x) {
                const float noise = __fmul_rn(__fmul_rn(sigma, nz[c]), d.temperature);
                d.x_out[i * d.Cx + d.start + c] =
                    __fadd_rn(__fadd_rn(__fmul_rn(sqrt_ap, x0), __fmul_rn(dir_c, e)), noise);
            }
        }
    }
}

// ---- stage hand-off -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void handoff_kernel(const FridoHandoff d) {
    const int bs = 1 << d.levels;
    const int nby = d.H / bs, nbx = d.W / bs, nc = d.c1 - d.c0;
    const int64_t total = (int64_t)d.B * nby * nbx * nc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = d.c0 + (int)(i % nc);
        int64_t r = i / nc;
        const int bx = (int)(r % nbx); r /= nbx;
        const int by = (int)(r % nby);
        const int64_t b = r / nby;
        float* base = d.x + ((b * d.H + (int64_t)by * bs) * d.W + (int64_t)bx * bs) * d.Cx + c;
        // hierarchical 2x2 means, exactly like avg_pool2d(2, 2) applied `levels` times (bs <= 4 supported)
        float m;
        auto px = [&](int y, int x) { return base[((int64_t)y * d.W + x) * d.Cx]; };
        auto mean4 = [&](int y, int x) {
            return __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(px(y, x), px(y, x + 1)), px(y + 1, x)), px(y + 1, x + 1)), 0.25f);
        };
        if (d.levels == 1) {
            m = mean4(0, 0);
        } else {
            const float m00 = mean4(0, 0), m01 = mean4(0, 2), m10 = mean4(2, 0), m11 = mean4(2, 2);
            m = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(m00, m01), m10), m11), 0.25f);
        }
        for (int y = 0; y < bs; ++y)
            for (int x = 0; x < bs; ++x) base[((int64_t)y * d.W + x) * d.Cx] = m;
    }
}

__global__ __launch_bounds__(256) void time_emb_kernel(const FridoTimeEmb d) {
    const int half = d.dim / 2;
    const int total = d.n * half;
    const float neg_log = -logf(d.max_period);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / half, k = i - r * half;
        const float f = expf(__fdiv_rn(__fmul_rn(neg_log, (float)k), (float)half));
        const float a = __fmul_rn((float)d.t[r], f);
        d.out[(int64_t)r * d.dim + k] = cosf(a);
        d.out[(int64_t)r * d.dim + half + k] = sinf(a);
        if ((d.dim & 1) && k == 0) d.out[(int64_t)r * d.dim + d.dim - 1] = 0.f;
    }
}

__global__ __launch_bounds__(256) void convt_kernel(const FridoConvT d) {
    const int H = d.h * 2, W = d.w * 2;
    const int64_t total = (int64_t)d.B * H * W * d.Cout;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % d.Cout);
        int64_t r = i / d.Cout;
        const int ox = (int)(r % W); r /= W;
        const int oy = (int)(r % H);
        const int64_t b = r / H;
        float acc = d.bias ? d.bias[co] : 0.f;
        for (int ky = 0; ky < 4; ++ky) {
            const int ty = oy + 1 - ky;                 // oy = iy*2 - 1 + ky
            if (ty < 0 || (ty & 1) || (ty >> 1) >= d.h) continue;
            for (int kx = 0; kx < 4; ++kx) {
                const int tx = ox + 1 - kx;
                if (tx < 0 || (tx & 1) || (tx >> 1) >= d.w) continue;
                const float* px = d.src + ((b * d.h + (ty >> 1)) * d.w + (tx >> 1)) * d.Cin;
                for (int ci = 0; ci < d.Cin; ++ci) acc = fmaf(px[ci], d.weight[((ci * d.Cout + co) * 4 + ky) * 4 + kx], acc);
            }
        }
        d.dst[i] = acc;
    }
}

__global__ __launch_bounds__(256) void place_kernel(const FridoPlace d) {
    const int H = d.h << d.up_shift, W = d.w << d.up_shift;
    const int64_t total = (int64_t)d.B * d.Cuse * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % W);
        int64_t r = i / W;
        const int oy = (int)(r % H); r /= H;
        const int c = (int)(r % d.Cuse);
        const int64_t b = r / d.Cuse;
        const float v = d.src[((b * d.h + (oy >> d.up_shift)) * d.w + (ox >> d.up_shift)) * d.Csrc + d.c0 + c];
        d.dst[((b * d.Cdst + d.d0 + c) * H + oy) * W + ox] = v * d.scale;
    }
}

__global__ __launch_bounds__(256) void embed_kernel(const FridoEmbed d) {
    const int D4 = d.D >> 2;
    const int64_t total = (int64_t)d.rows * D4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / D4;
        const int c = (int)(i - r * D4) * 4;
        int64_t tk = d.tokens[r];
        tk = tk < 0 ? 0 : (tk >= d.vocab ? d.vocab - 1 : tk);
        const float4 a = *reinterpret_cast<const float4*>(d.tok + tk * d.D + c);
        const float4 p = d.pos ? *reinterpret_cast<const float4*>(d.pos + (r % d.n) * d.D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(d.out + r * d.D + c) = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}

__global__ __launch_bounds__(256) void to_u8_kernel(const FridoToU8 d) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
        float v = __fmul_rn(__fadd_rn(d.src[i], 1.0f), 127.5f);
        v = fminf(fmaxf(v, 0.f), 255.f);
        d.dst[i] = (uint8_t)v;
    }
}

__global__ void step_add_kernel(const FridoStepAdd d) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *d.step += d.delta;
}

__global__ __launch_bounds__(256) void copy_kernel(const FridoCopy d) {
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(d.src);
    uint4* __restrict__ dst = reinterpret_cast<uint4*>(d.dst);
    const int64_t n16 = d.n >> 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void fill_kernel(const FridoFill d) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) d.dst[i] = d.value;
}

}  // namespace

extern "C" int frido_pack(const FridoPack* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->src && d->out_op, "null pointer");
    FRIDO_REQUIRE((d->Cpad & 3) == 0 && d->Cuse <= d->Cpad && d->c0 + d->Cuse <= d->Csrc && d->Cuse > 0, "bad channel slice");
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for((int64_t)d->B * d->HW * (d->Cpad >> 2))), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("pack");
}

extern "C" int frido_relayout(const FridoRelayout* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->src && d->dst, "null pointer");
    FRIDO_REQUIRE(d->c0 + d->Cuse <= d->Csrc && d->d0 + d->Cuse <= d->Cdst && d->Cuse > 0, "bad channel slice");
    hipLaunchKernelGGL(relayout_kernel, dim3(grid_for((int64_t)d->B * d->HW * d->Cuse)), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("relayout");
}

extern "C" int frido_vq(const FridoVq* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->x && d->codebook && d->zq, "null pointer");
    FRIDO_REQUIRE(d->e > 0 && d->e <= VQ_MAXE && d->n_codes > 0 && d->npix > 0, "bad sizes");
    hipLaunchKernelGGL(vq_kernel, dim3((d->npix + 255) / 256), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("vq");
}

extern "C" int frido_sampler_step(const FridoSamplerStep* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->x && d->eps_cond && d->coef, "null pointer");
    FRIDO_REQUIRE(d->nch > 0 && d->nch <= 12 && d->start >= 0 && d->start + d->nch <= d->Cx, "bad channel range");
    FRIDO_REQUIRE(!d->write_x || d->x_out, "x_out missing");
    FRIDO_REQUIRE(!d->hist2 || d->hist1, "history must be contiguous");
    FRIDO_REQUIRE(!d->hist_mode || (d->hist_ring && d->step && d->hist_stride >= (int64_t)d->B * d->HW * d->nch &&
                                    (d->hist_mode == 1 || d->hist_mode == 3) && !d->hist1 && !d->eps_out),
                  "hist ring: needs the device step counter, a [4][hist_stride] buffer and no explicit history pointers");
    hipLaunchKernelGGL(sampler_step_kernel, dim3(grid_for((int64_t)d->B * d->HW, 2048)), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("sampler_step");
}

extern "C" int frido_handoff(const FridoHandoff* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->x, "null pointer");
    FRIDO_REQUIRE(d->levels >= 1 && d->levels <= 2, "hand-off supports 2x2 and 4x4 block means");
    FRIDO_REQUIRE(d->H % (1 << d->levels) == 0 && d->W % (1 << d->levels) == 0 && d->c1 > d->c0 && d->c1 <= d->Cx, "bad geometry");
    const int bs = 1 << d->levels;
    hipLaunchKernelGGL(handoff_kernel, dim3(grid_for((int64_t)d->B * (d->H / bs) * (d->W / bs) * (d->c1 - d->c0))), dim3(256),
                       0, (hipStream_t)s, *d);
    return frido_check_launch("handoff");
}

extern "C" int frido_randn(const FridoRandn* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->dst && d->n > 0 && d->per_sample > 0 && (d->per_sample & 3) == 0, "bad arguments");
    hipLaunchKernelGGL(randn_kernel, dim3(grid_for((d->n + 3) >> 2)), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("randn");
}

extern "C" int frido_step_add(const FridoStepAdd* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->step, "null pointer");
    hipLaunchKernelGGL(step_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, *d);
    return frido_check_launch("step_add");
}

extern "C" int frido_time_emb(const FridoTimeEmb* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->t && d->out && d->n > 0 && d->dim >= 2, "bad arguments");
    hipLaunchKernelGGL(time_emb_kernel, dim3(grid_for((int64_t)d->n * (d->dim / 2))), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("time_emb");
}

extern "C" int frido_convt(const FridoConvT* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->src && d->dst && d->weight && d->Cin > 0 && d->Cout > 0 && d->B > 0, "bad arguments");
    hipLaunchKernelGGL(convt_kernel, dim3(grid_for((int64_t)d->B * d->h * d->w * 4 * d->Cout)), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("convt");
}

extern "C" int frido_place(const FridoPlace* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->src && d->dst && d->Cuse > 0 && d->c0 + d->Cuse <= d->Csrc && d->d0 + d->Cuse <= d->Cdst, "bad arguments");
    hipLaunchKernelGGL(place_kernel, dim3(grid_for(((int64_t)d->B * d->Cuse * d->h * d->w) << (2 * d->up_shift))), dim3(256), 0,
                       (hipStream_t)s, *d);
    return frido_check_launch("place");
}

extern "C" int frido_embed(const FridoEmbed* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->tokens && d->tok && d->out && d->rows > 0 && d->n > 0 && (d->D & 3) == 0 && d->vocab > 0,
                  "bad arguments");
    hipLaunchKernelGGL(embed_kernel, dim3(grid_for((int64_t)d->rows * (d->D >> 2))), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("embed");
}

extern "C" int frido_to_u8(const FridoToU8* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->src && d->dst && d->n > 0, "bad arguments");
    hipLaunchKernelGGL(to_u8_kernel, dim3(grid_for(d->n)), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("to_u8");
}

extern "C" int frido_copy(const FridoCopy* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->src && d->dst && d->n > 0 && (d->n & 15) == 0 && (((uintptr_t)d->src | (uintptr_t)d->dst) & 15) == 0,
                  "copy: 16-byte aligned pointers and size");
    hipLaunchKernelGGL(copy_kernel, dim3(grid_for(d->n >> 4, 2048)), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("copy");
}

extern "C" int frido_fill(const FridoFill* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->dst && d->n > 0, "bad arguments");
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(d->n)), dim3(256), 0, (hipStream_t)s, *d);
    return frido_check_launch("fill");
}
