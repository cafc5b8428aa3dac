// Fused GroupNorm-apply + 3x3 convolution of the two-plane (parity) arithmetic; shares the tile epilogue and the LDS / DMA helpers
// of the implicit-GEMM family (igemm_shared.h).
#include "igemm_shared.h"

namespace {

// =====================================================================================================================
// conv3x3_gn_kernel: two-plane (parity arithmetic) 3x3 stride-1 convolution whose A operand is PRODUCED IN THE KERNEL from
// the f32 residual stream:  out = conv3x3( act( GroupNorm(x) [* (1 + gamma) + beta] ) ) [+ conv1x1(raw)] + bias ...
// (pyunet.py:262-300 ResBlock: in_layers = GN -> SiLU -> conv, out_layers = GN -> SiLU -> conv, skip_connection;
//  spade_norm.py:44-60 for the gamma / beta form).  It replaces the pair  gn_apply_kernel -> igemm_kernel<.., CONV>  on the
// 64 x 64 and 32 x 32 planes: the normalised operand (hi / lo fp16 planes, 4 B per element written and 9 x 4 B per element
// pulled through L2 -> LDS by the ring kernel's tap walk) never exists in memory.
//
//   * tile BM x 192 on eight waves (4 x 2; wave tile BM/4 x 96), BM = 256 (64^2 planes: 4 image rows) or 128 (32^2: 4 rows);
//   * per 32-channel chunk the (R + 2) x (W + 2) pixel PATCH of the tile is staged ONCE: raw f32 -> registers ->
//     y = x * sc[c] + sh[c]  [-> y * (1 + gamma) + beta]  [-> SiLU]  -> hi / lo split -> ds_write_b128 into one of two patch
//     buffers (zero halo written as zeros: the conv pads the ACTIVATION); the nine taps read their pixel fragments from it
//     with a constant slot shift.  Arithmetic per element is gn_apply_kernel's, expression for expression: the operand bits
//     are the ones the two-kernel path feeds the MFMAs;
//   * the staging of chunk c + 1 is spread over the taps of chunk c: a round's loads go out two taps before its conversion, the
//     two waves of a SIMD convert at alternating taps (one does VALU work while the other issues MFMAs);
//   * only the weights stream per tap: [2 planes][192][32] = 24 KiB by LDS-DMA into a 2-stage ring (one tap ahead);
//   * optional appended K range (fused 1x1 skip conv, pyunet.py:248,300): the raw f32 input rows of the tile, split into
//     hi / lo, as dense [BM][32] tiles after the last conv chunk -- one k-step each, staged two steps ahead;
//   * epilogue = tile_epilogue (bias, timestep vector, residual, f32 stream out, GroupNorm partial sums for the NEXT norm).
//
// LDS (BM = 256): 2 x 49.5 KiB patch + 2 x 24 KiB weights + 7.5 KiB scale / shift table = 154.5 KiB, one workgroup per CU.

// Timing ablations (separate builds, results garbage): 1 = no convert / LDS write of staged units, 2 = no staging loads,
// 4 = no MFMAs, 8 = no fragment reads, 16 = no weight DMA inside the loop
#ifndef CG_ABLATE
#define CG_ABLATE 0
#endif
// (r06) 1: the step barrier sits INSIDE the k-step, at the head of its last weight slab (see "mid-step barrier" in the kernel); 0: r04's
// barrier at the top of every step.  Same MFMAs on the same operands in the same order: bit-identical results.
#ifndef CG_MIDBAR
#define CG_MIDBAR 0
#endif
// (r06) -DCG_PROF=1 (tools/cg_prof.py; never in the shipped build): every wave of the r04 loop form (CG_MIDBAR=0) sums, over its k-steps,
// the shader cycles (s_memtime) it spends  [0] waiting for the step's weights (end of the previous step -> arrival at the barrier),
// [1] inside the barrier, [2] on the weight DMA issue + staging (release -> first fragment address), [3] on fragment reads + MFMAs;
// [4] = the whole loop, [5] = kernel start -> loop, [6] = epilogue.  Stamps sit where lgkmcnt is 0 anyway (an s_memtime is an SMEM read).
#ifndef CG_PROF
#define CG_PROF 0
#endif
// (r06) 1: PING-PONG.  The two waves of a SIMD (w and w + 4) run HALF A STEP APART: waves 0-3 own the tile's columns 0-95, waves 4-7 columns
// 96-191 (so each group streams ITS OWN half of a step's weights and nobody else reads them), group X = waves 0-3 meets at the "A" barriers,
// group Y at the "B" barriers in between; a group's own barrier is the top of its step (weight DMA issue, staging), the other group's
// barrier falls between its weight slabs CG_PP_K - 1 and CG_PP_K.  While one wave of a SIMD is in its top-of-step work (800 - 1000 cycles
// without a single MFMA: tools/cg_prof.py) its partner is in the MFMA half of its own step.  r04 loop form only (CG_MIDBAR 0).
// Same MFMAs on the same operands in the same order per accumulator: bit-identical results.
#ifndef CG_PINGPONG
#define CG_PINGPONG 0
#endif
#ifndef CG_PP_K
#define CG_PP_K 3
#endif

#if CG_PROF
__device__ unsigned g_cg_prof[2048 * 8 * 8];       // [workgroup][wave][8]
#define CGP_NOW() ((unsigned)__builtin_readcyclecounter())
#endif

template <int BM>
struct CGeo {
    static constexpr int BN = 192, NW = 8, NT = NW * 64, WM = 4, WN = 2;
    static constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    static constexpr int MAXSLOTS = BM == 256 ? 396 : 204;          // (R + 2)(W + 2): 6 x 66 (64^2, BM 256), 10 x 34 (32^2, BM 256), 6 x 34 (32^2, BM 128)
    static constexpr int PPLANE = MAXSLOTS * 64;                     // bytes of one plane of a patch buffer (64 B per pixel slot)
    static constexpr int PBUF = 2 * PPLANE;
    static constexpr int WPLANE = BN * 64, WSTAGE = 2 * WPLANE;
    static constexpr int W0 = 2 * PBUF;
    static constexpr int TAB0 = W0 + 2 * WSTAGE;
    static constexpr int MAXC = 960;
    static constexpr int SMEM = TAB0 + 2 * MAXC * 4;
    static constexpr int NR = (MAXSLOTS * 4 + NT - 1) / NT;          // staging rounds per chunk (8-channel units per thread): 4 / 2
    static constexpr int RU = BM * 4 / NT;                           // units per thread of a dense raw [BM][32] tile: 2 / 1
    static_assert(BM * 64 * 2 <= PBUF, "a raw [BM][32] two-plane tile fits a patch buffer");
    static_assert(NW * 16 * (BN / 2 + 4) * 4 <= SMEM, "epilogue slabs");
    static_assert(SMEM <= 163840, "LDS budget");
    static_assert(WPLANE == 12 * 1024, "12 one-KiB weight chunks per plane");
};

__device__ __forceinline__ void lds_write128u(unsigned addr, u32x4 v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

struct CgUnit { f32x4 x0, x1, g0, g1, b0, b1; };

// Staging loads are issued and waited for BY HAND: hipcc fences a compiler-visible global load that is consumed after an LDS-DMA
// with vmcnt(0) (measured: ~1300 cycles per staging round, the round trip of the weight DMA issued just before).  The loads are
// inline asm (invisible to the compiler's counter) and the wait carries the loaded registers as in/out operands, so that no
// consumer can be scheduled above it.
// (r06) scalar-base form: the tensor's base is wave-uniform (an SGPR pair), the lane's part a 32-bit BYTE offset -- no 64-bit address
// arithmetic or address register pair per load (the HASRAW instantiations of BM = 256 spilled 172 / 224 B per lane on the 64-bit form);
// convgn_ok bounds every staged tensor to < 4 GiB
template <int IMM>
__device__ __forceinline__ f32x4 gload128(const float* base, unsigned off) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(off), "s"(base), "n"(IMM));
    return v;
}
template <int N, bool SPADE>
__device__ __forceinline__ void wait_unit(CgUnit& u) {       // at most N younger VMEM operations may stay in flight
    if constexpr (SPADE)
        asm volatile("s_waitcnt vmcnt(%6)" : "+v"(u.x0), "+v"(u.x1), "+v"(u.g0), "+v"(u.g1), "+v"(u.b0), "+v"(u.b1) : "n"(N));
    else
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(u.x0), "+v"(u.x1) : "n"(N));
}

// gn_apply_kernel's per-element arithmetic (norm.hip), expression for expression
template <bool SPADE>
__device__ __forceinline__ void cg_convert(const CgUnit& u, const float (&sc)[8], const float (&sh)[8], bool norm, bool silu, bool zero,
                                           u32x4& hi, u32x4& lo, bool& sat) {
    const float xin[8] = {u.x0[0], u.x0[1], u.x0[2], u.x0[3], u.x1[0], u.x1[1], u.x1[2], u.x1[3]};
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = xin[e];
    if (norm) {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = fmaf(xin[e], sc[e], sh[e]);
        if constexpr (SPADE) {
            const float ga[8] = {u.g0[0], u.g0[1], u.g0[2], u.g0[3], u.g1[0], u.g1[1], u.g1[2], u.g1[3]};
            const float be[8] = {u.b0[0], u.b0[1], u.b0[2], u.b0[3], u.b1[0], u.b1[1], u.b1[2], u.b1[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = fmaf(y[e], 1.f + ga[e], be[e]);
        }
        if (silu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = silu_f(y[e]);
        }
    }
    uint32_t hp[4], lp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_op2(y[2 * e], y[2 * e + 1], 2, hp[e], lp[e]);
    // status word (common.h): 4 v_max3 + 1 compare per 8 values, the flag lives in an SGPR pair.  Only values that are actually STAGED count:
    // a halo / unused slot converts whatever a stand-in pixel holds and then writes zeros (tools/find_saturation.py: those lanes raised the
    // bit in 5 - 10 of 400 steps of the benchmark, the real pixels in none)
    sat |= !zero && op_sat8(y);
    hi = u32x4{hp[0], hp[1], hp[2], hp[3]};
    lo = u32x4{lp[0], lp[1], lp[2], lp[3]};
    if (zero) { hi = u32x4{0u, 0u, 0u, 0u}; lo = hi; }
}

template <int BM, bool SPADE, bool HASRAW>
__global__ __launch_bounds__(512, 1) void conv3x3_gn_kernel(const FridoGemm d) {
    using G = CGeo<BM>;
    constexpr int BN = G::BN, NW = G::NW, NT = G::NT, WM = G::WM, TM = G::TM, TN = G::TN, NR = G::NR, RU = G::RU;
    constexpr int PPLANE = G::PPLANE, PBUF = G::PBUF, WPLANE = G::WPLANE, WSTAGE = G::WSTAGE, W0 = G::W0, TAB0 = G::TAB0, MAXC = G::MAXC;
    constexpr int NL = SPADE ? 6 : 2;                 // global loads per thread of one staging round
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = CG_PINGPONG ? wave & 3 : wave >> 1, wn = CG_PINGPONG ? wave >> 2 : wave & 1;      // (ping-pong: a SIMD's two waves w, w + 4 = the two column halves of one row block)
#if CG_PROF
    unsigned cgp[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const unsigned cgp_t0 = CGP_NOW();
    unsigned cgp_prev = cgp_t0, cgp_a = 0u;
#endif

    const int tiles_n = d.N / BN;
    const int nb = (d.M / BM) * tiles_n;
    // (k-slice, tile) work items, XCD-contiguous like igemm_kernel's: neighbouring tiles share halo rows, a k-slice's weights sit in ONE L2
    const int kz = xcd_item(nb) / nb;
    const int bid = xcd_item(nb) - kz * nb;
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    stagger_one_per_cu(d.flags);
    const int W = d.Ws, H = d.Hs, HW = H * W;
    const int R = BM / W, PW = W + 2, PS = (R + 2) * PW;
    const int img = m0 / HW, y0 = (m0 - img * HW) / W;
    const int C1 = d.gn_C1, C2 = d.gn_C2, C = C1 + C2;
    const float* __restrict__ x1 = d.gn_x1;
    const float* __restrict__ x2 = d.gn_x2;
    const float* __restrict__ gam = d.gn_gamma;
    const float* __restrict__ bet = d.gn_beta;
    const frido_bf16* __restrict__ Bb = d.B;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    float* const tab_sc = reinterpret_cast<float*>(smem + TAB0);
    float* const tab_sh = tab_sc + MAXC;

    // ---- GroupNorm statistics -> per-channel scale / shift table (gn_apply_kernel's prologue: same order, same expressions) ----
    if constexpr (CG_ABLATE & 64) {       // (timing only) no statistics prologue
        for (int c = t; c < C; c += NT) { tab_sc[c] = 1.f; tab_sh[c] = 0.f; }
        __syncthreads();
    } else
    {
        float* s_mean = reinterpret_cast<float*>(smem);
        float* s_rstd = s_mean + 64;
        const int cpg = C / d.gn_groups;
        double* s_part = reinterpret_cast<double*>(smem + 1024);            // [8][32][2], like gn_apply_kernel
        if (t < 256) {      // combine the per-split partials: 8 lanes per group in parallel, then a fixed-order sum of the 8
            const int g = t & 31, l8 = t >> 5;
            double s = 0.0, q = 0.0;
            if (g < d.gn_groups) {
                double2 pv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int sp = l8 + 8 * k;
                    pv[k] = sp < d.gn_nsplit_px
                                ? *reinterpret_cast<const double2*>(d.gn_partials + (((int64_t)img * d.gn_nsplit_px + sp) * d.gn_groups + g) * 2)
                                : make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) { s += pv[k].x; q += pv[k].y; }
                s_part[(l8 * 32 + g) * 2] = s;
                s_part[(l8 * 32 + g) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (t < d.gn_groups) {
            double s = 0.0, q = 0.0;
            for (int l = 0; l < 8; ++l) { s += s_part[(l * 32 + t) * 2]; q += s_part[(l * 32 + t) * 2 + 1]; }
            const double n = (double)HW * cpg;
            const double mean = s / n;
            double var = q / n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mean[t] = (float)mean;
            s_rstd[t] = (float)(1.0 / sqrt(var + (double)d.gn_eps));
            if (bid == 0) status_raise(false, stat_bad(s_mean[t], s_rstd[t]));      // (one workgroup per launch reports: tile 0 sees sample 0's statistics only -- gn_stats flags the others)
        }
        __syncthreads();
        for (int c = t; c < C; c += NT) {
            const int g = c / cpg;
            const float sc = s_rstd[g] * d.gn_weight[c];
            tab_sc[c] = sc;
            tab_sh[c] = d.gn_bias[c] - s_mean[g] * sc;
        }
        __syncthreads();
    }

    // ---- staging geometry of this thread: unit (round r) = pixel slot r * 128 + (t >> 2), 8 channels (t & 3) * 8 of the chunk ----
    const int cu = (t & 3) * 8;
    int pix[NR];                                      // global pixel index, -1: halo (zeros), -2: beyond the patch (no write)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int slot = r * 128 + (t >> 2);
        const int pr = slot / PW, px = slot - pr * PW;
        const int y = y0 + pr - 1, x = px - 1;
        const bool ok = y >= 0 && y < H && x >= 0 && x < W;
        pix[r] = slot >= PS ? -2 : (ok ? img * HW + y * W + x : -1);
    }
    // LDS byte offset of this thread's 16-byte piece in round 0 (piece q of slot s sits at q ^ (((s >> 2) & 1) << 1): conflict-free
    // ds_read_b128 fragments from ANY starting slot, which the tap shifts need); round r adds r * 128 slots = r * 8192 bytes
    const unsigned wslot = (unsigned)(((t >> 2) << 6) + (((t & 3) ^ ((((t >> 2) >> 2) & 1) << 1)) << 4));
    // split-K (gridDim.z slices, r04: the 16 x 16 planes, whose 48 tiles would leave the chip idle): slice kz takes an equal share of
    // the conv chunks AND of the raw chunks -- every slice has the same structure and at least one conv chunk (splitk <= C / 32);
    // partial sums go to the workspace through tile_epilogue, frido_gemm launches splitk_reduce behind this kernel
    const int nsl = (int)gridDim.z;
    const int nc_all = C >> 5, nraw_all = HASRAW ? d.K2 >> 5 : 0;       // (Cin == C: both multiples of 32)
    const int cb = nc_all * kz / nsl, nc = nc_all * (kz + 1) / nsl;     // conv chunks [cb, nc)
    const int rb = nraw_all * kz / nsl, re = nraw_all * (kz + 1) / nsl; // raw chunks [rb, re)
    const int nraw = re - rb;

    auto load_unit = [&](int c, int p, CgUnit& u) {   // chunk c of the (virtually concatenated) GroupNorm input at pixel p
        const int ch = c * 32 + cu;
        const bool first = c * 32 < C1;                // wave-uniform: a 32-channel chunk lies in ONE of the two concatenated tensors
        const float* base = first ? x1 : x2;
        const int Cs = first ? C1 : C2, c0s = first ? c * 32 : c * 32 - C1;      // (scalar selects: one multiply per lane, nothing per-tensor to hoist)
        const unsigned off = (unsigned)(p * Cs + c0s + cu) * 4u;
        u.x0 = gload128<0>(base, off);
        u.x1 = gload128<16>(base, off);
        if constexpr (SPADE) {
            const unsigned o = (unsigned)(p * C + ch) * 4u;
            u.g0 = gload128<0>(gam, o);
            u.g1 = gload128<16>(gam, o);
            u.b0 = gload128<0>(bet, o);
            u.b1 = gload128<16>(bet, o);
        }
    };
    bool sat = false;                                 // an operand value beyond the fp16 planes' range was staged (common.h status word)
    const int psafe = img * HW + y0 * W;              // a valid pixel for lanes whose slot is halo / unused (loaded, never written)
    const bool do_silu = d.gn_act == FRIDO_ACT_SILU;
    auto stage_load = [&](auto rc, int c, CgUnit& u) {
        constexpr int r = decltype(rc)::value;
        if constexpr (CG_ABLATE & 2) { u.x0 = u.x1 = u.g0 = u.g1 = u.b0 = u.b1 = f32x4{0.f, 0.f, 0.f, 0.f}; return; }
        load_unit(c, pix[r] >= 0 ? pix[r] : psafe, u);
    };
    // convert + write round r of chunk c into patch buffer c & 1; YOUNGER = VMEM operations issued after the round's loads
    auto stage_write = [&](auto rc, auto yc, int c, CgUnit& u) {
        constexpr int r = decltype(rc)::value, YOUNGER = decltype(yc)::value;
        if constexpr (!(CG_ABLATE & 2)) wait_unit<YOUNGER, SPADE>(u);
        if constexpr (CG_ABLATE & 1) { asm volatile("" ::"v"(u.x0), "v"(u.x1)); return; }
        float sc[8], sh[8];
        {
            const unsigned ta = lds0 + TAB0 + (unsigned)(c * 32 + cu) * 4u;
            const float4 a0 = lds_read128f<0>(ta), a1 = lds_read128f<16>(ta);
            const float4 b0 = lds_read128f<MAXC * 4>(ta), b1 = lds_read128f<MAXC * 4 + 16>(ta);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
            sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
        }
        u32x4 hi, lo;
        cg_convert<SPADE>(u, sc, sh, true, do_silu, pix[r] < 0, hi, lo, sat);
        if (pix[r] != -2) {
            const unsigned a = lds0 + (unsigned)((c & 1) * PBUF) + wslot + (unsigned)(r * 8192);
            lds_write128u(a, hi);
            lds_write128u(a + PPLANE, lo);
        }
    };

    // ---- raw (skip-conv) tiles: unit q of this thread = tile row q * 128 + (t >> 2), channels (t & 3) * 8 of raw chunk s ----
    const float* __restrict__ rx1 = d.raw_x1;
    const float* __restrict__ rx2 = d.raw_x2;
    const int RC1 = d.raw_C1, RC2 = d.raw_C2;
    auto raw_load = [&](int s, CgUnit (&u)[RU]) {      // s: index within this slice's raw chunks
        const int ch = (rb + s) * 32 + cu;
        const bool first = (rb + s) * 32 < RC1;        // wave-uniform, as in load_unit
        const float* base = first ? rx1 : rx2;
        const int RCs = first ? RC1 : RC2, rc0 = first ? (rb + s) * 32 : (rb + s) * 32 - RC1;
#pragma unroll
        for (int q = 0; q < RU; ++q) {
            const int p = m0 + q * 128 + (t >> 2);
            const unsigned off = (unsigned)(p * RCs + rc0 + cu) * 4u;
            u[q].x0 = gload128<0>(base, off);
            u[q].x1 = gload128<16>(base, off);
        }
    };
    auto raw_write = [&](auto yc, int buf, CgUnit (&u)[RU]) {
        constexpr int YOUNGER = decltype(yc)::value;
        const float none[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < RU; ++q) {
            wait_unit<YOUNGER, false>(u[q]);
            u32x4 hi, lo;
            cg_convert<false>(u[q], none, none, false, false, false, hi, lo, sat);
            const unsigned a = lds0 + (unsigned)(buf * PBUF) + wslot + (unsigned)(q * 8192);
            lds_write128u(a, hi);
            lds_write128u(a + PPLANE, lo);
        }
    };

    // ---- weight DMA: 24 one-KiB chunks per tap ([2 planes][12 x 16 rows]), three per wave; the ring kernel's B layout ----
    const int lrow = lane >> 2;
    const int lq = (lane & 3) ^ ((4 - ((lrow >> 2) & 3)) & 3);
    int b_off[3];                                     // element offsets (< 2^31: checked by the launcher)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        int q = wave + 8 * p, plane = q / 12, rc = q - 12 * plane;
        if constexpr (CG_PINGPONG) {      // this group's column half = chunks 6 wn .. 6 wn + 5 of each plane: twelve chunks on four waves
            q = (wave & 3) + 4 * p; plane = q / 6; rc = 6 * wn + (q - 6 * plane);
        }
        const int n = n0 + chan_of_pos(rc * 16 + lrow);               // permuted weight rows: see tile_epilogue
        b_off[p] = (int)((int64_t)plane * d.b_lo + (int64_t)n * d.ldb + lq * 8);
    }
    bool in_loop = false;
    auto issue_w = [&](int64_t koff, int stage) {
        if ((CG_ABLATE & 16) && in_loop) return;
        if constexpr (CG_PINGPONG) {
            unsigned char* dstp = smem + W0 + stage * WSTAGE;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int q = (wave & 3) + 4 * p, plane = q / 6, rc = 6 * wn + (q - 6 * plane);
                int bo = b_off[p];
                asm volatile("" : "+v"(bo));
                __builtin_amdgcn_global_load_lds((gptr_t)(Bb + koff + bo), (lptr_t)(dstp + plane * WPLANE + rc * 1024), 16, 0, 0);
            }
            return;
        }
        unsigned char* dst = smem + W0 + stage * WSTAGE + wave * 1024;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            int bo = b_off[p];
            asm volatile("" : "+v"(bo));
            __builtin_amdgcn_global_load_lds((gptr_t)(Bb + koff + bo), (lptr_t)(dst + p * 8192), 16, 0, 0);
        }
    };
    // ---- fragment addressing ----
    const int frow = lane & 15, kg = lane >> 4;
    // patch slot of this lane's output pixel of m-tile i (centre tap) = sb0 + soff[i], its row in a dense raw tile = rb0 + 16 i:
    // one VGPR each, the per-tile parts are wave-uniform (m-tile i starts 16 i pixels further: (16 i / W) rows down, 16 i % W along;
    // no wrap inside a row for W in {16, 32, 64} with 32- or 64-pixel wave tiles)
    const int ml0 = wm * (BM / WM) + frow;
    const int sb0 = (ml0 / W + 1) * PW + (ml0 % W + 1), rb0 = ml0;
    int soff[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) soff[i] = ((i * 16) / W) * PW + ((i * 16) % W);
    const unsigned bfrag = lds0 + W0 + (unsigned)((wn * (BN / 2) + frow) * 64 + (((kg ^ ((4 - ((frow >> 2) & 3)) & 3))) << 4));

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one k-step: A fragments from `abase` (a patch buffer or a dense tile, both planes PPLANE apart) at slots sb[i] + shift,
    // weight fragments from stage `ws`; hi*lo + lo*hi + hi*hi per tile, reads in order of first use (the ring kernel's plain loop)
    auto mma_step = [&](unsigned abase, bool dense, int shift, int ws, auto&& hook, auto&& after) {
        unsigned aa[TM];
        int s0 = dense ? rb0 : sb0;
        asm volatile("" : "+v"(s0));                   // recomputed per step on purpose: hoisted, the 9 x TM addresses spill
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int sl = s0 + (dense ? 16 * i : soff[i]) + shift;
            aa[i] = abase + (unsigned)(sl << 6) + (unsigned)((kg ^ (((sl >> 2) & 1) << 1)) << 4);
        }
        unsigned sbb = bfrag;
        asm volatile("" : "+v"(sbb));
        sbb += (unsigned)(ws * WSTAGE);
        bf16x8 fa[2][TM], fb[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) fb[0][p] = lds_read128(sbb + p * WPLANE);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if constexpr (CG_ABLATE & 8) { if (i == 0) fa[p][0] = lds_read128(aa[0] + p * PPLANE); else fa[p][i] = fa[p][0]; }
                else fa[p][i] = lds_read128(aa[i] + p * PPLANE);
            }
        if constexpr (CG_PINGPONG && CG_PP_K == 0) {      // the other group's boundary BEFORE this step's first MFMA: the fragment reads just issued stay in
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 + 2 * TM) : "memory");      // flight across it, this wave's patch / tile writes (older) have landed
            __builtin_amdgcn_s_barrier();
        }
        // weight slab j: its fragment was fetched one slab ahead; `hook` runs at the head of the LAST slab (every fragment of the step is
        // in registers: CG_MIDBAR's barrier), `after(j)` behind slab j's MFMAs (unused in the shipped forms: a hook for experiments)
        static_for<0, TN>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
#ifdef CG_FAIRPRIO
            // (A/B, r06) least progress first: a wave's priority falls as it works through the step's weight slabs (2, 2, 1, 1, 0, 0), so the wave
            // that is behind wins the MFMA arbitration -- age alone lets the older wave of a SIMD finish ~500 cycles early and leaves the younger
            // one to finish alone, at the ~60 % a lone wave reaches (tools/cg_prof.py)
            __builtin_amdgcn_s_setprio((TN - 1 - j) >> 1);
#endif
            if constexpr (CG_PINGPONG && j == CG_PP_K && CG_PP_K > 0) {      // the OTHER group's step boundary: this slab's weight fragment is the only read in flight
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (j + 1 < TN) {
#pragma unroll
                for (int p = 0; p < 2; ++p) fb[(j + 1) & 1][p] = lds_read128(sbb + p * WPLANE + (j + 1) * 16 * 64);
                if constexpr (j > 0) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                hook();
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (CG_ABLATE & 4) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fb[j & 1][0]), "v"(fb[j & 1][1]), "v"(fa[0][i]), "v"(fa[1][i]));
            } else if constexpr (j == 0) {
                // first slab: the MFMAs of pixel slabs i0 .. i0 + GI - 1 start as soon as THEIR fragments are back (outstanding after them: the
                // later slabs' + the prefetched weight fragment).  (r06) inside a group pass-major -- hi*lo of every slab, then lo*hi, then
                // hi*hi -- so that no MFMA follows one on its own accumulator (igemm_shared.h FRIDO_SLAB0; 0: groups of one = r04's order)
                constexpr int GI = FRIDO_SLAB0 == 0 ? 1 : (FRIDO_SLAB0 == 2 ? TM : 2);
                static_for<0, TM / GI>([&](auto gc) {
                    constexpr int i0 = decltype(gc)::value * GI;
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (TM - GI - i0) + 2) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ii = 0; ii < GI; ++ii) acc[i0 + ii][0] = mfma_op<2>(fb[0][0], fa[1][i0 + ii], acc[i0 + ii][0]);      // weights first: C^T tiles (see tile_epilogue)
#pragma unroll
                    for (int ii = 0; ii < GI; ++ii) acc[i0 + ii][0] = mfma_op<2>(fb[0][1], fa[0][i0 + ii], acc[i0 + ii][0]);
#pragma unroll
                    for (int ii = 0; ii < GI; ++ii) acc[i0 + ii][0] = mfma_op<2>(fb[0][0], fa[0][i0 + ii], acc[i0 + ii][0]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else if constexpr (FRIDO_SLAB0 != 0) {      // pass-major: a dependent MFMA always has TM - 1 independent ones in front
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = mfma_op<2>(fb[j & 1][0], fa[1][i], acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = mfma_op<2>(fb[j & 1][1], fa[0][i], acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = mfma_op<2>(fb[j & 1][0], fa[0][i], acc[i][j]);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i][j] = mfma_op<2>(fb[j & 1][0], fa[1][i], acc[i][j]);
                    acc[i][j] = mfma_op<2>(fb[j & 1][1], fa[0][i], acc[i][j]);
                    acc[i][j] = mfma_op<2>(fb[j & 1][0], fa[0][i], acc[i][j]);
                }
            }
            after(jc);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto no_hook = [] {};
    auto no_after = [](auto) {};

    // ---- prologue: weights of step 0 (mid-step barrier form: and of step 1) in flight, chunk 0 staged without overlap ----
    issue_w((int64_t)cb * 32, cb & 1);
    if constexpr (CG_MIDBAR) issue_w((int64_t)C + cb * 32, (cb + 1) & 1);      // tap 1 of the first chunk (a slice has >= 9 steps)
    {
        CgUnit pu[NR];                 // every round's loads in flight before the first conversion (the accumulators are not live yet)
        static_for<0, NR>([&](auto rc) { stage_load(rc, cb, pu[decltype(rc)::value]); });
        static_for<0, NR>([&](auto rc) { stage_write(rc, std::integral_constant<int, 0>{}, cb, pu[decltype(rc)::value]); });
    }

    in_loop = true;
#if CG_PROF
    cgp_prev = CGP_NOW();
    cgp[5] = cgp_prev - cgp_t0;
    const unsigned cgp_loop0 = cgp_prev;
#endif
#ifdef CG_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);     // (A/B) static priority for the second-dispatched half: MI355X_MICROARCH.md "Two waves per SIMD" item 4
#endif
    CgUnit su;                        // the staging round in flight (loaded at an even tap, written at the next odd one)
    CgUnit ru[RU];                    // the raw tile in flight
    const bool early = wave < 4;      // waves w and w + 4 share a SIMD: the first converts before its MFMA block, the second after

    // MODE 0: a conv chunk follows; 1: last chunk, nothing follows; 2: last chunk, raw tiles follow.
    // Staging schedule of chunk c + 1 inside chunk c (MODE 0), one unit register set per wave, TWO steps between a round's loads and
    // its conversion:  waves 0-3 ("early"): load round r at tap 2r, convert it at tap 2r + 2;
    //                  waves 4-7:           load round r at tap 2r + 1, convert it at tap 2r + 3 (the last round of four: at tap 8).
    // Waves w and w + 4 share a SIMD: at any tap only one of the two converts (VALU) while the other goes straight to its MFMAs.
    auto chunk = [&](auto modec, int c) {
        constexpr int MODE = decltype(modec)::value;
        static_for<0, 9>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            // loads issued in the PREVIOUS step after its weight DMA may stay in flight (loads retire in order)
            constexpr int PT = T == 0 ? 8 : T - 1;                       // previous tap
            if constexpr (T == 0) {
                wait_vmcnt<0>();                                         // previous step = tap 8 (no staging loads) or the prologue
            } else if constexpr (MODE == 0) {
                constexpr int E = (PT % 2 == 0 && PT / 2 < NR) ? NL : 0, L = (PT % 2 == 1 && (PT - 1) / 2 < NR) ? NL : 0;
                if constexpr (E == L) wait_vmcnt<E>();
                else if (early) wait_vmcnt<E>();
                else wait_vmcnt<L>();
            } else if constexpr (MODE == 2) {
                // (an inline-asm load must never be left unconsumed: its registers would be re-allocated while it is in flight --
                //  so raw tile 1 is only fetched when it exists, and the wait count follows)
                if constexpr (PT == 0) wait_vmcnt<2 * RU>();
                else if constexpr (PT == 6) { if (nraw > 1) wait_vmcnt<2 * RU>(); else wait_vmcnt<0>(); }
                else wait_vmcnt<0>();
            } else {
                wait_vmcnt<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's patch / tile writes have landed
#if CG_PROF
            cgp_a = CGP_NOW(); cgp[0] += cgp_a - cgp_prev;
#endif
            if constexpr (!(CG_ABLATE & 128)) __builtin_amdgcn_s_barrier();
#if CG_PROF
            cgp_prev = CGP_NOW(); cgp[1] += cgp_prev - cgp_a;
#endif
            // (a) weights of the next step
            if constexpr (T < 8) issue_w((int64_t)(T + 1) * C + c * 32, (c + T + 1) & 1);
            else if constexpr (MODE == 0) issue_w((int64_t)(c + 1) * 32, (c + 1) & 1);                 // tap 0 of chunk c + 1: stage (9 (c + 1)) & 1
            else if constexpr (MODE == 2) issue_w((int64_t)9 * C + rb * 32, (c + 1) & 1);               // this slice's first raw tile
            // (dependency only: the unit's loads completed at this step's top wait -- except the late waves' last round and the raw tile, which
            //  the count behind the three DMA pieces just issued really waits for)
            using Y3 = std::integral_constant<int, 3>;
            using YL = Y3;
            if constexpr (MODE == 0) {
                constexpr int EC = (T % 2 == 0 && T >= 2 && T / 2 - 1 < NR) ? T / 2 - 1 : -1;          // round the early waves convert
                constexpr int EL = (T % 2 == 0 && T / 2 < NR) ? T / 2 : -1;                            // ... and load
                constexpr int LC = (T % 2 == 1 && T >= 3 && (T - 3) / 2 < NR) ? (T - 3) / 2 : ((T == 8 && NR == 4) ? 3 : -1);
                constexpr int LL = (T % 2 == 1 && (T - 1) / 2 < NR) ? (T - 1) / 2 : -1;
                if (early) {
                    if constexpr (EC >= 0) stage_write(std::integral_constant<int, EC < 0 ? 0 : EC>{}, Y3{}, c + 1, su);
                    if constexpr (EL >= 0) stage_load(std::integral_constant<int, EL < 0 ? 0 : EL>{}, c + 1, su);
                } else {
                    if constexpr (LC >= 0) stage_write(std::integral_constant<int, LC < 0 ? 0 : LC>{}, YL{}, c + 1, su);
                    if constexpr (LL >= 0) stage_load(std::integral_constant<int, LL < 0 ? 0 : LL>{}, c + 1, su);
                }
            }
            if constexpr (MODE == 2 && T == 0) raw_load(0, ru);
            if constexpr (MODE == 2 && T == 6) { if (nraw > 1) raw_load(1, ru); }
            constexpr bool RAWW = MODE == 2 && T == 1;
            if (early) {
                if constexpr (RAWW) raw_write(Y3{}, (c + 1) & 1, ru);
            }
#if CG_PROF
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            cgp_a = CGP_NOW(); cgp[2] += cgp_a - cgp_prev;
#endif
            mma_step(lds0 + (unsigned)((c & 1) * PBUF), false, (T / 3 - 1) * PW + (T % 3 - 1), (c + T) & 1, no_hook, no_after);
#if CG_PROF
            cgp_prev = CGP_NOW(); cgp[3] += cgp_prev - cgp_a;
#endif
            if (!early) {
                if constexpr (RAWW) raw_write(YL{}, (c + 1) & 1, ru);
            }
        });
    };
    // ---- (r06) MID-STEP BARRIER form.  r04 met at the TOP of every step (54 - 66 times per launch): after the release a wave first issues
    // its weight DMA, converts a staging round, computes fragment addresses and waits out an LDS round trip -- with both waves of every SIMD
    // at the same point, the matrix pipe idles for those few hundred cycles of every ~2700-cycle step.  Here step g's barrier sits at the
    // head of its LAST weight slab: all of the wave's fragments of the step are in registers (its reads of weight stage g & 1 are over),
    // TM MFMA triples are ready to issue right after the release, and the top-of-step work of step g + 1 is no longer aligned across waves
    // (the older wave of a SIMD runs ahead into it while the younger one still issues MFMAs).  Ring protocol (2 weight stages, as before):
    //   hook of step g:  lgkmcnt(0) [mma_step]; my share of W(g + 1) has landed (vmcnt: only loads issued at THIS step's top are younger);
    //                    s_barrier  -> W(g + 1) and every patch / raw-tile write before it are published, stage g & 1 is free;
    //                    issue W(g + 2) into stage g & 1 (one full step ahead, like r04's tap-ahead DMA).
    // Patch buffers: a round of chunk c + 1 is written at a step's top into the buffer chunk c - 1 read; all of those reads (issued at a
    // step's top) lie before that chunk's tap-8 hook.  Raw tile s + 1 is written at the top of raw step s into the buffer step s - 1 read.
    // A staged unit loaded at the top of step g is converted at the top of step g + 2 (g + 1 for the late waves' last round): the hook of
    // step g + 1 has drained it (a wave never loads at two consecutive steps), so the conversion's wait (vmcnt(3): the DMA just issued) is
    // a dependency only, as in r04.
    if constexpr (CG_MIDBAR) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // chunk 0's patch written, both weight stages landed
        __builtin_amdgcn_s_barrier();
        using Y3 = std::integral_constant<int, 3>;
        auto chunk_mb = [&](auto modec, int c) {
            constexpr int MODE = decltype(modec)::value;
            static_for<0, 9>([&](auto tc) {
                constexpr int T = decltype(tc)::value;
                // ---- top of the step: staging of chunk c + 1 (MODE 0) / the raw tiles (MODE 2); no barrier ----
                constexpr int EL = (MODE == 0 && T % 2 == 0 && T / 2 < NR) ? T / 2 : -1;                       // round the early waves load
                constexpr int LL = (MODE == 0 && T % 2 == 1 && (T - 1) / 2 < NR) ? (T - 1) / 2 : -1;           // ... the late waves
                if constexpr (MODE == 0) {
                    constexpr int EC = (T % 2 == 0 && T >= 2 && T / 2 - 1 < NR) ? T / 2 - 1 : -1;              // round the early waves convert
                    constexpr int LC = (T % 2 == 1 && T >= 3 && (T - 3) / 2 < NR) ? (T - 3) / 2 : ((T == 8 && NR == 4) ? 3 : -1);
                    if (early) {
                        if constexpr (EC >= 0) stage_write(std::integral_constant<int, EC < 0 ? 0 : EC>{}, Y3{}, c + 1, su);
                        if constexpr (EL >= 0) stage_load(std::integral_constant<int, EL < 0 ? 0 : EL>{}, c + 1, su);
                    } else {
                        if constexpr (LC >= 0) stage_write(std::integral_constant<int, LC < 0 ? 0 : LC>{}, Y3{}, c + 1, su);
                        if constexpr (LL >= 0) stage_load(std::integral_constant<int, LL < 0 ? 0 : LL>{}, c + 1, su);
                    }
                }
                if constexpr (MODE == 2 && T == 0) raw_load(0, ru);
                if constexpr (MODE == 2 && T == 1) raw_write(Y3{}, (c + 1) & 1, ru);            // tile 0 -> the buffer chunk c - 1 read
                if constexpr (MODE == 2 && T == 6) { if (nraw > 1) raw_load(1, ru); }
                mma_step(lds0 + (unsigned)((c & 1) * PBUF), false, (T / 3 - 1) * PW + (T % 3 - 1), (c + T) & 1, [&] {
                    // my share of the NEXT step's weights has landed: only the loads issued at this step's top may stay in flight
                    if constexpr (MODE == 0) {
                        constexpr int E = EL >= 0 ? NL : 0, L = LL >= 0 ? NL : 0;
                        if constexpr (E == L) wait_vmcnt<E>();
                        else if (early) wait_vmcnt<E>();
                        else wait_vmcnt<L>();
                    } else if constexpr (MODE == 2 && T == 0) {
                        wait_vmcnt<2 * RU>();
                    } else if constexpr (MODE == 2 && T == 6) {
                        if (nraw > 1) wait_vmcnt<2 * RU>(); else wait_vmcnt<0>();
                    } else {
                        wait_vmcnt<0>();
                    }
                    if constexpr (!(CG_ABLATE & 128)) __builtin_amdgcn_s_barrier();
                    // refill the stage this step has finished reading with the weights of step + 2
                    if constexpr (T + 2 <= 8) issue_w((int64_t)(T + 2) * C + c * 32, (c + T) & 1);
                    else if constexpr (MODE == 0) issue_w((int64_t)(T + 2 - 9) * C + (c + 1) * 32, (c + T) & 1);
                    else if constexpr (MODE == 2) { if (T + 2 - 9 < nraw) issue_w((int64_t)9 * C + (rb + T + 2 - 9) * 32, (c + T) & 1); }
                }, no_after);
            });
        };
        for (int c = cb; c + 1 < nc; ++c) chunk_mb(std::integral_constant<int, 0>{}, c);
        if (HASRAW && nraw > 0) {
            chunk_mb(std::integral_constant<int, 2>{}, nc - 1);
            // raw tile s: weights in stage (nc + s) & 1, tile in patch buffer (nc + s) & 1
            for (int s = 0; s < nraw; ++s) {
                const bool nxt = s + 1 < nraw, nxt2 = s + 2 < nraw;
                if (nxt) {
                    raw_write(Y3{}, (nc + s + 1) & 1, ru);      // tile s + 1 -> the buffer step s - 1 read (loaded at the top of step s - 1 / tap 6)
                    if (nxt2) raw_load(s + 2, ru);
                }
                mma_step(lds0 + (unsigned)(((nc + s) & 1) * PBUF), true, 0, (nc + s) & 1, [&] {
                    if (nxt && nxt2) wait_vmcnt<2 * RU>(); else wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                    if (nxt2) issue_w((int64_t)9 * C + (rb + s + 2) * 32, (nc + s) & 1);
                }, no_after);
            }
        } else {
            chunk_mb(std::integral_constant<int, 1>{}, nc - 1);
        }
    } else {
    if constexpr (CG_PINGPONG) {
        static_assert(!CG_MIDBAR && CG_PP_K >= 0 && CG_PP_K < TN, "ping-pong: r04 loop form");
        if (!early) {      // group Y starts half a step late: it sits out barrier A_0 (group X's first top)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    for (int c = cb; c + 1 < nc; ++c) chunk(std::integral_constant<int, 0>{}, c);
    if (HASRAW && nraw > 0) {
        chunk(std::integral_constant<int, 2>{}, nc - 1);
        // raw tile s: weights in stage (9 nc + s) & 1 = (nc + s) & 1, tile in patch buffer (nc + s) & 1
        for (int s = 0; s < nraw; ++s) {
            // the previous step's loads after its weight DMA: raw tile s + 1 (fetched at step s - 1, when it exists)
            if (s > 0 && s + 1 < nraw) wait_vmcnt<2 * RU>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + 1 < nraw) {
                issue_w((int64_t)9 * C + (rb + s + 1) * 32, (nc + s + 1) & 1);
                raw_write(std::integral_constant<int, 3>{}, (nc + s + 1) & 1, ru);      // tile s + 1 -> the buffer step s - 1 read
                if (s + 2 < nraw) raw_load(s + 2, ru);
            }
            mma_step(lds0 + (unsigned)(((nc + s) & 1) * PBUF), true, 0, (nc + s) & 1, no_hook, no_after);
        }
    } else {
        chunk(std::integral_constant<int, 1>{}, nc - 1);
    }
    if constexpr (CG_PINGPONG) {
        if (early) __builtin_amdgcn_s_barrier();      // barrier A_n: the mid-step barrier of group Y's last step
    }
    }      // (!CG_MIDBAR)
    wait_vmcnt<0>();
#if CG_PROF
    const unsigned cgp_loop1 = CGP_NOW();
    cgp[4] = cgp_loop1 - cgp_loop0;
#endif
    status_raise(sat);
    if constexpr (CG_ABLATE & 32) {       // (timing only) no epilogue: keep the accumulators live, store nothing
        float sink = 0.f;
        for (auto& ai : acc) for (auto& aj : ai) sink += aj[0] + aj[1] + aj[2] + aj[3];
        if (sink == 1.2345e-30f) d.out_f32[0] = sink;
        return;
    }
    tile_epilogue<BM, BN, 2, WM, false>(d, acc, smem, m0, n0, CG_PINGPONG ? wm * 2 + wn : wave, lane, 0, 0, kz);      // (the epilogue's wave id = 2 wm + wn)
#if CG_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores have left the wave (not: reached memory)
    cgp[6] = CGP_NOW() - cgp_loop1;
    if (lane == 0 && bid < 2048 && kz == 0) {
        unsigned* o = g_cg_prof + (bid * 8 + wave) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = cgp[i];
    }
#endif
}

#if CG_PROF
extern "C" int frido_cg_prof_read(unsigned* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_cg_prof), (size_t)n * sizeof(unsigned), 0, hipMemcpyDeviceToHost);
}
#endif

// eligibility of the fused GroupNorm + conv kernel (tile ids 20 = BM 256, 21 = BM 128)
bool convgn_ok(const FridoGemm& d, int bm) {
    if (!d.conv || d.nsplit != 2 || d.batch != 1 || !d.gn_x1 || !d.gn_partials || !d.gn_weight || !d.gn_bias) return false;
    if (d.splitk > 1 && (!d.ws || d.sk_mode != 0 || d.gn_part || d.splitk > ((d.gn_C1 + d.gn_C2) >> 5))) return false;
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.padx != 1 || d.up_shift || d.dn_shift || d.up2_phase) return false;
    if (d.Ho != d.Hs || d.Wo != d.Ws || d.Hl != d.Hs || d.Wl != d.Ws) return false;
    const int W = d.Ws, HW = d.Hs * d.Ws, C = d.gn_C1 + d.gn_C2;
    if (W < 16 || W > 64 || (W & (W - 1)) || bm % W) return false;
    if (d.M % bm || HW % bm || d.N % 192) return false;
    const int R = bm / W;
    if ((R + 2) * (W + 2) > (bm == 256 ? 396 : 204)) return false;
    if ((d.gn_C1 & 31) || (d.gn_C2 & 31) || (d.gn_C2 && !d.gn_x2) || C != d.Cin || C > 960 || d.K != 9 * d.Cin) return false;
    // (<= 32 groups: the statistics prologue is laid out for them, like gn_apply_kernel's)
    if (d.gn_groups <= 0 || d.gn_groups > 32 || C % d.gn_groups || d.gn_nsplit_px < 1 || d.gn_nsplit_px > 64) return false;
    if ((d.gn_gamma == nullptr) != (d.gn_beta == nullptr)) return false;
    if (d.K2) {
        if (!d.raw_x1 || (d.raw_C1 & 31) || (d.raw_C2 & 31) || (d.raw_C2 && !d.raw_x2) || d.K2 != d.raw_C1 + d.raw_C2) return false;
    }
    if (d.geglu || d.out_u8 || (d.flags & 4)) return false;
    if (2 * d.b_lo + (int64_t)d.N * d.ldb >= (1ll << 31)) return false;      // 32-bit weight offsets in the kernel
    {   // (r06) the staging loads address every staged tensor as base (SGPR pair) + 32-bit BYTE offset: rows x widest row < 4 GiB
        int cw = C;                                                            // gamma / beta rows are C wide, x1 / x2 rows narrower
        if (d.raw_C1 > cw) cw = d.raw_C1;
        if (d.raw_C2 > cw) cw = d.raw_C2;
        if ((int64_t)d.M * cw >= (1ll << 30)) return false;
    }
    return true;
}

template <int BM>
int launch_convgn_bm(const FridoGemm& d, hipStream_t s) {
    const int tiles = (d.M / BM) * (d.N / 192), sk = d.splitk > 1 ? d.splitk : 1;
    constexpr int smem = CGeo<BM>::SMEM;
    const bool sp = d.gn_gamma != nullptr, raw = d.K2 > 0;
    if (sp && raw) hipLaunchKernelGGL((conv3x3_gn_kernel<BM, true, true>), dim3(tiles, 1, sk), dim3(512), smem, s, d);
    else if (sp) hipLaunchKernelGGL((conv3x3_gn_kernel<BM, true, false>), dim3(tiles, 1, sk), dim3(512), smem, s, d);
    else if (raw) hipLaunchKernelGGL((conv3x3_gn_kernel<BM, false, true>), dim3(tiles, 1, sk), dim3(512), smem, s, d);
    else hipLaunchKernelGGL((conv3x3_gn_kernel<BM, false, false>), dim3(tiles, 1, sk), dim3(512), smem, s, d);
    return frido_check_launch("conv3x3_gn");
}


// =====================================================================================================================
// conv3x3_gn_tiny_kernel (r06, FridoGemm tile 40): the denoiser's OUTPUT HEAD  eps = conv3x3( SiLU( GroupNorm32(h) ) )  with N = 3 or 4
// output channels (pyunet.py:775-803: `out` = normalization, SiLU, conv_nd(model_channels, out_channels, 3, padding=1); one per stage with
// the split head).  On the MFMA family this conv was the worst launch of the forward: a 64-column tile computes 61 dead columns, the
// normalised operand (50 MB of hi / lo planes at 64^2 x 192) is written by gn_apply and pulled through L2 -> LDS nine times, and the K
// walk needs split-K + a reduction launch -- 21 + 77 us per forward at 8.8 TFLOP/s.  Here the whole head is ONE launch on the f32 VALU:
//   * tile = 256 output pixels (R = 256 / W image rows); per 32-channel chunk the (R + 2) x (W + 2) patch is read ONCE from the f32 stream,
//     normalised (gn_apply_kernel's expressions: x * sc + sh, SiLU) and parked in LDS as f32 (zero halo = the conv pads the ACTIVATION);
//     the next chunk's global loads are in flight while the current one is multiplied;
//   * a thread owns one pixel and N accumulators: 9 taps x 32 channels x N fmaf per chunk; activations come from LDS (slot stride 36 floats:
//     conflict-free ds_read_b128 for consecutive pixels), WEIGHTS FROM THE SCALAR CACHE (s_load_dwordx8, wave-uniform addresses): they cost
//     no LDS and no VALU, an FMA takes them as its SGPR operand;
//   * plain f32 accumulation (exact products, 24-bit sums): at least as accurate as the two-plane path it replaces.
// 65536 x 192 -> 3: 340 MFLOP, 50 MB read once -> HBM / VALU balanced at ~15 us.
template <int NO>
__global__ __launch_bounds__(512) void conv3x3_gn_tiny_kernel(const FridoGemm d, const float* __restrict__ wq) {      // wq = d.w_f32
    // 256 pixels per workgroup on 512 threads: thread t and t + 256 own the SAME pixel and split a chunk's 32 channels 16 / 16 -- the
    // launch is only 65536 pixels = 1024 waves of one-pixel-per-lane work, one wave per SIMD; the channel split makes it two
    constexpr int TP = 256, NT = 512, SS = 36, MAXSLOTS = 396, MAXC = 960;      // SS: floats per patch slot (32 channels + 4 pad)
    constexpr int NRD = (MAXSLOTS * 4 + NT - 1) / NT;                  // staging rounds per chunk (8-channel units per thread): 4
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const patch = reinterpret_cast<float*>(smem);               // [MAXSLOTS][SS]
    float* const tab_sc = patch + MAXSLOTS * SS;
    float* const tab_sh = tab_sc + MAXC;
    float* const wl = tab_sh + MAXC;                                    // this chunk's weights [9 taps][128] (NO x 32 used)
    const int t = threadIdx.x;
    const int W = d.Ws, H = d.Hs, HW = H * W;
    const int R = TP / W, PW = W + 2, PS = (R + 2) * PW;
    const int m0 = (int)blockIdx.x * TP;
    const int img = m0 / HW, y0 = (m0 - img * HW) / W;
    const int C = d.gn_C1;
    const float* __restrict__ x1 = d.gn_x1;

    // ---- GroupNorm statistics -> per-channel scale / shift (conv3x3_gn_kernel's prologue: same order, same expressions) ----
    {
        float* s_mean = patch;
        float* s_rstd = s_mean + 64;
        const int cpg = C / d.gn_groups;
        double* s_part = reinterpret_cast<double*>(smem + 1024);
        if (t < 256) {
            const int g = t & 31, l8 = t >> 5;
            double s = 0.0, q = 0.0;
            if (g < d.gn_groups) {
                for (int k = 0; k < 8; ++k) {
                    const int sp = l8 + 8 * k;
                    if (sp < d.gn_nsplit_px) {
                        const double2 pv = *reinterpret_cast<const double2*>(d.gn_partials + (((int64_t)img * d.gn_nsplit_px + sp) * d.gn_groups + g) * 2);
                        s += pv.x;
                        q += pv.y;
                    }
                }
                s_part[(l8 * 32 + g) * 2] = s;
                s_part[(l8 * 32 + g) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (t < d.gn_groups) {
            double s = 0.0, q = 0.0;
            for (int l = 0; l < 8; ++l) { s += s_part[(l * 32 + t) * 2]; q += s_part[(l * 32 + t) * 2 + 1]; }
            const double n = (double)HW * cpg;
            const double mean = s / n;
            double var = q / n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mean[t] = (float)mean;
            s_rstd[t] = (float)(1.0 / sqrt(var + (double)d.gn_eps));
            if (blockIdx.x == 0) status_raise(false, stat_bad(s_mean[t], s_rstd[t]));
        }
        __syncthreads();
        for (int c = t; c < C; c += NT) {
            const int g = c / cpg;
            const float sc = s_rstd[g] * d.gn_weight[c];
            tab_sc[c] = sc;
            tab_sh[c] = d.gn_bias[c] - s_mean[g] * sc;
        }
        __syncthreads();
    }

    // ---- staging geometry: unit (round r) = patch slot r * 128 + (t >> 2), channels (t & 3) * 8 of the chunk ----
    const int cu = (t & 3) * 8;
    int pix[NRD];                                       // global pixel, -1: halo (zeros), -2: beyond the patch
#pragma unroll
    for (int r = 0; r < NRD; ++r) {
        const int slot = r * 128 + (t >> 2);
        const int pr = slot / PW, px = slot - pr * PW;
        const int y = y0 + pr - 1, x = px - 1;
        pix[r] = slot >= PS ? -2 : ((y >= 0 && y < H && x >= 0 && x < W) ? img * HW + y * W + x : -1);
    }
    const bool do_silu = d.gn_act == FRIDO_ACT_SILU;
    float4 pre[NRD][2];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int r = 0; r < NRD; ++r)
            if (pix[r] >= 0) {
                const float* src = x1 + (int64_t)pix[r] * C + c * 32 + cu;
                pre[r][0] = *reinterpret_cast<const float4*>(src);
                pre[r][1] = *reinterpret_cast<const float4*>(src + 4);
            }
    };
    auto park_chunk = [&](int c) {
        for (int i = t; i < 9 * NO * 32; i += NT) wl[(i / (NO * 32)) * 128 + i % (NO * 32)] = wq[(int64_t)c * (9 * NO * 32) + i];
        const float4 a0 = *reinterpret_cast<const float4*>(tab_sc + c * 32 + cu), a1 = *reinterpret_cast<const float4*>(tab_sc + c * 32 + cu + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(tab_sh + c * 32 + cu), b1 = *reinterpret_cast<const float4*>(tab_sh + c * 32 + cu + 4);
        const float sc[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, sh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int r = 0; r < NRD; ++r) {
            if (pix[r] == -2) continue;
            float y[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (pix[r] >= 0) {
                const float xin[8] = {pre[r][0].x, pre[r][0].y, pre[r][0].z, pre[r][0].w, pre[r][1].x, pre[r][1].y, pre[r][1].z, pre[r][1].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    y[e] = fmaf(xin[e], sc[e], sh[e]);
                    if (do_silu) y[e] = silu_f(y[e]);
                }
            }
            float* dst = patch + (r * 128 + (t >> 2)) * SS + cu;
            *reinterpret_cast<float4*>(dst) = make_float4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
    };

    // ---- this thread's output pixel: centre slot in the patch ----
    const int pxl = t & 255, half = t >> 8;               // this thread's pixel of the tile and its 16-channel half of every chunk
    const int prow = pxl / W, pcol = pxl - prow * W;
    const int sc0 = (prow + 1) * PW + (pcol + 1);
    float acc[NO][2];                                    // two independent chains per output: the FMA latency, not its rate, paced one chain
#pragma unroll
    for (int n = 0; n < NO; ++n) acc[n][0] = acc[n][1] = 0.f;
    const int nch = C >> 5;
    load_chunk(0);
    park_chunk(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) load_chunk(c + 1);              // in flight under this chunk's multiply
        // One tap = NO x 16 weights per channel half, the same for every pixel: lane l < NO * 16 of a wave fetches weight (n = l / 16,
        // channel 16 half + l % 16) of the tap from LDS (one conflict-free ds_read_b32) and every FMA takes its weight out of that register
        // with v_readlane_b32 -> SGPR operand.  (Tried first: s_load_dwordx16 from global -- the 20-KB weight set thrashes the scalar cache,
        // 97 us per launch; 24 broadcast ds_read_b128 per tap would make the launch LDS-bound.)  NOT unrolled over taps: registers.
        const int lane = t & 63;
        const int wsel = (lane >> 4) * 32 + half * 16 + (lane & 15);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const float w0 = wl[tap * 128 + (wsel & 127)];
            const float* ap = patch + (sc0 + (tap / 3 - 1) * PW + (tap % 3 - 1)) * SS + half * 16;
            float a[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(ap + 4 * q);
                a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int n = 0; n < NO; ++n) {
                    const float wv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w0), n * 16 + e));
                    acc[n][e & 1] = fmaf(a[e], wv, acc[n][e & 1]);
                }
        }
        __syncthreads();                                 // every thread is done reading the patch
        if (c + 1 < nch) {
            park_chunk(c + 1);
            __syncthreads();
        }
    }
    // the two channel halves of a pixel -> one: the upper half parks its sums in LDS (the patch is dead), the lower half adds and stores
    float* red = patch + pxl * 4;
    if (half == 1) {
#pragma unroll
        for (int n = 0; n < NO; ++n) red[n] = acc[n][0] + acc[n][1];
    }
    __syncthreads();
    const int m = m0 + pxl;
    if (half == 0 && m < d.M) {
#pragma unroll
        for (int n = 0; n < NO; ++n)
            if (n < d.N) d.out_f32[(int64_t)m * d.ldo + n] = d.alpha * ((acc[n][0] + acc[n][1]) + red[n]) + (d.bias ? d.bias[n] : 0.f);
    }
}

bool convgn_tiny_ok(const FridoGemm& d) {
    if (!d.conv || d.nsplit != 2 || d.batch != 1 || !d.gn_x1 || d.gn_x2 || d.gn_C2 || !d.gn_partials || !d.gn_weight || !d.gn_bias || !d.w_f32) return false;
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.padx != 1 || d.up_shift || d.dn_shift || d.up2_phase || d.splitk > 1) return false;
    if (d.Ho != d.Hs || d.Wo != d.Ws || d.Hl != d.Hs || d.Wl != d.Ws) return false;
    const int W = d.Ws, HW = d.Hs * d.Ws, C = d.gn_C1;
    if (W < 16 || W > 64 || (W & (W - 1)) || HW % 256 || d.M % 256 || (256 / W + 2) * (W + 2) > 396) return false;
    if ((C & 31) || C != d.Cin || C > 960 || d.K != 9 * d.Cin || d.K2) return false;
    if (d.gn_groups <= 0 || d.gn_groups > 32 || C % d.gn_groups || d.gn_nsplit_px < 1 || d.gn_nsplit_px > 64) return false;
    if (d.gn_gamma || d.gn_beta || (d.N != 3 && d.N != 4)) return false;
    if (!d.out_f32 || d.out_bf16 || d.out_op || d.out_u8 || d.residual || d.rowvec || d.row_bias || d.geglu || d.gn_part || d.act != FRIDO_ACT_NONE || d.ldo < d.N) return false;
    return true;
}
constexpr int CG_TINY_SMEM = (396 * 36 + 2 * 960 + 9 * 128) * 4;

}  // namespace

int frido_launch_convgn_tiny(const FridoGemm& d, hipStream_t s) {
    if (!convgn_tiny_ok(d)) {
        frido_set_error("igemm: tile 40 (fused GroupNorm + 3x3 conv with 3 / 4 output channels) does not apply to this descriptor");
        return FRIDO_EINVAL;
    }
    if (d.N == 3) hipLaunchKernelGGL((conv3x3_gn_tiny_kernel<3>), dim3(d.M / 256), dim3(512), CG_TINY_SMEM, s, d, d.w_f32);
    else hipLaunchKernelGGL((conv3x3_gn_tiny_kernel<4>), dim3(d.M / 256), dim3(512), CG_TINY_SMEM, s, d, d.w_f32);
    return frido_check_launch("conv3x3_gn_tiny");
}

int frido_launch_convgn(const FridoGemm& d, int bm, hipStream_t s) {
    if (!convgn_ok(d, bm)) {
        frido_set_error("igemm: tile 20 / 21 (fused GroupNorm + 3x3 conv) does not apply to this descriptor");
        return FRIDO_EINVAL;
    }
    return bm == 256 ? launch_convgn_bm<256>(d, s) : launch_convgn_bm<128>(d, s);
}

namespace {
template <int BM, bool SP, bool RAW>
int convgn_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_gn_kernel<BM, SP, RAW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               CGeo<BM>::SMEM) == hipSuccess ? 0 : 1;
}
}  // namespace

int frido_convgn_init() {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_gn_tiny_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, CG_TINY_SMEM) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_gn_tiny_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, CG_TINY_SMEM) != hipSuccess)
        return 1;
    return convgn_attr<256, false, false>() | convgn_attr<256, false, true>() | convgn_attr<256, true, false>() | convgn_attr<256, true, true>() |
           convgn_attr<128, false, false>() | convgn_attr<128, false, true>() | convgn_attr<128, true, false>() | convgn_attr<128, true, true>();
}
