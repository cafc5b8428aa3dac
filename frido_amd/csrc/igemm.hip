// Implicit-GEMM on the CDNA4 matrix pipe: C[m][n] = sum_k A[m][k] * B[n][k]  (both K-contiguous bf16).
//
//  * one workgroup = 256 threads = 4 waves (2 x 2) computing a BM x BN tile with
//    v_mfma_f32_16x16x32_bf16; K is walked in BK = 32 steps (one MFMA k-step per LDS stage);
//  * A is either a dense operand matrix or an NHWC image gathered on the fly (3x3 / 1x1 conv with
//    stride, zero padding, nearest x2 up-sampling or 2^k sub-sampling folded into the address);
//  * operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB = 16 rows x 64 B per wave
//    instruction) through a D-deep ring of LDS stages: D-1 k-tiles are in flight while one is consumed;
//    a stage is waited for with a COUNTED s_waitcnt vmcnt(N) and published with one raw s_barrier per
//    k-tile (loads stay in flight across the barrier); fragment ds_reads are inline asm so that hipcc does
//    not fence them with vmcnt(0) (cdna_hip_programming.md §5 "Pipelining across barriers", §5.7);
//  * LDS rows are 64 B (32 bf16); the 16-B slot q of row r lives at slot q ^ ((4 - (r>>2)) & 3): the
//    LDS-DMA image is lane-linear, so the swizzle is applied to the per-lane SOURCE address and to the
//    fragment read address (rule 21); it makes the ds_read_b128 fragment loads conflict-free for the
//    16x16x32 fragment layout (MI355X_MICROARCH.md §LDS lane groups); conv taps that fall into the
//    zero padding read a 16-byte zero page instead;
//  * nsplit == 2 ("bf16x3"): operands carry a second bf16 plane with the rounding residual and each
//    tile product is hi*hi + hi*lo + lo*hi (fp32 accumulate) — ~2^-17 relative error at 3 MFMAs;
//  * fused epilogue: alpha, bias[n], per-row bias, per-row-group vector (timestep embedding), ReLU/SiLU,
//    residual, f32 and/or operand (bf16 hi/lo) output.  Every MFMA takes the WEIGHT fragment as its first operand, so an
//    accumulator tile is C^T (a lane owns consecutive channels of one pixel) and the hot epilogues store straight from
//    registers -- see tile_epilogue;
//  * the BK = 64 bf16 tiles run a software-pipelined loop (fragments of the next k-step prefetched under the MFMAs);
//  * split-K work items (k-slice, tile) are dealt to the XCDs in contiguous ranges (xcd_item).
#include "igemm_shared.h"
#include <atomic>

// convgn.hip: the fused GroupNorm + 3x3 conv kernel (tiles 20 / 21)
int frido_launch_convgn(const FridoGemm& d, int bm, hipStream_t s);
int frido_launch_convgn_tiny(const FridoGemm& d, hipStream_t s);      // tile 40
int frido_convgn_init();

// Timing ablations of the bf16x3 main loop (tools/build_ablate.sh builds SEPARATE libraries with -DFRIDO_ABLATE=<mask>; the shipped
// library is built with 0 and contains none of this): 1 = no DMA refills inside the loop, 2 = no fragment reads, 4 = no barriers /
// vmcnt waits, 8 = no MFMAs.  Results are garbage; only the launch time means anything.
// 1024 (r05, results stay CORRECT): stagger experiment of DESIGN.md section 7 item 5 -- on the two-per-CU (4-wave) tiles the workgroups
// with dispatch ids 256 .. 511 (the second resident slot of the first round, if the dispatcher fills one slot per CU first) start
// FRIDO_STAGGER_US microseconds late, so that from then on one slot's prologue / epilogue runs under the other slot's k-loop.
#ifndef FRIDO_ABLATE
#define FRIDO_ABLATE 0
#endif
#ifndef FRIDO_STAGGER_US
#define FRIDO_STAGGER_US 8
#endif
// Run-time form of the same (for round 6; -DFRIDO_STAGGER_RT=1 builds only, the shipped library is built with 0 and is bit-identical to
// the one without this code): the delay in QUARTER microseconds comes from FridoGemm.flags bits 8..15 (0 = none), the smallest grid it
// applies to from bits 16..23 in units of 64 workgroups (0 = 768 workgroups), which workgroups wait from bits 24..25.
// Python: FRIDO_STAGGER_US (a float) / FRIDO_STAGGER_MIN_WG / FRIDO_STAGGER_MODE (engine.py).
// (FRIDO_STAGGER_RT itself: igemm_shared.h, next to the one-workgroup-per-CU form of the experiment)
// (r06) -DIG_PROF=1 (tools/igemm_prof.py; never in the shipped build): every wave of the two-plane virtual-step loop sums the shader cycles
// (s_memtime) it spends, per k-tile,  [0] waiting for the next stage's DMA in front of the barrier, [1] inside the barrier, [2] in the rest of
// its three virtual steps (MFMAs, fragment reads and their lgkmcnt waits, DMA issue);  [3] = the whole loop, [4] = kernel start -> loop,
// [5] = epilogue, [6] = k-tiles.  Stamps sit where lgkmcnt is 0 anyway.  convgn.hip has the same for the fused kernel (CG_PROF).
#ifndef IG_PROF
#define IG_PROF 0
#endif
#ifndef FRIDO_X3_PIPE_ALL
#define FRIDO_X3_PIPE_ALL 0      // 1: also run the six-n-tile bf16x3 tiles (128 x 192, 64 x 192) on the virtual-k-step loop
#endif

namespace {

#if IG_PROF
__device__ unsigned g_ig_prof[4096 * 8 * 8];       // [workgroup][wave][8]
#define IGP_NOW() ((unsigned)__builtin_readcyclecounter())
#endif
__device__ uint4 g_zero_page[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};

// (LDS / waitcnt helpers, xcd_item, static_for, chan_of_pos, tile_epilogue: igemm_shared.h)
// KG = 2 (r06, "intra-workgroup split-K", tiles 31 / 33 / 34 / 35 / 36): TWO 4-wave groups in one 8-wave workgroup, group g running the
// unmodified 4-wave loop over k-tiles [g * nk / 2, (g + 1) * nk / 2) on its own ring; group 1 hands its accumulators to group 0 through
// LDS and exits, group 0 adds them (acc0 + acc1: one fixed order) and runs the epilogue.  For the SUB-ROUND dense launches of the 8^2 / 16^2
// planes (160 - 320 workgroups of 64 x 64 ... 128 x 128 tiles: ONE wave per SIMD, every k-tile one exposed L2 -> LDS round trip): two
// waves per SIMD and half the k-steps per wave, with no second launch, no workspace and no agent-scope hand-off.
template <int BM, int BN, int NS, int BK, bool W8 = false, int KG = 1>
struct Geo {
    // waves along M (x 2 along N): 4 or 8 waves per workgroup.  W8 (r03, tile 18 = 128 x 192): EIGHT waves on a 128-row tile, wave
    // tile 32 x 96 -- one workgroup of 8 waves per CU where 128 x 192 tiles give exactly 256 or 512 workgroups
    static constexpr int WMW = (BM >= 256 || W8) ? 4 : 2;
    static constexpr int NW = WMW * 2, NT = NW * 64 * KG;      // NW: waves of ONE k-group (the DMA dealing and the wave grid are per group)
    static constexpr int ROWB = BK * 2;                     // bytes per LDS row
    static constexpr int CHR = 1024 / ROWB;                 // rows per 1-KiB LDS-DMA chunk (16 / 8)
    // 1-KiB chunks per wave per plane.  The weight panel may deal UNEVENLY: with JBR != 0 only waves 0 .. JBR - 1 carry chunk JB - 1
    static constexpr int JA = BM / (NW * CHR), JB = (BN / CHR + NW - 1) / NW, JBR = (BN / CHR) % NW;
    static_assert(JA * NW * CHR == BM && BN % CHR == 0, "tile not divisible into per-wave DMA chunks");
    static constexpr int LPT = (JA + JB) * NS;              // LDS-DMA instructions per thread per k-tile (waves >= JBR: NS fewer)
    static constexpr int LPT_L = JBR ? (JA + JB - 1) * NS : LPT;
    static constexpr int PLANE = (BM + BN) * ROWB;
    static constexpr int STAGE = NS * PLANE;
    // LDS ring depth.  4-wave tiles: 3-4 stages -- a deeper ring costs resident workgroups per CU, and on the U-Net's
    // many-workgroup shapes occupancy hides latency better than bytes in flight (measured: 8 stages = -20 % on
    // 16384x384x384).  8-wave tiles own the CU, so they take what fits.
    static constexpr int D8 = 147456 / STAGE > 6 ? 6 : 147456 / STAGE;
    // bf16x3 (NS == 2) runs the software-pipelined virtual-k-step loop, in which every ring slot carries loads: two
    // 4-wave workgroups per CU (<= 80 KiB each) with 2-4 stages, or one 8-wave workgroup with what fits
    static constexpr int DX = 81920 / STAGE < 2 ? 2 : (81920 / STAGE > 4 ? 4 : 81920 / STAGE);
    // (deeper rings with ONE workgroup per CU were measured on the small-M shapes, r03_x3_deep_ring_tiles.txt: no gain -- those
    //  launches are bound by their fixed costs, not by stages in flight)
    static constexpr int D = NW == 8 ? (D8 < 2 ? 2 : D8) : (NS == 2 ? DX : ((4 * STAGE <= 98304) ? 4 : 3));
    static constexpr int SLABS = NW * 16 * (BN / 2 + 4) * 4;   // epilogue transpose slabs (one per wave)
    // + the bf16 residual sub-tile of every wave, DMA'd into the idle ring at the start of the epilogue (bf16 mode)
    static constexpr bool RSTAGE = NS == 1 && SLABS + BM * BN * 2 <= 163840;
    static constexpr int EPI = SLABS + (RSTAGE ? BM * BN * 2 : 0);
    static constexpr int RED_OFF = (EPI + 15) / 16 * 16;      // KG = 2: group 1's accumulators are parked BEHIND the epilogue's slabs
    static constexpr int EPI_ALL = KG == 2 ? RED_OFF + BM * BN * 4 : EPI;
    static constexpr int SMEM = KG * D * STAGE > EPI_ALL ? KG * D * STAGE : EPI_ALL;
    static_assert(KG == 1 || (KG == 2 && NS == 2 && BK == 32 && !W8 && BM < 256), "k-groups: 4-wave two-plane tiles only");
    static_assert(SMEM <= 163840, "LDS budget");
};

// drain phase of the ring: `rem` (< D-2) younger tiles are still in flight, each LPT loads per thread
template <int R, int LPT>
__device__ __forceinline__ void wait_tail(int rem) {
    if constexpr (R <= 0) {
        wait_vmcnt<0>();
    } else {
        if (rem >= R) wait_vmcnt<R * LPT>();
        else wait_tail<R - 1, LPT>(rem);
    }
}

template <int BM, int BN, int NS, bool CONV, int BK, bool W8 = false, int KG = 1>
__global__ __launch_bounds__((Geo<BM, BN, NS, BK, W8, KG>::NT), ((Geo<BM, BN, NS, BK, W8, KG>::NW == 8 || KG == 2) ? 1 : 2)) void igemm_kernel(const FridoGemm d) {
    using G = Geo<BM, BN, NS, BK, W8, KG>;
#if IG_PROF
    unsigned igp[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const unsigned igp_t0 = IGP_NOW();
    unsigned igp_prev = igp_t0, igp_a = 0u, igp_loop0 = 0u;
#endif
    constexpr int JBR = G::JBR;
    constexpr int ROWB = G::ROWB, CHR = G::CHR, KS = BK / 32;
    constexpr int WM = G::WMW, WN = 2, NW = G::NW;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int D = G::D, JA = G::JA, JB = G::JB, PLANE = G::PLANE, STAGE = G::STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave_wg = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kg = KG == 2 ? wave_wg >> 2 : 0;                 // k-group of this wave (KG = 2: waves 0..3 / 4..7)
    const int wave = KG == 2 ? wave_wg & 3 : wave_wg;          // wave within its group: DMA dealing, wave grid, epilogue slabs
    const int wm = wave >> 1, wn = wave & 1;
    unsigned char* const ring = smem + kg * (D * STAGE);       // this group's LDS ring

    // ---- block -> tile (XCD-aware: consecutive logical tiles share an A row-panel and an XCD L2) ----
    const int tiles_n = (d.N + BN - 1) / BN;
    const int tiles_m = (d.M + BM - 1) / BM;
    const int nb = tiles_m * tiles_n;
    // Work items = (k-slice, tile), k-slice slowest; workgroups are dealt to the 8 XCDs round-robin in dispatch order
    // (x fastest, then z), and XCD x takes a CONTIGUOUS range of items: neighbouring tiles share an A row panel, and under
    // split-K an XCD works on ONE k-slice (or a few), so its L2 holds that slice of the weights once.  Before (k-slice =
    // blockIdx.z, every XCD walked all slices): the 8x8-plane convs (M = 1024, split 8) fetched 72 MB per launch from
    // MALL / HBM for 18.6 MB of operands (tools/_pmc_smallm.sh).
    const int kz = xcd_item(nb) / nb;                          // (KG = 2 launches have gridDim.z == 1: kz = 0)
    const int bid = xcd_item(nb) - kz * nb;
    int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    // (r06) COLUMN PANELS (FridoGemm.flags bit 27): with many tile columns an XCD's 64 concurrent workgroups cover a few tile rows x ALL
    // columns, i.e. the whole weight matrix -- beyond its 4-MB L2 (GEGLU projection 16384 x 3072 x 384: 4.7 MB of weights + the rows'
    // activations), so every group of tile rows streams the weights from MALL again: PMC (profiles/r06_pmc_l2_by_instance_*.json) L2 hit rate
    // 81 %, 162 MB fetched per launch for 30 MB of operands.  Tiles are therefore walked panel by panel -- P columns x all rows, P = 8, halved while its weight
    // panel exceeds 3 MB (64 concurrent workgroups = 8 rows x 8 columns minimise rows + columns) -- so the concurrent set is a block of rows x P columns: the panel stays L2-resident
    // while the rows stream through once.  Pure re-ordering of independent tiles: results unchanged bit for bit.
    if ((d.flags >> 27) & 1) {
        const int colbytes = BN * (d.K + d.K2) * 2 * NS;
        int P = 8;
        while (P > 1 && P * colbytes > (3 << 20)) P >>= 1;
        if (P > 1 && tiles_n > P) {
            const int full = tiles_n / P, per_panel = tiles_m * P;
            const int p = bid / per_panel;
            if (p < full) {
                const int r = bid - p * per_panel;
                tm = r / P;
                tn = p * P + (r - tm * P);
            } else {
                const int rem = tiles_n - full * P, r = bid - full * per_panel;
                tm = r / rem;
                tn = full * P + (r - tm * rem);
            }
        }
    }
    const int m0 = tm * BM, n0 = tn * BN;
    if constexpr (((FRIDO_ABLATE & 1024) || FRIDO_STAGGER_RT) && NW == 4) {
        const int id = ((int)blockIdx.y * (int)gridDim.z + (int)blockIdx.z) * (int)gridDim.x + (int)blockIdx.x;
        // run-time form: delay in QUARTER microseconds (a k-step of these tiles is ~1 us: sub-k-step offsets are the interesting ones for
        // one-round launches), smallest grid in units of 64 workgroups, and which workgroups wait: mode 0 = dispatch ids 256..511 (the
        // second slot of every CU if the dispatcher deals one workgroup per CU first), mode 1 = every other workgroup of an XCD among
        // the first 512 (the control: right only if the dispatcher fills a CU's two slots back to back)
        const int ticks = FRIDO_STAGGER_RT ? ((d.flags >> 8) & 255) * 25 : FRIDO_STAGGER_US * 100;          // 100 MHz
        const int min_wg = (FRIDO_STAGGER_RT && ((d.flags >> 16) & 255)) ? ((d.flags >> 16) & 255) * 64 : 768;
        const int mode = FRIDO_STAGGER_RT ? (d.flags >> 24) & 3 : 0;
        const bool late = mode == 0 ? (id >= 256 && id < 512) : (id < 512 && ((id >> 3) & 1));
        if (ticks && late && (int)(gridDim.x * gridDim.y * gridDim.z) >= min_wg) {
            const uint64_t t0 = wall_clock64();
            while (wall_clock64() - t0 < (uint64_t)ticks) __builtin_amdgcn_s_sleep(4);
        }
    }
    if constexpr (NW == 8) stagger_one_per_cu(d.flags);
    // batch index, optionally two-level (outer x inner, e.g. image x head)
    int zo = blockIdx.y, zi = 0;
    if (d.batch_inner > 1) {
        zo = blockIdx.y / d.batch_inner;
        zi = blockIdx.y - zo * d.batch_inner;
    }
    const frido_bf16* __restrict__ Ab = d.A + (int64_t)zo * d.a_bs + (int64_t)zi * d.a_bs2;
    const frido_bf16* __restrict__ Bb = d.B + (int64_t)zo * d.b_bs + (int64_t)zi * d.b_bs2;

    // ---- LDS-DMA assignments: wave w moves chunks w, w+4, ... ; lane l of a chunk lands at row l>>2, physical
    //      slot l&3, i.e. it must FETCH logical slot (l&3) ^ swz(row) ----
    // BK = 32: 64-B rows, 4 slots, slot q of row r at q ^ ((4 - (r>>2)) & 3);  BK = 64: 128-B rows, 8 slots, q ^ ((r>>1) & 7)
    // (ablation 64, dense bf16x3 timing only: a piece = 8 rows x 128 contiguous bytes of a plane-interleaved operand; the "lo" piece
    //  covers the chunk's rows 8..15 -- same instruction count and bytes as the shipped 16 rows x 64 B pieces, full 128-B lines)
    constexpr bool AB64 = (FRIDO_ABLATE & 64) && NS == 2 && !CONV;
    const int lrow = (BK == 32 && !AB64) ? lane >> 2 : lane >> 3;
    const int lq = AB64 ? (lane & 7) : BK == 32 ? (lane & 3) ^ ((4 - ((lrow >> 2) & 3)) & 3)
                            : (lane & 7) ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7);
    // conv: element offset of tap (0,0) of this row's receptive field + a bit mask of the taps that fall inside
    // the (logical) input; with resampling folded in (up/dn shifts) the per-tap offsets are tabulated instead
    // all four phases of an upsample conv in one launch (up2_phase == 5): phase = batch index, padding follows the phase
    const int pady = d.up2_phase == 5 ? 1 - (zo >> 1) : d.pad, padx = d.up2_phase == 5 ? 1 - (zo & 1) : d.padx;
    int64_t a_off[JA];
    unsigned a_mask[JA];
    const bool resample = CONV && (d.up_shift | d.dn_shift) != 0;
    int a_pos[JA];                       // (image << 20) | (oy << 10) | ox of this row's output pixel: only the resampling convs read it back
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int row = (wave + NW * j) * CHR + lrow;
        int m = m0 + row;
        if (CONV) {
            const bool okm = m < d.M;
            m = okm ? m : 0;
            const int hw = d.Ho * d.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
            a_pos[j] = (b << 20) | (oy << 10) | ox;
            unsigned mask = 0;
            for (int ty = 0; ty < d.kh; ++ty)
                for (int tx = 0; tx < d.kw; ++tx) {
                    const int iy = oy * d.stride + ty - pady, ix = ox * d.stride + tx - padx;
                    if (okm && iy >= 0 && iy < d.Hl && ix >= 0 && ix < d.Wl) mask |= 1u << (ty * d.kw + tx);
                }
            a_mask[j] = mask;
            a_off[j] = ((int64_t)(b * d.Hs + oy * d.stride - pady) * d.Ws + (ox * d.stride - padx)) * d.Cin + lq * 8;
        } else {
            m = m < d.M ? m : d.M - 1;      // rows past M are clamped (their outputs are masked)
            a_off[j] = (int64_t)m * d.lda * ((FRIDO_ABLATE & 16) ? 2 : 1) + lq * 8;
        }
    }
    const frido_bf16* __restrict__ A2b = d.A2;
    const int nk1 = d.K / BK;                 // k-tiles of the primary A operand
    int64_t b_off[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        constexpr bool BIL = (FRIDO_ABLATE & 512) && NS == 2;      // timing only: 8 rows x 128 B weight pieces
        int row = (wave + NW * j) * CHR + (BIL ? lane >> 3 : lrow);
        row = row < BN ? row : BN - 1;                            // (uneven dealing: this wave has no chunk j; never issued)
        int n = n0 + (d.geglu ? row : chan_of_pos(row));      // LDS row `row` holds channel chan_of_pos(row): see tile_epilogue
        n = n < d.N ? n : d.N - 1;
        if (BIL) n = n < d.N - 8 ? n : d.N - 9;      // (the "lo" piece reads 8 rows further down: stay inside the buffer)
        b_off[j] = BIL ? (int64_t)n * d.ldb * 2 + (lane & 7) * 8 : (int64_t)n * d.ldb * ((FRIDO_ABLATE & 16) && !CONV ? 2 : 1) + lq * 8;
    }

    // k-tile range of this workgroup (split-K: gridDim.z slices)
    const int nk_all = (d.K + d.K2) / BK;
    // (KG = 2: the two groups of the workgroup are the two "slices"; the launcher admits an even number of k-tiles only, so both
    //  groups run the same number of loop iterations and meet at the same workgroup-wide barriers)
    const int per = KG == 2 ? nk_all / 2 : (nk_all + (int)gridDim.z - 1) / (int)gridDim.z;
    const int kt0 = (KG == 2 ? kg : kz) * per;
    const int nk = max(0, min(nk_all, kt0 + per) - kt0);
    const int cin = d.Cin, kw = d.kw, ws = d.Ws;
    // keep the zero page's address in SGPRs (otherwise hipcc re-loads it from the GOT inside the k-loop)
    unsigned long long zero_addr = (unsigned long long)reinterpret_cast<const void*>(g_zero_page);
    asm volatile("" : "+s"(zero_addr));
    const int64_t b_lo = ((FRIDO_ABLATE & 512) && NS == 2) ? (int64_t)16 * d.ldb : AB64 ? (int64_t)16 * d.ldb : (FRIDO_ABLATE & 16) && !CONV ? 32 : d.b_lo;
    constexpr int KADV = (FRIDO_ABLATE & 16) && !CONV ? 2 * BK : BK;      // elements a dense source pointer advances per k-tile
    constexpr int KADVB = ((FRIDO_ABLATE & 512) && NS == 2) ? 2 * BK : KADV;

    // ---- incremental source pointers.  Inside one SEGMENT of the k-walk (the channel chunks of one conv tap, the whole K of
    //      a dense operand, the appended A2 range) every piece just advances by BK elements per k-tile; the per-piece
    //      base, the tap's validity bit and the zero-page substitution are evaluated once per segment (`retap`).  The
    //      previous form re-derived all of it per k-tile: ~100 scalar + vector instructions next to 24 MFMAs.
    const frido_bf16* aptr[JA];
    int astep[JA];                      // BK, or 0 for a piece parked on the zero page
    const frido_bf16* bptr[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) bptr[j] = Bb + b_off[j] + (int64_t)kt0 * KADVB;
    int ktn = kt0;                      // next k-tile to issue
    int seg_left = 0;                   // k-tiles left in the current segment
    int64_t alo_cur = d.a_lo;           // hi -> lo plane distance of the operand the segment reads
    int rt_tap = -1, rt_ky = 0, rt_kx = 0;      // conv k-walk state: tap of the current segment (-1: none yet) and its (ky, kx)
    auto retap = [&]() {
        if (ktn >= nk1) {               // appended dense operand (fused 1x1 skip conv): runs to the end of K
#pragma unroll
            for (int j = 0; j < JA; ++j) {
                int m2 = m0 + (wave + NW * j) * CHR + lrow;          // (recomputed here: once per launch, nothing to keep live)
                m2 = m2 < d.M ? m2 : d.M - 1;
                aptr[j] = A2b + (int64_t)m2 * d.lda2 + lq * 8 + (int64_t)(ktn - nk1) * BK;
                astep[j] = BK;
            }
            alo_cur = d.a2_lo;
            seg_left = 1 << 30;
        } else if (CONV) {
            // k-tile ktn -> (tap, first channel kc); a segment = the channel chunks of one tap.  After the first call a segment ends
            // exactly at a tap boundary, so the walk just steps to the next tap: no divisions on the hot path (the general form
            // below was ~250 instructions between the k-tile's barrier and its first DMA piece, once per tap).
            const int cpt = cin / BK;
            int tap, kc;
            if (rt_tap >= 0) {
                tap = rt_tap + 1;
                kc = 0;
            } else {
                tap = ktn / cpt;
                kc = (ktn - tap * cpt) * BK;
                rt_ky = tap / kw;
                rt_kx = tap - rt_ky * kw - 1;
            }
            rt_tap = tap;
            if (++rt_kx == kw) { rt_kx = 0; ++rt_ky; }
            const int ky = rt_ky, kx = rt_kx;
#pragma unroll
            for (int j = 0; j < JA; ++j) {
                const bool ok = (a_mask[j] >> tap) & 1u;
                int64_t off;
                if (resample) {   // Upsample / SPADE-resize convs: source pixel = ((iy >> up) << dn, (ix >> up) << dn)
                    const int a_b_ = a_pos[j] >> 20, a_oy_ = (a_pos[j] >> 10) & 1023, a_ox_ = a_pos[j] & 1023;
                    const int iy = a_oy_ * d.stride + ky - pady, ix = a_ox_ * d.stride + kx - padx;
                    const int sy = (iy >> d.up_shift) << d.dn_shift, sx = (ix >> d.up_shift) << d.dn_shift;
                    off = ((int64_t)(a_b_ * d.Hs + sy) * d.Ws + sx) * d.Cin + kc + lq * 8;
                } else {
                    off = a_off[j] + ((int64_t)ky * ws + kx) * cin + kc;
                }
                aptr[j] = ok ? Ab + off : reinterpret_cast<const frido_bf16*>(zero_addr);
                astep[j] = ok ? BK : 0;
            }
            const int in_tap = cpt - kc / BK, to_end = nk1 - ktn;
            seg_left = in_tap < to_end ? in_tap : to_end;
        } else {
#pragma unroll
            for (int j = 0; j < JA; ++j) {
                aptr[j] = Ab + a_off[j] + (int64_t)ktn * KADV;
                astep[j] = KADV;
            }
            if constexpr ((FRIDO_ABLATE & 16) && !CONV) alo_cur = AB64 ? (int64_t)16 * d.lda : 32;
            seg_left = nk1 - ktn;
        }
    };

    // one DMA piece of a stage (bf16 mode only: pieces 0..JA-1 = A chunks, JA..JA+JB-1 = B chunks); issue_begin / issue_end bracket a stage
    auto issue_begin = [&]() { if (seg_left == 0) retap(); };
    auto issue_piece = [&](auto pc, int buf) {
        constexpr int pi = decltype(pc)::value;
        unsigned char* sb = ring + buf * STAGE + wave * 1024;
        if constexpr (pi < JA) {
            __builtin_amdgcn_global_load_lds((gptr_t)aptr[pi], (lptr_t)(sb + pi * (NW * 1024)), 16, 0, 0);
            aptr[pi] += astep[pi];
        } else {
            constexpr int j = pi - JA;
            __builtin_amdgcn_global_load_lds((gptr_t)bptr[j], (lptr_t)(sb + BM * ROWB + j * (NW * 1024)), 16, 0, 0);
            bptr[j] += BK;
        }
    };
    auto advance = [&]() { ++ktn; --seg_left; };      // the stage of k-tile ktn has been issued
    auto issue_end = [&]() { advance(); };
    auto issue = [&](int buf) {
        if (seg_left == 0) retap();
        unsigned char* sb = ring + buf * STAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)aptr[j], (lptr_t)(sb + j * (NW * 1024)), 16, 0, 0);
            if (NS == 2) {
                const frido_bf16* lo = astep[j] ? aptr[j] + alo_cur : aptr[j];
                __builtin_amdgcn_global_load_lds((gptr_t)lo, (lptr_t)(sb + PLANE + j * (NW * 1024)), 16, 0, 0);
            }
            aptr[j] += astep[j];
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            if (JBR == 0 || j < JB - 1 || wave < JBR) {
                __builtin_amdgcn_global_load_lds((gptr_t)bptr[j], (lptr_t)(sb + BM * ROWB + j * (NW * 1024)), 16, 0, 0);
                if (NS == 2)
                    __builtin_amdgcn_global_load_lds((gptr_t)(bptr[j] + b_lo), (lptr_t)(sb + PLANE + BM * ROWB + j * (NW * 1024)), 16, 0, 0);
            }
            bptr[j] += KADVB;
        }
        advance();
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: row (lane & 15) of a 16-row MFMA tile, logical slot (lane >> 4)
    const int frow = lane & 15;
    const int fslot0 = BK == 32 ? ((lane >> 4) ^ ((4 - ((frow >> 2) & 3)) & 3)) << 4 : ((lane >> 4) ^ (frow >> 1)) << 4;
    const int fslot1 = ((4 + (lane >> 4)) ^ (frow >> 1)) << 4;          // second k-step of a BK = 64 row
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)ring;
    const unsigned a_frag = lds0 + (wm * (BM / WM) + frow) * ROWB;
    const unsigned b_frag = lds0 + BM * ROWB + (wn * (BN / WN) + frow) * ROWB;

    // uneven weight-chunk dealing (12 chunks on 8 waves): the virtual-step loop counts per wave; the plain loop is safe with a 2-slot
    // ring only (every wait is vmcnt(0))
    static_assert(JBR == 0 || (NS == 2 && W8) || (NS == 2 && D == 2), "uneven weight-chunk dealing: bf16x3 8-wave 192-column tiles only");
    if constexpr (NS == 1 && BK == 64) {
        // ---- software-pipelined main loop (bf16 mode, BK = 64 tiles).  The fragments of k-step t+1 are read from LDS into a SECOND register set
        //      while the MFMAs of k-step t issue from the first: with one workgroup of 8 waves per CU (or two of 4) the waves
        //      of a SIMD run in lockstep between barriers, so in the plain loop (below, kept for bf16x3 where the second set
        //      does not fit) the matrix pipe idled through every read phase -- 256 x 128 tile: 1100 cycles per k-tile for
        //      512 cycles of MFMA work.  Measured per tile (tools/gemm_sweep.sh -n 1 -t <all tiles> [-v "SPLITK ..."] <shapes>): BK = 64 tiles -8...-12 % per launch; BK = 32 tiles
        //      +-0 (128 x 128, 256 x 128) or spilling (128 x 192, 256 x 256), so they keep the plain loop.  Ring protocol: a stage is free as soon as every wave holds its fragments in
        //      registers, i.e. at the barrier that publishes the NEXT stage, so all D slots carry loads (one more in flight).
        auto wait_younger = [&](int younger) {                  // stages issued after the one waited for (loads retire in order)
            switch (younger) {
                case 0: wait_vmcnt<0>(); break;
                case 1: wait_vmcnt<(D > 1 ? 1 : 0) * G::LPT>(); break;
                case 2: wait_vmcnt<(D > 2 ? 2 : 0) * G::LPT>(); break;
                case 3: wait_vmcnt<(D > 3 ? 3 : 0) * G::LPT>(); break;
                case 4: wait_vmcnt<(D > 4 ? 4 : 0) * G::LPT>(); break;
                default: wait_vmcnt<(D > 5 ? 5 : 0) * G::LPT>(); break;
            }
        };
        static_assert(D <= 6 && (D - 1) * G::LPT < 64, "vmcnt immediate");
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (s < nk) issue(s);
        bf16x8 fa[2][TM], fb[2][TN];
        // fragment r of a k-step: r < TN -> weight tile r, else pixel tile r - TN
        auto read_frag = [&](auto setc, auto rc, unsigned sa, unsigned sbb) {
            constexpr int set = decltype(setc)::value, r = decltype(rc)::value;
            if constexpr (r < TN) fb[set][r] = lds_read128(sbb + r * 16 * ROWB);
            else fa[set][r - TN] = lds_read128(sa + (r - TN) * 16 * ROWB);
        };
        constexpr int NR = TM + TN;
        if (nk > 0) {
            wait_younger(nk - 1 < D - 1 ? nk - 1 : D - 1);
            __builtin_amdgcn_s_barrier();
            static_for<0, NR>([&](auto rc) { read_frag(std::integral_constant<int, 0>{}, rc, a_frag + fslot0, b_frag + fslot0); });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        int buf = 0;
        // one k-STEP (32 k) per call; P = register set holding this step's fragments.  The reads of the next step's fragments are
        // dealt over the TN MFMA groups of this step (a group = one weight tile x TM pixel tiles), the next stage's DMA
        // goes out after the first group: the matrix pipe has work queued while the wave issues memory instructions.
        auto step = [&](auto pc, auto prec, int kt, int ks) {
            constexpr int P = decltype(pc)::value;
            constexpr bool PRE = decltype(prec)::value;            // a next step exists: prefetch its fragments
            const bool last_ks = ks == KS - 1;
            const int nbuf = buf + 1 == D ? 0 : buf + 1;
            bool dma = false;                                      // refill this stage's slot during this step?
            int rbuf = buf, rks = ks + 1;
            if (last_ks) {
                rbuf = nbuf; rks = 0;
                if (PRE) {
                    // stage kt+1 must have landed; after this barrier every wave holds stage kt in registers: its slot is free
                    if (kt + D - 1 < nk) wait_vmcnt<(D - 2) * G::LPT>();
                    else wait_younger(nk - kt - 2);
                    __builtin_amdgcn_s_barrier();
                    dma = kt + D < nk;
                }
            }
            const int fs = rks ? fslot1 : fslot0;
            const unsigned sa = a_frag + rbuf * STAGE + fs, sbb = b_frag + rbuf * STAGE + fs;
            if (dma) issue_begin();
            constexpr int NPC = JA + JB;
            static_for<0, TN>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (PRE) {
                    static_for<j * NR / TN, (j + 1) * NR / TN>([&](auto rc) { read_frag(std::integral_constant<int, 1 - P>{}, rc, sa, sbb); });
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i][j] = mfma_op<NS>(fb[P][j], fa[P][i], acc[i][j]);   // weights first: C^T tiles
                __builtin_amdgcn_sched_barrier(0);
                if (dma) {
                    static_for<j * NPC / TN, (j + 1) * NPC / TN>([&](auto qc) { issue_piece(qc, buf); });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if (dma) issue_end();
            if constexpr (PRE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (last_ks) buf = nbuf;
        };
        const int nsteps = nk * KS;
        using T_ = std::true_type;
        using F_ = std::false_type;
        int st = 0;
        for (; st + 2 < nsteps; st += 2) {
            step(std::integral_constant<int, 0>{}, T_{}, st / KS, st % KS);
            step(std::integral_constant<int, 1>{}, T_{}, (st + 1) / KS, (st + 1) % KS);
        }
        if (st + 2 == nsteps) {
            step(std::integral_constant<int, 0>{}, T_{}, st / KS, st % KS);
            step(std::integral_constant<int, 1>{}, F_{}, (st + 1) / KS, (st + 1) % KS);
        } else if (st + 1 == nsteps) {
            step(std::integral_constant<int, 0>{}, F_{}, st / KS, st % KS);
        }
    } else if constexpr (NS == 2 && (BN / 2 / 16 <= 4 || W8 || FRIDO_X3_PIPE_ALL)) {
        // ---- bf16x3 main loop: software-pipelined over VIRTUAL k-steps (r03).  A stage holds the hi and lo planes of one 32-deep
        //      k-tile (= the LDS bytes of one BK = 64 bf16 stage); its product hi*hi + hi*lo + lo*hi is walked as three virtual
        //      k-steps, each TM x TN MFMAs on ONE pixel-fragment set and ONE weight-fragment set:
        //          v0: A_hi x B_hi   (meanwhile B_lo is read into the idle weight set)
        //          v1: A_hi x B_lo   (meanwhile A_lo is read into the idle pixel set)
        //          v2: A_lo x B_hi   (meanwhile the NEXT stage's A_hi / B_hi are read into the two sets v1 used)
        //      so two register sets per operand (the bf16 BK = 64 budget) carry three MFMA rounds per 2 (TM + TN) fragment reads,
        //      and the matrix pipe has TM x TN MFMAs queued behind every read / DMA / barrier phase.  The hi weights alternate
        //      between fb[0] and fb[1] from stage to stage (Q), so the loop is unrolled by two stages.  Ring protocol as in the
        //      bf16 pipelined loop: the barrier sits at the head of v2 (every wave then holds the whole stage in registers: its
        //      slot is refilled at once, all D slots carry loads); with D >= 3 the refill's DMA pieces are dealt over the MFMA
        //      groups of v2 and the next stage's v0 / v1, with D = 2 they go out in v2 (the stage must land one k-tile later).
        //      The old loop (one barrier, 2 (TM + TN) reads, then 3 TM TN MFMAs per k-tile, D = 3: ONE 4-wave workgroup per CU) left the
        //      pipe idle through every read phase: 270-300 TF/s (algorithmic) on the 64x64-plane convs.
        static_assert(BK == 32, "bf16x3 tiles: BK = 32 (two planes of 32 = the LDS stage of one plane of 64)");
        constexpr int NPC = (JA + JB) * 2;                           // DMA pieces per thread per stage (= G::LPT)
        constexpr bool SPREAD = D >= 3;
        constexpr int NG = SPREAD ? 3 * TN : TN;                     // MFMA groups one refill is dealt over
        static_assert(D <= 6 && (D - 1) * G::LPT < 64, "vmcnt immediate");
        const bool light = JBR != 0 && wave >= JBR;                  // this wave carries one weight chunk less per stage
        auto wait_stages = [&](auto kc) {                            // at most K whole stages of this wave's loads may stay in flight
            constexpr int K = decltype(kc)::value;
            if (light) wait_vmcnt<K * G::LPT_L>();
            else wait_vmcnt<K * G::LPT>();
        };
        auto wait_younger = [&](int younger) {                       // stages issued after the one waited for (loads retire in order)
            switch (younger) {
                case 0: wait_vmcnt<0>(); break;
                case 1: wait_stages(std::integral_constant<int, (D > 1 ? 1 : 0)>{}); break;
                case 2: wait_stages(std::integral_constant<int, (D > 2 ? 2 : 0)>{}); break;
                case 3: wait_stages(std::integral_constant<int, (D > 3 ? 3 : 0)>{}); break;
                case 4: wait_stages(std::integral_constant<int, (D > 4 ? 4 : 0)>{}); break;
                default: wait_stages(std::integral_constant<int, (D > 5 ? 5 : 0)>{}); break;
            }
        };
#pragma unroll
        for (int s = 0; s < D; ++s)
            if (s < nk) issue(s);
        bf16x8 fa[2][TM], fb[2][TN];                                 // fa[0] = A_hi, fa[1] = A_lo;  fb[Q] = B_hi, fb[1 - Q] = B_lo
        const unsigned afr = a_frag + fslot0, bfr = b_frag + fslot0;
        if (nk > 0) {
            wait_younger(nk - 1 < D - 1 ? nk - 1 : D - 1);
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = lds_read128(bfr + j * 16 * ROWB);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = lds_read128(afr + i * 16 * ROWB);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        int buf = 0;                                                 // ring slot of the stage being multiplied
        bool dma = false;                                            // a refill is in progress ...
        int dbuf = 0;                                                // ... into this slot
        // DMA piece pi of the refill in progress: (A chunk 0 hi, lo), (A chunk 1 hi, lo), ..., (B chunk 0 hi, lo), ...
        auto piece = [&](auto pc) {
            constexpr int pi = decltype(pc)::value, idx = pi >> 1, pl = pi & 1;
            unsigned char* sb = ring + dbuf * STAGE + pl * PLANE + wave * 1024;
            if constexpr (idx < JA) {
                const frido_bf16* src = (pl && astep[idx]) ? aptr[idx] + alo_cur : aptr[idx];
                if constexpr (FRIDO_ABLATE & 32) src = reinterpret_cast<const frido_bf16*>(zero_addr) + (lane & 3) * 8;   // same 64 bytes for every piece
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sb + idx * (NW * 1024)), 16, 0, 0);
            } else {
                constexpr int j = idx - JA;
                const frido_bf16* srcb = bptr[j] + (pl ? b_lo : 0);
                if constexpr (FRIDO_ABLATE & 32) srcb = reinterpret_cast<const frido_bf16*>(zero_addr) + (lane & 3) * 8;
                if (JBR == 0 || j < JB - 1 || wave < JBR)
                    __builtin_amdgcn_global_load_lds((gptr_t)srcb, (lptr_t)(sb + BM * ROWB + j * (NW * 1024)), 16, 0, 0);
            }
        };
        auto refill_end = [&]() {                                    // every piece of the stage is out: step the source pointers
#pragma unroll
            for (int j = 0; j < JA; ++j) aptr[j] += astep[j];
#pragma unroll
            for (int j = 0; j < JB; ++j) bptr[j] += KADVB;
            issue_end();
        };
        // (ablation 256: waves 4..7 of an 8-wave workgroup take the refill's pieces half a window later than waves 0..3)
        const bool late = (FRIDO_ABLATE & 256) && NW == 8 && wave >= 4;
        auto vstep = [&](auto vc, auto qc, auto prec, int kt) {
            constexpr int V = decltype(vc)::value, Q = decltype(qc)::value;
            constexpr bool PRE = decltype(prec)::value;             // v2 only: a next stage exists
            constexpr int BS = V == 1 ? 1 - Q : Q, AS = V == 2 ? 1 : 0;
            unsigned ra = afr + buf * STAGE + PLANE, rb = bfr + buf * STAGE + PLANE;     // the lo planes of this stage (v0, v1)
            int nbuf = buf;
            if constexpr (V == 2) {
                if constexpr (PRE) {
                    // stage kt + 1 must have landed; past this barrier every wave holds stage kt in registers: its slot is free
                    if constexpr (!(FRIDO_ABLATE & 4)) {
#if IG_PROF
                        igp_a = IGP_NOW(); igp[2] += igp_a - igp_prev;
#endif
                        if (kt + D - 1 < nk) wait_stages(std::integral_constant<int, D - 2>{});
                        else wait_younger(nk - kt - 2);
#if IG_PROF
                        igp_prev = IGP_NOW(); igp[0] += igp_prev - igp_a; igp_a = igp_prev;
#endif
                        __builtin_amdgcn_s_barrier();
#if IG_PROF
                        igp_prev = IGP_NOW(); igp[1] += igp_prev - igp_a; igp[6] += 1;
#endif
                    }
                    nbuf = buf + 1 == D ? 0 : buf + 1;
                    ra = afr + nbuf * STAGE;
                    rb = bfr + nbuf * STAGE;
                    dma = kt + D < nk && !(FRIDO_ABLATE & 1);
                    dbuf = buf;
                    if (dma) issue_begin();
                } else {
                    dma = false;
                }
            }
            constexpr int NR = V == 0 ? TN : (V == 1 ? TM : (PRE ? TM + TN : 0));      // fragment reads issued under this step
            // ... all of them under its FIRST groups (pixel fragments first: the next step needs every one of them at once, the weight
            // fragments one group at a time), so that the lgkmcnt(0) at the end of the step finds them landed
            constexpr int RG = TN > 2 ? TN - 2 : 1;
            static_for<0, TN>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
#ifdef IG_FAIRPRIO
                // (A/B, r06) least progress first within a k-tile: priority 3 .. 0 over its 3 TN MFMA groups (8-wave tiles: two waves per SIMD)
                if constexpr (NW == 8) __builtin_amdgcn_s_setprio(3 - ((V == 2 ? 0 : (V == 0 ? 1 : 2)) * TN + j) * 4 / (3 * TN));
#endif
                if constexpr (!(FRIDO_ABLATE & 2) && j < RG) static_for<j * NR / RG, (j + 1) * NR / RG>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    if constexpr (V == 0) fb[1 - Q][r] = lds_read128(rb + r * 16 * ROWB);
                    else if constexpr (V == 1) fa[1][r] = lds_read128(ra + r * 16 * ROWB);
                    else if constexpr (r < TM) fa[0][r] = lds_read128(ra + r * 16 * ROWB);
                    else fb[1 - Q][r - TM] = lds_read128(rb + (r - TM) * 16 * ROWB);
                });
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(FRIDO_ABLATE & 8)) {
                    if constexpr (FRIDO_ABLATE & 128) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i][j] = mfma_op<NS>(fb[BS][j], fa[AS][i], acc[i][j]);   // weights first: C^T tiles
                    if constexpr (FRIDO_ABLATE & 128) __builtin_amdgcn_s_setprio(0);
                } else {
                    asm volatile("" :: "v"(fb[BS][j]), "v"(fa[AS][0]), "v"(fa[AS][TM - 1]));      // keep the fragment reads alive
                }
                __builtin_amdgcn_sched_barrier(0);
                constexpr int g = V == 2 ? j : (V == 0 ? TN + j : 2 * TN + j);          // group index within the refill begun at v2
                if constexpr (g < NG) {
                    if (dma) {
                        constexpr int g2 = (g + NG / 2) % NG;
                        if (!late) static_for<g * NPC / NG, (g + 1) * NPC / NG>([&](auto pc) { piece(pc); });
                        else static_for<g2 * NPC / NG, (g2 + 1) * NPC / NG>([&](auto pc) { piece(pc); });
                        if constexpr (g == NG - 1) refill_end();
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (NR > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (V == 2) buf = nbuf;
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        int kt = 0;
#ifdef IG_PRIO
        // (A/B, r06) static priority for the second-dispatched half of an 8-wave workgroup: tools/igemm_prof.py shows it losing the MFMA
        // arbitration on every k-tile (2120 against 1620 cycles) while the older half waits for it at the barrier
        if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#if IG_PROF
        igp_prev = IGP_NOW(); igp[4] = igp_prev - igp_t0; igp_loop0 = igp_prev;
#endif
        for (; kt + 2 < nk; kt += 2) {
            vstep(I0{}, I0{}, T_{}, kt); vstep(I1{}, I0{}, T_{}, kt); vstep(I2{}, I0{}, T_{}, kt);
            vstep(I0{}, I1{}, T_{}, kt + 1); vstep(I1{}, I1{}, T_{}, kt + 1); vstep(I2{}, I1{}, T_{}, kt + 1);
        }
        if (kt + 2 == nk) {
            vstep(I0{}, I0{}, T_{}, kt); vstep(I1{}, I0{}, T_{}, kt); vstep(I2{}, I0{}, T_{}, kt);
            vstep(I0{}, I1{}, T_{}, kt + 1); vstep(I1{}, I1{}, T_{}, kt + 1); vstep(I2{}, I1{}, F_{}, kt + 1);
        } else if (kt + 1 == nk) {
            vstep(I0{}, I0{}, T_{}, kt); vstep(I1{}, I0{}, T_{}, kt); vstep(I2{}, I0{}, F_{}, kt);
        }
#if IG_PROF
        { const unsigned t1 = IGP_NOW(); igp[2] += t1 - igp_prev; igp[3] = t1 - igp_loop0; igp_prev = t1; }
#endif
    } else {
    // ---- plain loop (bf16 BK = 32 tiles; bf16x3 tiles with six n-tiles per wave, whose second fragment sets would spill: those run
    //      as TWO 4-wave workgroups per CU on a 2-slot ring, the other workgroup's MFMAs covering this one's read phase) ----
    // (r06) MID-STEP BARRIER form of the two-slot two-plane tiles (128 x 192, 64 x 192, 256 x 192; convgn.hip has the long version): k-tile kt's
    // barrier sits at the head of its LAST weight slab -- every fragment of the tile is in registers, TM MFMA triples are ready right after
    // the release -- instead of at its top, where the release is followed by the DMA issue and an LDS round trip with the matrix pipe idle.
    // Hook of k-tile kt: my share of tile kt + 1 has landed (vmcnt(0): it is all that is in flight), barrier (tile kt + 1 published, slot
    // kt & 1 free), issue tile kt + 2 into slot kt & 1.  Same MFMAs, same operands, same order: bit-identical results.
    constexpr bool MIDBAR = FRIDO_MIDBAR != 0 && NS == 2 && D == 2 && KG == 1 && TN > 1 && BK == 32;
    // ---- prologue: fill D-1 stages (mid-step form: both slots) ----
#pragma unroll
    for (int s = 0; s < D - 1 + (MIDBAR ? 1 : 0); ++s)
        if (s < nk) issue(s);
    if constexpr (MIDBAR) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
    }

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if constexpr (!MIDBAR) {
        // tile kt must have landed: at most the loads of the (D-2) younger tiles may stay in flight
        if (kt + D - 2 < nk) wait_vmcnt<(D - 2) * G::LPT>();
        else wait_tail<D - 3, G::LPT>(nk - 1 - kt);
        __builtin_amdgcn_s_barrier();      // every wave's share of tile kt is visible; stage (kt-1)%D is free
        if (kt + D - 1 < nk) {
            int nb_ = buf + D - 1;
            nb_ = nb_ >= D ? nb_ - D : nb_;
            issue(nb_);
        }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int fs = ks ? fslot1 : fslot0;
            const unsigned sa = a_frag + buf * STAGE + fs, sbb = b_frag + buf * STAGE + fs;
            // fragment reads in the order of their first use (LDS returns in order): the first weight fragment, then the pixel
            // fragments slab by slab, then the second weight fragment -- the MFMAs of slab i start as soon as ITS fragments are back
            // (r03; before, the first MFMA of a k-step waited for all TM * NS + NS reads of a wave, with all eight waves of the
            // workgroup reading at once right after the barrier)
            bf16x8 fa[NS][TM];
            bf16x8 fb[2][NS];
#pragma unroll
            for (int p = 0; p < NS; ++p) fb[0][p] = lds_read128(sbb + p * PLANE);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NS; ++p) fa[p][i] = lds_read128(sa + p * PLANE + i * 16 * ROWB);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (j + 1 < TN) {
#pragma unroll
                    for (int p = 0; p < NS; ++p) fb[(j + 1) & 1][p] = lds_read128(sbb + p * PLANE + (j + 1) * 16 * ROWB);
                    if (j > 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NS) : "memory");
                } else {
                    if (j > 0 || TN == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if constexpr (MIDBAR) {
                        wait_vmcnt<0>();
                        __builtin_amdgcn_s_barrier();
                        if (kt + 2 < nk) issue(buf);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (FRIDO_SLAB0 != 0 && NS == 2 && TN > 1 && (TM & 1) == 0) {
                    if (j == 0) {      // (r06) pixel slabs in groups of GI, pass-major inside a group (igemm_shared.h FRIDO_SLAB0)
                        constexpr int GI = FRIDO_SLAB0 == 2 ? TM : 2;
                        static_for<0, TM / GI>([&](auto gc) {
                            constexpr int i0 = decltype(gc)::value * GI;
                            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NS * (TM - GI - i0) + NS) : "memory");
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int ii = 0; ii < GI; ++ii) acc[i0 + ii][0] = mfma_op<NS>(fb[0][0], fa[1][i0 + ii], acc[i0 + ii][0]);
#pragma unroll
                            for (int ii = 0; ii < GI; ++ii) acc[i0 + ii][0] = mfma_op<NS>(fb[0][1], fa[0][i0 + ii], acc[i0 + ii][0]);
#pragma unroll
                            for (int ii = 0; ii < GI; ++ii) acc[i0 + ii][0] = mfma_op<NS>(fb[0][0], fa[0][i0 + ii], acc[i0 + ii][0]);
                            __builtin_amdgcn_sched_barrier(0);
                        });
                        continue;
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (j == 0 && TN > 1) {     // outstanding after slab i's fragments: the later slabs' + the prefetched weight fragment
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NS * (TM - 1 - i) + NS) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // weights first: the accumulator tile is C^T (a lane owns 4 consecutive channels of one pixel)
                    if (NS == 2) {
                        acc[i][j] = mfma_op<NS>(fb[j & 1][0], fa[1][i], acc[i][j]);
                        acc[i][j] = mfma_op<NS>(fb[j & 1][1], fa[0][i], acc[i][j]);
                    }
                    acc[i][j] = mfma_op<NS>(fb[j & 1][0], fa[0][i], acc[i][j]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        buf = buf + 1 == D ? 0 : buf + 1;
    }

    }

    // ---- in-kernel split-K reduction (FridoGemm.sk_mode 1, r03 A/B): every slice parks its accumulators in the workspace,
    //      fragment-major (16 coalesced bytes per lane and fragment); the LAST workgroup of the tile to arrive adds all slices in
    //      slice order -- the sum does not depend on who is last -- and runs the normal epilogue.  Nobody waits for anybody.
    bool reduced = false;
    if constexpr (KG == 2) {
        // ---- the two k-groups' partial tiles -> one: group 1 parks its accumulators (fragment-major: 16 contiguous bytes per lane and
        //      fragment, conflict-free) behind the epilogue's slab area and EXITS; group 0 adds them -- always acc0 + acc1 -- and goes
        //      on alone.  (s_barrier waits for the SURVIVING waves of a workgroup only, so the epilogue's own barriers stay valid.)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();           // both groups are done with their rings (the park area overlaps them)
        float* red = reinterpret_cast<float*>(smem + G::RED_OFF) + (t & 255) * 4;
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(red + (i * TN + j) * 1024) = acc[i][j];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(red + (i * TN + j) * 1024);
    }
    if (KG == 1 && gridDim.z > 1 && d.sk_mode == 1) {
        constexpr int NT = G::NT, SLAB = BM * BN;
        const int S = (int)gridDim.z;
        float* part = d.ws + SK_HDR + (int64_t)bid * S * SLAB;
        {
            float* mine = part + (int64_t)kz * SLAB + t * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(mine + (i * TN + j) * NT * 4) = acc[i][j];
        }
        // hand-off recipe of cdna_hip_programming.md section 6 G16 (counter form): every wave drains its slab stores, ONE lane
        // releases at agent scope (L2 write-back: the slices of a tile sit on different XCDs) and takes the ticket; the last
        // arriver's lane 0 acquires, then every wave reads the slabs with plain loads.  (__threadfence() in every lane -- the
        // first form tried -- made every workgroup pay ~0.25 us of serialized L2 maintenance: 2-6x slower launches.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                        // (also: every wave is done with the ring)
        volatile int* flag = reinterpret_cast<volatile int*>(smem);
        if (t == 0) {
            unsigned* tickets = reinterpret_cast<unsigned*>(d.ws);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned prev = __hip_atomic_fetch_add(tickets + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = prev == (unsigned)(S - 1);
            if (last) {
                __hip_atomic_store(tickets + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // header stays zero for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            *flag = last;
        }
        __syncthreads();
        const int last = *flag;
        __syncthreads();                        // the flag has been read by everyone before the epilogue reuses the ring
        if (!last) return;
        const float* src = part + t * 4;
        static_for<0, TM * TN / 2>([&](auto fc) {
            constexpr int f0 = 2 * decltype(fc)::value, f1 = f0 + 1;
            f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
            for (int z0 = 0; z0 < S; z0 += 8) {
                f32x4 p0[8], p1[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (z0 + u < S) {
                        p0[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)(z0 + u) * SLAB + f0 * NT * 4);
                        p1[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)(z0 + u) * SLAB + f1 * NT * 4);
                    }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (z0 + u < S) { s0 += p0[u]; s1 += p1[u]; }
            }
            acc[f0 / TN][f0 % TN] = s0;
            acc[f1 / TN][f1 % TN] = s1;
        });
        reduced = true;
    }
    tile_epilogue<BM, BN, NS, WM, G::RSTAGE>(d, acc, smem, m0, n0, wave, lane, zo, zi, kz, reduced);
#if IG_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores have left the wave (not: reached memory)
    igp[5] = IGP_NOW() - igp_prev;
    {
        const int wg = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
        if (lane == 0 && wg < 4096 && wave < 8) {
            unsigned* o = g_ig_prof + (wg * 8 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = igp[i];
        }
    }
#endif
}

#if IG_PROF
extern "C" int frido_ig_prof_read(unsigned* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ig_prof), (size_t)n * sizeof(unsigned), 0, hipMemcpyDeviceToHost);
}
#endif

// =====================================================================================================================
// Patch-staged 3x3 convolution (stride 1, pad 1, no resampling), bf16 mode, 256 x 192 tile, 8 waves.
//
// The ring kernel above re-fetches every input pixel once per tap (9x) and is bound by the ~20 B/clk/CU L2 -> LDS DMA rate
// (PMC: 37 % of wave time parked on vmcnt at 10 TB/s aggregate).  Here the 256 output pixels of a tile are R complete image
// rows (or 256 / HW whole small images), so their receptive field is a (R+2) x (W+2) pixel PATCH per 32-channel chunk: it is
// DMA'd into LDS ONCE (zero halo from the zero page) and serves all nine taps -- a tap is a constant LDS row shift of the A
// fragment address.  Only the weights stream per tap.  DMA bytes per FLOP drop ~2x vs the 256-row ring tile, ~3.5x vs 128 x 192.
//
//   LDS: 2 patch buffers x 32 KiB (512 pixel slots x 64 B) + DB weight stages x 12 KiB ([192][32] bf16).
//   Patch slot s holds its four 16-byte k-pieces at q ^ (((s >> 2) & 1) << 1): with the ds_read_b128 lane groups of gfx950
//   ({0-3,12-15 | 20-27} ...) sixteen CONSECUTIVE slots are conflict-free from ANY starting slot, which the tap shifts need.
//   k-walk: for every 32-channel chunk: 9 taps; then the chunks of the optional second operand A2 (fused 1x1 skip conv,
//   centre tap only).  vmcnt: loads retire in order, so "stage landed" = at most (pieces issued after it) outstanding; the
//   per-wave issue counter and the marks of the DB-1 youngest weight stages / two patches live in SGPRs.

// NW = 8: 256 x 192 tile (one workgroup per CU);  NW = 4: 128 x 192 tile (two per CU) for the planes where 256-row tiles
// would leave CUs idle.  Both: wave tile 64 x 96, 12 weight chunks of 1 KiB per tap dealt round-robin to the waves.
template <int BN, int NW_>
struct PGeo {
    static constexpr int NW = NW_, BM = NW * 32, NT = NW * 64, WM = NW / 2;
    static constexpr int PPW = NW == 8 ? 4 : 5;              // patch pieces (1-KiB chunks of 16 pixel slots) per wave
    static constexpr int PCH = NW * PPW;                     // chunks per patch buffer: 512 / 320 pixel slots
    static constexpr int PBUF = PCH * 1024, BSTAGE = BN * 64, DB = NW == 8 ? 6 : 3;
    static constexpr int BCH = BN / 16, LPBMAX = (BCH + NW - 1) / NW;   // weight chunks per tap; pieces per wave (upper bound)
    static constexpr int B0 = 2 * PBUF;
    static constexpr int SLABS = NW * 16 * (BN / 2 + 4) * 4;
    static constexpr bool RSTAGE = true;
    static constexpr int EPI = SLABS + BM * BN * 2;           // transpose slabs + staged bf16 residual tile
    static constexpr int SMEM = B0 + DB * BSTAGE > EPI ? B0 + DB * BSTAGE : EPI;
    static_assert(BN == 192 && (NW == 8 || NW == 4), "12 weight chunks: 8 + 4 (8 waves) or 3 x 4 (4 waves)");
    static_assert(SMEM <= 163840, "LDS budget");
};

__device__ __forceinline__ void wait_vmcnt_dyn(int n) {   // n = loads that may stay in flight (uniform); rounding down is safe
    switch (n) {
#define W_(N) case N: wait_vmcnt<N>(); break;
        W_(0) W_(1) W_(2) W_(3) W_(4) W_(5) W_(6) W_(7) W_(8) W_(9) W_(10) W_(11) W_(12) W_(13) W_(14) W_(15) W_(16)
        W_(17) W_(18) W_(19) W_(20) W_(21) W_(22) W_(23) W_(24) W_(25) W_(26) W_(27) W_(28) W_(29) W_(30) W_(31) W_(32)
#undef W_
        default: wait_vmcnt<32>(); break;
    }
}

// ---- statically scheduled main loop of the patch kernel (no appended operand) -------------------------------------------
// The nine taps of a 32-channel chunk are unrolled: tap shifts, weight-stage indices (3-stage ring, stage = tap % 3) and every
// vmcnt / lgkmcnt immediate are compile-time constants (the dynamic bookkeeping of the general loop -- issue counters, wait
// switch, k-walk -- was ~40 % of its step time).  (Prefetching the next tap's A fragments into a second register set was tried:
// with 96 accumulator VGPRs it spills, and every reload sits behind a vmcnt(0) that drains the weight DMA ring.)
template <int BN, int NW>
struct PatchCtx {
    unsigned char* smem;
    const frido_bf16* Ab;
    const frido_bf16* Bb;
    int pix[PGeo<BN, NW>::PPW];
    int64_t b_off[PGeo<BN, NW>::LPBMAX];
    int sbase[4];
    int pq, cin, PW, wave, kg;
    unsigned lds0, b_frag;
    unsigned long long zero_addr;
    // appended dense operand A2 (fused 1x1 skip conv): two 1-KiB pieces per wave of the [BM][32] tile, fragment offsets
    const frido_bf16* A2b;
    int arow0, lda2;            // first A2 row this lane fetches (the second is NW * 16 rows further), row stride
    int Kc, nc2;                // K of the conv part (weights of skip chunk s start at column Kc + 32 s); skip chunks
    int frow0;                  // wave row base + fragment row of this lane: dense-tile fragment addressing
};

template <int BN, int NW>
__device__ __forceinline__ void patch_issue_patch(const PatchCtx<BN, NW>& cx, int c, int pb) {
    using P = PGeo<BN, NW>;
    const frido_bf16* base = cx.Ab + c * 32 + cx.pq;
    unsigned char* dst = cx.smem + pb * P::PBUF + cx.wave * 1024;
#pragma unroll
    for (int j = 0; j < P::PPW; ++j) {
        int px = cx.pix[j];
        asm volatile("" : "+v"(px));
        const frido_bf16* src = px >= 0 ? base + (int64_t)px * cx.cin : reinterpret_cast<const frido_bf16*>(cx.zero_addr);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + j * (NW * 1024)), 16, 0, 0);
    }
}

template <int BN, int NW>
__device__ __forceinline__ void patch_issue_b(const PatchCtx<BN, NW>& cx, int c, int tap, int stage) {
    using P = PGeo<BN, NW>;
    const int64_t koff = (int64_t)tap * cx.cin + c * 32;
    unsigned char* dst = cx.smem + P::B0 + stage * P::BSTAGE + cx.wave * 1024;
#pragma unroll
    for (int j = 0; j < P::LPBMAX; ++j) {
        int64_t bo = cx.b_off[j];
        asm volatile("" : "+v"(bo));                            // no per-tap copies of these hoisted out of the chunk loop
        if (cx.wave + NW * j < P::BCH)
            __builtin_amdgcn_global_load_lds((gptr_t)(cx.Bb + bo + koff), (lptr_t)(dst + j * (NW * 1024)), 16, 0, 0);
    }
}

template <int BN, int NW, int TAP>
__device__ __forceinline__ void patch_read_a(const PatchCtx<BN, NW>& cx, bf16x8 (&fa)[4], int c) {
    using P = PGeo<BN, NW>;
    // The addresses are recomputed at every tap ON PURPOSE (the empty asm makes the inputs opaque): left alone, hipcc hoists
    // all 36 + 18 (tap, fragment) LDS addresses out of the chunk loop, spills them, and reloads each one behind a vmcnt(0).
    int pw = cx.PW;
    asm volatile("" : "+s"(pw));
    const int shift = (TAP / 3 - 1) * pw + (TAP % 3 - 1);
    const unsigned pbase = cx.lds0 + (c & 1) * P::PBUF + (cx.kg << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int sb = cx.sbase[i];
        asm volatile("" : "+v"(sb));
        const int sl = sb + shift;
        fa[i] = lds_read128((pbase + (sl << 6)) ^ ((sl & 4) << 3));
    }
}

// issue the weight stage and the dense A2 tile of skip chunk sidx (stage sidx % 3, tile slot sidx % 4: slots 0, 1 live in the
// patch buffer the LAST conv chunk does not use, slots 2, 3 in the other one)
template <int BN, int NW>
__device__ __forceinline__ void patch_issue_skip(const PatchCtx<BN, NW>& cx, int sidx, int cl) {
    using P = PGeo<BN, NW>;
    const int64_t koff = (int64_t)cx.Kc + sidx * 32;
    unsigned char* dst = cx.smem + P::B0 + (sidx % 3) * P::BSTAGE + cx.wave * 1024;
#pragma unroll
    for (int j = 0; j < P::LPBMAX; ++j) {
        int64_t bo = cx.b_off[j];
        asm volatile("" : "+v"(bo));
        if (cx.wave + NW * j < P::BCH)
            __builtin_amdgcn_global_load_lds((gptr_t)(cx.Bb + bo + koff), (lptr_t)(dst + j * (NW * 1024)), 16, 0, 0);
    }
    const int slot = sidx & 3;
    unsigned char* ad = cx.smem + (((cl + 1) & 1) ^ (slot >> 1)) * P::PBUF + (slot & 1) * (P::BM * 64) + cx.wave * 1024;
    int ar = cx.arow0;
    asm volatile("" : "+v"(ar));                                // recomputed per issue: nothing to keep live across the conv chunks
#pragma unroll
    for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(cx.A2b + (int64_t)(ar + j * NW * 16) * cx.lda2 + cx.pq + sidx * 32),
                                         (lptr_t)(ad + j * (NW * 1024)), 16, 0, 0);
}

// MODE 0: a conv chunk follows; 1: nothing follows; 2: last conv chunk, skip chunks follow
// The pixel fragments of tap T+1 come from the SAME resident patch (tap 8: from the next chunk's patch, published since its
// tap 3), so they are read into the other half of fa[2][4] under this tap's MFMAs; only the weight fragments, which need
// this tap's barrier, are read just in time.  After the barrier a wave waits for ONE ds_read instead of five.
template <int BN, int NW, int T, int MODE>
__device__ __forceinline__ void patch_tap(const PatchCtx<BN, NW>& cx, f32x4 (&acc)[4][BN / 32], bf16x8 (&fa)[2][4], bool have, int c) {
    using P = PGeo<BN, NW>;
    constexpr int TM = 4, TN = BN / 32;
    // loads issued after the weight stage of tap T (in-order retirement) may stay in flight: the next tap's stage, plus the next
    // chunk's patch when it was issued in between (taps 1, 2 of MODE 0), plus the first skip chunk's A2 tile (tap 8 of MODE 2)
    constexpr int NEXT = (MODE == 1 && T == 8) ? 0 : 1;
    constexpr int PATCH = (MODE == 0 && (T == 1 || T == 2)) ? P::PPW : ((MODE == 2 && T == 8) ? 2 : 0);
    if constexpr (NW == 8) {                 // waves 0-3 carry two weight pieces per tap, waves 4-7 one
        if (cx.wave < 4) wait_vmcnt<2 * NEXT + PATCH>();
        else wait_vmcnt<NEXT + PATCH>();
    } else {
        wait_vmcnt<3 * NEXT + PATCH>();
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (T + 2 <= 8) patch_issue_b<BN, NW>(cx, c, T + 2, (T + 2) % 3);
    else if constexpr (MODE == 0) patch_issue_b<BN, NW>(cx, c + 1, T - 7, (T + 2) % 3);
    else if constexpr (MODE == 2) { if (T - 7 < cx.nc2) patch_issue_skip<BN, NW>(cx, T - 7, c); }
    if constexpr (T == 0 && MODE == 0) patch_issue_patch<BN, NW>(cx, c + 1, (c + 1) & 1);
    constexpr int CUR = T & 1, NXT = 1 - CUR;
    // (not in the chunk that is followed by the appended operand: with that variant's extra state the second set spills)
    // Only the 8-wave tile (one workgroup per CU: its waves run in lockstep between barriers) gains from this -- 3...7 % per launch;
    // with two 4-wave workgroups per CU the other workgroup's waves already fill the read phase (A/B: -0.3 % end to end).
    constexpr bool PF = NW == 8 && MODE != 2;                   // this chunk prefetches
    constexpr bool PRE = PF && (T < 8 || MODE == 0);            // ... and a conv tap follows this one
    if constexpr (!PF && T > 0) patch_read_a<BN, NW, T>(cx, fa[CUR], c);
    if constexpr (T == 0) {
        if (!have || NW != 8) {                                              // very first tap of the workgroup: nothing was prefetched
            patch_read_a<BN, NW, 0>(cx, fa[0], c);
        } else {                                                  // prefetched by tap 8 of the previous chunk into the odd half
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[0][i] = fa[1][i];
        }
    }
    // fragment addresses of the next tap (recomputed per tap on purpose: see patch_read_a)
    unsigned na[4];
    if constexpr (PRE) {
        constexpr int TN_ = T < 8 ? T + 1 : 0;
        int pw = cx.PW;
        asm volatile("" : "+s"(pw));
        const int shift = (TN_ / 3 - 1) * pw + (TN_ % 3 - 1);
        const int cn = T < 8 ? c : c + 1;
        const unsigned pbase = cx.lds0 + (cn & 1) * P::PBUF + (cx.kg << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int sb = cx.sbase[i];
            asm volatile("" : "+v"(sb));
            const int sl = sb + shift;
            na[i] = (pbase + (sl << 6)) ^ ((sl & 4) << 3);
        }
    }
    unsigned sbb = cx.b_frag;
    asm volatile("" : "+v"(sbb));                               // see patch_read_a: keep the fragment addresses from being hoisted
    sbb += (T % 3) * P::BSTAGE;
    bf16x8 fb[2];
    fb[0] = lds_read128(sbb);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        // LDS reads return in order: after issuing this group's reads, everything up to fb[j] must be back; what may stay
        // outstanding is this group's own reads plus the previous group's prefetch read (issued after fb[j])
        const bool pf = PRE && j < 4;                            // this group carries prefetch read j
        if (j + 1 < TN) fb[(j + 1) & 1] = lds_read128(sbb + (j + 1) * 16 * 64);
        if (pf) fa[NXT][j < 4 ? j : 0] = lds_read128(na[j < 4 ? j : 0]);
        const int outstanding = (j + 1 < TN ? 1 : 0) + (pf ? 1 : 0) + ((PRE && j >= 1 && j <= 4) ? 1 : 0);
        if (outstanding == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else if (outstanding == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if (outstanding == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j & 1], fa[CUR][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int BN, int NW, int MODE>
__device__ __forceinline__ void patch_chunk(const PatchCtx<BN, NW>& cx, f32x4 (&acc)[4][BN / 32], bf16x8 (&fa)[2][4], bool have, int c) {
    patch_tap<BN, NW, 0, MODE>(cx, acc, fa, have, c); patch_tap<BN, NW, 1, MODE>(cx, acc, fa, true, c); patch_tap<BN, NW, 2, MODE>(cx, acc, fa, true, c);
    patch_tap<BN, NW, 3, MODE>(cx, acc, fa, true, c); patch_tap<BN, NW, 4, MODE>(cx, acc, fa, true, c); patch_tap<BN, NW, 5, MODE>(cx, acc, fa, true, c);
    patch_tap<BN, NW, 6, MODE>(cx, acc, fa, true, c); patch_tap<BN, NW, 7, MODE>(cx, acc, fa, true, c); patch_tap<BN, NW, 8, MODE>(cx, acc, fa, true, c);
}

// the appended dense operand: one k-tile per 32-channel chunk, weight stage s % 3, A2 tile slot s % 4, fetch distance 2
template <int BN, int NW>
__device__ __forceinline__ void patch_skip_loop(const PatchCtx<BN, NW>& cx, f32x4 (&acc)[4][BN / 32], int cl) {
    using P = PGeo<BN, NW>;
    constexpr int TM = 4, TN = BN / 32;
    for (int sidx = 0; sidx < cx.nc2; ++sidx) {
        // in flight after this step's loads: the next step's weight stage + A2 tile (if there is a next step)
        if (sidx + 1 < cx.nc2) {
            if constexpr (NW == 8) {
                if (cx.wave < 4) wait_vmcnt<2 + 2>();
                else wait_vmcnt<1 + 2>();
            } else {
                wait_vmcnt<3 + 2>();
            }
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (sidx + 2 < cx.nc2) patch_issue_skip<BN, NW>(cx, sidx + 2, cl);
        const int slot = sidx & 3;
        unsigned abase = cx.lds0 + (((cl + 1) & 1) ^ (slot >> 1)) * P::PBUF + (slot & 1) * (P::BM * 64);
        bf16x8 fa[4];
        int fr = cx.frow0;
        asm volatile("" : "+v"(fr));
        const unsigned doff = (unsigned)(fr * 64 + ((cx.kg ^ (((fr >> 2) & 1) << 1)) << 4));
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = lds_read128(abase + doff + i * 16 * 64);
        const unsigned sbb = cx.b_frag + (sidx % 3) * P::BSTAGE;
        bf16x8 fb[2];
        fb[0] = lds_read128(sbb);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (j + 1 < TN) {
                fb[(j + 1) & 1] = lds_read128(sbb + (j + 1) * 16 * 64);
                asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j & 1], fa[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int BN, int NW, bool HAS2>
__device__ __forceinline__ void patch_static_loop(const PatchCtx<BN, NW>& cx, f32x4 (&acc)[4][BN / 32], int c_begin, int nch) {
    // prologue: patch + two weight stages in flight   (nch = end of the CONV chunks of this workgroup)
    patch_issue_patch<BN, NW>(cx, c_begin, c_begin & 1);
    patch_issue_b<BN, NW>(cx, c_begin, 0, 0);
    patch_issue_b<BN, NW>(cx, c_begin, 1, 1);
    bf16x8 fa[2][4];
    bool have = false;
    for (int c = c_begin; c + 1 < nch; ++c) {
        patch_chunk<BN, NW, 0>(cx, acc, fa, have, c);
        have = true;
    }
    if constexpr (HAS2) {
        if (cx.nc2 > 0) {
            patch_chunk<BN, NW, 2>(cx, acc, fa, have, nch - 1);
            patch_skip_loop<BN, NW>(cx, acc, nch - 1);
            return;
        }
    }
    patch_chunk<BN, NW, 1>(cx, acc, fa, have, nch - 1);
}

// HAS2: instantiated separately for convolutions with an appended operand, so that the plain kernel does not carry its state
template <int BN, int NW, bool HAS2>
__global__ __launch_bounds__((NW * 64), (NW == 8 ? 1 : 2)) void conv3x3_patch_kernel(const FridoGemm d) {
    using P = PGeo<BN, NW>;
    constexpr int BM = P::BM, WM = P::WM, TM = BM / WM / 16, TN = BN / 2 / 16, DB = P::DB, LPBMAX = P::LPBMAX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (d.N + BN - 1) / BN;
    const int nb = (d.M / BM) * tiles_n;
    const int kz = xcd_item(nb) / nb;                  // (k-slice, tile) work items, XCD-contiguous: see igemm_kernel
    const int bid = xcd_item(nb) - kz * nb;
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- tile geometry ----
    const int W = d.Ws, H = d.Hs, HW = H * W;
    const int RW = HW < BM ? HW : BM;          // output pixels per group (one group = R rows of one image)
    const int R = RW / W, PW = W + 2, PS = (R + 2) * PW;
    const int ngrp = BM / RW;
    const int img0 = m0 / HW, y0 = (m0 - img0 * HW) / W;

    // ---- patch DMA pieces of this lane: chunks wave, wave+NW, wave+2NW, wave+3NW; slot = chunk*16 + lane/4 ----
    const int pq = ((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 8;      // logical k-piece this lane fetches (elements)
    int pix[P::PPW];
#pragma unroll
    for (int j = 0; j < P::PPW; ++j) {
        const int slot = (wave + NW * j) * 16 + (lane >> 2);
        const int g = slot / PS, rem = slot - g * PS;
        const int pr = rem / PW, px = rem - pr * PW;
        const int y = y0 + pr - 1, x = px - 1;
        const bool ok = g < ngrp && y >= 0 && y < H && x >= 0 && x < W;
        pix[j] = ok ? (img0 + g) * HW + y * W + x : -1;
    }
    // ---- weight DMA pieces: chunk `wave` (all waves) and chunk 8 + wave (waves 0-3); existing BK = 32 row swizzle ----
    const int lrow = lane >> 2;
    const int lq = (lane & 3) ^ ((4 - ((lrow >> 2) & 3)) & 3);
    const int lpb = NW == 8 ? (wave < 4 ? 2 : 1) : 3;
    int64_t b_off[LPBMAX];
#pragma unroll
    for (int j = 0; j < LPBMAX; ++j) {
        int n = n0 + chan_of_pos((wave + NW * j) * 16 + lrow);      // permuted weight rows: see tile_epilogue
        n = n < d.N ? n : d.N - 1;
        b_off[j] = (int64_t)n * d.ldb + lq * 8;
    }
    const frido_bf16* __restrict__ Ab = d.A;
    const frido_bf16* __restrict__ A2b = d.A2;
    const frido_bf16* __restrict__ Bb = d.B;
    const int cin = d.Cin;
    const int nc1 = cin >> 5, nc2 = HAS2 ? d.K2 >> 5 : 0, nch_all = nc1 + nc2;
    // split-K: gridDim.z slices of the 32-channel chunk sequence (a conv chunk = 9 k-tiles, an A2 chunk = 1)
    const int c_begin = (int)((long)nch_all * kz / (int)gridDim.z), nch = (int)((long)nch_all * (kz + 1) / (int)gridDim.z);
    const int ncv = (nch < nc1 ? nch : nc1) - (c_begin < nc1 ? c_begin : nc1);        // conv chunks in [c_begin, nch)
    const int S = 9 * ncv + (nch - c_begin - ncv);
    unsigned long long zero_addr = (unsigned long long)reinterpret_cast<const void*>(g_zero_page);
    asm volatile("" : "+s"(zero_addr));

    auto issue_patch = [&](int c, int pb) {
        const frido_bf16* base;
        int64_t ld;
        if (c < nc1) { base = Ab + c * 32 + pq; ld = cin; }
        else { base = A2b + (c - nc1) * 32 + pq; ld = d.lda2; }
        unsigned char* dst = smem + pb * P::PBUF + wave * 1024;
#pragma unroll
        for (int j = 0; j < P::PPW; ++j) {
            const frido_bf16* src = pix[j] >= 0 ? base + (int64_t)pix[j] * ld : reinterpret_cast<const frido_bf16*>(zero_addr);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + j * (NW * 1024)), 16, 0, 0);
        }
    };
    // issue-side k-walk (uniform): chunk / tap of the next weight stage to fetch
    int ic = c_begin, it = c_begin < nc1 ? 0 : 4;
    auto issue_b = [&](int stage) {
        const int64_t koff = ic < nc1 ? (int64_t)it * cin + ic * 32 : (int64_t)d.K + (ic - nc1) * 32;
        unsigned char* dst = smem + P::B0 + stage * P::BSTAGE + wave * 1024;
#pragma unroll
        for (int j = 0; j < LPBMAX; ++j)
            if (wave + NW * j < P::BCH)
                __builtin_amdgcn_global_load_lds((gptr_t)(Bb + b_off[j] + koff), (lptr_t)(dst + j * (NW * 1024)), 16, 0, 0);
        if (ic < nc1 && it < 8) ++it;
        else { ++ic; it = ic < nc1 ? 0 : 4; }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addressing ----
    const int frow = lane & 15, kg = lane >> 4;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    int sbase[TM];                                                    // patch slot of this lane's output pixel (centre tap)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm * (BM / WM) + i * 16 + frow;
        const int g = ml / RW, rem = ml - g * RW;
        const int r = rem / W, x = rem - r * W;
        sbase[i] = g * PS + (r + 1) * PW + (x + 1);
    }
    const unsigned b_frag = lds0 + P::B0 + (wn * (BN / 2) + frow) * 64 + ((kg ^ ((4 - ((frow >> 2) & 3)) & 3)) << 4);

    if (nc1 > 0 && (nc2 == 0 || gridDim.z == 1) && c_begin < (nch < nc1 ? nch : nc1)) {
        PatchCtx<BN, NW> cx;
        cx.smem = smem; cx.Ab = Ab; cx.Bb = Bb;
#pragma unroll
        for (int j = 0; j < P::PPW; ++j) cx.pix[j] = pix[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) cx.sbase[j] = sbase[j];
#pragma unroll
        for (int j = 0; j < LPBMAX; ++j) cx.b_off[j] = b_off[j];
        cx.pq = pq; cx.cin = cin; cx.PW = PW; cx.wave = wave; cx.kg = kg; cx.lds0 = lds0; cx.b_frag = b_frag; cx.zero_addr = zero_addr;
        cx.A2b = A2b; cx.Kc = d.K; cx.nc2 = gridDim.z == 1 ? nc2 : 0;
        cx.arow0 = m0 + wave * 16 + (lane >> 2); cx.lda2 = d.lda2; cx.frow0 = wm * (BM / WM) + frow;
        patch_static_loop<BN, NW, HAS2>(cx, acc, c_begin, nch < nc1 ? nch : nc1);
        tile_epilogue<BM, BN, 1, WM, true>(d, acc, smem, m0, n0, wave, lane, 0, 0, kz);
        return;
    }
    // ---- prologue ----
    int tot = 0;                      // DMA pieces this wave has issued
    issue_patch(c_begin, c_begin & 1);
    tot += P::PPW;
    int markp_cur = tot, markp_nxt = tot;      // issue count right after the current / next chunk's patch
    int mark[DB - 1];                          // ... right after the weight stages of steps s .. s+DB-2
#pragma unroll
    for (int s = 0; s < DB - 1; ++s) {
        if (s < S) { issue_b(s); tot += lpb; }
        mark[s] = tot;
    }

    int cc = c_begin, tc = c_begin < nc1 ? 0 : 4;   // compute-side k-walk
    int stage = 0;
    bool first = true;                         // first step of chunk cc
    for (int s = 0; s < S; ++s) {
        {
            const int need = mark[0] > markp_cur ? mark[0] : markp_cur;
            wait_vmcnt_dyn(tot - need);
        }
        __builtin_amdgcn_s_barrier();          // stage s (and patch cc) visible to every wave; stage s-1 / patch cc-1 are free
        int newmark;
        {
            if (s + DB - 1 < S) {
                int st = stage + DB - 1;
                st = st >= DB ? st - DB : st;
                issue_b(st);
                tot += lpb;
            }
            newmark = tot;
            if (first && cc + 1 < nch) {
                issue_patch(cc + 1, (cc + 1) & 1);
                tot += P::PPW;
                markp_nxt = tot;
            }
#pragma unroll
            for (int q = 0; q < DB - 2; ++q) mark[q] = mark[q + 1];
            mark[DB - 2] = newmark;
        }
        // ---- compute: tap shift = constant slot offset ----
        const int dy = tc / 3 - 1, dx = tc - (tc / 3) * 3 - 1;
        const int shift = dy * PW + dx;
        const unsigned pbase = lds0 + (cc & 1) * P::PBUF + (kg << 4);
        bf16x8 fa[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int sl = sbase[i] + shift;
            fa[i] = lds_read128((pbase + (sl << 6)) ^ ((sl & 4) << 3));
        }
        const unsigned sbb = b_frag + stage * P::BSTAGE;
        bf16x8 fb[2];
        fb[0] = lds_read128(sbb);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (j + 1 < TN) {
                fb[(j + 1) & 1] = lds_read128(sbb + (j + 1) * 16 * 64);
                asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j & 1], fa[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- advance ----
        stage = stage + 1 == DB ? 0 : stage + 1;
        first = false;
        if (cc < nc1 && tc < 8) ++tc;
        else {
            ++cc;
            tc = cc < nc1 ? 0 : 4;
            first = true;
            markp_cur = markp_nxt;
        }
    }
    tile_epilogue<BM, BN, 1, WM, true>(d, acc, smem, m0, n0, wave, lane, 0, 0, kz);
}

// split-K reduction + epilogue, 8 columns per thread (16-byte accesses; N % 8 == 0 and 8-element aligned strides)
__global__ __launch_bounds__(256) void splitk_reduce8_kernel(const FridoGemm d) {
    const int n8 = d.N >> 3;
    const unsigned total = (unsigned)d.M * (unsigned)n8;
    int vstep = 0;
    if (d.rowvec && d.rowvec_step) vstep = *d.rowvec_step;
    const int64_t plane = (int64_t)d.M * d.N;
    bool sat = false;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int m = (int)(i / (unsigned)n8), n = (int)(i - (unsigned)m * (unsigned)n8) * 8;
        const float* w = d.ws + SK_HDR + (int64_t)m * d.N + n;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        // up to 8 slices' loads in flight per lane (a rolled loop kept ONE pair in flight: S dependent round trips to partials that
        // other XCDs wrote, ~1 us each); the additions keep the slice order, so the sums are bit-identical to the rolled form
        for (int z0 = 0; z0 < d.splitk; z0 += 8) {
            float4 a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (z0 + u < d.splitk) {
                    a[u] = *reinterpret_cast<const float4*>(w + (z0 + u) * plane);
                    b[u] = *reinterpret_cast<const float4*>(w + (z0 + u) * plane + 4);
                }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (z0 + u < d.splitk) {
                    v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w;
                    v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
                }
        }
        float add[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) add[e] = d.row_bias ? d.row_bias[m] : 0.f;
        if (d.bias) {
            const float4 a = *reinterpret_cast<const float4*>(d.bias + n), b = *reinterpret_cast<const float4*>(d.bias + n + 4);
            add[0] += a.x; add[1] += a.y; add[2] += a.z; add[3] += a.w; add[4] += b.x; add[5] += b.y; add[6] += b.z; add[7] += b.w;
        }
        if (d.rowvec) {
            const float* rp = d.rowvec + (int64_t)(m / d.rows_per_vec + vstep) * d.ldv + n;
            const float4 a = *reinterpret_cast<const float4*>(rp), b = *reinterpret_cast<const float4*>(rp + 4);
            add[0] += a.x; add[1] += a.y; add[2] += a.z; add[3] += a.w; add[4] += b.x; add[5] += b.y; add[6] += b.z; add[7] += b.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * d.alpha + add[e];
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
            const int64_t ro = (int64_t)m * d.ldr + n;
            if (d.res_bf16) {
                const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const frido_bf16*>(d.residual) + ro);
                v[0] += __uint_as_float(u.x << 16); v[1] += __uint_as_float(u.x & 0xffff0000u);
                v[2] += __uint_as_float(u.y << 16); v[3] += __uint_as_float(u.y & 0xffff0000u);
                v[4] += __uint_as_float(u.z << 16); v[5] += __uint_as_float(u.z & 0xffff0000u);
                v[6] += __uint_as_float(u.w << 16); v[7] += __uint_as_float(u.w & 0xffff0000u);
            } else {
                const float* rp = reinterpret_cast<const float*>(d.residual) + ro;
                const float4 a = *reinterpret_cast<const float4*>(rp), b = *reinterpret_cast<const float4*>(rp + 4);
                v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
            }
        }
        if (d.out_f32) {
            const int64_t o = (int64_t)m * d.ldo + n;
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
            for (int e = 0; e < 8; e += 2) { split_op2(v[e], v[e + 1], d.nsplit, h[e], l[e]); h[e + 1] = 0u; l[e + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
            if (d.nsplit == 2) sat |= op_sat8(v);
            frido_bf16* op = d.out_op + (int64_t)m * d.ldoo + n;
            *reinterpret_cast<uint4*>(op) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            if (d.nsplit == 2)
                *reinterpret_cast<uint4*>(op + d.oo_lo) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
        }
    }
    status_raise(sat);
}

// launch the split-K reduction that matches the descriptor's alignment
void launch_splitk_reduce(const FridoGemm& d, hipStream_t s);

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const FridoGemm d) {
    const int64_t total = (int64_t)d.M * d.N;
    int vstep = 0;
    if (d.rowvec && d.rowvec_step) vstep = *d.rowvec_step;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / d.N), n = (int)(i - (int64_t)m * d.N);
        float a = 0.f;
        for (int z = 0; z < d.splitk; ++z) a += d.ws[SK_HDR + (int64_t)z * total + i];
        float v = a * d.alpha + (d.bias ? d.bias[n] : 0.f);
        if (d.row_bias) v += d.row_bias[m];
        if (d.rowvec) v += d.rowvec[(int64_t)(m / d.rows_per_vec + vstep) * d.ldv + n];
        if (d.act == FRIDO_ACT_RELU) v = fmaxf(v, 0.f);
        else if (d.act == FRIDO_ACT_SILU) v = silu_f(v);
        else if (d.act == FRIDO_ACT_GELU) v = gelu_f(v);
        else if (d.act == FRIDO_ACT_QUICKGELU) v = quickgelu_f(v);
        if (d.residual) v += load_act1(d.residual, (int64_t)m * d.ldr + n, d.res_bf16);
        if (d.out_f32) store_act1(d.out_f32, (int64_t)m * d.ldo + n, d.out_bf16, v);
        if (d.out_op) store_op1(d.out_op, d.oo_lo, d.nsplit, (int64_t)m * d.ldoo + n, v);
        if (d.out_op && d.nsplit == 2 && op_sat(v)) status_raise(true);
    }
}

void launch_splitk_reduce(const FridoGemm& d, hipStream_t s) {
    const bool vec8 = ((d.N | d.ldo | d.ldr | d.ldoo | d.ldv | d.oo_lo) & 7) == 0;
    const int64_t total = vec8 ? (int64_t)d.M * (d.N >> 3) : (int64_t)d.M * d.N;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (vec8) hipLaunchKernelGGL(splitk_reduce8_kernel, dim3(blocks), dim3(256), 0, s, d);
    else hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, d);
}

template <int BM, int BN, int NS, bool CONV, int BK, bool W8 = false, int KG = 1>
int set_attr() {
    constexpr int smem = Geo<BM, BN, NS, BK, W8, KG>::SMEM;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, NS, CONV, BK, W8, KG>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
        frido_set_error("igemm: cannot set dynamic LDS size %d", smem);
        return FRIDO_EHIP;
    }
    return FRIDO_OK;
}

template <int BM, int BN, int NS, bool CONV, int BK, bool W8 = false>
int launch(const FridoGemm& d, hipStream_t s) {
    constexpr int smem = Geo<BM, BN, NS, BK, W8>::SMEM;
    const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    const int sk = d.splitk > 1 ? d.splitk : 1;
    if (sk > 1 && d.sk_mode == 1 && tiles > SK_HDR) {
        frido_set_error("igemm: sk_mode 1 has %d ticket slots, the launch has %d output tiles", SK_HDR, tiles);
        return FRIDO_EINVAL;
    }
    if (sk > 1 && d.sk_mode == 1) {                         // the last workgroup of each tile reduces: no second launch
        hipLaunchKernelGGL((igemm_kernel<BM, BN, NS, CONV, BK, W8>), dim3(tiles, d.batch, sk), dim3(Geo<BM, BN, NS, BK, W8>::NT), smem, s, d);
        return frido_check_launch("igemm");
    }
    FridoGemm dd = d;
    dd.sk_mode = 0;
    if (sk > 1) dd.gn_part = nullptr;                       // (validated: only sk_mode 1 may carry gn_part under split-K)
    hipLaunchKernelGGL((igemm_kernel<BM, BN, NS, CONV, BK, W8>), dim3(tiles, d.batch, sk), dim3(Geo<BM, BN, NS, BK, W8>::NT), smem, s, dd);
    if (sk > 1 && d.sk_mode != 2) {                         // (sk_mode 2: the consuming GroupNorm launch adds the slices up)
        launch_splitk_reduce(dd, s);
    }
    return frido_check_launch("igemm");
}

// tiles 31 / 33 / 34 / 35 / 36 (r06): the 4-wave tile (id - 30) with K split over the two wave groups of an 8-wave workgroup (Geo, KG = 2)
bool kg2_ok(const FridoGemm& d) {
    const int nk = (d.K + d.K2) / 32;
    return d.nsplit == 2 && !d.conv && d.splitk <= 1 && nk >= 2 && (nk & 1) == 0 && !d.gn_x1;
}

template <int BM, int BN>
int launch_kg2(const FridoGemm& d, hipStream_t s) {
    if (!kg2_ok(d)) {
        frido_set_error("igemm: tiles 31..36 (K split inside the workgroup) take dense two-plane GEMMs with an EVEN number of 32-deep k-tiles and no split-K");
        return FRIDO_EINVAL;
    }
    using G = Geo<BM, BN, 2, 32, false, 2>;
    const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    FridoGemm dd = d;
    dd.sk_mode = 0;
    hipLaunchKernelGGL((igemm_kernel<BM, BN, 2, false, 32, false, 2>), dim3(tiles, d.batch, 1), dim3(G::NT), G::SMEM, s, dd);
    return frido_check_launch("igemm (k-groups)");
}

// eligibility of the patch-staged 3x3 kernel (tile id 9)
bool patch_ok(const FridoGemm& d, int bm) {
    if (!d.conv || d.nsplit != 1 || d.batch != 1) return false;
    if (d.splitk > 1 && (!d.ws || d.splitk > ((d.Cin + d.K2) >> 5))) return false;
    if (d.kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.padx != 1 || d.up_shift || d.dn_shift || d.up2_phase) return false;
    if (d.Ho != d.Hs || d.Wo != d.Ws || d.Hl != d.Hs || d.Wl != d.Ws) return false;
    const int W = d.Ws, HW = d.Hs * d.Ws;
    if (W < 8 || W > 64 || (W & (W - 1))) return false;
    if (d.M % bm || !((HW % bm) == 0 || (bm % HW) == 0)) return false;
    const int RW = HW < bm ? HW : bm, R = RW / W;
    if (R < 1 || (bm / RW) * (R + 2) * (W + 2) > (bm == 256 ? 512 : 320)) return false;      // pixel slots of a patch buffer (PGeo::PCH)
    if ((d.Cin & 31) || (d.K2 & 31) || d.K != 9 * d.Cin || (d.K2 && !d.A2)) return false;
    return true;
}

int launch_patch(const FridoGemm& d, int nw, hipStream_t s) {
    const int bm = nw * 32;
    if (!patch_ok(d, bm)) {
        frido_set_error("igemm: tile 9 / 10 (patch-staged 3x3) does not apply to this convolution");
        return FRIDO_EINVAL;
    }
    const int tiles = (d.M / bm) * ((d.N + 191) / 192);
    const int sk = d.splitk > 1 ? d.splitk : 1;
    constexpr int smem8 = PGeo<192, 8>::SMEM, smem4 = PGeo<192, 4>::SMEM;
    if (nw == 8 && d.K2) hipLaunchKernelGGL((conv3x3_patch_kernel<192, 8, true>), dim3(tiles, 1, sk), dim3(512), smem8, s, d);
    else if (nw == 8) hipLaunchKernelGGL((conv3x3_patch_kernel<192, 8, false>), dim3(tiles, 1, sk), dim3(512), smem8, s, d);
    else if (d.K2) hipLaunchKernelGGL((conv3x3_patch_kernel<192, 4, true>), dim3(tiles, 1, sk), dim3(256), smem4, s, d);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<192, 4, false>), dim3(tiles, 1, sk), dim3(256), smem4, s, d);
    if (sk > 1) {
        launch_splitk_reduce(d, s);
    }
    return frido_check_launch("conv3x3_patch");
}

template <int NS, bool CONV>
int dispatch_tile(const FridoGemm& d, int tile, hipStream_t s) {
    if constexpr (NS == 2 && CONV) {
        if (tile == 20 || tile == 21) {
            const int rc = frido_launch_convgn(d, tile == 20 ? 256 : 128, s);
            if (rc == FRIDO_OK && d.splitk > 1) {        // the slices' partial sums: added up (fixed order) with the epilogue applied
                launch_splitk_reduce(d, s);
                return frido_check_launch("conv3x3_gn split-K reduction");
            }
            return rc;
        }
    }
    if constexpr (NS == 2 && CONV) {
        if (tile == 40) return frido_launch_convgn_tiny(d, s);
    }
    if (d.gn_x1) {
        frido_set_error("igemm: a descriptor with a fused GroupNorm input (gn_x1) runs on tile 20, 21 or 40 only");
        return FRIDO_EINVAL;
    }
    switch (tile) {
        case 1: return launch<128, 128, NS, CONV, 32>(d, s);
        case 2: return launch<128, 192, NS, CONV, 32>(d, s);
        case 4: return launch<128, 64, NS, CONV, 32>(d, s);
        case 5: return launch<64, 192, NS, CONV, 32>(d, s);
        case 6: return launch<64, 128, NS, CONV, 32>(d, s);
        default: break;
    }
    if constexpr (NS == 1 && CONV) {
        if (tile == 9) return launch_patch(d, 8, s);
        if (tile == 10) return launch_patch(d, 4, s);
    }
    if (tile == 7) return launch<256, 128, NS, CONV, 32>(d, s);      // 8-wave tiles: half the L2->LDS bytes per FLOP of the 128-row tiles
    if constexpr (NS == 2 && !CONV) {
        switch (tile) {
            case 31: return launch_kg2<128, 128>(d, s);
            case 33: return launch_kg2<64, 64>(d, s);
            case 34: return launch_kg2<128, 64>(d, s);
            case 35: return launch_kg2<64, 192>(d, s);
            case 36: return launch_kg2<64, 128>(d, s);
            default: break;
        }
    }
    if (tile >= 31 && tile <= 36) {
        frido_set_error("igemm: tiles 31..36 (K split inside the workgroup) exist for dense two-plane GEMMs only");
        return FRIDO_EINVAL;
    }
    if constexpr (NS == 2) {
        if (tile == 18) return launch<128, 192, NS, CONV, 32, true>(d, s);     // 128 x 192 on eight waves (bf16x3)
        if (tile == 19) return launch<256, 192, NS, CONV, 32>(d, s);           // 256 x 192 on eight waves, wave tile 64 x 96 (bf16x3)
    }
    if constexpr (NS == 1) {
        if (tile == 8) return launch<256, 256, NS, CONV, 32>(d, s);
    }
    if constexpr (NS == 1) {      // BK = 64 variants (bf16 mode only: the bf16x3 planes would not fit the LDS budget)
        const bool k64 = (d.K & 63) == 0 && (d.K2 & 63) == 0 && (!CONV || (d.Cin & 63) == 0);
        if (k64) switch (tile) {
            case 11: return launch<128, 128, NS, CONV, 64>(d, s);
            case 12: return launch<128, 192, NS, CONV, 64>(d, s);
            case 13: return launch<64, 64, NS, CONV, 64>(d, s);
            case 14: return launch<128, 64, NS, CONV, 64>(d, s);
            case 15: return launch<64, 192, NS, CONV, 64>(d, s);
            case 16: return launch<64, 128, NS, CONV, 64>(d, s);
            case 17: return launch<256, 128, NS, CONV, 64>(d, s);      // 8 waves, 3-deep ring of 48-KiB stages
            default: break;
        }
    }
    return launch<64, 64, NS, CONV, 32>(d, s);
}

int pick_tile(const FridoGemm& d) {
    // widest tile whose padded-N waste is smallest, as long as it still yields >= 128 workgroups
    auto tiles = [&](int bm, int bn) { return (int64_t)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * d.batch; };
    if (d.M >= 128 && d.N >= 96) {
        const int w128 = (d.N + 127) / 128 * 128, w192 = (d.N + 191) / 192 * 192;
        const int best = w192 <= w128 ? 2 : 1;
        if (tiles(128, best == 2 ? 192 : 128) >= 128) return best;
    }
    return 3;
}

}  // namespace

int frido_igemm_init() {
    int rc = 0;
#define FRIDO_SET_ALL(BM, BN) \
    rc |= set_attr<BM, BN, 1, true, 32>() | set_attr<BM, BN, 1, false, 32>() | set_attr<BM, BN, 2, true, 32>() | \
          set_attr<BM, BN, 2, false, 32>() | set_attr<BM, BN, 1, true, 64>() | set_attr<BM, BN, 1, false, 64>()
    FRIDO_SET_ALL(128, 128); FRIDO_SET_ALL(128, 192); FRIDO_SET_ALL(64, 64);
    FRIDO_SET_ALL(128, 64); FRIDO_SET_ALL(64, 192); FRIDO_SET_ALL(64, 128);
#undef FRIDO_SET_ALL
    rc |= set_attr<256, 128, 2, true, 32>() | set_attr<256, 128, 2, false, 32>();
    rc |= set_attr<128, 128, 2, false, 32, false, 2>() | set_attr<64, 64, 2, false, 32, false, 2>() | set_attr<128, 64, 2, false, 32, false, 2>() |
          set_attr<64, 192, 2, false, 32, false, 2>() | set_attr<64, 128, 2, false, 32, false, 2>();      // K split inside the workgroup (tiles 31..36)
    rc |= set_attr<128, 192, 2, true, 32, true>() | set_attr<128, 192, 2, false, 32, true>();
    rc |= set_attr<256, 192, 2, true, 32>() | set_attr<256, 192, 2, false, 32>();
    rc |= set_attr<256, 128, 1, true, 32>() | set_attr<256, 128, 1, false, 32>() | set_attr<256, 256, 1, true, 32>() |
          set_attr<256, 256, 1, false, 32>() | set_attr<256, 128, 1, true, 64>() | set_attr<256, 128, 1, false, 64>();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_kernel<192, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            PGeo<192, 8>::SMEM) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_kernel<192, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            PGeo<192, 8>::SMEM) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_kernel<192, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            PGeo<192, 4>::SMEM) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_kernel<192, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            PGeo<192, 4>::SMEM) != hipSuccess) {
        frido_set_error("igemm: cannot set dynamic LDS size of the patch kernel");
        rc |= 1;
    }
    if (frido_convgn_init()) {
        frido_set_error("igemm: cannot set dynamic LDS size of the fused GroupNorm + conv kernel");
        rc |= 1;
    }
    return rc ? FRIDO_EHIP : FRIDO_OK;
}

// The > 64 KiB dynamic-LDS opt-in is a per-device function attribute: frido_init() sets it on the device that is current at
// load time; any other device of the process gets it on its first GEMM.
static int ensure_device_attrs() {
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FRIDO_EHIP;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return FRIDO_OK;
    const int rc = frido_igemm_init();
    if (rc == FRIDO_OK) done.fetch_or(bit, std::memory_order_release);
    return rc;
}

extern "C" int64_t frido_gemm_workspace_bytes(const FridoGemm* d) {
    if (!d || d->splitk <= 1) return 0;
    // sk_mode 1 stores whole tiles: M, N padded to the largest tile edge of the family is an upper bound for every tile choice
    const int64_t Mp = d->sk_mode == 1 ? ((int64_t)d->M + 255) / 256 * 256 + 0 : d->M;
    const int64_t Np = d->sk_mode == 1 ? ((int64_t)d->N + 383) / 384 * 384 : d->N;
    return (int64_t)SK_HDR * 4 + (int64_t)d->splitk * Mp * Np * (int64_t)sizeof(float);
}

extern "C" int frido_gemm(const FridoGemm* dp, frido_stream_t stream) {
    FRIDO_REQUIRE(dp != nullptr, "null descriptor");
    const FridoGemm& d = *dp;
    FRIDO_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.batch > 0, "empty problem");
    FRIDO_REQUIRE((d.K & 31) == 0 && (d.K2 & 63) == 0, "K must be a multiple of 32, K2 of 64 (zero-pad the operands)");
    FRIDO_REQUIRE(d.K2 == 0 || (((d.A2 && (d.lda2 & 7) == 0) || (d.gn_x1 && d.raw_x1)) && (d.K & 63) == 0 && d.batch == 1), "bad second A operand");
    FRIDO_REQUIRE(d.nsplit == 1 || d.nsplit == 2, "nsplit must be 1 or 2");
    FRIDO_REQUIRE((d.A || d.gn_x1) && (d.B || (d.tile == 40 && d.w_f32)), "null operand");
    FRIDO_REQUIRE(!d.gn_x1 || d.tile == 20 || d.tile == 21 || d.tile == 40, "a fused GroupNorm input (gn_x1) needs tile 20, 21 or 40");
    FRIDO_REQUIRE(d.out_f32 || d.out_op || d.out_u8, "no output");
    FRIDO_REQUIRE(!d.out_u8 || ((d.u8_mode == 1 || d.u8_mode == 2) && d.ldu8 >= d.N && d.splitk <= 1 && !d.gn_part && !d.geglu && !d.up2_phase &&
                                d.batch == 1),
                  "out_u8: u8_mode 1 (custom_to_np) or 2 (custom_to_pil), ldu8 >= N, no split-K / gn_part / geglu / phases / batching");
    FRIDO_REQUIRE((d.ldb & 7) == 0 && (d.b_bs & 7) == 0, "B rows must be 16-byte aligned");
    if (d.conv) {
        FRIDO_REQUIRE((d.Cin & 31) == 0 && d.Cin > 0, "conv Cin must be a multiple of 32");
        FRIDO_REQUIRE(d.K == d.kh * d.kw * d.Cin, "conv K != kh*kw*Cin");
        FRIDO_REQUIRE(d.Ho > 0 && d.Wo > 0 && d.M % (d.Ho * d.Wo) == 0, "conv M must be Bimg*Ho*Wo");
        FRIDO_REQUIRE(d.stride >= 1 && d.up_shift >= 0 && d.dn_shift >= 0, "bad conv geometry");
        FRIDO_REQUIRE(d.batch == 1 || d.up2_phase == 5, "conv mode is not batched (except the four upsample phases)");
    } else {
        FRIDO_REQUIRE((d.lda & 7) == 0 && (d.a_bs & 7) == 0, "A rows must be 16-byte aligned");
    }
    if (d.rowvec) FRIDO_REQUIRE(d.rows_per_vec > 0, "rows_per_vec");
    if (d.batch_inner > 1) FRIDO_REQUIRE(d.batch % d.batch_inner == 0 && !d.residual, "batch must be outer * inner; no residual");
    if (d.up2_phase) {
        FRIDO_REQUIRE(d.conv && d.up2_phase >= 1 && d.up2_phase <= 5 && d.splitk <= 1 && !d.residual && !d.rowvec && !d.geglu &&
                          d.batch == (d.up2_phase == 5 ? 4 : 1) && d.a_bs == 0 && d.of_bs == 0 && d.oo_bs == 0 && d.batch_inner <= 1,
                      "upsample phase conv: conv only, no split-K / residual / rowvec / geglu; 5 = all four phases as batch 4");
        FRIDO_REQUIRE((d.Ho & (d.Ho - 1)) == 0 && (d.Wo & (d.Wo - 1)) == 0 && d.M % (d.Ho * d.Wo) == 0 && d.kh == 2 && d.kw == 2,
                      "upsample phase conv: 2x2 taps, Ho, Wo powers of two");
    }
    if (d.geglu) {
        FRIDO_REQUIRE((d.N & 31) == 0 && d.out_op && !d.out_f32 && d.batch == 1 && d.splitk <= 1 && !d.residual && !d.rowvec,
                      "geglu epilogue: N % 32 == 0, operand output only, no split-K / residual / rowvec");
    }
    if (d.gn_part) {
        FRIDO_REQUIRE(d.nsplit == 2 && d.out_f32 && !d.out_bf16 && !d.out_op && d.act == FRIDO_ACT_NONE && !d.row_bias && !d.geglu &&
                          !d.up2_phase && (d.splitk <= 1 || d.sk_mode == 1) && d.batch == 1 && !(d.flags & 18) && (d.N & 7) == 0 && (d.M & 31) == 0 &&
                          ((d.ldo | d.ldr | d.ldv | d.of_bs | d.res_bs) & 7) == 0 && !(d.residual && d.res_bf16) &&
                          (!d.rowvec || d.rows_per_vec >= (1 << 29)),
                      "gn_part needs the store-from-registers f32 epilogue (see frido_hip.h)");
    }
    if (d.splitk > 1) {
        FRIDO_REQUIRE(d.batch == 1 && d.ws != nullptr, "split-K needs batch == 1 and a workspace");
        FRIDO_REQUIRE(d.splitk <= ((d.K + d.K2) >> 6), "more K slices than k-tiles");
        FRIDO_REQUIRE(d.sk_mode >= 0 && d.sk_mode <= 2, "sk_mode must be 0, 1 or 2");
        if (d.sk_mode == 2) {
            const int tl = d.tile ? d.tile : pick_tile(d);
            FRIDO_REQUIRE(tl != 9 && tl != 10 && tl != 20 && tl != 21, "sk_mode 2 (reduction left to the consumer): ring kernels only");
            FRIDO_REQUIRE(d.out_f32 && !d.out_bf16 && !d.out_op && !d.out_u8 && d.act == FRIDO_ACT_NONE && !d.row_bias && !d.geglu &&
                              !d.up2_phase && d.ldo == d.N && (d.N & 7) == 0 && !(d.residual && d.res_bf16),
                          "sk_mode 2: plain f32 output [M][N] whose epilogue splitk_reduce8 could run (see frido_hip.h)");
        }
    }
    if (const int arc = ensure_device_attrs()) return arc;
    const int tile = d.tile ? d.tile : pick_tile(d);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d.nsplit == 1) return d.conv ? dispatch_tile<1, true>(d, tile, s) : dispatch_tile<1, false>(d, tile, s);
    return d.conv ? dispatch_tile<2, true>(d, tile, s) : dispatch_tile<2, false>(d, tile, s);
}
