// Flash-style single-head attention for LONG key sequences (gfx950): O = softmax(alpha * Q K^T) V without ever forming the
// [Nq][Nk] score matrix -- online softmax over 32-key tiles.  Covers the self-attention of the U-Net's large planes
// (frido/modules/attention.py:170-193: 1024 / 4096 tokens, d = C = 384 / 576) and the VQGAN AttnBlocks
// (taming/modules/diffusionmodules/model.py:168-192: 4096 tokens at 256^2, 16384 at 512^2, d = 128 .. 512), which the
// three-kernel path (QK^T GEMM -> f32 scores -> row softmax -> PV GEMM) served before: 1 GiB of scores per image at
// 16384 keys.  The heads here are SINGLE heads with d = C up to 576, so the tiling is over d, not over heads.
//
// Everything is computed TRANSPOSED so that one query lives in one MFMA column (= lane & 15) through the whole kernel:
//     S^T[key][q] = K_tile . Q^T      A = K fragment (LDS), B = Q fragment (registers, loaded once)
//     O^T[c][q]  += V^T_tile . P^T    A = V^T fragment (LDS), B = P fragment (registers)
//   * v_mfma_f32_16x16x32_bf16 leaves lane l with D[row = 4 (l>>4) + e][col = l & 15]: every lane's S values belong to ONE
//     query, so the row max / row sum are 8 in-register ops + two cross-lane steps (xor 16, 32), and the running
//     (max, sum, rescale factor) of a query sit in the lane that owns its O^T column: no broadcast, no LDS round trip;
//   * the K rows of a 32-key tile are stored PERMUTED in LDS (key 8a + 4t + b at row 16t + 4a + b; the permutation rides
//     on the per-lane SOURCE address of the LDS-DMA) so that the two S^T fragments of a lane hold keys 8g .. 8g+7 -- exactly
//     the 8 k-slots of the B operand of the PV MFMA: P goes from accumulator registers to operand registers by a pack;
//   * workgroup = NW waves x 16 queries.  Two 2-stage phases per key tile, each covered by the other's DMA:
//       phase 1: wait K_j, barrier, issue V^T_j, S^T = K_j Q^T, online softmax, rescale O^T
//       phase 2: wait V^T_j, barrier, issue K_{j+1}, O^T += V^T_j P^T
//     K / V^T tiles go L2 -> LDS by global_load_lds_dwordx4 in the [rows][64 B] sub-tile layout of igemm.hip (XOR slot
//     swizzle on the source address, conflict-free ds_read_b128 fragments); fragment reads are inline asm so that hipcc does
//     not drain the DMA queue in front of them;
//   * bf16x3 mode (NS = 2): Q, K, V^T and P carry hi + lo bf16 planes, products are hi*hi + hi*lo + lo*hi in f32.
#include "common.h"
#include <atomic>
#include <cstdlib>

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
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int NS>
__device__ __forceinline__ f32x4 mma3(const bf16x8 (&a)[NS], const bf16x8 (&b)[NS], f32x4 acc) {
    if constexpr (NS == 2) {
        acc = mfma_op<NS>(a[1], b[0], acc);
        acc = mfma_op<NS>(a[0], b[1], acc);
    }
    return mfma_op<NS>(a[0], b[0], acc);
}

__device__ __forceinline__ float xor_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// OS > 1: the O^T rows (output channels) are split over gridDim.y workgroups, each recomputing S^T (bit-identically) -- used
// where hi + lo planes of Q plus all D / 16 accumulator fragments do not fit the register file (bf16x3 mode, d >= 512).
template <int D, int NS, int NW, int OS>
struct FGeo {
    static constexpr int BKV = 32;                       // keys per tile
    static constexpr int KS = D / 32, CT = D / 16 / OS;  // QK^T k-steps; O^T row fragments of this workgroup
    static constexpr int KPL = BKV * D * 2;              // bytes of one K plane of a tile  ([KS][32 rows][64 B])
    static constexpr int VPL = D / OS * BKV * 2;         // bytes of one V^T plane of a tile ([D / OS rows][64 B])
    static constexpr int KBUF = NS * KPL, VBUF = NS * VPL;
    // two tiles in LDS where they fit: K / V^T of tile j + 1 stream in during ALL of tile j (one barrier per tile).  With one
    // tile (bf16x3 mode at d >= 384) the two phases of a tile cover each other's DMA instead.
    static constexpr bool DBUF = 2 * (KBUF + VBUF) <= 163840;
    static constexpr int TILE = KBUF + VBUF;
    static constexpr int SMEM = (DBUF ? 2 : 1) * TILE;
    static constexpr int KP = KS * 2, VP = D / 16 / OS;  // 1-KiB DMA pieces per plane
    static_assert(D % (32 * OS) == 0 && SMEM <= 163840, "LDS budget");
    static_assert(NW % 2 == 0, "a wave's K pieces all belong to one 16-row half");
};

// FridoAttnSmall descriptor (include/frido_hip.h); requirements checked by the launcher.
template <int D, int NS, int NW, int OS>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 1) void flash_attn_kernel(const FridoAttnSmall d) {
    using G = FGeo<D, NS, NW, OS>;
    constexpr int KS = G::KS, CT = G::CT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, g = lane >> 4;
    constexpr int BQ = NW * 16;

    // workgroup -> (sample, query block); consecutive logical ids (one sample's blocks share K / V) land on one XCD's L2
    const int qblocks = (d.Nq + BQ - 1) / BQ;
    const int nb = qblocks * d.B;
    int bid = blockIdx.x;
    {
        const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    }
    const int z = bid / qblocks, qb = bid - z * qblocks;
    const int q0 = qb * BQ + wave * 16;                  // first query of this wave (within the sample)
    int qrow = q0 + r;
    const bool q_ok = qrow < d.Nq;
    qrow = q_ok ? qrow : d.Nq - 1;
    const int64_t grow = (int64_t)z * d.Nq + qrow;       // global row of Q / output

    // ---- Q fragments: lane holds Q[q = r][32 ks + 8 g .. +8] for every k-step (B operand of S^T = K Q^T) ----
    bf16x8 qf[KS][NS];
    {
        const frido_bf16* qp = d.Q + grow * d.ldq + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks][0] = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
            if constexpr (NS == 2) qf[ks][1] = *reinterpret_cast<const bf16x8*>(qp + d.q_lo + ks * 32);
        }
    }

    // ---- DMA source offsets.  A 1-KiB piece = 16 LDS rows x 64 B; lane l lands at row l >> 2, physical slot l & 3, i.e. it
    //      must fetch logical 16-byte slot (l & 3) ^ swz(row) of that row (igemm.hip's BK = 32 swizzle).
    const int lrow = lane >> 2;
    const int lq = (lane & 3) ^ ((4 - ((lrow >> 2) & 3)) & 3);
    // K piece p of a plane: sub-tile ks = p >> 1, half h = p & 1 (LDS rows 16 h .. 16 h + 15 of the sub-tile); LDS row
    // 16 h + 4 a + b holds key 8 a + 4 h + b.  This wave's pieces are p = wave + NW i: h = wave & 1 for all of them.
    const int kh = wave & 1;
    const int key_l = 8 * (lrow >> 2) + 4 * kh + (lrow & 3);            // key (within the tile) this lane fetches
    const frido_bf16* Kb = d.K + (int64_t)z * d.k_bs + lq * 8;
    const int c_base = (int)blockIdx.y * (D / OS);       // first output channel of this workgroup
    const frido_bf16* Vb = d.VT + (int64_t)z * d.vt_bs + (int64_t)(c_base + lrow) * d.ldvt + lq * 8;
    const int ntiles = (d.Nk + G::BKV - 1) / G::BKV;

    auto issue_k = [&](int j, int buf = 0) {
        int key = j * G::BKV + key_l;
        key = key < d.Nk ? key : d.Nk - 1;               // rows past Nk: any valid row (their scores are masked)
        const frido_bf16* src = Kb + (int64_t)key * d.ldk;
#pragma unroll
        for (int p = 0; p < NS; ++p)
#pragma unroll
            for (int i = 0; i < (G::KP + NW - 1) / NW; ++i) {
                const int piece = wave + NW * i;
                if (piece < G::KP) {
                    const int ks = piece >> 1;
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + (p ? d.k_lo : 0) + ks * 32),
                                                     (lptr_t)(smem + buf * G::TILE + p * G::KPL + piece * 1024), 16, 0, 0);
                }
            }
    };
    auto issue_v = [&](int j, int buf = 0) {
        const frido_bf16* src = Vb + j * G::BKV;
#pragma unroll
        for (int p = 0; p < NS; ++p)
#pragma unroll
            for (int i = 0; i < (G::VP + NW - 1) / NW; ++i) {
                const int piece = wave + NW * i;
                if (piece < G::VP)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + (p ? d.vt_lo : 0) + (int64_t)piece * 16 * d.ldvt),
                                                     (lptr_t)(smem + buf * G::TILE + G::KBUF + p * G::VPL + piece * 1024), 16, 0, 0);
            }
    };

    // the same pieces one SLOT at a time (slot s < NPK: K piece wave + NW s, else V^T piece wave + NW (s - NPK)): the
    // double-buffered loop deals the next tile's slots over the k-steps of S^T = K Q^T instead of issuing all of them
    // (12 per wave at d = 384, 60-185 cycles each) in front of the first MFMA
    constexpr int NPK = (G::KP + NW - 1) / NW, NPV = (G::VP + NW - 1) / NW, NSLOT = NPK + NPV;
    auto issue_slot = [&](int sl, const frido_bf16* ksrc, const frido_bf16* vsrc, int buf) {      // sl: constant after unrolling
        if (sl < NPK) {
            const int piece = wave + NW * sl;
            if (piece < G::KP) {
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    __builtin_amdgcn_global_load_lds((gptr_t)(ksrc + (p ? d.k_lo : 0) + (piece >> 1) * 32),
                                                     (lptr_t)(smem + buf * G::TILE + p * G::KPL + piece * 1024), 16, 0, 0);
            }
        } else {
            const int piece = wave + NW * (sl - NPK);
            if (piece < G::VP) {
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    __builtin_amdgcn_global_load_lds((gptr_t)(vsrc + (p ? d.vt_lo : 0) + (int64_t)piece * 16 * d.ldvt),
                                                     (lptr_t)(smem + buf * G::TILE + G::KBUF + p * G::VPL + piece * 1024), 16, 0, 0);
            }
        }
    };
    // fragment read address inside a [16 rows][64 B] chunk: row r, logical slot g
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned frag = (unsigned)(r * 64 + ((g ^ ((4 - ((r >> 2) & 3)) & 3)) << 4));
    const unsigned k_frag0 = lds0 + frag, v_frag0 = lds0 + G::KBUF + frag;

    f32x4 o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1.0e30f, l_run = 0.f;
    const float alpha = d.alpha;

    issue_k(0);
    if constexpr (G::DBUF) issue_v(0);
    for (int j = 0; j < ntiles; ++j) {
        // ================= phase 1: S^T = K_j Q^T, online softmax =================
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();              // K_j (and V^T_j) visible; every wave has finished tile j - 1
        unsigned k_frag = k_frag0, v_frag = v_frag0;
        if constexpr (G::DBUF) {
            k_frag += (j & 1) * G::TILE;
            v_frag += (j & 1) * G::TILE;
        } else {
            issue_v(j);
        }
        const bool nxt = G::DBUF && j + 1 < ntiles;          // deal tile j + 1's DMA slots over the k-steps below
        const frido_bf16* ksrc_n = Kb;
        const frido_bf16* vsrc_n = Vb + (j + 1) * G::BKV;
        if (nxt) {
            int key = (j + 1) * G::BKV + key_l;
            key = key < d.Nk ? key : d.Nk - 1;
            ksrc_n = Kb + (int64_t)key * d.ldk;
        }
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
        {
            // the fragments of step ks + 1 are fetched under the MFMAs of step ks.  (r03: in bf16x3 mode too -- the 4-wave form runs
            // one wave per SIMD, which may use all 512 registers; before, every k-step paid a full LDS round trip and then three
            // DEPENDENT MFMAs per accumulator: 160 us for the 77 GFLOP of a 32x32-plane self-attention.)
            constexpr int NB = 2;
            bf16x8 kf[NB][2][NS];                  // [buffer][half][plane]
#pragma unroll
            for (int p = 0; p < NS; ++p) {
                kf[0][0][p] = lds_read128(k_frag + p * G::KPL);
                kf[0][1][p] = lds_read128(k_frag + p * G::KPL + 1024);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int p = 0; p < NS; ++p) {
                        kf[(ks + 1) & 1][0][p] = lds_read128(k_frag + p * G::KPL + (ks + 1) * 2048);
                        kf[(ks + 1) & 1][1][p] = lds_read128(k_frag + p * G::KPL + (ks + 1) * 2048 + 1024);
                    }
                    wait_lgkm<2 * NS>();
                } else {
                    wait_lgkm<0>();
                }
                if constexpr (NS == 2) {
                    // hi*lo, lo*hi, hi*hi of the two score fragments interleaved: consecutive MFMAs never share an accumulator
                    const bf16x8(&ka)[2] = kf[ks & 1][0];
                    const bf16x8(&kb)[2] = kf[ks & 1][1];
                    s0 = mfma_op<NS>(ka[1], qf[ks][0], s0);
                    s1 = mfma_op<NS>(kb[1], qf[ks][0], s1);
                    s0 = mfma_op<NS>(ka[0], qf[ks][1], s0);
                    s1 = mfma_op<NS>(kb[0], qf[ks][1], s1);
                    s0 = mfma_op<NS>(ka[0], qf[ks][0], s0);
                    s1 = mfma_op<NS>(kb[0], qf[ks][0], s1);
                } else {
                    s0 = mma3<NS>(kf[ks & 1][0], qf[ks], s0);
                    s1 = mma3<NS>(kf[ks & 1][1], qf[ks], s1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (nxt) {
#pragma unroll
                    for (int sl = ks * NSLOT / KS; sl < (ks + 1) * NSLOT / KS; ++sl) issue_slot(sl, ksrc_n, vsrc_n, (j + 1) & 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // lane holds the scores of query r against keys 32 j + 8 g + {0..3} (s0) and + {4..7} (s1)
        float sv[8] = {s0[0] * alpha, s0[1] * alpha, s0[2] * alpha, s0[3] * alpha, s1[0] * alpha, s1[1] * alpha, s1[2] * alpha, s1[3] * alpha};
        if ((j + 1) * G::BKV > d.Nk) {             // ragged last tile (uniform branch)
            const int k0 = j * G::BKV + 8 * g;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (k0 + i >= d.Nk) sv[i] = -1.0e30f;
        }
        float tmax = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
        tmax = xor_max(tmax);
        const float m_new = fmaxf(m_run, tmax);
        const float corr = __expf(m_run - m_new);
        float pv[8], rs = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pv[i] = __expf(sv[i] - m_new);
            rs += pv[i];
        }
        rs = xor_sum(rs);
        l_run = l_run * corr + rs;
        m_run = m_new;
        // P fragment (B operand of O^T += V^T P^T): k-slot 8 g + i <-> key 32 j + 8 g + i
        bf16x8 pf[NS];
        {
            uint32_t h[8], l[8];
#pragma unroll
            for (int i = 0; i < 8; i += 2) { split_op2(pv[i], pv[i + 1], NS, h[i], l[i]); h[i + 1] = 0u; l[i + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
            u32x4 ph = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
            pf[0] = __builtin_bit_cast(bf16x8, ph);
            if constexpr (NS == 2) {
                u32x4 pl = {l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
                pf[1] = __builtin_bit_cast(bf16x8, pl);
            }
        }
        if (__any(corr != 1.0f)) {                 // the running max moved for some query of this wave: rescale O^T
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                o[c][0] *= corr; o[c][1] *= corr; o[c][2] *= corr; o[c][3] *= corr;
            }
        }
        // ================= phase 2: O^T += V^T_j P^T =================
        if constexpr (!G::DBUF) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();          // V^T_j visible; every wave has finished reading K_j
            if (j + 1 < ntiles) issue_k(j + 1);
        }
        if constexpr (NS == 1) {
            bf16x8 vf[2][NS];
            vf[0][0] = lds_read128(v_frag);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (c + 1 < CT) {
                    vf[(c + 1) & 1][0] = lds_read128(v_frag + (c + 1) * 1024);
                    wait_lgkm<1>();
                } else {
                    wait_lgkm<0>();
                }
                o[c] = mma3<NS>(vf[c & 1], pf, o[c]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // bf16x3 (r03): output row fragments in PAIRS -- the next pair's hi / lo V^T fragments are fetched under this pair's six
            // MFMAs, whose three product terms alternate between the two accumulators
            static_assert(CT % 2 == 0, "output row fragments are processed in pairs");
            bf16x8 vf[2][2][NS];                   // [buffer][fragment of the pair][plane]
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int p = 0; p < NS; ++p) vf[0][q][p] = lds_read128(v_frag + p * G::VPL + q * 1024);
#pragma unroll
            for (int c = 0; c < CT; c += 2) {
                const int cur = (c >> 1) & 1;
                if (c + 2 < CT) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int p = 0; p < NS; ++p) vf[cur ^ 1][q][p] = lds_read128(v_frag + p * G::VPL + (c + 2 + q) * 1024);
                    wait_lgkm<2 * NS>();
                } else {
                    wait_lgkm<0>();
                }
                o[c] = mfma_op<NS>(vf[cur][0][1], pf[0], o[c]);
                o[c + 1] = mfma_op<NS>(vf[cur][1][1], pf[0], o[c + 1]);
                o[c] = mfma_op<NS>(vf[cur][0][0], pf[1], o[c]);
                o[c + 1] = mfma_op<NS>(vf[cur][1][0], pf[1], o[c + 1]);
                o[c] = mfma_op<NS>(vf[cur][0][0], pf[0], o[c]);
                o[c + 1] = mfma_op<NS>(vf[cur][1][0], pf[0], o[c + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: lane holds O^T[c = 16 ct + 4 g + e][q = r]: four consecutive channels of its query per fragment ----
    if (!q_ok) return;
    const float inv = 1.0f / l_run;
    // (validated: f32 stream output, this workgroup owns whole rows.  Instantiated for the U-Net's head widths only: in the d = 128
    //  kernel the extra live range pushed hipcc into scratch)
    const bool ln = OS == 1 && NS == 2 && D >= 256 && d.ln_op != nullptr;
    float lsum = 0.f;
    bool sat = false;
    status_raise(false, g == 0 && stat_bad(m_run, inv));       // softmax statistics of this query
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = c_base + c * 16 + g * 4;
        float4 v = make_float4(o[c][0] * inv, o[c][1] * inv, o[c][2] * inv, o[c][3] * inv);
        if (d.out_act) {                             // residual-stream form: O + bias + residual
            if (d.bias) {
                const float4 bb = *reinterpret_cast<const float4*>(d.bias + col);
                v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            }
            if (d.residual) {
                const float4 rr = load_act4(d.residual, grow * d.ldr + col, d.act_bf16);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            const int64_t oo = grow * d.ld_act + col;
            if (d.act_bf16)
                *reinterpret_cast<uint2*>(reinterpret_cast<frido_bf16*>(d.out_act) + oo) =
                    make_uint2(f32_to_bf16_bits(v.x) | (f32_to_bf16_bits(v.y) << 16), f32_to_bf16_bits(v.z) | (f32_to_bf16_bits(v.w) << 16));
            else
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.out_act) + oo) = v;
            if (ln) {                                    // keep the stored row: its LayerNorm follows
                o[c] = f32x4{v.x, v.y, v.z, v.w};
                lsum += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            const float vv[4] = {v.x, v.y, v.z, v.w};
            store_op4(d.out_op, d.out_lo, NS, grow * d.ldo + col, vv);
            if (NS == 2) sat |= op_sat4(vv);
        }
    }
    if (ln) {
        // LayerNorm of the stream row (r03: norm2 of the transformer block, attention.py:225): the four lanes r, r + 16, r + 32,
        // r + 48 hold the D channels of query r, so mean / variance are two in-wave reductions; same two-pass arithmetic as
        // layernorm_kernel
        const float mean = xor_sum(lsum) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float a0 = o[c][0] - mean, a1 = o[c][1] - mean, a2 = o[c][2] - mean, a3 = o[c][3] - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
        const float rstd = 1.0f / sqrtf(xor_sum(q) * (1.0f / D) + d.ln_eps);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int col = c * 16 + g * 4;
            const float4 w = *reinterpret_cast<const float4*>(d.ln_w + col), bb = *reinterpret_cast<const float4*>(d.ln_b + col);
            const float y[4] = {(o[c][0] - mean) * rstd * w.x + bb.x, (o[c][1] - mean) * rstd * w.y + bb.y,
                                (o[c][2] - mean) * rstd * w.z + bb.z, (o[c][3] - mean) * rstd * w.w + bb.w};
            store_op4(d.ln_op, d.ln_lo, NS, grow * d.ld_ln + col, y);
            if (NS == 2) sat |= op_sat4(y);
        }
        status_raise(false, g == 0 && stat_bad(mean, rstd));
    }
    status_raise(sat);
}

// =====================================================================================================================
// (r05) flash_ds_kernel<D>: the two-plane (parity) form with the HEAD DIMENSION SPLIT OVER A WAVE PAIR -- eight waves over the
// same 64 queries, two waves per SIMD.  r04's PMC of the 4-wave form above (one wave per SIMD, 256+ registers): 21.5 % MFMA-busy,
// 37 % of the wave-cycles parked -- every 32-key phase starts with twelve LDS-DMA instructions per wave (60 - 185 issue cycles
// each, MI355X_MICROARCH.md) and a softmax that nothing overlaps, because the SIMD has no second wave to issue MFMAs from.
// Here waves 2 p and 2 p + 1 share the 16 queries of pair p:
//     phase 1   wave h (= wave & 1) forms the PARTIAL scores over its half of d (k-steps h KS/2 ..), parks them in LDS;
//     barrier   (the one that publishes V^T_j anyway) -- then both waves add the two halves (a + b == b + a: identical bits in
//               both), run the same online softmax and hold the same P fragment;
//     phase 2   wave h accumulates its HALF of the output channels: O^T[h D/2 ..] += V^T_j[h D/2 ..] P^T.
// Per wave: half the Q fragments (KS/2 x 2 planes), half the accumulators, half the DMA pieces, half the fragment reads -- < 200
// registers, so two waves share a SIMD and one's DMA issue / softmax / LDS waits run under the other's MFMAs.  The K / V^T tile
// and its LDS traffic per query are unchanged (one 98-KiB tile at d = 384 + a 16-KiB exchange buffer: one workgroup per CU).
// d = 512 fits too (131 KiB + 16): the 4-wave form needs two workgroups per query block there, each recomputing S^T (OS = 2).
// The fused LayerNorm (ln_op) exchanges the pair's row sums through the same buffer: mean / variance over all D channels, two-pass.
template <int D>
struct FDGeo {
    static constexpr int NS = 2, NW = 8, BKV = 32;
    static constexpr int KS = D / 32, KSH = KS / 2;      // QK^T k-steps, per wave
    static constexpr int CT = D / 16, CTH = CT / 2;      // O^T row fragments, per wave
    static constexpr int KPL = BKV * D * 2, VPL = D * BKV * 2;
    static constexpr int KBUF = NS * KPL, VBUF = NS * VPL, TILE = KBUF + VBUF;
    static constexpr int XCH = NW * 64 * 8 * 4;          // partial-score exchange: 8 floats per lane
    // K DOUBLE-buffered where it fits (d = 384: 2 x 48 + 48 + 16 = exactly 160 KiB; d = 256): K_{j+1} then streams in during ALL of tile j
    // instead of having to land inside the 36-MFMA P V phase (measured: the single-buffered form spends ~1.3 us per phase waiting for
    // the other buffer's 48 KiB -- both phases are shorter than one DMA round trip); only V^T_j's load stays exposed to phase 1
#ifndef FLASH_DS_KDB
#define FLASH_DS_KDB 1
#endif
#ifndef FLASH_DS_COUNTED
#define FLASH_DS_COUNTED 1
#endif
    static constexpr bool KDB = FLASH_DS_KDB && 2 * KBUF + VBUF + XCH <= 163840;
    static constexpr int KSTRIDE = KDB ? KBUF : 0;       // byte distance between the two K buffers
    static constexpr int V0 = (KDB ? 2 : 1) * KBUF;      // V^T tile offset
    static constexpr int X0 = V0 + VBUF;                 // exchange buffer offset
    static constexpr int SMEM = X0 + XCH;
    static constexpr int KP = KS * 2, VP = D / 16;       // 1-KiB DMA pieces per plane
    static_assert(KS % 2 == 0 && CTH % 2 == 0 && SMEM <= 163840, "d-split geometry / LDS budget");
    static_assert(!KDB || KP % NW == 0, "the counted vmcnt wait of the double-buffered form needs every wave to issue the same number of K pieces");
};

template <int D>
__global__ __launch_bounds__(512, 1) void flash_ds_kernel(const FridoAttnSmall d) {
    using G = FDGeo<D>;
    constexpr int NS = 2, NW = 8, KSH = G::KSH, CTH = G::CTH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pair = wave >> 1, hf = wave & 1;
    const int r = lane & 15, g = lane >> 4;
    constexpr int BQ = 64;

    const int qblocks = (d.Nq + BQ - 1) / BQ;
    const int nb = qblocks * d.B;
    int bid = blockIdx.x;
    {
        const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    }
    const int z = bid / qblocks, qb = bid - z * qblocks;
    const int q0 = qb * BQ + pair * 16;
    int qrow = q0 + r;
    const bool q_ok = qrow < d.Nq;
    qrow = q_ok ? qrow : d.Nq - 1;
    const int64_t grow = (int64_t)z * d.Nq + qrow;

    // ---- Q fragments of this wave's half of d ----
    bf16x8 qf[KSH][NS];
    {
        const frido_bf16* qp = d.Q + grow * d.ldq + hf * (D / 2) + g * 8;
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
            qf[ks][0] = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
            qf[ks][1] = *reinterpret_cast<const bf16x8*>(qp + d.q_lo + ks * 32);
        }
    }

    // ---- DMA (the 4-wave form's piece layout, dealt over eight waves: piece = wave + 8 i keeps h = wave & 1) ----
    const int lrow = lane >> 2;
    const int lq = (lane & 3) ^ ((4 - ((lrow >> 2) & 3)) & 3);
    const int key_l = 8 * (lrow >> 2) + 4 * hf + (lrow & 3);
    const frido_bf16* Kb = d.K + (int64_t)z * d.k_bs + lq * 8;
    const frido_bf16* Vb = d.VT + (int64_t)z * d.vt_bs + (int64_t)lrow * d.ldvt + lq * 8;
    const int ntiles = (d.Nk + G::BKV - 1) / G::BKV;
    auto issue_k = [&](int j) {
        int key = j * G::BKV + key_l;
        key = key < d.Nk ? key : d.Nk - 1;
        const frido_bf16* src = Kb + (int64_t)key * d.ldk;
        unsigned char* dst = smem + (j & 1) * G::KSTRIDE;
#pragma unroll
        for (int i = 0; i < (G::KP + NW - 1) / NW; ++i) {
            const int piece = wave + NW * i;
            if (piece < G::KP) {
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + (p ? d.k_lo : 0) + (piece >> 1) * 32),
                                                     (lptr_t)(dst + p * G::KPL + piece * 1024), 16, 0, 0);
            }
        }
    };
    auto issue_v = [&](int j) {
        const frido_bf16* src = Vb + j * G::BKV;
#pragma unroll
        for (int i = 0; i < (G::VP + NW - 1) / NW; ++i) {
            const int piece = wave + NW * i;
            if (piece < G::VP) {
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + (p ? d.vt_lo : 0) + (int64_t)piece * 16 * d.ldvt),
                                                     (lptr_t)(smem + G::V0 + p * G::VPL + piece * 1024), 16, 0, 0);
            }
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const unsigned frag = (unsigned)(r * 64 + ((g ^ ((4 - ((r >> 2) & 3)) & 3)) << 4));
    const unsigned k_frag0 = lds0 + frag + (unsigned)(hf * KSH * 2048);              // this wave's k-steps (K buffer 0)
    const unsigned v_frag = lds0 + G::V0 + frag + (unsigned)(hf * CTH * 1024);        // this wave's output channels
    const unsigned x_mine = lds0 + G::X0 + (unsigned)((wave * 64 + lane) * 32);
    const unsigned x_peer = lds0 + G::X0 + (unsigned)(((wave ^ 1) * 64 + lane) * 32);

    f32x4 o[CTH];
#pragma unroll
    for (int c = 0; c < CTH; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1.0e30f, l_run = 0.f;
    const float alpha = d.alpha;

    issue_k(0);
    for (int j = 0; j < ntiles; ++j) {
        // ================= phase 1: partial S^T over this wave's half of d =================
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();              // K_j visible; every wave has finished tile j - 1 (its V^T reads, its exchange reads)
        issue_v(j);
        if constexpr (G::KDB) {                    // the other K buffer was last read in phase 1 of tile j - 1
            if (j + 1 < ntiles) issue_k(j + 1);
        }
        const unsigned k_frag = k_frag0 + (unsigned)((j & 1) * G::KSTRIDE);
        f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
        {
            bf16x8 kf[2][2][NS];                   // [buffer][half][plane]
#pragma unroll
            for (int p = 0; p < NS; ++p) {
                kf[0][0][p] = lds_read128(k_frag + p * G::KPL);
                kf[0][1][p] = lds_read128(k_frag + p * G::KPL + 1024);
            }
#pragma unroll
            for (int ks = 0; ks < KSH; ++ks) {
                if (ks + 1 < KSH) {
#pragma unroll
                    for (int p = 0; p < NS; ++p) {
                        kf[(ks + 1) & 1][0][p] = lds_read128(k_frag + p * G::KPL + (ks + 1) * 2048);
                        kf[(ks + 1) & 1][1][p] = lds_read128(k_frag + p * G::KPL + (ks + 1) * 2048 + 1024);
                    }
                    wait_lgkm<2 * NS>();
                } else {
                    wait_lgkm<0>();
                }
                const bf16x8(&ka)[2] = kf[ks & 1][0];
                const bf16x8(&kb)[2] = kf[ks & 1][1];
                s0 = mfma_op<NS>(ka[1], qf[ks][0], s0);
                s1 = mfma_op<NS>(kb[1], qf[ks][0], s1);
                s0 = mfma_op<NS>(ka[0], qf[ks][1], s0);
                s1 = mfma_op<NS>(kb[0], qf[ks][1], s1);
                s0 = mfma_op<NS>(ka[0], qf[ks][0], s0);
                s1 = mfma_op<NS>(kb[0], qf[ks][0], s1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // park the partial scores for the partner wave.  The stores are inline asm (a C++ LDS store would be fenced with vmcnt(0) behind the
        // DMA just issued), so the compiler's hazard recogniser does NOT see that they read registers the last MFMAs are still writing:
        // the matrix pipe has no interlock towards an LDS / VMEM read of its destination, and without the wait states below the store of
        // s1 could pick up the accumulator BEFORE its last product term landed -- a timing-dependent error of one term of the scores
        // (found as run-to-run differences of 1e-5 when a second process shared the GPU; 2 x 16 wait states cover the 8-pass MFMA's 18).
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(s0), "+v"(s1));
        asm volatile("ds_write_b128 %0, %1" ::"v"(x_mine), "v"(s0) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:16" ::"v"(x_mine), "v"(s1) : "memory");
        // ================= phase 2 head: V^T_j and the partner's partials visible =================
        if constexpr (G::KDB) {
            // V^T_j's pieces were issued BEFORE K_{j+1}'s: loads retire in order, so the younger K pieces may stay in flight
            if (FLASH_DS_COUNTED && j + 1 < ntiles) wait_vmcnt<NS * ((G::KP + NW - 1) / NW)>();
            else wait_vmcnt<0>();
        } else {
            wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // every wave has finished reading K_j
        if constexpr (!G::KDB) {
            if (j + 1 < ntiles) issue_k(j + 1);
        }
        f32x4 p0, p1;
        asm volatile("ds_read_b128 %0, %1" : "=v"(p0) : "v"(x_peer));
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(p1) : "v"(x_peer));
        // (the V^T fragments of the first channel pair are fetched under the softmax)
        bf16x8 vf[2][2][NS];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int p = 0; p < NS; ++p) vf[0][q][p] = lds_read128(v_frag + p * G::VPL + q * 1024);
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(p0), "+v"(p1) : "n"(2 * NS));      // the partner's partials have landed
        __builtin_amdgcn_sched_barrier(0);
        float sv[8] = {(s0[0] + p0[0]) * alpha, (s0[1] + p0[1]) * alpha, (s0[2] + p0[2]) * alpha, (s0[3] + p0[3]) * alpha,
                       (s1[0] + p1[0]) * alpha, (s1[1] + p1[1]) * alpha, (s1[2] + p1[2]) * alpha, (s1[3] + p1[3]) * alpha};
        if ((j + 1) * G::BKV > d.Nk) {             // ragged last tile (uniform branch)
            const int k0 = j * G::BKV + 8 * g;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (k0 + i >= d.Nk) sv[i] = -1.0e30f;
        }
        float tmax = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
        tmax = xor_max(tmax);
        const float m_new = fmaxf(m_run, tmax);
        const float corr = __expf(m_run - m_new);
        float pv[8], rs = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pv[i] = __expf(sv[i] - m_new);
            rs += pv[i];
        }
        rs = xor_sum(rs);
        l_run = l_run * corr + rs;
        m_run = m_new;
        bf16x8 pf[NS];
        {
            uint32_t h[8], l[8];
#pragma unroll
            for (int i = 0; i < 8; i += 2) { split_op2(pv[i], pv[i + 1], NS, h[i], l[i]); h[i + 1] = 0u; l[i + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
            u32x4 ph = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
            u32x4 pl = {l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
            pf[0] = __builtin_bit_cast(bf16x8, ph);
            pf[1] = __builtin_bit_cast(bf16x8, pl);
        }
        if (__any(corr != 1.0f)) {
#pragma unroll
            for (int c = 0; c < CTH; ++c) {
                o[c][0] *= corr; o[c][1] *= corr; o[c][2] *= corr; o[c][3] *= corr;
            }
        }
        // ================= phase 2: O^T[this half] += V^T_j P^T =================
#pragma unroll
        for (int c = 0; c < CTH; c += 2) {
            const int cur = (c >> 1) & 1;
            if (c + 2 < CTH) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int p = 0; p < NS; ++p) vf[cur ^ 1][q][p] = lds_read128(v_frag + p * G::VPL + (c + 2 + q) * 1024);
                wait_lgkm<2 * NS>();
            } else {
                wait_lgkm<0>();
            }
            o[c] = mfma_op<NS>(vf[cur][0][1], pf[0], o[c]);
            o[c + 1] = mfma_op<NS>(vf[cur][1][1], pf[0], o[c + 1]);
            o[c] = mfma_op<NS>(vf[cur][0][0], pf[1], o[c]);
            o[c + 1] = mfma_op<NS>(vf[cur][1][0], pf[1], o[c + 1]);
            o[c] = mfma_op<NS>(vf[cur][0][0], pf[0], o[c]);
            o[c + 1] = mfma_op<NS>(vf[cur][1][0], pf[0], o[c + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: lane holds O^T[c = 16 (hf CTH + ct) + 4 g + e][q = r] ----
    const float inv = 1.0f / l_run;
    const bool ln = d.ln_op != nullptr;
    float lsum = 0.f;
    bool sat = false;
    if (q_ok) status_raise(false, g == 0 && hf == 0 && stat_bad(m_run, inv));
#pragma unroll
    for (int c = 0; c < CTH; ++c) {
        const int col = (hf * CTH + c) * 16 + g * 4;
        float4 v = make_float4(o[c][0] * inv, o[c][1] * inv, o[c][2] * inv, o[c][3] * inv);
        if (d.out_act) {
            if (d.bias) {
                const float4 bb = *reinterpret_cast<const float4*>(d.bias + col);
                v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            }
            if (d.residual) {
                const float4 rr = load_act4(d.residual, grow * d.ldr + col, d.act_bf16);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            const int64_t oo = grow * d.ld_act + col;
            if (q_ok) {
                if (d.act_bf16)
                    *reinterpret_cast<uint2*>(reinterpret_cast<frido_bf16*>(d.out_act) + oo) =
                        make_uint2(f32_to_bf16_bits(v.x) | (f32_to_bf16_bits(v.y) << 16), f32_to_bf16_bits(v.z) | (f32_to_bf16_bits(v.w) << 16));
                else
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.out_act) + oo) = v;
            }
            if (ln) {
                o[c] = f32x4{v.x, v.y, v.z, v.w};
                lsum += (v.x + v.y) + (v.z + v.w);
            }
        } else if (q_ok) {
            const float vv[4] = {v.x, v.y, v.z, v.w};
            store_op4(d.out_op, d.out_lo, NS, grow * d.ldo + col, vv);
            sat |= op_sat4(vv);
        }
    }
    if (ln) {       // (uniform: every wave of the workgroup takes the two exchange barriers)
        // LayerNorm of the stream row: the pair's half-row sums meet in the exchange buffer; mean first, then the centred sum of
        // squares (layernorm_kernel's two passes), both added as half 0 + half 1 so that the two waves hold identical statistics
        float* xs = reinterpret_cast<float*>(smem + G::X0);
        const float hs = xor_sum(lsum);
        __builtin_amdgcn_s_barrier();              // (every wave is past its last exchange read of the main loop)
        if (g == 0) xs[wave * 16 + r] = hs;
        __syncthreads();
        const float mean = (xs[(pair * 2) * 16 + r] + xs[(pair * 2 + 1) * 16 + r]) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CTH; ++c) {
            const float a0 = o[c][0] - mean, a1 = o[c][1] - mean, a2 = o[c][2] - mean, a3 = o[c][3] - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
        const float hq = xor_sum(q);
        if (g == 0) xs[128 + wave * 16 + r] = hq;
        __syncthreads();
        const float rstd = 1.0f / sqrtf((xs[128 + (pair * 2) * 16 + r] + xs[128 + (pair * 2 + 1) * 16 + r]) * (1.0f / D) + d.ln_eps);
        if (q_ok) {
#pragma unroll
            for (int c = 0; c < CTH; ++c) {
                const int col = (hf * CTH + c) * 16 + g * 4;
                const float4 w = *reinterpret_cast<const float4*>(d.ln_w + col), bb = *reinterpret_cast<const float4*>(d.ln_b + col);
                const float y[4] = {(o[c][0] - mean) * rstd * w.x + bb.x, (o[c][1] - mean) * rstd * w.y + bb.y,
                                    (o[c][2] - mean) * rstd * w.z + bb.z, (o[c][3] - mean) * rstd * w.w + bb.w};
                store_op4(d.ln_op, d.ln_lo, NS, grow * d.ld_ln + col, y);
                sat |= op_sat4(y);
            }
            status_raise(false, g == 0 && hf == 0 && stat_bad(mean, rstd));
        }
    }
    status_raise(sat);
}

template <int D>
int flash_ds_launch(const FridoAttnSmall& d, hipStream_t s) {
    constexpr int smem = FDGeo<D>::SMEM;
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FRIDO_EHIP;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_ds_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
            frido_set_error("attn_flash (d-split): cannot set dynamic LDS size %d", smem);
            return FRIDO_EHIP;
        }
        done.fetch_or(bit, std::memory_order_release);
    }
    const int blocks = d.B * ((d.Nq + 63) / 64);
    hipLaunchKernelGGL((flash_ds_kernel<D>), dim3(blocks), dim3(512), smem, s, d);
    return frido_check_launch("attn_flash(d-split)");
}


template <int D, int NS, int NW, int OS>
int flash_launch(const FridoAttnSmall& d, hipStream_t s) {
    constexpr int smem = FGeo<D, NS, NW, OS>::SMEM;
    static std::atomic<uint64_t> done{0};           // > 64 KiB dynamic LDS is a per-device function attribute
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FRIDO_EHIP;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_kernel<D, NS, NW, OS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                smem) != hipSuccess) {
            frido_set_error("attn_flash: cannot set dynamic LDS size %d", smem);
            return FRIDO_EHIP;
        }
        done.fetch_or(bit, std::memory_order_release);
    }
    const int blocks = d.B * ((d.Nq + NW * 16 - 1) / (NW * 16));
    hipLaunchKernelGGL((flash_attn_kernel<D, NS, NW, OS>), dim3(blocks, OS), dim3(NW * 64), smem, s, d);
    return frido_check_launch("attn_flash");
}

bool flash_dsplit_on() {
    static const bool on = !(getenv("FRIDO_FLASH_DSPLIT") && atoi(getenv("FRIDO_FLASH_DSPLIT")) == 0);
    return on;
}

template <int D>
int flash_dispatch(const FridoAttnSmall& d, hipStream_t s) {
    if (d.nsplit == 2) {
        // (r05) two-plane mode: the d-split 8-wave form (two waves per SIMD) for 256 <= d <= 512 (d = 576 would need exactly the 160 KiB
        // of a CU and 8 B of scratch: measured SLOWER than the GEMM chain on the 16 x 16 plane it would serve, 43.6 vs 36.4 us, and not kept);
        // FRIDO_FLASH_DSPLIT=0 keeps r04's 4-wave form (A/B switch)
        if constexpr (D >= 256 && D <= 512) {
            if (flash_dsplit_on()) return flash_ds_launch<D>(d, s);
        }
        return flash_launch<D, 2, 4, (D >= 512 ? 2 : 1)>(d, s);   // hi + lo planes of Q and P: one wave per SIMD
    }
    // 8 waves (128 queries) per workgroup halve the L2 -> LDS bytes per FLOP; 4 waves when that would leave CUs idle or the
    // accumulators do not fit 256 registers
    if constexpr (D <= 384) {
        static const int min_wgs = getenv("FRIDO_FLASH_NW8_MIN_WGS") ? atoi(getenv("FRIDO_FLASH_NW8_MIN_WGS")) : 200;
        if ((int64_t)d.B * ((d.Nq + 127) / 128) >= min_wgs) return flash_launch<D, 1, 8, 1>(d, s);
    }
    return flash_launch<D, 1, 4, 1>(d, s);
}

}  // namespace

extern "C" int frido_attn_flash_supported(int32_t dd) { return dd == 128 || dd == 256 || dd == 384 || dd == 512 || dd == 576; }
// head widths whose two-plane launch can produce the LayerNorm of its stream rows (FridoAttnSmall.ln_op): the workgroup owns whole rows
extern "C" int frido_attn_flash_ln_supported(int32_t dd) { return dd == 256 || dd == 384 || (flash_dsplit_on() && dd == 512); }

extern "C" int frido_attn_flash(const FridoAttnSmall* d, frido_stream_t s) {
    FRIDO_REQUIRE(d && d->Q && d->K && d->VT && (d->out_op || d->out_act), "null pointer");
    FRIDO_REQUIRE(!d->out_act || ((d->ld_act & 3) == 0 && (d->ldr & 3) == 0), "stream strides must be multiples of 4");
    FRIDO_REQUIRE(d->B > 0 && d->Nq > 0 && d->Nk > 0, "empty problem");
    FRIDO_REQUIRE(d->d == d->dv && frido_attn_flash_supported(d->d), "d = dv must be one of 128, 256, 384, 512, 576");
    FRIDO_REQUIRE(d->ldvt >= ((d->Nk + 31) & ~31) && (d->ldvt & 7) == 0, "V^T rows must be zero-padded to a multiple of 32 keys");
    FRIDO_REQUIRE((d->ldq & 7) == 0 && (d->ldk & 7) == 0 && (d->ldo & 3) == 0 && (d->q_lo & 7) == 0 && (d->k_lo & 7) == 0 &&
                      (d->vt_lo & 7) == 0 && (d->out_lo & 3) == 0 && (d->k_bs & 7) == 0 && (d->vt_bs & 7) == 0,
                  "strides and plane offsets must keep 16-byte alignment");
    FRIDO_REQUIRE(d->nsplit == 1 || d->nsplit == 2, "nsplit must be 1 or 2");
    FRIDO_REQUIRE(!d->skip_act_store, "skip_act_store is served by frido_attn_small only");
    FRIDO_REQUIRE(!d->ln_op || (d->out_act && !d->act_bf16 && d->nsplit == 2 && frido_attn_flash_ln_supported(d->d) && d->ln_w && d->ln_b && (d->ld_ln & 3) == 0 &&
                                (d->ln_lo & 3) == 0),
                  "ln_op: bf16x3 f32-stream output with a head width whose workgroup owns whole rows (frido_attn_flash_ln_supported), weight and bias given");
    hipStream_t st = (hipStream_t)s;
    switch (d->d) {
        case 128: return flash_dispatch<128>(*d, st);
        case 256: return flash_dispatch<256>(*d, st);
        case 384: return flash_dispatch<384>(*d, st);
        case 512: return flash_dispatch<512>(*d, st);
        default: return flash_dispatch<576>(*d, st);
    }
}
