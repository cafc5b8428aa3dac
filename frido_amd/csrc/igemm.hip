// Implicit-GEMM on the CDNA4 matrix pipe: C[m][n] = sum_k A[m][k] * B[n][k]  (both K-contiguous bf16).
//
//  * one workgroup = 256 threads = 4 waves (2 x 2) computing a BM x BN tile with
//    v_mfma_f32_16x16x32_bf16; K is walked in BK = 32 steps (one MFMA k-step per LDS tile);
//  * A is either a dense operand matrix or an NHWC image gathered on the fly (3x3 / 1x1 conv with
//    stride, zero padding, nearest x2 up-sampling or 2^k sub-sampling folded into the address);
//  * tiles are register-staged: the global loads of k-tile t+1 are issued before the MFMAs of tile t
//    and written to the other LDS buffer afterwards (one barrier per k-tile);
//  * LDS rows are 64 B (32 bf16); the 16-B slot q of row r lives at slot q ^ ((4 - (r>>2)) & 3), which
//    makes both the ds_write_b128 staging stores and the ds_read_b128 fragment loads conflict-free
//    for the 16x16x32 fragment layout (MI355X_MICROARCH.md §LDS lane groups);
//  * nsplit == 2 ("bf16x3"): operands carry a second bf16 plane with the rounding residual and each
//    tile product is hi*hi + hi*lo + lo*hi (fp32 accumulate) — ~2^-17 relative error at 3 MFMAs;
//  * fused epilogue: alpha, bias[n], per-row-group vector (timestep embedding), ReLU/SiLU, residual,
//    f32 and/or operand (bf16 hi/lo) output.
#include "common.h"

namespace {

struct RowInfo {      // per staged A row (conv mode)
    int b, oy, ox;
    bool ok;
};

__device__ __forceinline__ int swz(int row, int q) { return q ^ ((4 - ((row >> 2) & 3)) & 3); }

template <int BM, int BN, int NS, bool CONV>
__global__ __launch_bounds__(256) void igemm_kernel(const FridoGemm d) {
    constexpr int WM = 2, WN = 2;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int SA = BM * 4 / 256, SB = BN * 4 / 256;      // 16-B slots per thread per plane
    constexpr int PLANE = (BM + BN) * 64;                    // bytes per plane per buffer
    constexpr int BUF = NS * PLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block -> tile (XCD-aware: consecutive logical tiles share an A row-panel and an XCD L2) ----
    const int tiles_n = (d.N + BN - 1) / BN;
    const int tiles_m = (d.M + BM - 1) / BM;
    const int nb = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.y;

    const frido_bf16* __restrict__ Ab = d.A + (int64_t)z * d.a_bs;
    const frido_bf16* __restrict__ Bb = d.B + (int64_t)z * d.b_bs;

    // ---- per-thread staging assignments ----
    int64_t a_off[SA];
    RowInfo a_ri[SA];
    int a_lds[SA];
#pragma unroll
    for (int j = 0; j < SA; ++j) {
        const int s = t + j * 256, row = s >> 2, q = s & 3;
        const int m = m0 + row;
        a_lds[j] = row * 64 + (swz(row, q) << 4);
        a_ri[j].ok = m < d.M;
        if (CONV) {
            const int hw = d.Ho * d.Wo;
            const int mm = a_ri[j].ok ? m : 0;
            const int b = mm / hw, rem = mm - b * hw;
            a_ri[j].b = b;
            a_ri[j].oy = rem / d.Wo;
            a_ri[j].ox = rem - a_ri[j].oy * d.Wo;
            a_off[j] = q * 8;
        } else {
            a_off[j] = (int64_t)(a_ri[j].ok ? m : 0) * d.lda + q * 8;
        }
    }
    int64_t b_off[SB];
    int b_lds[SB];
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        const int s = t + j * 256, row = s >> 2, q = s & 3;
        int n = n0 + row;
        n = n < d.N ? n : d.N - 1;
        b_off[j] = (int64_t)n * d.ldb + q * 8;
        b_lds[j] = BM * 64 + row * 64 + (swz(row, q) << 4);
    }

    const int nk = d.K >> 5;
    // conv k-walk state (uniform): channel offset inside the tap, tap coordinates
    int kc = 0, ky = 0, kx = 0;

    u32x4 ra[NS][SA], rb[NS][SB];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < SA; ++j) {
            bool ok = a_ri[j].ok;
            int64_t off;
            if (CONV) {
                const int iy = a_ri[j].oy * d.stride + ky - d.pad;
                const int ix = a_ri[j].ox * d.stride + kx - d.pad;
                ok = ok && iy >= 0 && iy < d.Hl && ix >= 0 && ix < d.Wl;
                const int sy = (iy >> d.up_shift) << d.dn_shift;
                const int sx = (ix >> d.up_shift) << d.dn_shift;
                off = ((int64_t)(a_ri[j].b * d.Hs + sy) * d.Ws + sx) * d.Cin + kc + a_off[j];
            } else {
                off = a_off[j] + (int64_t)kt * 32;
            }
#pragma unroll
            for (int p = 0; p < NS; ++p)
                ra[p][j] = ok ? *reinterpret_cast<const u32x4*>(Ab + (p ? d.a_lo : 0) + off) : zero4;
        }
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const int64_t off = b_off[j] + (int64_t)kt * 32;
#pragma unroll
            for (int p = 0; p < NS; ++p) rb[p][j] = *reinterpret_cast<const u32x4*>(Bb + (p ? d.b_lo : 0) + off);
        }
        if (CONV) {   // advance the (tap, channel) walk
            kc += 32;
            if (kc == d.Cin) {
                kc = 0;
                if (++kx == d.kw) { kx = 0; ++ky; }
            }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int p = 0; p < NS; ++p) {
#pragma unroll
            for (int j = 0; j < SA; ++j) *reinterpret_cast<u32x4*>(base + p * PLANE + a_lds[j]) = ra[p][j];
#pragma unroll
            for (int j = 0; j < SB; ++j) *reinterpret_cast<u32x4*>(base + p * PLANE + b_lds[j]) = rb[p][j];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: row (lane & 15) of a 16-row MFMA tile, logical slot (lane >> 4)
    const int frow = lane & 15;
    const int fslot = ((lane >> 4) ^ ((4 - ((frow >> 2) & 3)) & 3)) << 4;
    const int a_frag = (wm * (BM / WM) + frow) * 64 + fslot;
    const int b_frag = BM * 64 + (wn * (BN / WN) + frow) * 64 + fslot;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* base = smem + cur * BUF;
        bf16x8 fa[NS][TM];
#pragma unroll
        for (int p = 0; p < NS; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[p][i] = *reinterpret_cast<const bf16x8*>(base + p * PLANE + a_frag + i * 16 * 64);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bf16x8 fb[NS];
#pragma unroll
            for (int p = 0; p < NS; ++p) fb[p] = *reinterpret_cast<const bf16x8*>(base + p * PLANE + b_frag + j * 16 * 64);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (NS == 2) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[1], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[0], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds D[row = (lane>>4)*4 + e][col = lane & 15] of each 16x16 tile ----
    const int col_l = lane & 15, row_l = (lane >> 4) * 4;
    int vstep = 0;
    if (d.rowvec && d.rowvec_step) vstep = *d.rowvec_step;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 16 + col_l;
        if (n >= d.N) continue;
        const float bias = d.bias ? d.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm * (BM / WM) + i * 16 + row_l + e;
                if (m >= d.M) continue;
                float v = acc[i][j][e] * d.alpha + bias;
                if (d.row_bias) v += d.row_bias[m];
                if (d.rowvec) v += d.rowvec[(int64_t)(m / d.rows_per_vec + vstep) * d.ldv + n];
                if (d.act == FRIDO_ACT_RELU) v = fmaxf(v, 0.f);
                else if (d.act == FRIDO_ACT_SILU) v = silu_f(v);
                if (d.residual) v += d.residual[(int64_t)z * d.res_bs + (int64_t)m * d.ldr + n];
                if (d.out_f32) d.out_f32[(int64_t)z * d.of_bs + (int64_t)m * d.ldo + n] = v;
                if (d.out_op) store_op1(d.out_op + (int64_t)z * d.oo_bs, d.oo_lo, d.nsplit, (int64_t)m * d.ldoo + n, v);
            }
        }
    }
}

template <int BM, int BN, int NS, bool CONV>
int set_attr() {
    constexpr int smem = 2 * NS * (BM + BN) * 64;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, NS, CONV>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
        frido_set_error("igemm: cannot set dynamic LDS size %d", smem);
        return FRIDO_EHIP;
    }
    return FRIDO_OK;
}

template <int BM, int BN, int NS, bool CONV>
int launch(const FridoGemm& d, hipStream_t s) {
    constexpr int smem = 2 * NS * (BM + BN) * 64;
    const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, NS, CONV>), dim3(tiles, d.batch), dim3(256), smem, s, d);
    return frido_check_launch("igemm");
}

template <int NS, bool CONV>
int dispatch_tile(const FridoGemm& d, int tile, hipStream_t s) {
    switch (tile) {
        case 1: return launch<128, 128, NS, CONV>(d, s);
        case 2: return launch<128, 192, NS, CONV>(d, s);
        default: return launch<64, 64, NS, CONV>(d, s);
    }
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
    rc |= set_attr<128, 128, 1, true>() | set_attr<128, 192, 1, true>() | set_attr<64, 64, 1, true>();
    rc |= set_attr<128, 128, 1, false>() | set_attr<128, 192, 1, false>() | set_attr<64, 64, 1, false>();
    rc |= set_attr<128, 128, 2, true>() | set_attr<128, 192, 2, true>() | set_attr<64, 64, 2, true>();
    rc |= set_attr<128, 128, 2, false>() | set_attr<128, 192, 2, false>() | set_attr<64, 64, 2, false>();
    return rc ? FRIDO_EHIP : FRIDO_OK;
}

extern "C" int frido_gemm(const FridoGemm* dp, frido_stream_t stream) {
    FRIDO_REQUIRE(dp != nullptr, "null descriptor");
    const FridoGemm& d = *dp;
    FRIDO_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.batch > 0, "empty problem");
    FRIDO_REQUIRE((d.K & 31) == 0, "K must be a multiple of 32 (zero-pad the operands)");
    FRIDO_REQUIRE(d.nsplit == 1 || d.nsplit == 2, "nsplit must be 1 or 2");
    FRIDO_REQUIRE(d.A && d.B, "null operand");
    FRIDO_REQUIRE(d.out_f32 || d.out_op, "no output");
    FRIDO_REQUIRE((d.ldb & 7) == 0 && (d.b_bs & 7) == 0, "B rows must be 16-byte aligned");
    if (d.conv) {
        FRIDO_REQUIRE((d.Cin & 31) == 0 && d.Cin > 0, "conv Cin must be a multiple of 32");
        FRIDO_REQUIRE(d.K == d.kh * d.kw * d.Cin, "conv K != kh*kw*Cin");
        FRIDO_REQUIRE(d.Ho > 0 && d.Wo > 0 && d.M % (d.Ho * d.Wo) == 0, "conv M must be Bimg*Ho*Wo");
        FRIDO_REQUIRE(d.stride >= 1 && d.up_shift >= 0 && d.dn_shift >= 0, "bad conv geometry");
        FRIDO_REQUIRE(d.batch == 1, "conv mode is not batched");
    } else {
        FRIDO_REQUIRE((d.lda & 7) == 0 && (d.a_bs & 7) == 0, "A rows must be 16-byte aligned");
    }
    if (d.rowvec) FRIDO_REQUIRE(d.rows_per_vec > 0, "rows_per_vec");
    const int tile = d.tile ? d.tile : pick_tile(d);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (d.nsplit == 1) return d.conv ? dispatch_tile<1, true>(d, tile, s) : dispatch_tile<1, false>(d, tile, s);
    return d.conv ? dispatch_tile<2, true>(d, tile, s) : dispatch_tile<2, false>(d, tile, s);
}
