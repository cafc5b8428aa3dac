// Shared device helpers for libfrido_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frido_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define FRIDO_WAVE 64

void frido_set_error(const char* fmt, ...);
int frido_check_launch(const char* what);

#define FRIDO_REQUIRE(cond, msg)                                            \
    do {                                                                    \
        if (!(cond)) {                                                      \
            frido_set_error("%s: %s (%s)", __func__, msg, #cond);           \
            return FRIDO_EINVAL;                                            \
        }                                                                   \
    } while (0)

// round-to-nearest-even fp32 -> bf16 bits (inputs are finite on this path)
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
// two fp32 -> one packed bf16 pair {lo, hi}, round-to-nearest-even, as ONE v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// ---- operand planes.  One-plane (throughput) mode: bf16.  Two-plane (parity, "x3") mode, r03: FP16 hi + FP16 lo planes --
//      v ~= hi + lo to 2^-22 relative (bf16 pairs: 2^-17) at the same bytes and the same three MFMA passes
//      (v_mfma_f32_16x16x32_f16 runs at the bf16 rate; gfx950 MFMA keeps fp16 denormals, which the lo plane of |v| < 0.25 needs);
//      Error of the pair: max(2^-22 |v|, 2^-25) -- the lo plane is an fp16 SUBNORMAL for |v| < 2^-3, so the absolute floor is
//      half of fp16's smallest subnormal step (2^-24), not a relative bound, below that magnitude.
//      Range: values beyond fp16's +-65504 are CLAMPED there by split_op (r04; they used to become inf hi / -inf lo planes and a
//      NaN product): an out-of-range operand degrades to a saturated one instead of poisoning the image.  Normalised
//      activations, probabilities and weights sit far inside; the un-normalised producers (raw residual stream into a 1x1
//      skip / resample conv, GEGLU / MLP hidden activations) are the ones a trained checkpoint could push out, and a model
//      whose operands do saturate needs the bf16-pair build: -DFRIDO_X3_F16=0 (rounds 1-2's planes, fp32's range).
#ifndef FRIDO_X3_F16
#define FRIDO_X3_F16 1
#endif
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) { return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f); }   // v_cvt_f16_f32: RNE
__device__ __forceinline__ float f16_bits_to_f32(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)h); }
// split v into the hi / lo planes of an `ns`-plane operand (ns == 1: hi = bf16, lo unused)
__device__ __forceinline__ void split_op(float v, int ns, uint32_t& hi, uint32_t& lo) {
    if (FRIDO_X3_F16 && ns == 2) {
        v = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);       // saturate at fp16's range (one VALU op) instead of inf - inf = NaN
        hi = f32_to_f16_bits(v);
        lo = f32_to_f16_bits(v - f16_bits_to_f32(hi));
    } else {
        hi = f32_to_bf16_bits(v);
        lo = f32_to_bf16_bits(v - bf16_bits_to_f32(hi));
    }
}
// (r06) two values at once, planes already PACKED (low half = a): on gfx950 the fp16 path is 2 x v_med3 + v_cvt_pk_f16_f32 + 2 x v_cvt_f32_f16 +
// v_pk_add_f32 + v_cvt_pk_f16_f32 = 7 instructions per pair against 14 with split_op + shift / or packing.  Same roundings (RNE), same exact
// residual: the same bits as split_op.
typedef _Float16 frido_h2 __attribute__((ext_vector_type(2)));
typedef float frido_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_op2(float a, float b, int ns, uint32_t& hi, uint32_t& lo) {
    if (FRIDO_X3_F16 && ns == 2) {
        frido_f2 v = {__builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f)};
        const frido_h2 h = __builtin_convertvector(v, frido_h2);
        const frido_f2 r = v - __builtin_convertvector(h, frido_f2);
        const frido_h2 l = __builtin_convertvector(r, frido_h2);
        hi = __builtin_bit_cast(uint32_t, h);
        lo = __builtin_bit_cast(uint32_t, l);
    } else {
        uint32_t h0, l0, h1, l1;
        split_op(a, ns, h0, l0);
        split_op(b, ns, h1, l1);
        hi = h0 | (h1 << 16);
        lo = l0 | (l1 << 16);
    }
}
// ---- sticky status word (r05).  The library cannot throw from a kernel; instead every operand producer and every normalisation
//      kernel ORs a bit into a per-translation-unit device word when it meets a value the arithmetic cannot represent, and
//      frido_status_flags() (runtime.hip) ORs the words together for the host:
//        bit 0 (FRIDO_STATUS_SATURATED): a two-plane fp16 operand producer clamped a value at +-65504 (split_op's range) -- the
//              product that consumes it is no longer fp32-class; such a model needs the bf16-pair planes;
//        bit 1 (FRIDO_STATUS_NONFINITE): a GroupNorm / LayerNorm / softmax statistic was NaN or infinite -- a NaN / inf reached
//              the residual stream (split_op's clamp turns a NaN OPERAND into a finite one, so this is where NaNs stay visible:
//              every stream tensor of the path is normalised within a block or two).
//      One word per .hip file (no -fgpu-rdc: device globals are not shared between translation units); each file registers an
//      accessor with the runtime at load time.  The checks cost one v_max3 per two values and one compare per eight.
namespace {
__device__ unsigned g_frido_status_word = 0;
}
typedef int (*frido_status_accessor)(unsigned* word, int clear);
typedef unsigned* (*frido_status_address)();                      // device address of the file's word (r06: frido_status_poll's gather kernel)
void frido_register_status_word(frido_status_accessor fn, frido_status_address addr);       // runtime.hip
namespace {
inline int frido_status_rw(unsigned* word, int clear) {
    unsigned w = 0;
    if (hipMemcpyFromSymbol(&w, HIP_SYMBOL(g_frido_status_word), sizeof(w), 0, hipMemcpyDeviceToHost) != hipSuccess) return FRIDO_EHIP;
    if (clear && w) {
        const unsigned z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_frido_status_word), &z, sizeof(z), 0, hipMemcpyHostToDevice) != hipSuccess) return FRIDO_EHIP;
    }
    *word = w;
    return FRIDO_OK;
}
inline unsigned* frido_status_addr() {
    void* p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(g_frido_status_word)) == hipSuccess ? (unsigned*)p : nullptr;
}
struct FridoStatusRegistrar {
    FridoStatusRegistrar() { frido_register_status_word(&frido_status_rw, &frido_status_addr); }
};
FridoStatusRegistrar g_frido_status_registrar;
}  // namespace
// |v| beyond the fp16 planes' range (false for NaN: see bit 1); the 8-value form is four v_max3_f32 and one compare
__device__ __forceinline__ bool op_sat(float v) { return FRIDO_X3_F16 && fabsf(v) > 65504.0f; }
__device__ __forceinline__ bool op_sat4(const float* v) {
    return FRIDO_X3_F16 && fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) > 65504.0f;
}
__device__ __forceinline__ bool op_sat8(const float* v) {
    const float m0 = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2])), m1 = fmaxf(fmaxf(fabsf(v[3]), fabsf(v[4])), fabsf(v[5]));
    return FRIDO_X3_F16 && fmaxf(fmaxf(m0, m1), fmaxf(fabsf(v[6]), fabsf(v[7]))) > 65504.0f;
}
__device__ __forceinline__ bool stat_bad(float mean, float rstd) { return !(fabsf(mean) <= 3.0e38f) || !(rstd > 0.0f) || !(rstd <= 3.0e38f); }
// at the end of a kernel (or of a thread's work): one atomic per lane that saw something, none otherwise
__device__ __forceinline__ void status_raise(bool saturated, bool nonfinite = false) {
    if (saturated || nonfinite) atomicOr(&g_frido_status_word, (saturated ? 1u : 0u) | (nonfinite ? 2u : 0u));
}

// one MFMA pass on operand fragments of an NS-plane operand format
template <int NS>
__device__ __forceinline__ f32x4 mfma_op(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (NS == 2 && FRIDO_X3_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// (r06) x * sigmoid(x) with v_rcp_f32 (1 ulp) instead of an IEEE division: the division was 11 of the ~24 VALU instructions per element of
// every normalise + SiLU + split conversion (tools/cg_prof.py: that conversion is the top-of-step work the fused GroupNorm + conv kernel's
// matrix pipe waits for; ISA count: 196 -> ~115 instructions per 8-channel unit).  One more ulp on a value that is split into two fp16
// planes afterwards: 1e-7 relative, two orders below the parity bounds.  -DFRIDO_SILU_DIV=1 restores the division.
#ifndef FRIDO_SILU_DIV
#define FRIDO_SILU_DIV 0
#endif
__device__ __forceinline__ float silu_f(float v) {
    if constexpr (FRIDO_SILU_DIV) return v / (1.0f + __expf(-v));
    else return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}
// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7): one rcp, one exp, six FMAs.  libm's erff costs ~3x as many
// VALU cycles, and the GEGLU epilogue evaluates it 2e8 times per denoiser forward (it was VALU-, not store-bound).
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    // (r06) v_rcp_f32: HIP's __frcp_rn is a correctly rounded 1 / x -- the full v_div_scale / v_div_fmas / v_div_fixup sequence, 11 instructions
    // where the approximation's own error (1.5e-7) is 2.5 ulp
    const float t = FRIDO_SILU_DIV ? __frcp_rn(fmaf(0.3275911f, ax, 1.0f)) : __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
// OpenAI CLIP's QuickGELU (clip/model.py): x * sigmoid(1.702 x)
__device__ __forceinline__ float quickgelu_f(float v) {
    if constexpr (FRIDO_SILU_DIV) return v / (1.0f + __expf(-1.702f * v));
    else return v * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v));
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erf_fast(v * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// store up to 4 consecutive operand elements (bf16 hi / optional lo plane)
__device__ __forceinline__ void store_op4(frido_bf16* op, int64_t lo_off, int nsplit, int64_t idx, const float v[4]) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; i += 2) { split_op2(v[i], v[i + 1], nsplit, h[i], l[i]); h[i + 1] = 0u; l[i + 1] = 0u; }      // (r06) packed pair: h[even] | (0 << 16)
    uint2 ph = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    *reinterpret_cast<uint2*>(op + idx) = ph;
    if (nsplit == 2) {
        uint2 pl = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
        *reinterpret_cast<uint2*>(op + lo_off + idx) = pl;
    }
}
__device__ __forceinline__ void store_op1(frido_bf16* op, int64_t lo_off, int nsplit, int64_t idx, float v) {
    uint32_t h, l;
    split_op(v, nsplit, h, l);
    op[idx] = (frido_bf16)h;
    if (nsplit == 2) op[lo_off + idx] = (frido_bf16)l;
}

// activation loads: f32 or bf16 residual stream
__device__ __forceinline__ float4 load_act4(const void* base, int64_t idx, int is_bf16) {
    if (is_bf16) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const frido_bf16*>(base) + idx);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
}
__device__ __forceinline__ float load_act1(const void* base, int64_t idx, int is_bf16) {
    if (is_bf16) return bf16_bits_to_f32(reinterpret_cast<const frido_bf16*>(base)[idx]);
    return reinterpret_cast<const float*>(base)[idx];
}
__device__ __forceinline__ void store_act1(void* base, int64_t idx, int is_bf16, float v) {
    if (is_bf16) reinterpret_cast<frido_bf16*>(base)[idx] = (frido_bf16)f32_to_bf16_bits(v);
    else reinterpret_cast<float*>(base)[idx] = v;
}
