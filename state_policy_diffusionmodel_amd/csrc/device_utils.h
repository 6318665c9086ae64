// device_utils.h -- small device helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace spdm {

// mean / rstd of GroupNorm(1, C) for sample b from the fp64 partials (eps 1e-5, biased
// variance: nn.GroupNorm(1, C) at models/Unet_FiLmLayer.py:105).
__device__ __forceinline__ void sample_mean_rstd(const StatsRef& st, int b, float& mean, float& rstd) {
    const long long r0 = (long long)b * st.HW;
    const int first = (int)(r0 / st.m_tile);
    const int last = (int)((r0 + st.HW - 1) / st.m_tile);
    const int n = (last - first + 1) * st.n_tiles;
    const double* p = st.p + (size_t)b * st.slots * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < n; ++i) { s1 += p[2 * i]; s2 += p[2 * i + 1]; }
    const double m = s1 * st.inv_count;
    double var = s2 * st.inv_count - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + 1e-5));
}

// own + the value of lane ^ 32 (or ^ 16), in every lane, by gfx950's v_permlane32_swap / v_permlane16_swap: with both operands the
// same register the two results hold {own, partner} in one order or the other, so their sum (or max) is the same in both lanes
// and bit-identical to v + __shfl_xor(v, 32) -- a VALU operation instead of a round trip through the LDS crossbar (ds_bpermute).
__device__ __forceinline__ float sum_xor32(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float max_xor32(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float sum_xor16(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ double sum_xor32_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double a = __builtin_bit_cast(double, ((unsigned long long)(unsigned)rh[0] << 32) | (unsigned)rl[0]);
    const double b = __builtin_bit_cast(double, ((unsigned long long)(unsigned)rh[1] << 32) | (unsigned)rl[1]);
    return a + b;
}
__device__ __forceinline__ double sum_xor16_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double a = __builtin_bit_cast(double, ((unsigned long long)(unsigned)rh[0] << 32) | (unsigned)rl[0]);
    const double b = __builtin_bit_cast(double, ((unsigned long long)(unsigned)rh[1] << 32) | (unsigned)rl[1]);
    return a + b;
}

// The operand split of every contraction, two values at once, packed: x = hi + lo with hi = fp16(x) (round to nearest)
// and lo = fp16(x - hi).  v_cvt_pk_f16_f32, two v_fma_mix_f32 (fp16 half * -1 + fp32: x - hi straight from the packed halves, one
// rounding of an exactly representable difference -- the same bits as x - (float)hi), v_cvt_pk_f16_f32: 4 instructions per pair;
// the plain C++ form compiles to 6-7 (v_cvt_f32_f16 back and forth).  VALU work is not hidden under the MFMAs on this chip
// (tools/probes/coexec_probe.hip), so this is time on every launch that converts operands.
__device__ __forceinline__ void split_pair_f16(float a, float b, unsigned& hi, unsigned& lo) {
    unsigned h;
    float la, lb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(la) : "v"(h), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(h), "v"(b));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(la), "v"(lb));
    hi = h;
}

// Sum of a double over the 64 lanes of a wave, every lane ends with the total: the four in-row steps move the two halves by
// DPP, rows are joined by v_permlane16_swap / v_permlane32_swap: VALU only.  Fixed order: deterministic.
__device__ __forceinline__ double wave_sum_f64(double v) {
#define DPP_ADD64(ctrl_)                                                                                        \
    {                                                                                                           \
        const unsigned long long u_ = __builtin_bit_cast(unsigned long long, v);                                \
        const int lo_ = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u_, ctrl_, 0xf, 0xf, false);             \
        const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u_ >> 32), ctrl_, 0xf, 0xf, false);     \
        v += __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi_ << 32) | (unsigned long long)(unsigned)lo_); \
    }
    DPP_ADD64(0xB1)      // quad_perm [1,0,3,2]
    DPP_ADD64(0x4E)      // quad_perm [2,3,0,1]
    DPP_ADD64(0x141)     // row_half_mirror
    DPP_ADD64(0x140)     // row_mirror
#undef DPP_ADD64
    v = sum_xor16_f64(v);
    v = sum_xor32_f64(v);
    return v;
}

// Butterfly sum of a double inside aligned power-of-two lane segments of `seg` lanes (wave-uniform), steps 1, 2, 4, ... in that
// order: bit-identical to `for (o = 1; o < seg; o <<= 1) v += __shfl_xor(v, o)`, without the LDS crossbar.
__device__ __forceinline__ double seg_sum_f64(double v, int seg) {
#define DPP_ADD64(ctrl_)                                                                                        \
    {                                                                                                           \
        const unsigned long long u_ = __builtin_bit_cast(unsigned long long, v);                                \
        const int lo_ = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u_, ctrl_, 0xf, 0xf, false);             \
        const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u_ >> 32), ctrl_, 0xf, 0xf, false);     \
        v += __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi_ << 32) | (unsigned long long)(unsigned)lo_); \
    }
    if (seg > 1) DPP_ADD64(0xB1)
    if (seg > 2) DPP_ADD64(0x4E)
    if (seg > 4) DPP_ADD64(0x141)
    if (seg > 8) DPP_ADD64(0x140)
#undef DPP_ADD64
    if (seg > 16) v = sum_xor16_f64(v);
    if (seg > 32) v = sum_xor32_f64(v);
    return v;
}

// The same for a whole wave (all 64 lanes call it, all get the result).  The fine tilings of the small-grid kernels leave up to
// ~130 partial slots per sample (16-row x 32-column tiles); one thread adding them one after the other was 3-6 us of a 9-15 us
// launch at batch 1.  Few slots: lane order as above (same bits as the serial form); many: lane i adds slots i, i + 64, ...
// and a fixed butterfly adds the lanes -- deterministic, independent of the batch.
__device__ __forceinline__ void sample_mean_rstd_wave(const StatsRef& st, int b, int lane, float& mean, float& rstd) {
    const long long r0 = (long long)b * st.HW;
    const int first = (int)(r0 / st.m_tile);
    const int last = (int)((r0 + st.HW - 1) / st.m_tile);
    const int n = (last - first + 1) * st.n_tiles;
    if (n <= 8) {                       // (wave-uniform)
        sample_mean_rstd(st, b, mean, rstd);
        return;
    }
    const double* p = st.p + (size_t)b * st.slots * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < n; i += 64) { s1 += p[2 * i]; s2 += p[2 * i + 1]; }
    s1 = wave_sum_f64(s1);
    s2 = wave_sum_f64(s2);
    const double m = s1 * st.inv_count;
    double var = s2 * st.inv_count - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + 1e-5));
}

// The FiLM tail of a Down/UpSample block (models/Unet_FiLmLayer.py:165-177: x = GN(x); x = x + emb_t; x = scale x + bias) as
// the affine y = A x + B of the raw conv output, for sample b: dst[0, C) = A = scale rstd gamma, dst[C, 2C) = B =
// scale (beta - mean rstd gamma + emb_t) + bias.  One WAVE per sample (all 64 lanes call it); same arithmetic as
// film_coef_kernel (elementwise.hip), which stays as the fallback.
__device__ __forceinline__ void film_coef_row_wave(const FilmSpec& f, int b, int lane, float* dst) {
    const int C = f.C;
    const int t = (f.temb != nullptr) ? f.t_dev[f.t_count == 1 ? 0 : b] : 0;
    float mean = 0.f, rstd = 1.f;
    if (f.st.p != nullptr) sample_mean_rstd_wave(f.st, b, lane, mean, rstd);
    for (int c = lane; c < C; c += 64) {
        float sc = 1.f, be = 0.f;
        if (f.st.p != nullptr) { sc = rstd * f.gamma[c]; be = f.beta[c] - mean * sc; }
        if (f.temb != nullptr) be += f.temb[(size_t)t * C + c];
        float fs = 1.f, fb = 0.f;
        if (f.film != nullptr) { fs = f.film[(size_t)b * 2 * C + c]; fb = f.film[(size_t)b * 2 * C + C + c]; }
        dst[c] = fs * sc;
        dst[C + c] = fs * be + fb;
    }
}

// Sum over the 16 lanes of a DPP row (quad swaps, half-mirror, mirror): every lane ends with the row total.  Four VALU
// operations, no LDS traffic; the same additions as the xor butterfly 1, 2, 4, 8 (a + b == b + a), so bit-identical to it.
__device__ __forceinline__ float row16_sum_dpp(float v) {
#define DPP_ADD(ctrl_)                                                                                          \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl_, 0xf, 0xf, false));
    DPP_ADD(0xB1)       // quad_perm [1,0,3,2]
    DPP_ADD(0x4E)       // quad_perm [2,3,0,1]
    DPP_ADD(0x141)      // row_half_mirror
    DPP_ADD(0x140)      // row_mirror
#undef DPP_ADD
    return v;
}

// erf(x), branch-free.  Same two minimax pieces the ROCm device library's erff uses (|x| < 1: odd
// polynomial; |x| >= 1: 1 - exp(-q(|x|))), but both evaluated and SELECTED instead of branched: inside a
// wave the library version almost always executes both sides of its divergent branch anyway, plus the
// exec-mask bookkeeping.  The exponential goes through v_exp_f32 with a compensated argument.
// Max deviation from the library erff measured on the GPU (tests/test_gpu_ops.py): <= 2 ulp.
__device__ __forceinline__ float erf_bf(float x) {
    const float t = fabsf(x);
    const float p = t * t;
    float r = __builtin_bit_cast(float, 0xba1345e1u) * p + __builtin_bit_cast(float, 0x3ba10414u);
    r = r * p + __builtin_bit_cast(float, 0xbcdac9b8u);
    r = r * p + __builtin_bit_cast(float, 0x3de703beu);
    r = r * p + __builtin_bit_cast(float, 0xbec09330u);
    r = r * p + __builtin_bit_cast(float, 0x3e0375d0u);
    const float small = t * r + t;
    float q = t * __builtin_bit_cast(float, 0x378e98abu) + __builtin_bit_cast(float, 0xb9c68948u);
    q = t * q + __builtin_bit_cast(float, 0x3b7cd369u);
    q = t * q + __builtin_bit_cast(float, 0xbcc618b2u);
    q = t * q + __builtin_bit_cast(float, 0x3dda74e4u);
    q = t * q + __builtin_bit_cast(float, 0x3f228afdu);
    q = t * q + __builtin_bit_cast(float, 0x3e03c728u);
    q = t * q + t;
    // exp(-q) = 2^(-q log2 e), argument split hi/lo
    const float a = -q * 1.44269504f;
    const float al = __fmaf_rn(-q, 1.44269504f, -a) - q * 1.925963033e-8f;
    const float e = __builtin_amdgcn_exp2f(a) * (1.0f + al * 0.69314718f);
    const float large = 1.0f - e;
    const float m = (t < 1.0f) ? small : large;
    return copysignf(m, x);
}

// exact (erf) GELU, nn.GELU() default -- models/Unet_FiLmLayer.py:104,65:  v * Phi(v),
// Phi(v) = 0.5 (1 + erf(v / sqrt 2)).
// Evaluated through the complementary form  0.5 erfc(z) = 0.5 t P(t) exp(-z^2),  t = 1 / (1 + p z),  z = |v| / sqrt 2
// (Abramowitz & Stegun 7.1.26, |error of erf| <= 1.5e-7), which needs ONE branch-free piece, one v_rcp_f32 and
// one v_exp_f32: 13 VALU instructions against 34 for the two-piece erf_bf above.  Both signs are covered by
// max(v, 0) - |v| 0.5 erfc(z) (for v < 0 that is v * 0.5 erfc(z): no 1 - x cancellation in the tail).  |GELU error| <= 0.5 |v| * 1.5e-7 + rounding:
// measured <= 4e-7 against fp64 on [-12, 12] (tests/test_gpu_ops.py), the same level as torch's own fp32 kernel.
// The GroupNorm -> GELU prologue of every second convolution runs this once per activation element.
__device__ __forceinline__ float gelu_erf(float v) {
    const float z = fabsf(v);
    const float t = __builtin_amdgcn_rcpf(__fmaf_rn(0.3275911f * 0.70710678118654752440f, z, 1.0f));
    float p = __fmaf_rn(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    p = __fmaf_rn(p, t, 0.5f * 1.421413741f);
    p = __fmaf_rn(p, t, 0.5f * -0.284496736f);
    p = __fmaf_rn(p, t, 0.5f * 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(v * v * -0.72134752044448170368f);     // exp(-v^2 / 2)
    const float q = (p * t) * e;                                                   // 0.5 erfc(|v| / sqrt 2)
    // v Phi(v) = v (1 - q) for v >= 0 and v q for v < 0, i.e. max(v, 0) - |v| q in both cases: no select
    return __fmaf_rn(-z, q, fmaxf(v, 0.f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the 32 lanes of this lane's half-wave (xor offsets < 32 stay inside the half)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace spdm
