// attention.hip -- core of nn.MultiheadAttention(C, 4, batch_first=True) as SelfAttention uses it
// (models/Unet_FiLmLayer.py:60,79): per (sample, head), softmax(q k^T / sqrt(d)) v over the
// L = H_l*W_l tokens of one trajectory (L <= 512, d in {16, 32, 64}).
//
// One workgroup per (sample, head).  K and V of the head (L x d each, <= 64 KB) are staged in LDS
// once; each thread owns one query row, keeps q and the output accumulator in registers and walks
// the keys with an online softmax -- every lane of a wave reads the same K/V row, i.e. an LDS
// broadcast, conflict-free.  fp32 VALU FMA runs at the same chip rate as fp32 MFMA on gfx950, and
// the whole attention core is ~2.5 % of the step's FLOPs.
#include <algorithm>
#include <cstdlib>

#include "device_utils.h"

namespace spdm {

// exp(x) for x <= 0 through v_exp_f32 with a compensated argument (~1e-7 relative error up to |x| ~ 80)
__device__ __forceinline__ float att_exp_neg(float x) {
    const float t = x * 1.44269504f;
    const float tl = __fmaf_rn(x, 1.44269504f, -t) + x * 1.925963033e-8f;
    return __builtin_amdgcn_exp2f(t) * (1.0f + tl * 0.69314718f);
}

// Short sequences (L = 4..32 tokens: the coarse levels) pack G = 64 / L (sample, head) pairs into one wave, so that
// every lane owns a query; longer ones run one pair per workgroup.  K / V rows are padded by 4 floats and the groups
// are skewed by 4 banks so that the G distinct rows a wave instruction reads never share a bank.
template <int D>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        int L, int C, int heads, int G, int GS, int npairs) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // per group: K [L][D + 4], V [L][D + 4]
    constexpr int RS = D + 4;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const size_t ld = (size_t)3 * C;
    constexpr int D4 = D / 4;
    const int pair0 = blockIdx.x * G;
    float* Ksm = sm;
    float* Vsm = sm + (size_t)G * GS;
    for (int i = tid; i < G * L * D4; i += nthr) {
        const int g = i / (L * D4), r = i - g * (L * D4);
        const int j = r / D4, k4 = r - j * D4;
        const int bh = min(pair0 + g, npairs - 1), b = bh / heads, hd = bh - b * heads;
        const float* base = qkv + (size_t)b * L * ld + hd * D;
        *reinterpret_cast<float4*>(Ksm + g * GS + j * RS + k4 * 4) = *reinterpret_cast<const float4*>(base + j * ld + C + k4 * 4);
        *reinterpret_cast<float4*>(Vsm + g * GS + j * RS + k4 * 4) = *reinterpret_cast<const float4*>(base + j * ld + 2 * C + k4 * 4);
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)D);
    const int per = (G > 1) ? L : nthr;                  // queries handled side by side per group
    const int g = (G > 1) ? tid / L : 0;
    const int bh = pair0 + g;
    if (g >= G || bh >= npairs) return;
    const int b = bh / heads, hd = bh - b * heads;
    const float* base = qkv + (size_t)b * L * ld + hd * D;
    const float* Ks = Ksm + g * GS;
    const float* Vs = Vsm + g * GS;
    for (int qi = (G > 1) ? tid - g * L : tid; qi < L; qi += per) {
        float q[D], o[D];
#pragma unroll
        for (int k4 = 0; k4 < D4; ++k4) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)qi * ld + k4 * 4);
            q[4 * k4] = v.x * scale; q[4 * k4 + 1] = v.y * scale; q[4 * k4 + 2] = v.z * scale; q[4 * k4 + 3] = v.w * scale;
        }
#pragma unroll
        for (int k = 0; k < D; ++k) o[k] = 0.f;
        float m = -1e30f, l = 0.f;
        for (int j = 0; j < L; ++j) {
            const float* kr = Ks + j * RS;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int k = 0; k < D; k += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(kr + k);
                s0 += q[k] * kv.x; s1 += q[k + 1] * kv.y; s2 += q[k + 2] * kv.z; s3 += q[k + 3] * kv.w;
            }
            const float sc = (s0 + s1) + (s2 + s3);
            const float mn = fmaxf(m, sc);
            const float alpha = att_exp_neg(m - mn);     // first key: exp(-1e30) = 0
            const float p = att_exp_neg(sc - mn);
            l = l * alpha + p;
            const float* vr = Vs + j * RS;
#pragma unroll
            for (int k = 0; k < D; k += 4) {
                const float4 vv = *reinterpret_cast<const float4*>(vr + k);
                o[k] = o[k] * alpha + p * vv.x; o[k + 1] = o[k + 1] * alpha + p * vv.y;
                o[k + 2] = o[k + 2] * alpha + p * vv.z; o[k + 3] = o[k + 3] * alpha + p * vv.w;
            }
            m = mn;
        }
        const float inv = 1.0f / l;
        float* orow = out + ((size_t)b * L + qi) * C + hd * D;
#pragma unroll
        for (int k4 = 0; k4 < D4; ++k4)
            *reinterpret_cast<float4*>(orow + k4 * 4) =
                make_float4(o[4 * k4] * inv, o[4 * k4 + 1] * inv, o[4 * k4 + 2] * inv, o[4 * k4 + 3] * inv);
    }
}

// Short sequences with a wide head (L <= 16 tokens, d = 32 / 64: sa2, sa3, sa4).  attention_kernel gives every query one
// thread, i.e. L threads per (sample, head) pair, and a pair's K / V rows cost 2 L (d + 4) floats of LDS: at L = 16, d = 64
// that is 8.7 KB for 16 threads -- 4-5 waves per CU, and the kernel crawled at ~2 TB/s on latency.  Here DS = 4 lanes share a
// query, each owning d / 4 of its features (partial dot products, two xor-shuffles per key): 4 x the threads on the same
// LDS bytes.  The softmax arithmetic per query is the same sequence of operations as in attention_kernel.
template <int D, int DS>
__global__ __launch_bounds__(256) void attention_ds_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           int L, int C, int heads, int G, int GS, int npairs) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // per pair: K [L][D + 4], V [L][D + 4]
    constexpr int RS = D + 4, D4 = D / 4, DP = D / DS, DP4 = DP / 4;
    static_assert(DS == 4 && DP % 4 == 0, "4 lanes per query");
    const int tid = threadIdx.x, nthr = blockDim.x;
    const size_t ld = (size_t)3 * C;
    const int pair0 = blockIdx.x * G;
    float* Ksm = sm;
    float* Vsm = sm + (size_t)G * GS;
    for (int i = tid; i < G * L * D4; i += nthr) {
        const int g = i / (L * D4), r = i - g * (L * D4);
        const int j = r / D4, k4 = r - j * D4;
        const int bh = min(pair0 + g, npairs - 1), b = bh / heads, hd = bh - b * heads;
        const float* base = qkv + (size_t)b * L * ld + hd * D;
        *reinterpret_cast<float4*>(Ksm + g * GS + j * RS + k4 * 4) = *reinterpret_cast<const float4*>(base + j * ld + C + k4 * 4);
        *reinterpret_cast<float4*>(Vsm + g * GS + j * RS + k4 * 4) = *reinterpret_cast<const float4*>(base + j * ld + 2 * C + k4 * 4);
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)D);
    // thread -> (pair g, query qi, feature slice ds): DS consecutive lanes share a query
    const int ds = tid & (DS - 1), qg = tid / DS;
    const int g = qg / L, qi = qg - g * L;
    const int bh = pair0 + g;
    const bool live = (g < G) && (bh < npairs);
    const int bhc = live ? bh : min(pair0, npairs - 1);
    const int gc = live ? g : 0, qc = live ? qi : 0;
    const int b = bhc / heads, hd = bhc - b * heads;
    const float* base = qkv + (size_t)b * L * ld + hd * D + ds * DP;
    const float* Ks = Ksm + gc * GS + ds * DP;
    const float* Vs = Vsm + gc * GS + ds * DP;
    float q[DP], o[DP];
#pragma unroll
    for (int k4 = 0; k4 < DP4; ++k4) {
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)qc * ld + k4 * 4);
        q[4 * k4] = v.x * scale; q[4 * k4 + 1] = v.y * scale; q[4 * k4 + 2] = v.z * scale; q[4 * k4 + 3] = v.w * scale;
    }
#pragma unroll
    for (int k = 0; k < DP; ++k) o[k] = 0.f;
    float m = -1e30f, l = 0.f;
    for (int j = 0; j < L; ++j) {
        const float* kr = Ks + j * RS;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int k = 0; k < DP; k += 4) {
            const float4 kv = *reinterpret_cast<const float4*>(kr + k);
            s0 += q[k] * kv.x; s1 += q[k + 1] * kv.y; s2 += q[k + 2] * kv.z; s3 += q[k + 3] * kv.w;
        }
        float sc = (s0 + s1) + (s2 + s3);
        // the four feature slices of this query (fixed order: deterministic): quad swaps by DPP -- the same additions as
        // __shfl_xor 1 and 2, without two LDS-crossbar round trips on the per-key dependency chain of the online softmax
        sc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sc), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
        sc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sc), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
        const float mn = fmaxf(m, sc);
        const float alpha = att_exp_neg(m - mn);
        const float p = att_exp_neg(sc - mn);
        l = l * alpha + p;
        const float* vr = Vs + j * RS;
#pragma unroll
        for (int k = 0; k < DP; k += 4) {
            const float4 vv = *reinterpret_cast<const float4*>(vr + k);
            o[k] = o[k] * alpha + p * vv.x; o[k + 1] = o[k + 1] * alpha + p * vv.y;
            o[k + 2] = o[k + 2] * alpha + p * vv.z; o[k + 3] = o[k + 3] * alpha + p * vv.w;
        }
        m = mn;
    }
    if (live) {
        const float inv = 1.0f / l;
        float* orow = out + ((size_t)b * L + qi) * C + hd * D + ds * DP;
#pragma unroll
        for (int k4 = 0; k4 < DP4; ++k4)
            *reinterpret_cast<float4*>(orow + k4 * 4) =
                make_float4(o[4 * k4] * inv, o[4 * k4 + 1] * inv, o[4 * k4 + 2] * inv, o[4 * k4 + 3] * inv);
    }
}

hipError_t launch_attention(const float* qkv, float* out, int B, int L, int C, int heads, hipStream_t s) {
    if (B <= 0 || L <= 0 || heads <= 0 || C % heads != 0) return hipErrorInvalidValue;
    const int d = C / heads;
    const int npairs = B * heads;
    if (L <= 16 && (L & (L - 1)) == 0 && (d == 32 || d == 64)) {
        // four lanes per query: G pairs per 256-thread workgroup, G L 4 = 256
        const int G = 64 / L;
        const int RS = d + 4;
        const int GS = L * RS + ((4 - (L * RS) % 64 + 64) % 64);
        const size_t lds = (size_t)2 * G * GS * sizeof(float);
        if (lds <= 64 * 1024) {
            const dim3 grid((npairs + G - 1) / G), block(256);
            if (d == 64) hipLaunchKernelGGL((attention_ds_kernel<64, 4>), grid, block, lds, s, qkv, out, L, C, heads, G, GS, npairs);
            else hipLaunchKernelGGL((attention_ds_kernel<32, 4>), grid, block, lds, s, qkv, out, L, C, heads, G, GS, npairs);
            return hipGetLastError();
        }
    }
    const int G = (L <= 32 && (L & (L - 1)) == 0) ? 64 / L : 1;          // (sample, head) pairs per workgroup
    const int RS = d + 4;
    const int GS = L * RS + ((4 - (L * RS) % 64 + 64) % 64);               // group stride = 4 (mod 64 banks)
    const size_t lds = (size_t)2 * G * GS * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    int threads = ((L + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    if (G > 1) threads = 64;
    const dim3 grid((npairs + G - 1) / G), block(threads);
#define SPDM_ATT(DD)                                                                                          \
    {                                                                                                         \
        auto kern = attention_kernel<DD>;                                                                     \
        if (lds > 64 * 1024)                                                                                  \
            if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return e; \
        hipLaunchKernelGGL(kern, grid, block, lds, s, qkv, out, L, C, heads, G, GS, npairs);                  \
    }
    switch (d) {
        case 16: SPDM_ATT(16) break;
        case 32: SPDM_ATT(32) break;
        case 64: SPDM_ATT(64) break;
        default: return hipErrorInvalidValue;
    }
#undef SPDM_ATT
    return hipGetLastError();
}


// =================================================================================================
// MFMA attention core (L >= 32): flash-style, one wave per 32 queries of one (sample, head).
//
//   S^T = K Q^T  per 32-key block:  A = K rows (keys), B = Q^T (queries on the MFMA's lanes)
//         -> each lane owns ONE query column: its 16 accumulator registers are 16 of the block's 32 keys,
//            the other 16 sit in lane^32.  Softmax over keys = register reduction + one cross-half
//            shuffle; the running max / sum / rescale factor are per-lane scalars.
//   O^T += V^T P^T: the P^T block is ALREADY in B-operand position (rows = keys = the summed index live
//            in the registers, columns = queries on the lanes): registers 8s..8s+7 are the fragment of
//            k-step s; element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3) of the block, and the
//            V^T fragment is gathered from LDS in that same key order.  No LDS round-trip for P.
//   All products use the split-fp16 scheme of conv_gemm.hip (x*2^s = hi + lo, hi*hi + hi*lo + lo*hi,
//   fp32 accumulate): q,k,v pre-scaled by 16, p by 1024.
// K (row-major, rows padded to d+8 halfs) and V^T (rows padded to Lp+4 halfs) of the head are staged in LDS
// as fp16 hi/lo once per workgroup; both paddings make the fragment reads bank-conflict-free.
typedef float f32x16a __attribute__((ext_vector_type(16)));
typedef float f32x4a __attribute__((ext_vector_type(4)));
typedef float f32x2a __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8a __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4a __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2a __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}
typedef unsigned u32x4a __attribute__((ext_vector_type(4)));
// hi / lo fragments of 8 values f(0..7): four split_pair_f16 (device_utils.h: 4 instructions per pair)
template <typename F>
__device__ __forceinline__ void split8(f16x8a& h, f16x8a& l, F f) {
    u32x4a hu, lu;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned a, b;
        split_pair_f16(f(2 * q), f(2 * q + 1), a, b);
        hu[q] = a;
        lu[q] = b;
    }
    h = __builtin_bit_cast(f16x8a, hu);
    l = __builtin_bit_cast(f16x8a, lu);
}
// exp(x) for x <= 0 through v_exp_f32 with a compensated argument (keeps ~1e-7 relative error up to |x| ~ 80)
__device__ __forceinline__ float exp_neg(float x) {
    const float t = x * 1.44269504f;
    const float tl = __fmaf_rn(x, 1.44269504f, -t) + x * 1.925963033e-8f;
    return __builtin_amdgcn_exp2f(t) * (1.0f + tl * 0.69314718f);
}

template <int D, bool FULL>
__global__ __launch_bounds__(256) void attention_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                             int L, int C, int heads, int qblocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    constexpr int KS = D / 16;             // k-steps of the S product
    constexpr int MO = (D + 31) / 32;      // 32-row tiles of O^T
    constexpr int KROW = D + 8;            // halfs per K row
    const int Lp = (L + 31) & ~31;
    const int VROW = Lp + 8;                 // halfs per V^T row: 16-byte aligned rows, 4-bank skew between rows
    _Float16* Khi = reinterpret_cast<_Float16*>(smraw);
    _Float16* Klo = Khi + (size_t)Lp * KROW;
    _Float16* Vhi = Klo + (size_t)Lp * KROW;
    _Float16* Vlo = Vhi + (size_t)32 * MO * VROW;

    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    int bid = blockIdx.x;
    const int qb = bid % qblocks; bid /= qblocks;
    const int hd = bid % heads;
    const int b = bid / heads;
    const size_t ld = (size_t)3 * C;
    const float* base = qkv + (size_t)b * L * ld + hd * D;

    // ---- stage K and V^T of this (sample, head) as split fp16 ----
    constexpr int D4 = D / 4;
    for (int i = tid; i < Lp * D4; i += nthr) {
        const int j = i / D4, c = i - j * D4;
        // position of key j in a V^T row: inside each group of 16 keys the order is [0-3, 8-11, 4-7, 12-15] (bits 2 and 3
        // swapped), which makes the 8 keys a lane half feeds to one P.V MFMA contiguous (one ds_read_b128)
        const int jp = (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1);
        f32x4a kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
        if (j < L) {
            kv = *reinterpret_cast<const f32x4a*>(base + j * ld + C + c * 4);
            vv = *reinterpret_cast<const f32x4a*>(base + j * ld + 2 * C + c * 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 h, l2;
            split1(kv[e] * 16.0f, h, l2);
            Khi[j * KROW + c * 4 + e] = h;
            Klo[j * KROW + c * 4 + e] = l2;
            split1(vv[e] * 16.0f, h, l2);
            Vhi[(c * 4 + e) * VROW + jp] = h;
            Vlo[(c * 4 + e) * VROW + jp] = l2;
        }
    }
    if (D < 32 * MO) {      // zero the padding rows of V^T (d = 16: rows 16..31)
        for (int i = tid; i < (32 * MO - D) * Lp; i += nthr) {
            const int r = D + i / Lp, j = i % Lp;
            Vhi[r * VROW + j] = (_Float16)0.f;
            Vlo[r * VROW + j] = (_Float16)0.f;
        }
    }
    __syncthreads();

    const int q0 = qb * 128 + wave * 32;
    if (q0 >= L) return;                                   // (whole wave) nothing to do
    const int q = q0 + li;
    const int qc = min(q, L - 1);

    // ---- Q fragments (B operand): Q[q][16 ks + 8 kh + j], pre-multiplied by 1/sqrt(d) like torch does ----
    const float scale = 1.0f / sqrtf((float)D);
    f16x8a qh[KS], ql[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const float* qp = base + (size_t)qc * ld + 16 * ks + 8 * kh;
        const f32x4a a0 = *reinterpret_cast<const f32x4a*>(qp), a1 = *reinterpret_cast<const f32x4a*>(qp + 4);
        split8(qh[ks], ql[ks], [&](int e) { return ((e < 4 ? a0[e & 3] : a1[e & 3]) * scale) * 16.0f; });
    }

    f32x16a acc_o[MO];
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[mo][r] = 0.f;
    float m = -1e30f, lsum = 0.f;

    for (int kb = 0; kb < Lp / 32; ++kb) {
        // S^T block = K_blk Q^T
        f32x16a acc_s;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8a k_h = *reinterpret_cast<const f16x8a*>(Khi + (kb * 32 + li) * KROW + 16 * ks + 8 * kh);
            const f16x8a k_l = *reinterpret_cast<const f16x8a*>(Klo + (kb * 32 + li) * KROW + 16 * ks + 8 * kh);
            acc_s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k_h, qh[ks], acc_s, 0, 0, 0);
            acc_s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k_h, ql[ks], acc_s, 0, 0, 0);
            acc_s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k_l, qh[ks], acc_s, 0, 0, 0);
        }
        // online softmax over the block's 32 keys of this lane's query, through v_exp_f32 directly:
        // p x 1024 = exp2(acc_s c + (10 - m)), c = log2(e) / 256 (undoes the 16 x 16 operand pre-scale), running max m
        // in log2 units -- one max, one fma, one exp2 per score
        constexpr float SC = 1.44269504088896340736f / 256.0f;
        float sc[16];
        float mraw = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = acc_s[r];
            if (!FULL) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (key >= L) sc[r] = -3.0e38f;
            }
            mraw = fmaxf(mraw, sc[r]);
        }
        mraw = max_xor32(mraw);
        const float m_new = fmaxf(m, mraw * SC);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        const float off = 10.0f - m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = __builtin_amdgcn_exp2f(__fmaf_rn(sc[r], SC, off));
            psum += sc[r];
        }
        lsum = lsum * alpha + psum;
        m = m_new;
        // P^T fragments straight from the registers (k-step s = registers 8s .. 8s+7)
        f16x8a p_h[2], p_l[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) split8(p_h[s2], p_l[s2], [&](int j) { return sc[8 * s2 + j]; });
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[mo][r] *= alpha;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const f16x8a v_h = *reinterpret_cast<const f16x8a*>(Vhi + (mo * 32 + li) * VROW + kb * 32 + 16 * s2 + 8 * kh);
                const f16x8a v_l = *reinterpret_cast<const f16x8a*>(Vlo + (mo * 32 + li) * VROW + kb * 32 + 16 * s2 + 8 * kh);
                acc_o[mo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_h, p_h[s2], acc_o[mo], 0, 0, 0);
                acc_o[mo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_h, p_l[s2], acc_o[mo], 0, 0, 0);
                acc_o[mo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_l, p_h[s2], acc_o[mo], 0, 0, 0);
            }
        }
    }
    lsum = sum_xor32(lsum);
    const float inv = 1.0f / (lsum * 16.0f);             // lsum carries the x1024 of p; v scale 16
    if (q < L) {
        float* orow = out + ((size_t)b * L + q) * C + hd * D;
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int drow = mo * 32 + 8 * g + 4 * kh;     // d index of registers 4g .. 4g+3
                if (drow < D) {
                    const f32x4a v = {acc_o[mo][4 * g] * inv, acc_o[mo][4 * g + 1] * inv, acc_o[mo][4 * g + 2] * inv,
                                      acc_o[mo][4 * g + 3] * inv};
                    *reinterpret_cast<f32x4a*>(orow + drow) = v;
                }
            }
    }
}

template <int D>
static hipError_t launch_attention_mfma(const float* qkv, float* out, int B, int L, int C, int heads, hipStream_t s) {
    const int Lp = (L + 31) & ~31;
    constexpr int MO = (D + 31) / 32;
    const size_t lds = ((size_t)2 * Lp * (D + 8) + (size_t)2 * 32 * MO * (Lp + 8)) * sizeof(_Float16);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = (L % 32 == 0) ? attention_mfma_kernel<D, true> : attention_mfma_kernel<D, false>;
    if (lds > 64 * 1024)
        if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return e;
    const int qblocks = (L + 127) / 128;
    const int waves = std::min(4, (L + 31) / 32);
    hipLaunchKernelGGL(kern, dim3(B * heads * qblocks), dim3(64 * waves), lds, s, qkv, out, L, C, heads, qblocks);
    return hipGetLastError();
}

hipError_t launch_attention_auto(const float* qkv, float* out, int B, int L, int C, int heads, unsigned sw, hipStream_t s) {
    if (B <= 0 || L <= 0 || heads <= 0 || C % heads != 0 || C % 4 != 0) return hipErrorInvalidValue;
    const int d = C / heads;
    if (L >= 32 && !(sw & SW_ATTN_VALU)) {
        switch (d) {
            case 16: return launch_attention_mfma<16>(qkv, out, B, L, C, heads, s);
            case 32: return launch_attention_mfma<32>(qkv, out, B, L, C, heads, s);
            case 64: return launch_attention_mfma<64>(qkv, out, B, L, C, heads, s);
            default: break;
        }
    }
    return launch_attention(qkv, out, B, L, C, heads, s);
}

}  // namespace spdm
