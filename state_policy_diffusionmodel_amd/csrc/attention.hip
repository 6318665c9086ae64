// attention.hip -- core of nn.MultiheadAttention(C, 4, batch_first=True) as SelfAttention uses it
// (models/Unet_FiLmLayer.py:60,79): per (sample, head), softmax(q k^T / sqrt(d)) v over the
// L = H_l*W_l tokens of one trajectory (L <= 512, d in {16, 32, 64}).
//
// One workgroup per (sample, head).  K and V of the head (L x d each, <= 64 KB) are staged in LDS
// once; each thread owns one query row, keeps q and the output accumulator in registers and walks
// the keys with an online softmax -- every lane of a wave reads the same K/V row, i.e. an LDS
// broadcast, conflict-free.  fp32 VALU FMA runs at the same chip rate as fp32 MFMA on gfx950, and
// the whole attention core is ~2.5 % of the step's FLOPs.
#include "device_utils.h"

namespace spdm {

template <int D>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        int L, int C, int heads) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // K [L][D], V [L][D]
    float* Ks = sm;
    float* Vs = sm + (size_t)L * D;
    const int bh = blockIdx.x, b = bh / heads, hd = bh - b * heads;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const size_t ld = (size_t)3 * C;
    const float* base = qkv + (size_t)b * L * ld + hd * D;
    constexpr int D4 = D / 4;
    for (int i = tid; i < L * D4; i += nthr) {
        const int j = i / D4, k4 = i - j * D4;
        *reinterpret_cast<float4*>(Ks + j * D + k4 * 4) = *reinterpret_cast<const float4*>(base + j * ld + C + k4 * 4);
        *reinterpret_cast<float4*>(Vs + j * D + k4 * 4) = *reinterpret_cast<const float4*>(base + j * ld + 2 * C + k4 * 4);
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)D);
    for (int qi = tid; qi < L; qi += nthr) {
        float q[D], o[D];
#pragma unroll
        for (int k4 = 0; k4 < D4; ++k4) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)qi * ld + k4 * 4);
            q[4 * k4] = v.x * scale; q[4 * k4 + 1] = v.y * scale; q[4 * k4 + 2] = v.z * scale; q[4 * k4 + 3] = v.w * scale;
        }
#pragma unroll
        for (int k = 0; k < D; ++k) o[k] = 0.f;
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j < L; ++j) {
            const float* kr = Ks + j * D;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int k = 0; k < D; k += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(kr + k);
                s0 += q[k] * kv.x; s1 += q[k + 1] * kv.y; s2 += q[k + 2] * kv.z; s3 += q[k + 3] * kv.w;
            }
            const float sc = (s0 + s1) + (s2 + s3);
            const float mn = fmaxf(m, sc);
            const float alpha = expf(m - mn);      // exp(-inf) = 0 on the first key
            const float p = expf(sc - mn);
            l = l * alpha + p;
            const float* vr = Vs + j * D;
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

hipError_t launch_attention(const float* qkv, float* out, int B, int L, int C, int heads, hipStream_t s) {
    if (B <= 0 || L <= 0 || heads <= 0 || C % heads != 0) return hipErrorInvalidValue;
    const int d = C / heads;
    const size_t lds = (size_t)2 * L * d * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    int threads = ((L + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    const dim3 grid(B * heads), block(threads);
#define SPDM_ATT(DD)                                                                                          \
    {                                                                                                         \
        auto kern = attention_kernel<DD>;                                                                     \
        if (lds > 64 * 1024) {                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                           \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)); \
            if (e != hipSuccess) return e;                                                                    \
        }                                                                                                     \
        hipLaunchKernelGGL(kern, grid, block, lds, s, qkv, out, L, C, heads);                                 \
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

}  // namespace spdm
