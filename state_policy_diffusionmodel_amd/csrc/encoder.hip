// Observation front end: the encoder half of the reference's lightweight autoencoder
// (models/encoder/autoencoder.py:11-20), which Diffusion_DDPM.prepare_obs_cond_vectors applies to every observed
// frame once per sample() call (models/diffusion_ddpm.py:317-321):
//     Conv2d(3,16,k2,s2,p1) ReLU  Conv2d(16,32,k2,s2) ReLU  Conv2d(32,64,k2,s2) ReLU  Flatten  Linear(9216,128)
//     (N,3,96,96) -> (N,16,49,49) -> (N,32,24,24) -> (N,64,12,12) -> (N,9216) -> (N,128)
// The three stride-2 2x2 convolutions have non-overlapping windows, so a conv-2 output needs exactly a 4x4 input patch:
// conv 1 is evaluated on the fly in registers (its row / column 48 is never read by conv 2), conv 2's map lives in LDS,
// conv 3 reads it from there and writes the flattened (c, h, w) feature row the Linear layer consumes through the
// product's GEMM (launch_gemm, exact fp32 MFMA path).  Once per call, off the per-step path: sized for clarity, not tuned.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace spdm {
namespace {

constexpr int E_IN = 96, E_C1 = 16, E_C2 = 32, E_S2 = 24, E_C3 = 64, E_S3 = 12;

// one workgroup per image
__global__ __launch_bounds__(256) void encoder_convs_kernel(const float* __restrict__ img,    // [n][3][96][96]
                                                            const float* __restrict__ w1, const float* __restrict__ b1,   // (16,3,2,2)
                                                            const float* __restrict__ w2, const float* __restrict__ b2,   // (32,16,2,2)
                                                            const float* __restrict__ w3, const float* __restrict__ b3,   // (64,32,2,2)
                                                            float* __restrict__ feat) {                                  // [n][64*12*12]
    extern __shared__ float s2[];                     // [32][24][24] conv-2 map after ReLU
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* im = img + (size_t)n * 3 * E_IN * E_IN;
    // ---- conv 1 (+ReLU) in registers -> conv 2 (+ReLU) -> LDS, one conv-2 position per thread per pass ----
    for (int p = tid; p < E_S2 * E_S2; p += 256) {
        const int py = p / E_S2, px = p - py * E_S2;
        float px_in[3][4][4];                         // input rows 4 py - 1 .. 4 py + 2 (zero padding outside the image)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                for (int dx = 0; dx < 4; ++dx) {
                    const int iy = 4 * py - 1 + dy, ix = 4 * px - 1 + dx;
                    px_in[c][dy][dx] = (iy >= 0 && iy < E_IN && ix >= 0 && ix < E_IN) ? im[((size_t)c * E_IN + iy) * E_IN + ix] : 0.f;
                }
        float acc[E_C2];
#pragma unroll
        for (int o = 0; o < E_C2; ++o) acc[o] = b2[o];
        for (int c1 = 0; c1 < E_C1; ++c1) {           // conv-1 channel c1 at the 2x2 positions conv 2 reads
            float v[2][2];
#pragma unroll
            for (int ky = 0; ky < 2; ++ky)
#pragma unroll
                for (int kx = 0; kx < 2; ++kx) {
                    float a = b1[c1];
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int jy = 0; jy < 2; ++jy)
#pragma unroll
                            for (int jx = 0; jx < 2; ++jx)
                                a = fmaf(w1[((c1 * 3 + c) * 2 + jy) * 2 + jx], px_in[c][2 * ky + jy][2 * kx + jx], a);
                    v[ky][kx] = fmaxf(a, 0.f);
                }
#pragma unroll
            for (int o = 0; o < E_C2; ++o) {
                const float* w = w2 + ((size_t)o * E_C1 + c1) * 4;
                acc[o] = fmaf(w[0], v[0][0], acc[o]);
                acc[o] = fmaf(w[1], v[0][1], acc[o]);
                acc[o] = fmaf(w[2], v[1][0], acc[o]);
                acc[o] = fmaf(w[3], v[1][1], acc[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < E_C2; ++o) s2[o * (E_S2 * E_S2) + p] = fmaxf(acc[o], 0.f);
    }
    __syncthreads();
    // ---- conv 3 (+ReLU) from LDS, written in Flatten order (c, h, w) ----
    float* f = feat + (size_t)n * (E_C3 * E_S3 * E_S3);
    for (int idx = tid; idx < E_C3 * E_S3 * E_S3; idx += 256) {
        const int co = idx / (E_S3 * E_S3), q = idx - co * (E_S3 * E_S3);
        const int qy = q / E_S3, qx = q - qy * E_S3;
        float a = b3[co];
        const float* w = w3 + (size_t)co * E_C2 * 4;
        for (int ci = 0; ci < E_C2; ++ci) {
            const float* sp = s2 + ci * (E_S2 * E_S2) + (2 * qy) * E_S2 + 2 * qx;
            a = fmaf(w[ci * 4 + 0], sp[0], a);
            a = fmaf(w[ci * 4 + 1], sp[1], a);
            a = fmaf(w[ci * 4 + 2], sp[E_S2], a);
            a = fmaf(w[ci * 4 + 3], sp[E_S2 + 1], a);
        }
        f[idx] = fmaxf(a, 0.f);
    }
}

}  // namespace

hipError_t launch_encoder_convs(const float* img, const float* w1, const float* b1, const float* w2, const float* b2,
                                const float* w3, const float* b3, float* feat, int n_images, hipStream_t s) {
    if (n_images <= 0 || !img || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !feat) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * E_C2 * E_S2 * E_S2;                 // 73,728 bytes
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(encoder_convs_kernel)); e != hipSuccess) return e;
    hipLaunchKernelGGL(encoder_convs_kernel, dim3(n_images), dim3(256), lds, s, img, w1, b1, w2, b2, w3, b3, feat);
    return hipGetLastError();
}

}  // namespace spdm
