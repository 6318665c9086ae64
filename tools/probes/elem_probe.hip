// The product's streaming kernels (film_apply, pool, upcat) timed alone on synthetic tensors, on a warm working set
// (same buffers every launch: Infinity-Cache resident) and on a cold one (8 rotating buffer sets).
// build: hipcc --offload-arch=gfx950 -O3 -I../../state_policy_diffusionmodel_amd/csrc -o elem_probe elem_probe.hip \
//        -L../../state_policy_diffusionmodel_amd -lspdm_hip -Wl,-rpath,'$ORIGIN/../../state_policy_diffusionmodel_amd'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels.h"
using namespace spdm;

// an MFMA-only kernel (no memory traffic) that puts the chip into the power state the conv kernels leave it in
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void burner(float* out, int iters) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
    f4 acc[8] = {};
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const int B = 4096, NB = 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < 3; ++i) launch(i);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 24; ++i) launch(i);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %7.1f us  %5.2f TB/s\n", name, ms / 24 * 1e3, bytes / (ms / 24 * 1e-3) / 1e12);
    };
    // level-0 shapes: HW = 128
    const size_t n64 = (size_t)B * 128 * 64, n128 = (size_t)B * 128 * 128, n64h = (size_t)B * 32 * 64;
    std::vector<float*> a64(NB), b64(NB), c128(NB), h64(NB);
    for (int i = 0; i < NB; ++i) {
        (void)hipMalloc(&a64[i], n64 * 4); (void)hipMalloc(&b64[i], n64 * 4); (void)hipMalloc(&c128[i], n128 * 4); (void)hipMalloc(&h64[i], n64h * 4);
        (void)hipMemset(a64[i], 0, n64 * 4); (void)hipMemset(h64[i], 0, n64h * 4);
    }
    double* st; float *gamma, *beta, *temb, *film; int* t_dev;
    (void)hipMalloc(&st, (size_t)B * 8 * 16); (void)hipMemset(st, 0, (size_t)B * 8 * 16);
    (void)hipMalloc(&gamma, 1024); (void)hipMalloc(&beta, 1024); (void)hipMemset(gamma, 0, 1024); (void)hipMemset(beta, 0, 1024);
    (void)hipMalloc(&temb, 1000 * 64 * 4); (void)hipMemset(temb, 0, 1000 * 64 * 4);
    (void)hipMalloc(&film, (size_t)B * 128 * 4); (void)hipMemset(film, 0, (size_t)B * 128 * 4);
    (void)hipMalloc(&t_dev, 4); (void)hipMemset(t_dev, 0, 4);
    auto src = [&](const float* x, int C, int HW, bool gn) {
        AffineSrc s{}; s.x = x; s.C = C;
        if (gn) { s.st.p = st; s.st.slots = 1; s.st.m_tile = HW; s.st.n_tiles = 1; s.st.HW = HW; s.st.inv_count = 1.0 / (C * (double)HW); s.gamma = gamma; s.beta = beta; }
        return s;
    };
    // streaming launches timed one by one, each right behind `burn_us` of MFMA work (as in the denoise step)
    float* sink; (void)hipMalloc(&sink, 64);
    auto time_after_burn = [&](const char* name, double bytes, int burn_iters, auto launch) {
        float tot = 0.f, totb = 0.f;
        hipEvent_t eb; (void)hipEventCreate(&eb);
        for (int i = 0; i < 27; ++i) {
            (void)hipEventRecord(eb);
            if (burn_iters) hipLaunchKernelGGL(burner, dim3(2048), dim3(256), 0, 0, sink, burn_iters);
            (void)hipEventRecord(e0);
            launch(i);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms, msb; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipEventElapsedTime(&msb, eb, e0);
            if (i >= 3) { tot += ms; totb += msb; }
        }
        printf("%-44s behind %6.0f us MFMA: %7.1f us  %5.2f TB/s\n", name, totb / 24 * 1e3, tot / 24 * 1e3, bytes / (tot / 24 * 1e-3) / 1e12);
    };
    for (int burn : {0, 2000, 8000, 30000}) {
        time_after_burn("film_apply L0 C=64 cold", 2.0 * n64 * 4, burn, [&](int i) {
            (void)launch_film_apply(src(a64[i % NB], 64, 128, true), temb, t_dev, 1, film, b64[i % NB], nullptr, B, 128, 0); });
        time_after_burn("upcat L1 -> L0 64 + 64 cold", (n64h + n64 + n128) * 4.0, burn, [&](int i) {
            (void)launch_upcat(src(h64[i % NB], 64, 32, true), src(a64[i % NB], 64, 128, true), c128[i % NB], B, 16, 2, 0); });
    }
    for (int cold = 0; cold < 2; ++cold) {
        const int m = cold ? NB : 1;
        printf("---- %s working set ----\n", cold ? "cold (8 rotating sets)" : "warm");
        time("film_apply  L0 C=64 (GN + temb + film)", 2.0 * n64 * 4, [&](int i) {
            (void)launch_film_apply(src(a64[i % m], 64, 128, true), temb, t_dev, 1, film, b64[i % m], nullptr, B, 128, 0); });
        time("film_apply  L0 C=64 (plain copy: no GN/temb/film)", 2.0 * n64 * 4, [&](int i) {
            (void)launch_film_apply(src(a64[i % m], 64, 128, false), nullptr, nullptr, 1, nullptr, b64[i % m], nullptr, B, 128, 0); });
        time("pool        L0 -> L1 C=64 (GN)", 1.25 * n64 * 4, [&](int i) {
            (void)launch_pool(src(a64[i % m], 64, 128, true), h64[i % m], B, 32, 4, 0); });
        time("upcat       L1 -> L0 64 + 64 (GN both)", (n64h + n64 + n128) * 4.0, [&](int i) {
            (void)launch_upcat(src(h64[i % m], 64, 32, true), src(a64[i % m], 64, 128, true), c128[i % m], B, 16, 2, 0); });
    }
    return 0;
}
