// Does the matrix pipe of a gfx950 SIMD run under another wave's VALU work?  One 512-thread workgroup per CU (two waves per SIMD:
// wave w and wave w + 4 share SIMD w % 4).  Waves 0-3 run role A, waves 4-7 role B; a role is MFMA-only, VALU-only (one of several
// instruction kinds), both interleaved in one stream, or idle.  Time(A = MFMA, B = VALU) against Time(A = MFMA, B = idle) and
// Time(A = idle, B = VALU) says whether the two overlap or add.
// build: hipcc --offload-arch=gfx950 -O3 -o coexec_probe coexec_probe.hip ; run: ./coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum { IDLE = 0, MFMA = 1, FMA = 2, PKFMA = 3, DPP = 4, EXP = 5, CVT = 6, MIX = 7, MFMA_AGPR = 8, MIX_DPP = 9 };

template <int ROLE>
__device__ __forceinline__ void run(int iters, float* out) {
    const int lane = threadIdx.x & 63;
    if constexpr (ROLE == IDLE) return;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.25f + i;
    for (int it = 0; it < iters; ++it) {
        if constexpr (ROLE == MFMA || ROLE == MFMA_AGPR) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);       // 32 MFMAs
        } else if constexpr (ROLE == MIX || ROLE == MIX_DPP) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                    if constexpr (ROLE == MIX) {
                        asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %2, %2, %1, %1\n\tv_fma_f32 %3, %3, %1, %1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
                    } else {
                        asm volatile("v_mov_b32_dpp %0, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                     "v_mov_b32_dpp %2, %3 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                     "v_mov_b32_dpp %1, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
                    }
                }
        } else {
            // 96 VALU instructions per iteration (3 per MFMA of the other role)
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                if constexpr (ROLE == FMA)
                    asm volatile("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
                                 "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8"
                                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(0.999f));
                if constexpr (ROLE == PKFMA) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2* p = reinterpret_cast<f32x2*>(v);
                    asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4\n\t"
                                 "v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4"
                                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(f32x2{0.999f, 0.999f}));
                }
                if constexpr (ROLE == DPP)
                    asm volatile("v_mov_b32_dpp %0, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                 "v_mov_b32_dpp %2, %3 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %3, %4 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                 "v_mov_b32_dpp %4, %5 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %5, %6 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                 "v_mov_b32_dpp %6, %7 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %7, %0 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
                if constexpr (ROLE == EXP)
                    asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                                 "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7"
                                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
                if constexpr (ROLE == CVT)
                    asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n\tv_cvt_pk_f16_f32 %1, %1, %2\n\tv_cvt_pk_f16_f32 %2, %2, %3\n\tv_cvt_pk_f16_f32 %3, %3, %4\n\t"
                                 "v_cvt_pk_f16_f32 %4, %4, %5\n\tv_cvt_pk_f16_f32 %5, %5, %6\n\tv_cvt_pk_f16_f32 %6, %6, %7\n\tv_cvt_pk_f16_f32 %7, %7, %0"
                                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int RA, int RB>
__global__ __launch_bounds__(512, 1) void probe(int iters, float* out) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) run<RA>(iters, out); else run<RB>(iters, out);
}

int main() {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    auto time = [&](const char* name, auto kern) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, 100, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per iteration and SIMD: 32 MFMAs (512 cycles at 4 passes) and / or 96 VALU instructions (384 issue cycles)
        printf("%-34s %8.1f us   %6.1f ns per iteration\n", name, ms * 1e3, ms * 1e6 / iters);
    };
    time("A = MFMA            B = idle", probe<MFMA, IDLE>);
    time("A = MFMA            B = MFMA", probe<MFMA, MFMA>);
    time("A = idle            B = v_fma_f32", probe<IDLE, FMA>);
    time("A = v_fma_f32       B = v_fma_f32", probe<FMA, FMA>);
    time("A = MFMA            B = v_fma_f32", probe<MFMA, FMA>);
    time("A = idle            B = v_pk_fma_f32", probe<IDLE, PKFMA>);
    time("A = MFMA            B = v_pk_fma_f32", probe<MFMA, PKFMA>);
    time("A = idle            B = v_mov_dpp", probe<IDLE, DPP>);
    time("A = MFMA            B = v_mov_dpp", probe<MFMA, DPP>);
    time("A = idle            B = v_exp_f32", probe<IDLE, EXP>);
    time("A = MFMA            B = v_exp_f32", probe<MFMA, EXP>);
    time("A = idle            B = v_cvt_pk_f16", probe<IDLE, CVT>);
    time("A = MFMA            B = v_cvt_pk_f16", probe<MFMA, CVT>);
    time("A = MFMA+3 fma/MFMA B = idle", probe<MIX, IDLE>);
    time("A = MFMA+3 fma/MFMA B = same", probe<MIX, MIX>);
    time("A = MFMA+3 dpp/MFMA B = idle", probe<MIX_DPP, IDLE>);
    time("A = MFMA+3 dpp/MFMA B = same", probe<MIX_DPP, MIX_DPP>);
    return 0;
}
