// Streaming-pattern probe for the elementwise kernels: which loop shape reaches the HBM rate on [B][HW][C] fp32?
// build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip ; run: ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void per_sample(const float* __restrict__ src, float* __restrict__ dst, const double* __restrict__ st,
                                                  int HW, int C, int rpb, int prologue) {
    __shared__ float sm[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    float mean = 0.f, rstd = 1.f;
    if (prologue) {
        if (tid == 0) {
            double s = 0, q = 0;
            for (int i = 0; i < prologue; ++i) { s += st[(b * 8 + i) * 2]; q += st[(b * 8 + i) * 2 + 1]; }
            sm[0] = (float)s; sm[1] = (float)(1.0 / sqrt(q + 1.0));
        }
        __syncthreads();
        mean = sm[0]; rstd = sm[1];
    }
    const int C4 = C >> 2, c4 = tid % C4, rl = tid / C4, rpp = 256 / C4;
    const int r_end = min(HW, (int)(blockIdx.y + 1) * rpb);
    for (int r = blockIdx.y * rpb + rl; r < r_end; r += U * rpp) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = min(r + u * rpp, r_end - 1);
            v[u] = *reinterpret_cast<const float4*>(src + ((size_t)b * HW + rr) * C + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rpp;
            float4 t = v[u];
            t.x = (t.x - mean) * rstd; t.y = (t.y - mean) * rstd; t.z = (t.z - mean) * rstd; t.w = (t.w - mean) * rstd;
            if (rr < r_end) *reinterpret_cast<float4*>(dst + ((size_t)b * HW + rr) * C + c4 * 4) = t;
        }
    }
}

template <int U>
__global__ __launch_bounds__(256) void flat(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[min(i + u * stride, n4 - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n4) dst[i + u * stride] = v[u];
    }
}

int main() {
    const int B = 4096, HW = 128, C = 64;
    const size_t n = (size_t)B * HW * C;
    float *src, *dst; double* st;
    hipMalloc(&src, n * 4); hipMalloc(&dst, n * 4); hipMalloc(&st, B * 16 * 8);
    hipMemset(src, 0, n * 4); hipMemset(st, 0, B * 16 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.1f us  %5.2f TB/s\n", name, ms / 20 * 1e3, 2.0 * n * 4 / (ms / 20 * 1e-3) / 1e12);
    };
    for (int grid : {2048, 8192, 32768}) {
        char nm[64];
        snprintf(nm, 64, "flat U=1 grid %d", grid); time(nm, [&] { hipLaunchKernelGGL(flat<1>, dim3(grid), dim3(256), 0, 0, (const float4*)src, (float4*)dst, n / 4); });
        snprintf(nm, 64, "flat U=4 grid %d", grid); time(nm, [&] { hipLaunchKernelGGL(flat<4>, dim3(grid), dim3(256), 0, 0, (const float4*)src, (float4*)dst, n / 4); });
    }
    for (int prologue : {0, 2}) for (int chunks : {1, 2, 4}) {
        const int rpb = HW / chunks; char nm[64];
        snprintf(nm, 64, "per-sample U=1 chunks %d prologue %d", chunks, prologue);
        time(nm, [&] { hipLaunchKernelGGL(per_sample<1>, dim3(B, chunks), dim3(256), 0, 0, src, dst, st, HW, C, rpb, prologue); });
        snprintf(nm, 64, "per-sample U=2 chunks %d prologue %d", chunks, prologue);
        time(nm, [&] { hipLaunchKernelGGL(per_sample<2>, dim3(B, chunks), dim3(256), 0, 0, src, dst, st, HW, C, rpb, prologue); });
        snprintf(nm, 64, "per-sample U=4 chunks %d prologue %d", chunks, prologue);
        time(nm, [&] { hipLaunchKernelGGL(per_sample<4>, dim3(B, chunks), dim3(256), 0, 0, src, dst, st, HW, C, rpb, prologue); });
        snprintf(nm, 64, "per-sample U=8 chunks %d prologue %d", chunks, prologue);
        time(nm, [&] { hipLaunchKernelGGL(per_sample<8>, dim3(B, chunks), dim3(256), 0, 0, src, dst, st, HW, C, rpb, prologue); });
    }
    return 0;
}
