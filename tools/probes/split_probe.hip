// split_pair_f16 (device_utils.h) against the plain C++ split on 2^24 floats of every magnitude: same bits?
// build: hipcc --offload-arch=gfx950 -O3 -I../../state_policy_diffusionmodel_amd/csrc -o split_probe split_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "device_utils.h"
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    unsigned h, l;
    spdm::split_pair_f16(a, b, h, l);
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const f16x2 h2 = {ha, hb};
    const f16x2 l2 = {(_Float16)(a - (float)ha), (_Float16)(b - (float)hb)};
    o[4 * i] = h; o[4 * i + 1] = l; o[4 * i + 2] = __builtin_bit_cast(unsigned, h2); o[4 * i + 3] = __builtin_bit_cast(unsigned, l2);
}
int main() {
    const int n = 1 << 24;
    std::vector<float> x(n);
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        unsigned bits = (s & 0x807fffffu) | ((90u + (s >> 8) % 60u) << 23);       // exponents 2^-37 .. 2^22
        if ((i & 1023) == 0) bits = s;                                             // anything, incl. inf / nan / denormals
        memcpy(&x[i], &bits, 4);
    }
    float* dx; unsigned* d_o;
    hipMalloc(&dx, n * 4); hipMalloc(&d_o, (size_t)n * 2 * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, d_o, n);
    std::vector<unsigned> o((size_t)n * 2);
    hipMemcpy(o.data(), d_o, (size_t)n * 2 * 4, hipMemcpyDeviceToHost);
    long bad_h = 0, bad_l = 0; int shown = 0;
    for (int i = 0; i < n / 2; ++i) {
        if (o[4 * i] != o[4 * i + 2]) ++bad_h;
        if (o[4 * i + 1] != o[4 * i + 3]) {
            ++bad_l;
            if (shown++ < 8) printf("x = %a %a  hi %08x  lo asm %08x  lo c++ %08x\n", x[2 * i], x[2 * i + 1], o[4 * i], o[4 * i + 1], o[4 * i + 3]);
        }
    }
    printf("pairs %d  hi mismatches %ld  lo mismatches %ld\n", n / 2, bad_h, bad_l);
    return 0;
}
