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

// exact (erf) GELU, nn.GELU() default -- models/Unet_FiLmLayer.py:104,65
__device__ __forceinline__ float gelu_erf(float v) {
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
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
