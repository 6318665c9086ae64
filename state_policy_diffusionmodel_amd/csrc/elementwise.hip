// elementwise.hip -- the streaming kernels around the MFMA GEMM: first conv (Cin = 1, fused
// zero-pad), max-pool, bilinear-x2 + concat, GroupNorm/time-embedding/FiLM finishing pass,
// LayerNorm, Mish/SiLU, and the fused "1x1 out conv + unpad + DDPM/DDIM update + inpaint" step
// kernel.  All of them are HBM-bound: channels-last rows, float4 (16 B) per lane, every wave
// instruction touches whole 128-byte lines.
#include <algorithm>

#include "device_utils.h"

namespace spdm {

// -------------------------------------------------------------------------------------------------
// inc.first: Conv2d(1, 64, 3, padding=1, bias=False) on pad_to(x, 8)
// (models/Unet_FiLmLayer.py:286,288 -> :101,111).  One workgroup per trajectory; the padded
// (Hp, Wp) image sits in LDS; thread = (row lane, 4 output channels).  Also emits the
// GroupNorm partial sums of its output (one slot per row part of a sample: gridDim.y parts, so that a batch of one is
// not a single workgroup).  Being the first kernel of a denoise step it also does the loop bookkeeping of
// advance_kernel (adv: -2 none, -1 step <- step + 1, >= 0 step <- adv): nothing that runs beside it reads those words.
struct StepAdvance { int* step_dev; int* t_dev; const int* timesteps; int n_steps; int adv; };
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ dst, double* __restrict__ stats,
                                                      int slots, int H0, int D, int Hp, int Wp, int lh, int lw,
                                                      const StepAdvance sa) {
    extern __shared__ __attribute__((aligned(16))) float sx[];   // [Hp*Wp] + 8 floats reduction scratch
    const int b = blockIdx.x, tid = threadIdx.x;
    const int HW = Hp * Wp;
    if (sa.adv >= -1 && b == 0 && blockIdx.y == 0 && tid == 0) {
        const int i = (sa.adv >= 0) ? sa.adv : (*sa.step_dev + 1);
        *sa.step_dev = i;
        *sa.t_dev = sa.timesteps[min(max(i, 0), sa.n_steps - 1)];
    }
    const int rows_part = HW / (int)gridDim.y;             // rows of this part (host: gridDim.y divides HW / 16... see launch)
    const int r_begin = blockIdx.y * rows_part, r_end = r_begin + rows_part;
    for (int i = tid; i < HW; i += 256) {
        const int h = i / Wp, c = i - h * Wp;
        const int h0 = h - lh, d = c - lw;
        sx[i] = (h0 >= 0 && h0 < H0 && d >= 0 && d < D) ? x[((size_t)b * H0 + h0) * D + d] : 0.f;
    }
    __syncthreads();
    const int c4 = tid & 15, rl = tid >> 4;
    float4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(w + t * 64 + c4 * 4);
    float s1 = 0.f, s2 = 0.f;
    for (int r = r_begin + rl; r < r_end; r += 16) {
        const int h = r / Wp, c = r - h * Wp;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hh = h + t / 3 - 1, cc = c + t % 3 - 1;
            if (hh >= 0 && hh < Hp && cc >= 0 && cc < Wp) {
                const float xv = sx[hh * Wp + cc];
                acc.x += xv * wv[t].x; acc.y += xv * wv[t].y; acc.z += xv * wv[t].z; acc.w += xv * wv[t].w;
            }
        }
        *reinterpret_cast<float4*>(dst + ((size_t)b * HW + r) * 64 + c4 * 4) = acc;
        s1 += (acc.x + acc.y) + (acc.z + acc.w);
        s2 += (acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    float* red = sx + HW;
    __syncthreads();
    if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s1; red[(tid >> 6) * 2 + 1] = s2; }
    __syncthreads();
    if (tid == 0) {
        double a1 = 0.0, a2 = 0.0;
        for (int i = 0; i < 4; ++i) { a1 += (double)red[2 * i]; a2 += (double)red[2 * i + 1]; }
        stats[((size_t)b * slots + blockIdx.y) * 2] = a1;             // slot of row part blockIdx.y (m_tile = rows_part, n_tiles = 1)
        stats[((size_t)b * slots + blockIdx.y) * 2 + 1] = a2;
    }
}

// row parts per sample: 4 below batch 1024 (Hp is a multiple of 8, so HW / 4 is a multiple of 16 rows; whole step, same box: -7 us at
// B = 256, -8..-16 us at 512, 0 at 1024), else 1 (every part re-loads the whole padded image: 60 -> 71 us at B = 4096) -- StatsRef{m_tile = HW / parts, n_tiles = 1}
int conv_in_parts(int Hp, int Wp, int B) { return (B < spdm_tune(21, 1024) && (Hp * Wp) % 64 == 0) ? 4 : 1; }

hipError_t launch_conv_in(const float* x, const float* w, float* dst, double* stats, int B, int H0, int D,
                          int Hp, int Wp, int lh, int lw, int* step_dev, int* t_dev, const int* timesteps, int n_steps,
                          int adv, hipStream_t s, int B_geom) {
    const int HW = Hp * Wp;
    const size_t lds = (size_t)(HW + 8) * sizeof(float);
    if (lds > 64 * 1024 || B <= 0) return hipErrorInvalidValue;
    if (adv >= -1 && (!step_dev || !t_dev || !timesteps || n_steps < 1)) return hipErrorInvalidValue;
    const int parts = conv_in_parts(Hp, Wp, B_geom > 0 ? B_geom : B);
    const StepAdvance sa{step_dev, t_dev, timesteps, n_steps, adv};
    hipLaunchKernelGGL(conv_in_kernel, dim3(B, parts), dim3(256), lds, s, x, w, dst, stats, stats_slots(HW, HW / parts, 1), H0, D,
                       Hp, Wp, lh, lw, sa);
    return hipGetLastError();
}

// per-block GroupNorm mean / rstd of sample b into LDS (identity when the tensor is already final)
__device__ __forceinline__ void block_sample_stats(const AffineSrc& src, int b, float* sm, float& mean, float& rstd) {
    mean = 0.f;
    rstd = 1.f;
    if (src.st.p != nullptr) {
        if (threadIdx.x < 64) {                                   // the first wave adds the partial slots together
            float m, r;
            sample_mean_rstd_wave(src.st, b, (int)threadIdx.x, m, r);
            if (threadIdx.x == 0) {
                sm[0] = m;
                sm[1] = r;
            }
        }
        __syncthreads();
        mean = sm[0];
        rstd = sm[1];
    }
}

struct Affine4 { float4 sc, be; float mu; bool on; };
__device__ __forceinline__ Affine4 make_affine(const AffineSrc& src, int c, float mean, float rstd) {
    Affine4 a;
    a.on = (src.st.p != nullptr);
    a.mu = mean;
    a.sc = make_float4(1.f, 1.f, 1.f, 1.f);
    a.be = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.on) {
        const float4 g = *reinterpret_cast<const float4*>(src.gamma + c);
        a.be = *reinterpret_cast<const float4*>(src.beta + c);
        a.sc = make_float4(rstd * g.x, rstd * g.y, rstd * g.z, rstd * g.w);
    }
    return a;
}
// the same in two halves: gamma / beta requested BEFORE the statistics are summed (one round trip less on the launch's chain)
struct AffineParams { float4 g, be; bool on; };
__device__ __forceinline__ AffineParams load_affine_params(const AffineSrc& src, int c) {
    AffineParams p;
    p.on = (src.st.p != nullptr);
    p.g = make_float4(1.f, 1.f, 1.f, 1.f);
    p.be = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.on) {
        p.g = *reinterpret_cast<const float4*>(src.gamma + c);
        p.be = *reinterpret_cast<const float4*>(src.beta + c);
    }
    return p;
}
__device__ __forceinline__ Affine4 finish_affine(const AffineParams& p, float mean, float rstd) {
    Affine4 a;
    a.on = p.on;
    a.mu = mean;
    a.sc = make_float4(1.f, 1.f, 1.f, 1.f);
    a.be = p.be;
    if (a.on) a.sc = make_float4(rstd * p.g.x, rstd * p.g.y, rstd * p.g.z, rstd * p.g.w);
    return a;
}
__device__ __forceinline__ float4 apply_affine(const Affine4& a, float4 v) {
    if (a.on) {
        v.x = (v.x - a.mu) * a.sc.x + a.be.x;
        v.y = (v.y - a.mu) * a.sc.y + a.be.y;
        v.z = (v.z - a.mu) * a.sc.z + a.be.z;
        v.w = (v.w - a.mu) * a.sc.w + a.be.w;
    }
    return v;
}

// -------------------------------------------------------------------------------------------------
// nn.MaxPool2d(2) (models/Unet_FiLmLayer.py:132,159) of the GroupNorm-finished producer.
// The max does not commute with an affine of negative gain, so the affine is applied per tap.
__global__ __launch_bounds__(256) void pool_kernel(const AffineSrc src, float* __restrict__ dst, int H, int W,
                                                   int rows_per_block) {
    __shared__ float sm[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int C = src.C, C4 = C >> 2;
    const int c4 = tid % C4, rl = tid / C4, rpp = 256 / C4;
    const int Ho = H >> 1, Wo = W >> 1, HWo = Ho * Wo;
    const AffineParams ap = load_affine_params(src, c4 * 4);
    float mean, rstd;
    block_sample_stats(src, b, sm, mean, rstd);
    const Affine4 af = finish_affine(ap, mean, rstd);
    const int r_end = min(HWo, (int)(blockIdx.y + 1) * rows_per_block);
    for (int ro = blockIdx.y * rows_per_block + rl; ro < r_end; ro += rpp) {
        const int ho = ro / Wo, wo = ro - ho * Wo;
        const float* base = src.x + (((size_t)b * H + 2 * ho) * W + 2 * wo) * C + c4 * 4;
        float4 v00 = apply_affine(af, *reinterpret_cast<const float4*>(base));
        float4 v01 = apply_affine(af, *reinterpret_cast<const float4*>(base + C));
        float4 v10 = apply_affine(af, *reinterpret_cast<const float4*>(base + (size_t)W * C));
        float4 v11 = apply_affine(af, *reinterpret_cast<const float4*>(base + (size_t)W * C + C));
        float4 m;
        m.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
        m.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
        m.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
        m.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
        *reinterpret_cast<float4*>(dst + ((size_t)b * HWo + ro) * C + c4 * 4) = m;
    }
}

static inline int rows_per_block_for(int HW, int B) {
    // enough blocks to fill 256 CUs several times over, but no fewer than 16 rows per block
    int chunks = 1;
    while ((long long)B * chunks < 2048 && HW / (chunks * 2) >= 16) chunks *= 2;
    return (HW + chunks - 1) / chunks;
}

hipError_t launch_pool(const AffineSrc& src, float* dst, int B, int H, int W, hipStream_t s) {
    const int C4 = src.C / 4;
    if (src.C % 4 != 0 || C4 > 256 || 256 % C4 != 0 || (H & 1) || (W & 1) || B <= 0) return hipErrorInvalidValue;
    const int HWo = (H / 2) * (W / 2);
    const int rpb = rows_per_block_for(HWo, B);
    hipLaunchKernelGGL(pool_kernel, dim3(B, (HWo + rpb - 1) / rpb), dim3(256), 0, s, src, dst, H, W, rpb);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) + torch.cat([x, x_res], dim=1)
// (models/Unet_FiLmLayer.py:191,217-218): upsampled channels first, skip channels after.
__global__ __launch_bounds__(256) void upcat_kernel(const AffineSrc up, const AffineSrc skip, float* __restrict__ dst,
                                                    int Hin, int Win, int rows_per_block) {
    __shared__ float sm_u[2], sm_s[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int Cu = up.C, Cs = skip.C, C = Cu + Cs, C4 = C >> 2;
    // thread -> (row, 4-channel group).  When both halves are equally wide (every level of this U-Net) the first 128
    // threads take the upsampled channels and the last 128 the skip channels, so that a wave runs ONE of the two
    // paths (4 gathered loads + blend, or 1 load) instead of both with half its lanes masked.
    int c4, rl;
    const int rpp = 256 / C4;
    if (Cu == Cs && C4 <= 128) {
        const int h4 = C4 >> 1, part = tid >> 7, t2 = tid & 127;
        c4 = part * h4 + t2 % h4;
        rl = t2 / h4;
    } else {
        c4 = tid % C4;
        rl = tid / C4;
    }
    const int Ho = 2 * Hin, Wo = 2 * Win, HWo = Ho * Wo;
    const bool is_up = (c4 * 4 < Cu);
    const int cl = is_up ? c4 * 4 : c4 * 4 - Cu;
    const AffineParams ap = is_up ? load_affine_params(up, cl) : load_affine_params(skip, cl);      // before the statistics
    // the statistics of both sources at once: wave 0 sums the slots of `up`, wave 1 those of `skip`; one barrier
    float mu_u = 0.f, rs_u = 1.f, mu_s = 0.f, rs_s = 1.f;
    if (up.st.p != nullptr || skip.st.p != nullptr) {
        const int wv = tid >> 6, ln = tid & 63;
        if (wv == 0 && up.st.p != nullptr) {
            float m, r;
            sample_mean_rstd_wave(up.st, b, ln, m, r);
            if (ln == 0) { sm_u[0] = m; sm_u[1] = r; }
        }
        if (wv == 1 && skip.st.p != nullptr) {
            float m, r;
            sample_mean_rstd_wave(skip.st, b, ln, m, r);
            if (ln == 0) { sm_s[0] = m; sm_s[1] = r; }
        }
        __syncthreads();
        if (up.st.p != nullptr) { mu_u = sm_u[0]; rs_u = sm_u[1]; }
        if (skip.st.p != nullptr) { mu_s = sm_s[0]; rs_s = sm_s[1]; }
    }
    const Affine4 af = is_up ? finish_affine(ap, mu_u, rs_u) : finish_affine(ap, mu_s, rs_s);
    const float sch = (Ho > 1) ? (float)(Hin - 1) / (float)(Ho - 1) : 0.f;
    const float scw = (Wo > 1) ? (float)(Win - 1) / (float)(Wo - 1) : 0.f;
    const int r_end = min(HWo, (int)(blockIdx.y + 1) * rows_per_block);
    for (int ro = blockIdx.y * rows_per_block + rl; ro < r_end; ro += rpp) {
        const int ho = ro / Wo, wo = ro - ho * Wo;
        float4 o;
        if (is_up) {
            const float fh = sch * (float)ho, fw = scw * (float)wo;
            const int h0 = (int)fh, w0 = (int)fw;
            const int h1 = h0 + (h0 < Hin - 1 ? 1 : 0), w1 = w0 + (w0 < Win - 1 ? 1 : 0);
            const float lh1 = fminf(fmaxf(fh - (float)h0, 0.f), 1.f), lw1 = fminf(fmaxf(fw - (float)w0, 0.f), 1.f);
            const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
            const float* base = up.x + (size_t)b * Hin * Win * Cu + cl;
            const float4 v00 = apply_affine(af, *reinterpret_cast<const float4*>(base + ((size_t)h0 * Win + w0) * Cu));
            const float4 v01 = apply_affine(af, *reinterpret_cast<const float4*>(base + ((size_t)h0 * Win + w1) * Cu));
            const float4 v10 = apply_affine(af, *reinterpret_cast<const float4*>(base + ((size_t)h1 * Win + w0) * Cu));
            const float4 v11 = apply_affine(af, *reinterpret_cast<const float4*>(base + ((size_t)h1 * Win + w1) * Cu));
            o.x = lh0 * (lw0 * v00.x + lw1 * v01.x) + lh1 * (lw0 * v10.x + lw1 * v11.x);
            o.y = lh0 * (lw0 * v00.y + lw1 * v01.y) + lh1 * (lw0 * v10.y + lw1 * v11.y);
            o.z = lh0 * (lw0 * v00.z + lw1 * v01.z) + lh1 * (lw0 * v10.z + lw1 * v11.z);
            o.w = lh0 * (lw0 * v00.w + lw1 * v01.w) + lh1 * (lw0 * v10.w + lw1 * v11.w);
        } else {
            // (the last reader of the skip tensor: streamed past the caches)
            typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
            const nt_f32x4 sv = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(skip.x + ((size_t)b * HWo + ro) * Cs + cl));
            o = apply_affine(af, make_float4(sv.x, sv.y, sv.z, sv.w));
        }
        *reinterpret_cast<float4*>(dst + ((size_t)b * HWo + ro) * C + c4 * 4) = o;
    }
}

hipError_t launch_upcat(const AffineSrc& up, const AffineSrc& skip, float* dst, int B, int Hin, int Win,
                        hipStream_t s) {
    const int C = up.C + skip.C, C4 = C / 4;
    if (up.C % 4 != 0 || skip.C % 4 != 0 || C4 > 256 || 256 % C4 != 0 || B <= 0) return hipErrorInvalidValue;
    const int HWo = 4 * Hin * Win;
    const int rpb = rows_per_block_for(HWo, B);
    hipLaunchKernelGGL(upcat_kernel, dim3(B, (HWo + rpb - 1) / rpb), dim3(256), 0, s, up, skip, dst, Hin, Win, rpb);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Tail of DownSample/UpSample.forward (models/Unet_FiLmLayer.py:165-177, :222-234), applied to the
// raw output of the block's last conv:  y = scale * (GN(x) + emb_t) + bias.
//   emb_t = temb_table[t][c]  (Linear(SiLU(pos_encoding(t))) tabulated for every t at load time)
//   [scale | bias] = film[b][0:C | C:2C]  (Linear(Mish(cond)), step-invariant, hoisted out of the loop)
__global__ __launch_bounds__(256) void film_apply_kernel(const AffineSrc src, const float* __restrict__ temb,
                                                         const int* __restrict__ t_dev, int t_count,
                                                         const float* __restrict__ film, float* __restrict__ dst,
                                                         double* __restrict__ row_stats, int HW, int rows_per_block) {
    __shared__ float sm[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    float mean, rstd;
    block_sample_stats(src, b, sm, mean, rstd);
    const int C = src.C, C4 = C >> 2;
    const int c4 = tid % C4, rl = tid / C4, rpp = 256 / C4;
    const Affine4 af = make_affine(src, c4 * 4, mean, rstd);
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f), fs = make_float4(1.f, 1.f, 1.f, 1.f), fb = e;
    if (temb != nullptr) {
        const int t = t_dev[t_count == 1 ? 0 : b];
        e = *reinterpret_cast<const float4*>(temb + (size_t)t * C + c4 * 4);
    }
    if (film != nullptr) {
        fs = *reinterpret_cast<const float4*>(film + (size_t)b * 2 * C + c4 * 4);
        fb = *reinterpret_cast<const float4*>(film + (size_t)b * 2 * C + C + c4 * 4);
    }
    const int r_end = min(HW, (int)(blockIdx.y + 1) * rows_per_block);
    for (int r = blockIdx.y * rows_per_block + rl; r < r_end; r += rpp) {
        const size_t o = ((size_t)b * HW + r) * C + c4 * 4;
        float4 v = apply_affine(af, *reinterpret_cast<const float4*>(src.x + o));
        if (temb != nullptr) { v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
        if (film != nullptr) {
            v.x = fs.x * v.x + fb.x; v.y = fs.y * v.y + fb.y; v.z = fs.z * v.z + fb.z; v.w = fs.w * v.w + fb.w;
        }
        *reinterpret_cast<float4*>(dst + o) = v;
        if (row_stats != nullptr) {      // a token row lives on C4 consecutive lanes of one wave (C4 in {16,32,64})
            float s1 = (v.x + v.y) + (v.z + v.w);
            float s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            for (int o2 = C4 >> 1; o2 > 0; o2 >>= 1) {
                s1 += __shfl_xor(s1, o2, 64);
                s2 += __shfl_xor(s2, o2, 64);
            }
            if (c4 == 0) {
                double* rs = row_stats + ((size_t)b * HW + r) * 2;
                rs[0] = (double)s1;
                rs[1] = (double)s2;
            }
        }
    }
}

// The same tail as a per-(sample, channel) affine  y = A x + B  for consumers that apply it while LOADING the raw conv
// output (sa_fused64_kernel, sa_qkv128_kernel, sa_tail128_kernel) instead of reading a materialised y:
//   A = scale * rstd * gamma,   B = scale * (beta - mean * rstd * gamma + emb_t) + bias;   ab[b] = [A (C) | B (C)]
__global__ __launch_bounds__(256) void film_coef_kernel(const AffineSrc src, const float* __restrict__ temb,
                                                        const int* __restrict__ t_dev, int t_count,
                                                        const float* __restrict__ film, float* __restrict__ ab) {
    __shared__ float sm[2];
    const int b = blockIdx.x;
    const int C = src.C;
    // everything that does not depend on the statistics is requested BEFORE they are summed (the launch is a chain of dependent
    // round trips otherwise: slots -> barrier -> t -> temb row): channel c0 of this thread, the usual case C <= blockDim
    const int c0 = threadIdx.x;
    const bool has0 = c0 < C;
    float g0 = 1.f, b0 = 0.f, e0 = 0.f, fs0 = 1.f, fb0 = 0.f;
    if (has0) {
        if (src.st.p != nullptr) { g0 = src.gamma[c0]; b0 = src.beta[c0]; }
        if (temb != nullptr) e0 = temb[(size_t)t_dev[t_count == 1 ? 0 : b] * C + c0];
        if (film != nullptr) { fs0 = film[(size_t)b * 2 * C + c0]; fb0 = film[(size_t)b * 2 * C + C + c0]; }
    }
    float mean, rstd;
    block_sample_stats(src, b, sm, mean, rstd);
    if (has0) {
        float sc = 1.f, be = 0.f;
        if (src.st.p != nullptr) { sc = rstd * g0; be = b0 - mean * sc; }
        be += e0;
        ab[(size_t)b * 2 * C + c0] = fs0 * sc;
        ab[(size_t)b * 2 * C + C + c0] = fs0 * be + fb0;
    }
    for (int c = c0 + blockDim.x; c < C; c += blockDim.x) {
        float sc = 1.f, be = 0.f;
        if (src.st.p != nullptr) { sc = rstd * src.gamma[c]; be = src.beta[c] - mean * sc; }
        if (temb != nullptr) be += temb[(size_t)t_dev[t_count == 1 ? 0 : b] * C + c];
        float fs = 1.f, fb = 0.f;
        if (film != nullptr) { fs = film[(size_t)b * 2 * C + c]; fb = film[(size_t)b * 2 * C + C + c]; }
        ab[(size_t)b * 2 * C + c] = fs * sc;
        ab[(size_t)b * 2 * C + C + c] = fs * be + fb;
    }
}
hipError_t launch_film_coef(const AffineSrc& src, const float* temb_table, const int* t_dev, int t_count, const float* film,
                            float* ab, int B, hipStream_t s) {
    if (B <= 0 || src.C <= 0 || ab == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(film_coef_kernel, dim3(B), dim3(std::min(256, ((src.C + 63) / 64) * 64)), 0, s, src, temb_table, t_dev,
                       t_count, film, ab);
    return hipGetLastError();
}

hipError_t launch_film_apply(const AffineSrc& src, const float* temb_table, const int* t_dev, int t_count,
                             const float* film, float* dst, double* row_stats, int B, int HW, hipStream_t s) {
    const int C4 = src.C / 4;
    if (src.C % 4 != 0 || C4 > 256 || 256 % C4 != 0 || B <= 0) return hipErrorInvalidValue;
    if (row_stats != nullptr && !(C4 == 16 || C4 == 32 || C4 == 64)) return hipErrorInvalidValue;
    const int rpb = rows_per_block_for(HW, B);
    hipLaunchKernelGGL(film_apply_kernel, dim3(B, (HW + rpb - 1) / rpb), dim3(256), 0, s, src, temb_table, t_dev,
                       t_count, film, dst, row_stats, HW, rpb);
    return hipGetLastError();
}

hipError_t launch_gn_apply(const AffineSrc& src, float* dst, int B, int HW, hipStream_t s) {
    return launch_film_apply(src, nullptr, nullptr, 1, nullptr, dst, nullptr, B, HW, s);
}

// -------------------------------------------------------------------------------------------------
// nn.LayerNorm([C]) over the channel dim of each token (models/Unet_FiLmLayer.py:61,63,78,81).
// One wave per token row, two-pass mean / variance in registers.
template <int EPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ bt, float* __restrict__ y,
                                                        int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    constexpr int C = EPL * 64;
    const float* xr = x + (size_t)row * C;
    float v[EPL];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) { v[j] = xr[lane + 64 * j]; s += v[j]; }
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) { const float d = v[j] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + 1e-5f);
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int c = lane + 64 * j;
        y[(size_t)row * C + c] = (v[j] - mean) * rstd * g[c] + bt[c];
    }
}

hipError_t launch_layernorm(const float* x, const float* g, const float* b, float* y, int rows, int C,
                            hipStream_t s) {
    if (rows <= 0) return hipErrorInvalidValue;
    const dim3 grid((rows + 3) / 4), block(256);
    switch (C) {
        case 64:  hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, g, b, y, rows); break;
        case 128: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, g, b, y, rows); break;
        case 256: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, x, g, b, y, rows); break;
        case 512: hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, x, g, b, y, rows); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// nn.Mish() + Flatten of the FiLM cond encoder (models/Unet_FiLmLayer.py:150-151), zero-padded to Kp
__global__ void mish_pad_kernel(const float* __restrict__ cond, float* __restrict__ dst, int B, int cond_dim, int Kp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * Kp) return;
    const int b = (int)(i / Kp), k = (int)(i - (size_t)b * Kp);
    float v = 0.f;
    if (k < cond_dim) {
        const float x = cond[(size_t)b * cond_dim + k];
        const float sp = (x > 20.f) ? x : log1pf(expf(x));
        v = x * tanhf(sp);
    }
    dst[i] = v;
}
hipError_t launch_mish_pad(const float* cond, float* dst, int B, int cond_dim, int Kp, hipStream_t s) {
    const size_t n = (size_t)B * Kp;
    hipLaunchKernelGGL(mish_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cond, dst, B, cond_dim, Kp);
    return hipGetLastError();
}

// device gelu_erf exposed for an op-level test (spdm_op_gelu)
__global__ void gelu_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = gelu_erf(x[i]);
}
hipError_t launch_gelu(const float* x, float* y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}

// nn.SiLU() of emb_layer (models/Unet_FiLmLayer.py:137)
__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = v / (1.0f + expf(-v)); }
}
hipError_t launch_silu(const float* x, float* y, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(silu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Philox4x32-10 -> 4 normals (Box-Muller); oracle/philox_ref.py is the CPU restatement.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned sample, unsigned step, unsigned elem) {
    unsigned w[4];
    philox4x32_10(elem >> 2, sample, step, 0u, (unsigned)seed, (unsigned)(seed >> 32), w);
    const unsigned comp = elem & 3u;
    const unsigned wa = (comp < 2) ? w[0] : w[2], wb = (comp < 2) ? w[1] : w[3];
    const float u1 = ((float)(wa >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float u2 = ((float)(wb >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float r = sqrtf(-2.0f * logf(u1));
    const float ang = 6.283185307179586f * u2;
    return (comp & 1u) ? r * sinf(ang) : r * cosf(ang);
}

// -------------------------------------------------------------------------------------------------
// outc (Conv2d(64,1,1) + bias, models/Unet_FiLmLayer.py:264,308) + unpad (:36-41,310) + scheduler
// step (diffusers 0.17.1 DDPMScheduler/DDIMScheduler.step, called at models/diffusion_ddpm.py:274)
// + add_constraints inpainting (:216-219), one launch.  16 lanes cooperate on one (b, h, d) element:
// each reads 4 of the 64 feature channels, a 4-step xor-shuffle finishes the dot product.  The
// scheduler arithmetic uses explicitly rounded fp32 operations (no FMA contraction) in the
// library's operation order, so that with identical eps the update is bit-identical to torch's.
__global__ __launch_bounds__(256) void out_step_kernel(const StepArgs a) {
    const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t e = gt >> 4;
    const int l16 = (int)(gt & 15);
    const size_t E = (size_t)a.B * a.H0 * a.D;
    const bool live = e < E;
    const size_t ee = live ? e : 0;
    const int HD = a.H0 * a.D;
    const int b = (int)(ee / HD), he = (int)(ee - (size_t)b * HD);
    const int h0 = he / a.D, d = he - h0 * a.D;
    const int p = (h0 + a.lh) * a.Wp + (d + a.lw);
    const float4 f = *reinterpret_cast<const float4*>(a.feat + ((size_t)b * a.Hp * a.Wp + p) * 64 + l16 * 4);
    const float4 w = *reinterpret_cast<const float4*>(a.w + l16 * 4);
    float dot = (f.x * w.x + f.y * w.y) + (f.z * w.z + f.w * w.w);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    const float eps = dot + a.bias;
    if (!live || l16 != 0) return;
    if (a.eps_out != nullptr) {
        a.eps_out[e] = eps;
        if (!(fabsf(eps) <= 3.0e38f)) *a.flag_dev = 1;
        return;
    }

    const int i = *a.step_dev;
    const float* c = a.coef + (size_t)i * 6;
    const float x = a.x[e];
    const float* inpaint = a.inpaint;
    const float* noise = a.noise;
    float* history = a.history;
    if (a.ptrs_dev != nullptr) {           // uniform (scalar) loads of the session's buffers
        inpaint = static_cast<const float*>(a.ptrs_dev[0]);
        noise = static_cast<const float*>(a.ptrs_dev[1]);
        history = const_cast<float*>(static_cast<const float*>(a.ptrs_dev[2]));
    }
    const float x0 = __fdiv_rn(__fsub_rn(x, __fmul_rn(c[0], eps)), c[1]);
    float prev;
    if (a.kind == 0) {   // DDPM
        prev = __fadd_rn(__fmul_rn(c[2], x0), __fmul_rn(c[3], x));
        if (c[5] != 0.f) {
            const float z = (noise != nullptr)
                                ? noise[((size_t)i * a.B + b) * HD + he]
                                : philox_normal(a.rng_dev[0], (unsigned)(a.rng_dev[1] + (unsigned long long)b),
                                                (unsigned)i, (unsigned)he);
            prev = __fadd_rn(prev, __fmul_rn(c[5], z));
        }
    } else {             // DDIM, eta = 0
        prev = __fadd_rn(__fmul_rn(c[2], x0), __fmul_rn(c[4], eps));
    }
    if (h0 < a.inp_h && inpaint != nullptr)
        prev = inpaint[(a.inpaint_per_sample ? (size_t)b * a.inp_h * a.D : 0) + (size_t)h0 * a.D + d];
    a.x[e] = prev;
    // overflow guard of the split-precision contractions (|activation| < 4094): surfaced by spdm_sample_nonfinite
    if (!(fabsf(prev) <= 3.0e38f)) *a.flag_dev = 1;
    if (history != nullptr) history[((size_t)(i + 1) * a.B + b) * HD + he] = prev;
}

hipError_t launch_out_step(const StepArgs& a, hipStream_t s) {
    const size_t E = (size_t)a.B * a.H0 * a.D;
    if (E == 0) return hipErrorInvalidValue;
    const size_t threads = E * 16;
    hipLaunchKernelGGL(out_step_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Split-K combine (kernels.h): dst = sum over ks (ascending: deterministic) of the partial slabs a split-K implicit-GEMM
// launch left, plus the GroupNorm(1,C) partial sums of dst -- the epilogue the GEMM itself could not run on partial
// sums.  One workgroup = `rows` consecutive rows of ONE sample x all N channels (a contiguous run of rows * N floats);
// per-thread fp32 sums of <= 32 values -> fp64 -> fixed-order reduction.
__global__ __launch_bounds__(256) void splitk_combine_kernel(const float* __restrict__ partial, int ksplit, size_t slab,
                                                             float* __restrict__ dst, int N, int HW, int rows,
                                                             double* __restrict__ stats, int slots) {
    __shared__ double red[2][4];
    const int per_sample = HW / rows;
    const int b = blockIdx.x / per_sample, r = blockIdx.x - b * per_sample;
    const size_t base = ((size_t)b * HW + (size_t)r * rows) * N;
    const int n4 = rows * N / 4;
    double d1 = 0.0, d2 = 0.0;
    for (int i0 = 0; i0 < n4; i0 += 256 * 8) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + j * 256 + threadIdx.x;
            if (i < n4) {
                const float* p = partial + base + (size_t)i * 4;
                float4 v = *reinterpret_cast<const float4*>(p);
                int k = 1;
                for (; k + 4 <= ksplit; k += 4) {          // four slabs in flight per round trip; the ADDS stay in slab order
                    const float4 w0 = *reinterpret_cast<const float4*>(p + (size_t)k * slab);
                    const float4 w1 = *reinterpret_cast<const float4*>(p + (size_t)(k + 1) * slab);
                    const float4 w2 = *reinterpret_cast<const float4*>(p + (size_t)(k + 2) * slab);
                    const float4 w3 = *reinterpret_cast<const float4*>(p + (size_t)(k + 3) * slab);
                    v.x += w0.x; v.y += w0.y; v.z += w0.z; v.w += w0.w;
                    v.x += w1.x; v.y += w1.y; v.z += w1.z; v.w += w1.w;
                    v.x += w2.x; v.y += w2.y; v.z += w2.z; v.w += w2.w;
                    v.x += w3.x; v.y += w3.y; v.z += w3.z; v.w += w3.w;
                }
                for (; k < ksplit; ++k) {
                    const float4 w = *reinterpret_cast<const float4*>(p + (size_t)k * slab);
                    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                }
                *reinterpret_cast<float4*>(dst + base + (size_t)i * 4) = v;
                s1 += (v.x + v.y) + (v.z + v.w);
                s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
        }
        d1 += (double)s1;
        d2 += (double)s2;
    }
    d1 = wave_sum_f64(d1);                          // (DPP + permlane swaps: fixed order, no LDS crossbar)
    d2 = wave_sum_f64(d2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d1; red[1][threadIdx.x >> 6] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = stats + ((size_t)b * slots + r) * 2;       // slot of tile mt = b HW / rows + r: (mt - b HW / rows) * 1 + 0
        o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

hipError_t launch_splitk_combine(const float* partial, int ksplit, float* dst, int M, int N, int HW, double* stats,
                                 hipStream_t s) {
    if (!partial || !dst || !stats || ksplit < 2 || M <= 0 || N % 4 != 0 || HW < 1 || M % HW != 0) return hipErrorInvalidValue;
    const int rows = combine_rows(HW, N);
    if (HW % rows != 0) return hipErrorInvalidValue;
    const int B = M / HW;
    hipLaunchKernelGGL(splitk_combine_kernel, dim3((unsigned)(B * (HW / rows))), dim3(256), 0, s, partial, ksplit,
                       (size_t)M * N, dst, N, HW, rows, stats, stats_slots(HW, rows, 1));
    return hipGetLastError();
}

// loop bookkeeping kept on the device so that one denoise step is the same launch sequence for
// every iteration (hipGraph-replayable): step <- step + 1, t <- timesteps[step]
__global__ void advance_kernel(int* step_dev, int* t_dev, const int* timesteps, int n_steps, int set_to) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int i = (set_to >= 0) ? set_to : (*step_dev + 1);
        *step_dev = i;
        *t_dev = timesteps[min(max(i, 0), n_steps - 1)];
    }
}
hipError_t launch_advance(int* step_dev, int* t_dev, const int* timesteps_dev, int n_steps, hipStream_t s) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, s, step_dev, t_dev, timesteps_dev, n_steps, -1);
    return hipGetLastError();
}
hipError_t launch_set_step(int* step_dev, int* t_dev, const int* timesteps_dev, int n_steps, int i, hipStream_t s) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, s, step_dev, t_dev, timesteps_dev, n_steps, i);
    return hipGetLastError();
}

}  // namespace spdm
