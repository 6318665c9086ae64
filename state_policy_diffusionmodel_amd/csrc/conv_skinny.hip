// conv_skinny.hip -- 3x3 / 3x1 convolution for launches whose grid would not fill the chip: first written for a handful of rows
// (the closed-loop caller: run_predictions.py:140-175 calls sample() at batch 1, so the coarse levels have M = B H_l W_l = 16 or
// 4 rows against K = 9 Cin up to 4608), now every layer whose (rows / 16 | 32 | 64) x (N / 64) tiles stay within 256 workgroups
// (conv_skinny_geometry below: all 31 convolutions at batch 1, the coarse levels up to batch 1024).
//
// Same math, operand formats and epilogue contract as conv_wide.hip / conv_gemm.hip (split-fp16 operands, hi*hi + hi*lo + lo*hi
// into fp32, GroupNorm(1,C) -> GELU prologue applied at staging, GroupNorm partial sums as the epilogue; replaces
// nn.Conv2d(k=3, padding=1, bias=False) + the GroupNorm/GELU around it, models/Unet_FiLmLayer.py:101-115).  What such a launch
// is bound by is how fast its weights (up to 9.4 MB) stream into the chip, through how many wave fronts -- the matrix
// work is nothing.  So the shape is the opposite of the big-batch kernels:
//
//   * one workgroup = 8 waves = one (<= 64 rows) x (64 | 32 columns) output tile over the WHOLE K range;
//   * the input slab of ALL chunks (rows + halo, every 32-channel chunk, prologue applied once) is staged in LDS up
//     front -- it is tiny (<= 87 KB);
//   * the 8 waves split K: wave w takes the (chunk, tap) items i = w, w + 8, ... and streams their weight fragments
//     straight from global memory in fragment order (frag_order_weights, kernels.h), one item ahead; every wave
//     accumulates the whole tile;
//   * the 8 partial tiles are added through LDS in the fixed order w = 0..7 (deterministic, no atomics) and the epilogue --
//     output rows, per-sample fp64 GroupNorm partial sums -- runs in the same kernel: no partial slabs in HBM, no combine
//     launch (the split-K path of the larger kernels pays ~10 us per layer for those at this size).
#include <algorithm>

#include "device_utils.h"

namespace spdm {

namespace {

typedef float k_f32x4 __attribute__((ext_vector_type(4)));
typedef float k_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 k_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 k_f16x2 __attribute__((ext_vector_type(2)));

constexpr float K_ACT_SCALE = 16.0f;          // the split scheme's operand scales (2^4 activations, 2^7 weights)
constexpr float K_DESCALE = 1.0f / 2048.0f;
constexpr int K_CK = 32, K_LDK = 36;          // channels per chunk; floats per slab row (128 + 16 pad bytes)
constexpr int K_WAVES = 8, K_NTHR = 512;

__device__ __forceinline__ k_f32x2 ksplit2(float a, float b) {
    unsigned h, l;
    split_pair_f16(a * K_ACT_SCALE, b * K_ACT_SCALE, h, l);
    return k_f32x2{__builtin_bit_cast(float, h), __builtin_bit_cast(float, l)};
}

#ifdef SPDM_DIAG_SKINNY
// diagnostic builds: phase stamps (s_memrealtime, 10-ns ticks) of workgroup 0, thread 0 of the launch with (K, N) == g_skinny_sel
__device__ unsigned long long g_skinny_stamps[16];
__device__ int g_skinny_sel[2];
#define SKINNY_STAMP(k_) if (blockIdx.x == 0 && threadIdx.x == 0 && a.K == g_skinny_sel[0] && a.N == g_skinny_sel[1]) { \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g_skinny_stamps[k_] = __builtin_amdgcn_s_memrealtime(); }
#else
#define SKINNY_STAMP(k_)
#endif

__device__ __forceinline__ k_f32x4 k_affine(k_f32x4 v, float mu, k_f32x4 sc, k_f32x4 be) {
    return k_f32x4{(v.x - mu) * sc.x + be.x, (v.y - mu) * sc.y + be.y, (v.z - mu) * sc.z + be.z, (v.w - mu) * sc.w + be.w};
}
// One 16-byte piece (channels ch 32 + c4 4 ..) of input row m of a FUSED source (kernels.h, GemmArgs): the value the
// materialising kernels (pool_kernel / upcat_kernel, elementwise.hip) would have written there -- same arithmetic, tap by tap.
// smean / srstd: statistics of src per sample (index b - bh_first), smean1 / srstd1: of the skip tensor (PRO_UPCAT).
template <int PRO>
__device__ __forceinline__ k_f32x4 skinny_fused_piece(const GemmArgs& a, int m, int ch, int c4, int bh_first, const float* smean,
                                                      const float* srstd, const float* smean1, const float* srstd1) {
    const int HW = a.HW, W = a.W, H = a.H;
    const int b = m / HW, pp = m - b * HW;
    const int h = pp / W, w = pp - h * W;
    const int c = ch * K_CK + c4 * 4;
    if (PRO == PRO_UPCAT && c >= a.up_C) {          // skip half: one row of the same level
        k_f32x4 v = *reinterpret_cast<const k_f32x4*>(a.skip + (size_t)m * a.skip_ld + (c - a.up_C));
        if (a.skip_stats.p != nullptr) {
            const float rs = srstd1[b - bh_first], mu = smean1[b - bh_first];
            const k_f32x4 g4 = *reinterpret_cast<const k_f32x4*>(a.skip_gamma + (c - a.up_C));
            const k_f32x4 b4 = *reinterpret_cast<const k_f32x4*>(a.skip_beta + (c - a.up_C));
            v = k_affine(v, mu, k_f32x4{rs * g4.x, rs * g4.y, rs * g4.z, rs * g4.w}, b4);
        }
        return v;
    }
    const bool gn = a.pro_stats.p != nullptr;
    float mu = 0.f;
    k_f32x4 sc = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    if (gn) {
        const float rs = srstd[b - bh_first];
        mu = smean[b - bh_first];
        const k_f32x4 g4 = *reinterpret_cast<const k_f32x4*>(a.pro_gamma + c);
        be = *reinterpret_cast<const k_f32x4*>(a.pro_beta + c);
        sc = k_f32x4{rs * g4.x, rs * g4.y, rs * g4.z, rs * g4.w};
    }
    if (PRO == PRO_POOL) {                            // MaxPool2d(2) of the (2 H) x (2 W) map
        const int Ws = 2 * W;
        const float* base = a.src + ((size_t)b * 4 * HW + (size_t)(2 * h) * Ws + 2 * w) * a.src_ld + c;
        k_f32x4 v00 = *reinterpret_cast<const k_f32x4*>(base);
        k_f32x4 v01 = *reinterpret_cast<const k_f32x4*>(base + a.src_ld);
        k_f32x4 v10 = *reinterpret_cast<const k_f32x4*>(base + (size_t)Ws * a.src_ld);
        k_f32x4 v11 = *reinterpret_cast<const k_f32x4*>(base + (size_t)Ws * a.src_ld + a.src_ld);
        if (gn) { v00 = k_affine(v00, mu, sc, be); v01 = k_affine(v01, mu, sc, be); v10 = k_affine(v10, mu, sc, be); v11 = k_affine(v11, mu, sc, be); }
        return k_f32x4{fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x)), fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y)),
                       fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z)), fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w))};
    }
    // bilinear x2, align_corners=True, of the (H / 2) x (W / 2) map (upcat_kernel's arithmetic)
    const int Hin = H >> 1, Win = W >> 1;
    const float sch = (H > 1) ? (float)(Hin - 1) / (float)(H - 1) : 0.f;
    const float scw = (W > 1) ? (float)(Win - 1) / (float)(W - 1) : 0.f;
    const float fh = sch * (float)h, fw = scw * (float)w;
    const int h0 = (int)fh, w0 = (int)fw;
    const int h1 = h0 + (h0 < Hin - 1 ? 1 : 0), w1 = w0 + (w0 < Win - 1 ? 1 : 0);
    const float lh1 = fminf(fmaxf(fh - (float)h0, 0.f), 1.f), lw1 = fminf(fmaxf(fw - (float)w0, 0.f), 1.f);
    const float lh0 = 1.f - lh1, lw0 = 1.f - lw1;
    const float* base = a.src + (size_t)b * Hin * Win * a.src_ld + c;
    k_f32x4 v00 = *reinterpret_cast<const k_f32x4*>(base + ((size_t)h0 * Win + w0) * a.src_ld);
    k_f32x4 v01 = *reinterpret_cast<const k_f32x4*>(base + ((size_t)h0 * Win + w1) * a.src_ld);
    k_f32x4 v10 = *reinterpret_cast<const k_f32x4*>(base + ((size_t)h1 * Win + w0) * a.src_ld);
    k_f32x4 v11 = *reinterpret_cast<const k_f32x4*>(base + ((size_t)h1 * Win + w1) * a.src_ld);
    if (gn) { v00 = k_affine(v00, mu, sc, be); v01 = k_affine(v01, mu, sc, be); v10 = k_affine(v10, mu, sc, be); v11 = k_affine(v11, mu, sc, be); }
    return k_f32x4{lh0 * (lw0 * v00.x + lw1 * v01.x) + lh1 * (lw0 * v10.x + lw1 * v11.x),
                   lh0 * (lw0 * v00.y + lw1 * v01.y) + lh1 * (lw0 * v10.y + lw1 * v11.y),
                   lh0 * (lw0 * v00.z + lw1 * v01.z) + lh1 * (lw0 * v10.z + lw1 * v11.z),
                   lh0 * (lw0 * v00.w + lw1 * v01.w) + lh1 * (lw0 * v10.w + lw1 * v11.w)};
}

// RT row tiles of 16 (M_T = 16 RT), CT column tiles of 16 (N_T = 16 CT)
template <int RT, int CT, int PRO>
__global__ __launch_bounds__(K_NTHR, 1) void conv_skinny_kernel(const GemmArgs a, const int epi_slots, const int NS) {
    constexpr int M_T = RT * 16, N_T = CT * 16;
    constexpr bool fused = (PRO == PRO_POOL || PRO == PRO_UPCAT);     // the input is read through MaxPool / upsample + concat
    constexpr bool pro = (PRO == PRO_GN || PRO == PRO_GN_GELU), pro_gelu = (PRO == PRO_GN_GELU);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SKINNY_STAMP(0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = a.HW, W = a.W, H = a.H, M = a.M, K = a.K, N = a.N, taps = a.taps;
    const int halo = W + 1;
    const int QA = M_T + 2 * halo;            // slab rows of one chunk; row QA is all zero (what a masked tap reads)
    const int QZ = QA + 1;
    const int nch = K / K_CK;
    const int n_ntiles = N / N_T;
    const int mtile = blockIdx.x / n_ntiles, ntile = blockIdx.x - mtile * n_ntiles;
    const int m0 = mtile * M_T, n0 = ntile * N_T;

    float* Abuf = smem;                                   // [nch][QZ][K_LDK]
    float* smean = Abuf + (size_t)nch * QZ * K_LDK;       // [NS]
    float* srstd = smean + NS;                            // [NS]
    float* smean1 = srstd + NS;                           // [NS] x 2: statistics of the skip tensor (PRO_UPCAT)
    float* srstd1 = smean1 + NS;

    // ---- the loads nothing depends on go out first: this wave's first weight item and the first batch of slab pieces (a launch of
    //      this kernel is a chain of dependent memory round trips -- statistics, slab, weights -- so they are overlapped) ----
    const int l16 = lane & 15, kg = lane >> 4;
    const int nitem = nch * taps;
    // B operands: fragment-order weights, block ((tap nch + chunk) N/16 + nb16) x {hi, lo} of 256 floats
    const float* wfl = a.wgt_frag + ((size_t)(n0 >> 4) * 2) * 256 + lane * 4;
    const size_t wtap = (size_t)(N >> 4) * 2 * 256;          // floats per (tap, chunk)
    k_f16x8 fb[2][CT][2];
#define SKINNY_LOAD_B(slot_, item_)                                                                  \
    {                                                                                                \
        const int it_ = min((item_), nitem - 1);                                                     \
        const int ch_ = it_ / taps, tp_ = it_ - ch_ * taps;                                          \
        const float* p_ = wfl + (size_t)(tp_ * nch + ch_) * wtap;                                    \
        _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_) {                                         \
            fb[slot_][c_][0] = *reinterpret_cast<const k_f16x8*>(p_ + c_ * 512);                     \
            fb[slot_][c_][1] = *reinterpret_cast<const k_f16x8*>(p_ + c_ * 512 + 256);               \
        }                                                                                            \
    }
    SKINNY_LOAD_B(0, wave)
    const int npiece = nch * QA * 8;                          // slab pieces: (chunk, row q, 16-byte part c4)
    k_f32x4 v[8];
    int qs[8], cs[8];
#define SKINNY_LOAD_A(p0_)                                                                           \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                 \
        const int p = min((p0_) + j * K_NTHR + tid, npiece - 1);                                     \
        const int c4 = p & 7, rq = p >> 3;                                                           \
        const int q = rq % QA, ch = rq / QA;                                                         \
        const int m = min(max(m0 - halo + q, 0), M - 1);                                             \
        qs[j] = q; cs[j] = ch * 8 + c4;                                                              \
        v[j] = *reinterpret_cast<const k_f32x4*>(a.src + (size_t)m * a.src_ld + ch * K_CK + c4 * 4); \
    }
    if (!fused) SKINNY_LOAD_A(0)

    // ---- statistics of the samples the slab touches ----
    int bh_first = 0;
    if (fused) {
        const int lo = max(m0 - halo, 0), hi = min(m0 + M_T + halo, M) - 1;
        bh_first = lo / HW;
        const int ns = hi / HW - bh_first + 1;
        // one wave per (tensor, sample): the source's statistics, then the skip tensor's
        for (int t = wave; t < 2 * ns; t += K_WAVES) {
            const bool second = t >= ns;
            const StatsRef& st = second ? a.skip_stats : a.pro_stats;
            if ((second && PRO != PRO_UPCAT) || st.p == nullptr) continue;
            float mean, rstd;
            sample_mean_rstd_wave(st, bh_first + (second ? t - ns : t), lane, mean, rstd);
            if (lane == 0) {
                (second ? smean1 : smean)[second ? t - ns : t] = mean;
                (second ? srstd1 : srstd)[second ? t - ns : t] = rstd;
            }
        }
    }
    if (pro) {
        const int lo = max(m0 - halo, 0), hi = min(m0 + M_T + halo, M) - 1;
        bh_first = lo / HW;
        const int bh_last = hi / HW;
        for (int t = wave; t <= bh_last - bh_first; t += K_WAVES) {          // one wave per sample
            float mean, rstd;
            sample_mean_rstd_wave(a.pro_stats, bh_first + t, lane, mean, rstd);
            if (lane == 0) {
                smean[t] = mean;
                srstd[t] = rstd;
            }
        }
    }
    for (int t = tid; t < nch * K_LDK; t += K_NTHR) Abuf[((t / K_LDK) * QZ + QA) * K_LDK + t % K_LDK] = 0.f;
    if (pro || fused) __syncthreads();
    SKINNY_STAMP(1)

    if (fused) {
        // ---- the slab through the resampling op: one piece at a time per thread (4 gathered loads each; a small launch) ----
#pragma unroll 2
        for (int p = tid; p < npiece; p += K_NTHR) {
            const int c4 = p & 7, rq = p >> 3;
            const int q = rq % QA, ch = rq / QA;
            const int m = m0 - halo + q;
            k_f32x4 x = skinny_fused_piece<PRO>(a, min(max(m, 0), M - 1), ch, c4, bh_first, smean, srstd, smean1, srstd1);
            if (m < 0 || m >= M) x = k_f32x4{0.f, 0.f, 0.f, 0.f};
            const k_f32x2 p0_ = ksplit2(x.x, x.y), p1_ = ksplit2(x.z, x.w);
            float* row = Abuf + ((size_t)ch * QZ + q) * K_LDK;
            *reinterpret_cast<k_f32x2*>(row + c4 * 2) = k_f32x2{p0_.x, p1_.x};        // hi
            *reinterpret_cast<k_f32x2*>(row + 16 + c4 * 2) = k_f32x2{p0_.y, p1_.y};   // lo
        }
    }
    // ---- stage the whole slab: 8 pieces in flight per thread (the first batch was issued above) ----
    for (int p0 = 0; !fused && p0 < npiece; p0 += K_NTHR * 8) {
        if (p0) SKINNY_LOAD_A(p0)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (p0 + j * K_NTHR + tid < npiece) {
                const int q = qs[j], ch = cs[j] >> 3, c4 = cs[j] & 7;
                const int m = m0 - halo + q;
                k_f32x4 x = v[j];
                if (pro) {
                    const int bi = min(max(m, 0), M - 1) / HW - bh_first;
                    const float rs = srstd[bi], mu = smean[bi];
                    const k_f32x4 g4 = *reinterpret_cast<const k_f32x4*>(a.pro_gamma + ch * K_CK + c4 * 4);
                    const k_f32x4 b4 = *reinterpret_cast<const k_f32x4*>(a.pro_beta + ch * K_CK + c4 * 4);
                    x.x = (x.x - mu) * (rs * g4.x) + b4.x;
                    x.y = (x.y - mu) * (rs * g4.y) + b4.y;
                    x.z = (x.z - mu) * (rs * g4.z) + b4.z;
                    x.w = (x.w - mu) * (rs * g4.w) + b4.w;
                    if (pro_gelu) { x.x = gelu_erf(x.x); x.y = gelu_erf(x.y); x.z = gelu_erf(x.z); x.w = gelu_erf(x.w); }
                }
                if (m < 0 || m >= M) x = k_f32x4{0.f, 0.f, 0.f, 0.f};
                const k_f32x2 p0_ = ksplit2(x.x, x.y), p1_ = ksplit2(x.z, x.w);
                float* row = Abuf + ((size_t)ch * QZ + q) * K_LDK;
                *reinterpret_cast<k_f32x2*>(row + c4 * 2) = k_f32x2{p0_.x, p1_.x};        // hi
                *reinterpret_cast<k_f32x2*>(row + 16 + c4 * 2) = k_f32x2{p0_.y, p1_.y};   // lo
            }
        }
    }
#undef SKINNY_LOAD_A

    // ---- per-lane A rows and tap-validity masks (bit t of am[rt]: tap t of this lane's row in row tile rt is inside the image) ----
    unsigned am[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int m = m0 + rt * 16 + l16;
        unsigned mask = 0u;
        if (m < M) {
            const int p = m % HW;
            const int h = p / W, w = p - h * W;
            for (int t = 0; t < taps; ++t) {
                const int dh = (taps == 9) ? t / 3 - 1 : t - 1;
                const int dw = (taps == 9) ? t % 3 - 1 : 0;
                const bool ok = (h + dh >= 0) && (h + dh < H) && (w + dw >= 0) && (w + dw < W);
                mask |= (ok ? 1u : 0u) << t;
            }
        }
        am[rt] = mask;
    }
    const int aoff0 = (l16 + halo) * K_LDK + kg * 4;
    const int zoff = QA * K_LDK + kg * 4;

    k_f32x4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = k_f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();                                          // slab complete
    SKINNY_STAMP(2)

    // ---- this wave's items: i = wave, wave + 8, ... over (chunk, tap), chunk-major (item `wave` is already in flight) ----
    // The (chunk, tap) pair of an item is STEPPED, not divided out: per item a wave has only 6 RT MFMAs, and the two runtime
    // integer divisions (~80 scalar instructions) were most of the 0.26 us an item took (phase stamps + ISA, tools/probes/skinny_stamps.py).
    const int is9 = (taps == 9) ? 1 : 0;
    const int dch = K_WAVES / taps, dtp = K_WAVES - dch * taps;       // one step of K_WAVES items in (chunk, tap)
#define SKINNY_STEP(ch_, tp_) { tp_ += dtp; ch_ += dch; if (tp_ >= taps) { tp_ -= taps; ++ch_; } }
    int cch = wave / taps, ctp = wave - cch * taps;                    // the item being multiplied
    int wch = cch, wtp = ctp;                                          // the item whose weights are fetched next
    SKINNY_STEP(wch, wtp)
#define SKINNY_LOAD_B2(slot_)                                                                        \
    {                                                                                                \
        const float* p_ = wfl + (size_t)(wtp * nch + min(wch, nch - 1)) * wtap;                      \
        _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_) {                                         \
            fb[slot_][c_][0] = *reinterpret_cast<const k_f16x8*>(p_ + c_ * 512);                     \
            fb[slot_][c_][1] = *reinterpret_cast<const k_f16x8*>(p_ + c_ * 512 + 256);               \
        }                                                                                            \
        SKINNY_STEP(wch, wtp)                                                                        \
    }
    for (int item = wave; item < nitem; item += 2 * K_WAVES) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int it = item + half * K_WAVES;
            SKINNY_LOAD_B2(1 - half)                           // unconditional prefetch (clamped past the end)
            __builtin_amdgcn_sched_barrier(0);                 // (pinned: the compiler otherwise sinks the prefetch below the MFMAs)
            if (it < nitem) {
                const int q3 = (ctp * 11) >> 5;                // ctp / 3 for ctp < 9
                const int dh = is9 * q3 + (1 - is9) * ctp - 1, dw = is9 * (ctp - 3 * q3 - 1);
                const float* Ab = Abuf + (size_t)cch * QZ * K_LDK;
                const int shift = (dh * W + dw) * K_LDK;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int o = ((am[rt] >> ctp) & 1u) ? aoff0 + rt * 16 * K_LDK + shift : zoff;
                    const k_f16x8 a_h = *reinterpret_cast<const k_f16x8*>(Ab + o);
                    const k_f16x8 a_l = *reinterpret_cast<const k_f16x8*>(Ab + o + 16);
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, fb[half][c][0], acc[rt][c], 0, 0, 0);
                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, fb[half][c][1], acc[rt][c], 0, 0, 0);
                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l, fb[half][c][0], acc[rt][c], 0, 0, 0);
                    }
                }
            }
            SKINNY_STEP(cch, ctp)
        }
    }
#undef SKINNY_LOAD_B2
#undef SKINNY_STEP
#undef SKINNY_LOAD_B


    // ---- cross-wave reduction in the fixed order w = 0..7 (the slab is dead) ----
    SKINNY_STAMP(3)
    __syncthreads();
    SKINNY_STAMP(4)
    float* red = smem;                                        // [8 waves][M_T][N_T] fp32
    // accumulator layout: lane (l16, kg), register j of tile (rt, ct) = row 16 rt + 4 kg + j, column 16 ct + l16
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                red[((size_t)wave * M_T + rt * 16 + 4 * kg + j) * N_T + ct * 16 + l16] = acc[rt][ct][j];
    __syncthreads();
    SKINNY_STAMP(5)

    // thread -> (row, 4 columns); M_T * N_T / 4 pieces over 512 threads
    constexpr int NPIECE = M_T * N_T / 4;
    float* srow = red + (size_t)K_WAVES * M_T * N_T;          // [M_T][N_T / 4][2] fp32 partials of the final values
    for (int p = tid; p < NPIECE; p += K_NTHR) {
        const int r = p / (N_T / 4), c4 = p - r * (N_T / 4);
        k_f32x4 s = *reinterpret_cast<const k_f32x4*>(red + (size_t)r * N_T + c4 * 4);
#pragma unroll
        for (int w = 1; w < K_WAVES; ++w) s += *reinterpret_cast<const k_f32x4*>(red + ((size_t)w * M_T + r) * N_T + c4 * 4);
        s *= K_DESCALE;
        const int row = m0 + r;
        if (row < M) *reinterpret_cast<k_f32x4*>(a.dst + (size_t)row * a.dst_ld + n0 + c4 * 4) = s;
        srow[(r * (N_T / 4) + c4) * 2] = (s.x + s.y) + (s.z + s.w);
        srow[(r * (N_T / 4) + c4) * 2 + 1] = (s.x * s.x + s.y * s.y) + (s.z * s.z + s.w * s.w);
    }
    __syncthreads();
    SKINNY_STAMP(6)
    // per-sample totals in fp64, one WAVE per sample the tile touches: lane i adds the entries i, i + 64, ... of the sample's
    // rows, a fixed butterfly adds the lanes (deterministic; one thread walking 128 entries was ~1.5 us of every launch)
    {
        const int t_lo = m0, t_hi = min(m0 + M_T, M);
        if (t_hi > t_lo) {
            const int b_first = t_lo / HW, b_last = (t_hi - 1) / HW;
            for (int t = wave; t <= b_last - b_first; t += K_WAVES) {
                const int b = b_first + t;
                const int r_lo = max(b * HW, t_lo) - m0, r_hi = min((b + 1) * HW, t_hi) - m0;
                const int e_lo = r_lo * (N_T / 4), e_hi = r_hi * (N_T / 4);          // srow entries [e_lo, e_hi) are contiguous
                double s1 = 0.0, s2 = 0.0;
                for (int e = e_lo + lane; e < e_hi; e += 64) {
                    s1 += (double)srow[e * 2];
                    s2 += (double)srow[e * 2 + 1];
                }
                s1 = wave_sum_f64(s1);
                s2 = wave_sum_f64(s2);
                if (lane == 0) {
                    const int slot2 = (mtile - (b * HW) / M_T) * n_ntiles + ntile;
                    double* o = a.epi_stats + ((size_t)b * epi_slots + slot2) * 2;
                    o[0] = s1;
                    o[1] = s2;
                }
            }
        }
    }
    SKINNY_STAMP(7)
}

template <int RT, int CT>
hipError_t launch_skinny_rc(const GemmArgs& a, const GemmGeom& g, hipStream_t s) {
    constexpr int M_T = RT * 16, N_T = CT * 16;
    if (g.m_tile != M_T || g.n_tile != N_T) return hipErrorInvalidValue;
    const int halo = a.W + 1;
    const int QA = M_T + 2 * halo;
    const int nch = a.K / K_CK;
    const int NS = (((QA - 1) / a.HW + 2) + 3) & ~3;
    size_t lds = ((size_t)nch * (QA + 1) * K_LDK + 4 * NS) * sizeof(float);
    lds = std::max(lds, ((size_t)K_WAVES * M_T * N_T + (size_t)M_T * (N_T / 4) * 2) * sizeof(float));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int grid = ((a.M + M_T - 1) / M_T) * (a.N / N_T);
    const void* kern = nullptr;
#define SKINNY_GO(PRO_)                                                                              \
    {                                                                                                \
        auto k_ = conv_skinny_kernel<RT, CT, PRO_>;                                                  \
        kern = reinterpret_cast<const void*>(k_);                                                    \
        if (hipError_t e = allow_full_lds(kern); e != hipSuccess) return e;                          \
        hipLaunchKernelGGL(k_, dim3(grid), dim3(K_NTHR), lds, s, a, g.slots, NS);                    \
    }
    if (a.pro == PRO_NONE) SKINNY_GO(PRO_NONE)
    else if (a.pro == PRO_GN) SKINNY_GO(PRO_GN)
    else if (a.pro == PRO_POOL) SKINNY_GO(PRO_POOL)
    else if (a.pro == PRO_UPCAT) SKINNY_GO(PRO_UPCAT)
    else SKINNY_GO(PRO_GN_GELU)
#undef SKINNY_GO
    return hipGetLastError();
}

}  // namespace

#ifdef SPDM_DIAG_SKINNY
extern "C" int spdm_debug_skinny_select(int K, int N) {
    const int sel[2] = {K, N};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_skinny_sel), sel, sizeof(sel)) == hipSuccess ? 0 : -1;
}
extern "C" int spdm_debug_skinny_stamps(unsigned long long* out16) {
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_skinny_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

// LDS bytes of the up-front slab for a tile of m_tile rows
static size_t skinny_slab_bytes(int m_tile, int W, int K) {
    return (size_t)(K / K_CK) * (m_tile + 2 * (W + 1) + 1) * K_LDK * sizeof(float);
}

// Shape rule (host): which launches run here, and their tile.  Returns false when the launch belongs to the other kernels.
bool conv_skinny_geometry(int M, int N, int K, int HW, int W, int taps, int split, unsigned sw, int* m_tile, int* n_tile) {
    if (!split || (sw & SW_NO_SKINNY) || M <= 0 || K % K_CK != 0 || N % 64 != 0 || W < 1 || W > 8) return false;
    if (!(taps == 9 || (taps == 3 && W == 1)) || M % HW != 0) return false;
    // Measured per layer at batch 1-256 against the split-K + combine launches of the larger kernels (tools/bench_convs.py with
    // forced tiles, profiles/r02_skinny_rows.txt): this kernel wins wherever its grid -- 8-wave workgroups that each walk the WHOLE
    // K range of a (rows x 64 columns) tile -- stays within one workgroup per CU, and the MORE workgroups the better: 16-row
    // tiles beat 32- and 64-row ones at the same shape (level 0-1 at batch 1: 15 -> 8.5 us per layer, 64-row tiles 19; level 3
    // at 1024 rows: 9-10 us against 14-17 with 64-row tiles), although every m-tile streams the layer's weights again (they come
    // from L2: up to 300 MB per launch at 20 us).  Past 256 workgroups it loses (level 0 at batch 32+: 30 vs 25 us).
    // So: the smallest row tile whose grid fits 256 workgroups; 64-row tiles only for the 3-tap convolutions of the W == 1 level
    // (M <= 4096, <= ~110 MB of weight reads: K = N = 512 at M = 2048 is 201 MB and a tie; 9-tap 64-row tiles lose everywhere).
    const int gmax[3] = {spdm_tune(8, 256), spdm_tune(12, 256), spdm_tune(13, 256)};        // grid limit of the 16 / 32 / 64-row tiles
    int mt = 0;
    for (int i = 0; i < 3 && mt == 0; ++i) {
        const int cand = 16 << i;
        if ((long long)((M + cand - 1) / cand) * (N / 64) <= gmax[i] && skinny_slab_bytes(cand, W, K) <= (size_t)96 * 1024) mt = cand;
    }
    if (const int cap = spdm_tune(10, 0); mt != 0 && (cap == 16 || cap == 32 || cap == 64) && cap >= mt) mt = cap;   // (tuning experiments: a coarser tile)
    if (mt == 0) return false;
    const int mtiles = (M + mt - 1) / mt;
    if (mt == 64) {
        const double mb = (double)mtiles * N * K * taps * 4.0 / 1.0e6;
        if (taps != 3 || M > 4096 || mb > (double)spdm_tune(9, 110)) return false;
    }
    // 32-wide tiles unless 64-wide ones already fill the chip (256 workgroups): the weight stream of a launch moves through more
    // CUs.  Timed on the WHOLE step (graph replay, tools/probes/step_tune.sh) -- the per-layer micro-benchmark re-reads hot weights
    // and misleads here: it preferred 16-wide tiles (batch 1: 290 -> 269 us over the 31 layers), which make the step SLOWER
    // (0.653 -> 0.677 ms at batch 1: 2-KB weight bursts from cold HBM/MALL); so they stay off (SPDM_TUNE14).  Step time with the
    // 32-wide limit at 32 / 128 / 256 workgroups: batch 8 0.725 / 0.694 / 0.688, batch 32 0.810 / 0.789 / 0.771, batch 64
    // - / 0.878 / 0.858 ms; 32-wide at exactly 256 (512 workgroups) costs 4-6 %.
    const int g64 = mtiles * (N / 64);
    const int nt = (g64 < spdm_tune(14, 0)) ? 16 : (g64 < spdm_tune(11, 256)) ? 32 : 64;
    *m_tile = mt;
    *n_tile = nt;
    return true;
}

hipError_t launch_conv_skinny(const GemmArgs& a, const GemmGeom& g, hipStream_t s) {
    if (!a.split || a.wgt_frag == nullptr || a.epi != EPI_STATS || a.epi_stats == nullptr || a.row_stats != nullptr || a.ksplit > 1)
        return hipErrorInvalidValue;
    if (a.M % a.HW != 0 || a.K % K_CK != 0 || a.N % g.n_tile != 0 || a.dst_ld % 4 != 0 || a.src_ld % 4 != 0) return hipErrorInvalidValue;
    const bool fused = a.pro == PRO_POOL || a.pro == PRO_UPCAT;
    if (!fused && a.pro != PRO_NONE && (a.pro_stats.p == nullptr || a.pro_gamma == nullptr || a.pro_beta == nullptr)) return hipErrorInvalidValue;
    if (fused) {        // shape contract of the fused sources (kernels.h): checked here so that a bad plan cannot fault the GPU
        if (a.H < 1 || a.W < 1 || a.HW != a.H * a.W) return hipErrorInvalidValue;
        if (a.pro_stats.p != nullptr && (a.pro_gamma == nullptr || a.pro_beta == nullptr)) return hipErrorInvalidValue;
        if (a.pro == PRO_POOL && a.src_ld < a.K) return hipErrorInvalidValue;
        if (a.pro == PRO_UPCAT) {
            if ((a.H & 1) || (a.W & 1) || a.up_C <= 0 || a.up_C >= a.K || a.up_C % K_CK != 0 || a.src_ld < a.up_C || a.skip == nullptr ||
                a.skip_ld < a.K - a.up_C || a.skip_ld % 4 != 0) return hipErrorInvalidValue;
            if (a.skip_stats.p != nullptr && (a.skip_gamma == nullptr || a.skip_beta == nullptr)) return hipErrorInvalidValue;
        }
    }
    if (g.n_tile == 64) {
        if (g.m_tile == 64) return launch_skinny_rc<4, 4>(a, g, s);
        if (g.m_tile == 32) return launch_skinny_rc<2, 4>(a, g, s);
        if (g.m_tile == 16) return launch_skinny_rc<1, 4>(a, g, s);
    } else if (g.n_tile == 32) {
        if (g.m_tile == 64) return launch_skinny_rc<4, 2>(a, g, s);
        if (g.m_tile == 32) return launch_skinny_rc<2, 2>(a, g, s);
        if (g.m_tile == 16) return launch_skinny_rc<1, 2>(a, g, s);
    } else if (g.n_tile == 16) {
        if (g.m_tile == 32) return launch_skinny_rc<2, 1>(a, g, s);
        if (g.m_tile == 16) return launch_skinny_rc<1, 1>(a, g, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace spdm
