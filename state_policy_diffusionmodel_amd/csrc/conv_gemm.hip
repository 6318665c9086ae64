// conv_gemm.hip -- the dominant kernel: 3x3 convolution / Linear as an implicit GEMM on the
// gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 157 TFLOP/s chip peak).
//
// Replaces, on the reference's path: every nn.Conv2d(k=3, padding=1, bias=False) of
// DoubleConvolution (models/Unet_FiLmLayer.py:101-103) together with the GroupNorm(1,C) -> GELU
// that precedes it (:112-113, fused as the load PROLOGUE) and the GroupNorm statistics of what it
// produces (:112/:115, fused as the EPILOGUE); and, with one tap, the Linear layers of
// SelfAttention (:60-67), emb_layer (:136-142) and cond_encoder (:149-154).
//
//   out[m][n] = sum_tap sum_ci  f(in[m + shift(tap)][ci]) * w[tap][n][ci]          m = b*HW + h*W + w
//
// Tiling (one workgroup = 4 wave64, one 128 x N_T output tile, N_T = 64 or 128):
//   * K is walked in chunks of 32 input channels.  Per chunk the workgroup stages ONE halo'd slab
//     of the input -- rows [m0-(W+1), m0+128+(W+1)) x 32 channels -- into LDS, applying f (GroupNorm
//     affine from the producer's fp64 partial sums, optional erf-GELU) once; the 9 taps are 9
//     shifted views of that slab, so the input is read from L2/HBM once per chunk, not 9 times.
//     Out-of-image taps (zero padding, sample boundaries) are masked per lane at fragment read.
//   * per (chunk, tap) the 128 x 32 weight slab w[tap][n0..][k0..] is staged to LDS.
//   * both slabs are double-buffered through registers: global loads for the next slab are issued
//     before the MFMA block of the current one and written to LDS after it; one barrier per tap.
//   * LDS rows are padded 32 -> 36 floats: a half-wave's 32 lanes then read 32 distinct rows with
//     ds_read_b128 conflict-free (row stride 144 B = 9 x 16 B, 9 odd).
//   * each lane fetches 4 consecutive k of its row (one ds_read_b128) and feeds them to 4 MFMAs;
//     lane half h supplies k = 8g + 4h + s in step s, identically for A and B, so the reduction
//     order is a fixed permutation of k.
//   * epilogue: GroupNorm partial sums (fp32 per 4-row unit -> fp64 per sample, fixed order) or
//     bias / GELU / residual for Linear layers; rows on registers, channels on lanes -> each store
//     instruction writes two full 128-byte lines.
//   * blockIdx is remapped so that the n-tiles of one m-tile run on the same XCD (shared L2).
#include <algorithm>

#include "device_utils.h"

namespace spdm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// PREC_F32   : operands stay fp32, v_mfma_f32_32x32x2_f32 (bit-exact fp32 FMA chain).
// PREC_SPLIT : every fp32 operand x is split on the fly into two fp16 numbers
//                  hi = fp16(x),   lo = fp16((x - hi) * 2^11)            (x = hi + lo * 2^-11 to ~2^-22 |x|)
//              and a product a*b is evaluated as  ah*bh  +  2^-11 (ah*bl + al*bh)  with three
//              v_mfma_f32_32x32x16_f16 (fp16 x fp16 products are exact in fp32; accumulation is fp32).
//              The dropped al*bl term is 2^-22 relative, i.e. 4 fp32 ulps per product, random sign.
//              The fp16 matrix pipe runs 16x the fp32 one per clock, so 3 MFMAs per K=16 are 5.3x the
//              fp32-MFMA rate.  The scaling keeps lo in fp16's normal range for |x| down to ~1e-5 * 2^-11.
//              Weights are split once at load time (host); activations at LDS-staging time.

constexpr int CK = 32;    // channels per K chunk
constexpr int LDK = 36;   // padded LDS row length (floats)
enum { PREC_F32 = 0, PREC_SPLIT = 1 };

template <bool HALO, int PREC, int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const GemmArgs a, const int epi_slots) {
    constexpr int M_T = WM * MT * 32, N_T = WN * NT * 32;
    constexpr int APASS = HALO ? (M_T + 18 + 31) / 32 : M_T / 32;   // halo <= 9 rows each side (W <= 8)
    constexpr int WPASS = N_T / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, kh = lane >> 5;
    const int HW = a.HW, W = a.W, H = a.H, M = a.M, K = a.K, N = a.N, taps = a.taps;
    const int halo = HALO ? (W + 1) : 0;
    const int QA = M_T + 2 * halo;
    const int NSP = (QA + 4) & ~3;

    // ---- tile of this workgroup (XCD-aware, bijective remap) ----
    const int n_ntiles = N / N_T;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int mtile = logical / n_ntiles, ntile = logical - mtile * n_ntiles;
    const int m0 = mtile * M_T, n0 = ntile * N_T;

    const int QZ = QA + 1;                    // + one all-zero row: what a masked (out-of-image) tap reads
    float* Abuf = smem;                       // [2][QZ][LDK]
    float* Wbuf = Abuf + 2 * QZ * LDK;        // [2][N_T][LDK]
    float* smean = Wbuf + 2 * N_T * LDK;      // [NSP]
    float* srstd = smean + NSP;               // [NSP]

    if (tid < 2 * LDK) Abuf[(tid / LDK) * QZ * LDK + QA * LDK + tid % LDK] = 0.f;

    // ---- prologue statistics of the samples this slab touches ----
    const bool pro = (a.pro != PRO_NONE);
    const bool pro_gelu = (a.pro == PRO_GN_GELU);
    int bh_first = 0;
    if (pro) {
        const int lo = max(m0 - halo, 0), hi = min(m0 + M_T + halo, M) - 1;
        bh_first = lo / HW;
        const int bh_last = hi / HW;
        for (int t = tid; t <= bh_last - bh_first; t += 256) {
            float mean, rstd;
            sample_mean_rstd(a.pro_stats, bh_first + t, mean, rstd);
            smean[t] = mean;
            srstd[t] = rstd;
        }
        __syncthreads();
    }

    // ---- per-thread staging assignment: 8 threads x float4 cover one 32-channel row ----
    const int srow_t = tid >> 3, c4 = tid & 7;
    bool aval[APASS];
    const float* aptr[APASS];
    float amean[APASS], arstd[APASS];
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
        const int q = p * 32 + srow_t;
        const int m = m0 - halo + q;
        const bool v = (q < QA) && (m >= 0) && (m < M);
        aval[p] = v;
        aptr[p] = a.src + (size_t)min(max(m, 0), M - 1) * a.src_ld + c4 * 4;   // clamped: always loadable
        amean[p] = 0.f;
        arstd[p] = 1.f;
        if (pro && v) {
            const int b = m / HW;
            amean[p] = smean[b - bh_first];
            arstd[p] = srstd[b - bh_first];
        }
    }
    const float* wptr = a.wgt + (size_t)(n0 + srow_t) * K + c4 * 4;   // + (tap*N + p*32)*K + chunk*32

    // ---- per-lane fragment rows and tap masks ----
    // A lane reads row (r + halo + shift(tap)) of the slab, or the all-zero row when the tap falls
    // outside the image / the sample for its output position.
    int aoff[MT], boff[NT];
    unsigned amask[MT];
    const int koff = kh * 4;    // floats: 16 B per lane half in both slab formats
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = wm * MT * 32 + mt * 32 + li;
        aoff[mt] = (r + halo) * LDK + koff;
        unsigned mask = HALO ? 0u : 1u;
        if (HALO) {
            const int m = m0 + r;
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
        }
        amask[mt] = mask;
    }
    const int zoff = QA * LDK + koff;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) boff[nt] = (wn * NT * 32 + nt * 32 + li) * LDK + koff;

    constexpr int NACC = (PREC == PREC_SPLIT) ? 2 : 1;     // [1]: the 2^-11-scaled cross terms
    f32x16 acc[NACC][MT][NT];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][mt][nt][r] = 0.f;

    // Staging registers.  Everything below is written with compile-time indices and UNCONDITIONAL
    // loads (rows outside the tensor are clamped to a valid address and zeroed at LDS-write time):
    // a predicated load makes hipcc wait for it right where it is issued, and HIP's float4 struct
    // arrays end up in scratch -- both defeat the overlap of the loads with the MFMA block.
    f32x4 areg[APASS], wreg[WPASS];
    f32x4 g4r = {1.f, 1.f, 1.f, 1.f}, b4r = {0.f, 0.f, 0.f, 0.f};   // GroupNorm gain / offset of this thread's 4 channels

#define SPDM_LOAD_A(chunk_)                                                                         \
    {                                                                                               \
        _Pragma("unroll") for (int p_ = 0; p_ < APASS; ++p_)                                       \
            areg[p_] = *reinterpret_cast<const f32x4*>(aptr[p_] + (chunk_) * CK);                   \
        if (pro) {                                                                                  \
            g4r = *reinterpret_cast<const f32x4*>(a.pro_gamma + (chunk_) * CK + c4 * 4);            \
            b4r = *reinterpret_cast<const f32x4*>(a.pro_beta + (chunk_) * CK + c4 * 4);             \
        }                                                                                           \
    }
#define SPDM_LOAD_W(chunk_, tap_)                                                                   \
    {                                                                                               \
        const float* wb_ = wptr + (size_t)(tap_) * N * K + (chunk_) * CK;                           \
        _Pragma("unroll") for (int p_ = 0; p_ < WPASS; ++p_)                                       \
            wreg[p_] = *reinterpret_cast<const f32x4*>(wb_ + (size_t)p_ * 32 * K);                  \
    }
#define SPDM_STORE_W(buf_)                                                                          \
    {                                                                                               \
        float* wd_ = Wbuf + (buf_) * N_T * LDK + srow_t * LDK + c4 * 4;                             \
        _Pragma("unroll") for (int p_ = 0; p_ < WPASS; ++p_)                                       \
            *reinterpret_cast<f32x4*>(wd_ + p_ * 32 * LDK) = wreg[p_];                              \
    }
    // fp32 slab row: 32 floats.  split slab row: [32 x fp16 hi | 32 x fp16 lo] (same 128 bytes).
#define SPDM_STORE_A(chunk_, buf_)                                                                  \
    {                                                                                               \
        const f32x4 g4_ = g4r, b4_ = b4r;                                                           \
        float* ad_ = Abuf + (buf_) * QZ * LDK;                                                      \
        _Pragma("unroll") for (int p_ = 0; p_ < APASS; ++p_) {                                     \
            f32x4 v_ = areg[p_];                                                                    \
            if (pro) {                                                                              \
                const float rs_ = arstd[p_], mu_ = amean[p_];                                       \
                v_.x = (v_.x - mu_) * (rs_ * g4_.x) + b4_.x;                                        \
                v_.y = (v_.y - mu_) * (rs_ * g4_.y) + b4_.y;                                        \
                v_.z = (v_.z - mu_) * (rs_ * g4_.z) + b4_.z;                                        \
                v_.w = (v_.w - mu_) * (rs_ * g4_.w) + b4_.w;                                        \
                if (pro_gelu) {                                                                     \
                    v_.x = gelu_erf(v_.x); v_.y = gelu_erf(v_.y);                                   \
                    v_.z = gelu_erf(v_.z); v_.w = gelu_erf(v_.w);                                   \
                }                                                                                   \
            }                                                                                       \
            if (!aval[p_]) v_ = f32x4{0.f, 0.f, 0.f, 0.f};                                          \
            if (p_ * 32 + srow_t < QA) {                                                            \
                float* row_ = ad_ + (p_ * 32 + srow_t) * LDK;                                       \
                if (PREC == PREC_SPLIT) {                                                           \
                    f16x4 hi_, lo_;                                                                 \
                    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                             \
                        const _Float16 h_ = (_Float16)v_[e_];                                       \
                        hi_[e_] = h_;                                                               \
                        lo_[e_] = (_Float16)((v_[e_] - (float)h_) * 2048.0f);                       \
                    }                                                                               \
                    *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(row_) + c4 * 4) = hi_;     \
                    *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(row_) + 32 + c4 * 4) = lo_; \
                } else {                                                                            \
                    *reinterpret_cast<f32x4*>(row_ + c4 * 4) = v_;                                  \
                }                                                                                   \
            }                                                                                       \
        }                                                                                           \
    }

    const int nchunks = K / CK;
    const int niter = nchunks * taps;

    SPDM_LOAD_A(0)
    SPDM_LOAD_W(0, 0)
    SPDM_STORE_A(0, 0)
    SPDM_STORE_W(0)
    __syncthreads();

    int chunk = 0, tap = 0;
    for (int it = 0; it < niter; ++it) {
        // next (chunk, tap)
        int ntap = tap + 1, nchunk = chunk;
        if (ntap == taps) { ntap = 0; nchunk = chunk + 1; }
        const bool have_next = (it + 1 < niter);
        const bool next_A = have_next && (ntap == 0);
        // A before W: hipcc guards the re-use of the A staging registers with a vmcnt wait that would
        // otherwise also wait for the W loads issued just before it
        if (next_A) SPDM_LOAD_A(nchunk)
        if (have_next) SPDM_LOAD_W(nchunk, ntap)

        // ---- MFMA block on Abuf[chunk&1], Wbuf[it&1] ----
        const float* Ab = Abuf + (chunk & 1) * QZ * LDK;
        const float* Wb = Wbuf + (it & 1) * N_T * LDK;
        int shift = 0;
        if (HALO) {
            const int dh = (taps == 9) ? tap / 3 - 1 : tap - 1;
            const int dw = (taps == 9) ? tap % 3 - 1 : 0;
            shift = (dh * W + dw) * LDK;
        }
        const float* ap[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ap[mt] = Ab + (((amask[mt] >> tap) & 1u) ? aoff[mt] + shift : zoff);

        if (PREC == PREC_SPLIT) {
            // row = [hi: 32 fp16 | lo: 32 fp16]; K=16 step s, lane half kh: 8 fp16 at byte 32 s + 16 kh
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    ah[mt] = *reinterpret_cast<const f16x8*>(ap[mt] + s2 * 8);
                    al[mt] = *reinterpret_cast<const f16x8*>(ap[mt] + 16 + s2 * 8);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bh[nt] = *reinterpret_cast<const f16x8*>(Wb + boff[nt] + s2 * 8);
                    bl[nt] = *reinterpret_cast<const f16x8*>(Wb + boff[nt] + 16 + s2 * 8);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bh[nt], acc[0][mt][nt], 0, 0, 0);
                        acc[NACC - 1][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], bl[nt], acc[NACC - 1][mt][nt], 0, 0, 0);
                        acc[NACC - 1][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], bh[nt], acc[NACC - 1][mt][nt], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 av[MT], bv[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(ap[mt] + g * 8);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const f32x4*>(Wb + boff[nt] + g * 8);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].x, bv[nt].x, acc[0][mt][nt], 0, 0, 0);
                        acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].y, bv[nt].y, acc[0][mt][nt], 0, 0, 0);
                        acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].z, bv[nt].z, acc[0][mt][nt], 0, 0, 0);
                        acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].w, bv[nt].w, acc[0][mt][nt], 0, 0, 0);
                    }
            }
        }

        if (have_next) SPDM_STORE_W((it + 1) & 1)
        if (next_A) SPDM_STORE_A(nchunk, nchunk & 1)
        __syncthreads();
        tap = ntap;
        chunk = nchunk;
    }
#undef SPDM_LOAD_A
#undef SPDM_LOAD_W
#undef SPDM_STORE_A
#undef SPDM_STORE_W

    if (PREC == PREC_SPLIT) {   // fold the scaled cross terms in: x = hi*hi + 2^-11 (hi*lo + lo*hi)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[0][mt][nt][r] = acc[0][mt][nt][r] + acc[NACC - 1][mt][nt][r] * (1.0f / 2048.0f);
    }

    // ---- epilogue ----
    // The accumulator layout has channels on lanes and rows on registers, i.e. one dword per lane per
    // store.  Going through LDS (the A/W slabs are dead now) turns the tile into whole rows so that
    // every lane stores 16 bytes and a wave instruction writes two 512-byte row pieces: 4x fewer
    // store instructions (the store tail is issue-bound, not bandwidth-bound).
    float* srow = smem;                                  // [M_T / unit][WN][2] GroupNorm partials
    float* otile = smem + 1024;                          // [M_T][N_T] fp32 output tile
    const bool unit4 = (HW & 3) == 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col_l = wn * NT * 32 + nt * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row_l = wm * MT * 32 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                otile[row_l * N_T + col_l] = acc[0][mt][nt][r];
            }
        }

    if (a.epi == EPI_STATS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r0 = wm * MT * 32 + mt * 32 + 8 * g + 4 * kh;     // tile-local row of register 4g
                if (unit4) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = acc[0][mt][nt][4 * g + j];
                            s1 += v;
                            s2 += v * v;
                        }
                    s1 = half_sum(s1);
                    s2 = half_sum(s2);
                    if (li == 0) {
                        srow[((r0 >> 2) * WN + wn) * 2] = s1;
                        srow[((r0 >> 2) * WN + wn) * 2 + 1] = s2;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const float v = acc[0][mt][nt][4 * g + j];
                            s1 += v;
                            s2 += v * v;
                        }
                        s1 = half_sum(s1);
                        s2 = half_sum(s2);
                        if (li == 0) {
                            srow[((r0 + j) * WN + wn) * 2] = s1;
                            srow[((r0 + j) * WN + wn) * 2 + 1] = s2;
                        }
                    }
                }
            }
    }
    __syncthreads();

    if (a.epi == EPI_STATS) {
        const int t_lo = m0, t_hi = min(m0 + M_T, M);
        if (t_hi > t_lo) {
            const int b_first = t_lo / HW, b_last = (t_hi - 1) / HW;
            const int ush = unit4 ? 2 : 0;
            for (int t = tid; t <= b_last - b_first; t += 256) {
                const int b = b_first + t;
                const int r_lo = max(b * HW, t_lo) - m0, r_hi = min((b + 1) * HW, t_hi) - m0;
                double s1 = 0.0, s2 = 0.0;
                for (int u = r_lo >> ush; u < ((r_hi + (unit4 ? 3 : 0)) >> ush); ++u)
                    for (int w2 = 0; w2 < WN; ++w2) {
                        s1 += (double)srow[(u * WN + w2) * 2];
                        s2 += (double)srow[(u * WN + w2) * 2 + 1];
                    }
                const int slot = (mtile - (b * HW) / M_T) * n_ntiles + ntile;
                double* o = a.epi_stats + ((size_t)b * epi_slots + slot) * 2;
                o[0] = s1;
                o[1] = s2;
            }
        }
    }

    {
        constexpr int TPR = N_T / 4, RPP = 256 / TPR;    // threads per row, rows per pass
        const int c4o = tid % TPR, r0 = tid / TPR;
        const bool has_bias = (a.epi == EPI_BIAS || a.epi == EPI_BIAS_GELU || a.epi == EPI_BIAS_RESID);
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (has_bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + n0 + c4o * 4);
#pragma unroll 4
        for (int p = 0; p < M_T / RPP; ++p) {
            const int row_l = p * RPP + r0;
            const int row = m0 + row_l;
            if (row < M) {
                f32x4 v = *reinterpret_cast<const f32x4*>(otile + row_l * N_T + c4o * 4);
                v += bias4;
                if (a.epi == EPI_BIAS_GELU) {
                    v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
                }
                if (a.epi == EPI_BIAS_RESID)
                    v += *reinterpret_cast<const f32x4*>(a.resid + (size_t)row * a.resid_ld + n0 + c4o * 4);
                *reinterpret_cast<f32x4*>(a.dst + (size_t)row * a.dst_ld + n0 + c4o * 4) = v;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
GemmGeom gemm_geometry(int M, int N, int HW) {
    GemmGeom g;
    g.m_tile = 128;
    g.n_tile = (N % 128 == 0) ? 128 : 64;
    g.n_tiles = N / g.n_tile;
    g.slots = stats_slots(HW, g.m_tile, g.n_tiles);
    (void)M;
    return g;
}

double gemm_flops(const GemmArgs& a) { return 2.0 * (double)a.M * (double)a.N * (double)a.K * (double)a.taps; }

template <bool HALO, int PREC, int WM, int WN, int MT, int NT>
static hipError_t launch_cfg(const GemmArgs& a, const GemmGeom& g, hipStream_t s) {
    constexpr int M_T = WM * MT * 32, N_T = WN * NT * 32;
    const int halo = HALO ? a.W + 1 : 0;
    const int QA = M_T + 2 * halo;
    const int NSP = (QA + 4) & ~3;
    size_t lds = (size_t)(2 * (QA + 1) * LDK + 2 * N_T * LDK + 2 * NSP) * sizeof(float);
    lds = std::max(lds, (size_t)(1024 + M_T * N_T) * sizeof(float));     // epilogue staging: srow + output tile
    auto kern = conv_gemm_kernel<HALO, PREC, WM, WN, MT, NT>;
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        lds_set = 160 * 1024;
    }
    const int n_mtiles = (a.M + M_T - 1) / M_T;
    const int grid = n_mtiles * g.n_tiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, g.slots);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs& a, hipStream_t s) {
    // shape contract of the kernel -- checked on the host so that a bad plan can never fault the GPU
    if (a.M <= 0 || a.K <= 0 || a.N <= 0) return hipErrorInvalidValue;
    if (a.K % CK != 0 || a.N % 64 != 0) return hipErrorInvalidValue;
    if (a.src_ld % 4 != 0 || a.src_ld < a.K || a.dst_ld % 4 != 0) return hipErrorInvalidValue;
    if (a.epi == EPI_BIAS_RESID && a.resid_ld % 4 != 0) return hipErrorInvalidValue;
    if (!(a.taps == 1 || a.taps == 3 || a.taps == 9)) return hipErrorInvalidValue;
    if (a.taps != 1 && (a.W < 1 || a.W > 8 || a.H < 1 || a.HW != a.H * a.W || a.M % a.HW != 0)) return hipErrorInvalidValue;
    if (a.taps == 3 && a.W != 1) return hipErrorInvalidValue;
    if (a.HW < 1) return hipErrorInvalidValue;
    if (a.pro != PRO_NONE && (a.pro_stats.p == nullptr || a.pro_gamma == nullptr || a.pro_beta == nullptr ||
                              a.pro_stats.HW != a.HW)) return hipErrorInvalidValue;
    if (a.epi == EPI_STATS && a.epi_stats == nullptr) return hipErrorInvalidValue;
    if ((a.epi == EPI_BIAS || a.epi == EPI_BIAS_GELU || a.epi == EPI_BIAS_RESID) && a.bias == nullptr) return hipErrorInvalidValue;
    if (a.epi == EPI_BIAS_RESID && a.resid == nullptr) return hipErrorInvalidValue;
    const GemmGeom g = gemm_geometry(a.M, a.N, a.HW);
    if (a.split) {
        if (a.taps == 1) {
            if (g.n_tile == 128) return launch_cfg<false, PREC_SPLIT, 2, 2, 2, 2>(a, g, s);
            return launch_cfg<false, PREC_SPLIT, 2, 2, 2, 1>(a, g, s);
        }
        if (g.n_tile == 128) return launch_cfg<true, PREC_SPLIT, 2, 2, 2, 2>(a, g, s);
        return launch_cfg<true, PREC_SPLIT, 2, 2, 2, 1>(a, g, s);
    }
    if (a.taps == 1) {
        if (g.n_tile == 128) return launch_cfg<false, PREC_F32, 2, 2, 2, 2>(a, g, s);
        return launch_cfg<false, PREC_F32, 2, 2, 2, 1>(a, g, s);
    }
    if (g.n_tile == 128) return launch_cfg<true, PREC_F32, 2, 2, 2, 2>(a, g, s);
    return launch_cfg<true, PREC_F32, 2, 2, 2, 1>(a, g, s);
}

}  // namespace spdm
