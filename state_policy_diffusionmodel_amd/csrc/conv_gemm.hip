// conv_gemm.hip -- the dominant kernel: 3x3 convolution / Linear as an implicit GEMM on the gfx950
// matrix cores.
//
// Replaces, on the reference's path: every nn.Conv2d(k=3, padding=1, bias=False) of
// DoubleConvolution (models/Unet_FiLmLayer.py:101-103) together with the GroupNorm(1,C) -> GELU
// that precedes it (:112-113, fused as the load PROLOGUE) and the GroupNorm statistics of what it
// produces (:112/:115, fused as the EPILOGUE); and, with one tap, the Linear layers of
// SelfAttention (:60-67), emb_layer (:136-142) and cond_encoder (:149-154).
//
//   out[m][n] = sum_tap sum_ci  f(in[m + shift(tap)][ci]) * w[tap][n][ci]          m = b*HW + h*W + w
//
// Structure (one workgroup = WM x WN wave64, one (WM*MT*32) x (WN*NT*32) output tile):
//   * K is walked in chunks of 32 input channels.  Per chunk the workgroup stages ONE halo'd slab
//     of the input -- rows [m0-(W+1), m0+M_T+(W+1)) x 32 channels -- into LDS, applying f (GroupNorm
//     affine from the producer's fp64 partial sums, optional erf-GELU) once; the 9 taps are 9
//     shifted views of that slab, so the input is read from L2/HBM once per chunk, not 9 times.
//     Out-of-image taps (zero padding, sample boundaries) read a dedicated all-zero LDS row.
//   * weight slabs w[tap][n0..][k0..] are staged TPI taps at a time and double-buffered through
//     registers: global loads for the next group are issued before the MFMA block of the current
//     one and written to LDS after it; ONE workgroup barrier per group of TPI taps.  The main
//     configuration (8 waves, 256 x 128 tile, TPI = 3) runs 72 MFMAs per wave between barriers --
//     the measured cost of a barrier + staging round is ~600-800 cycles, so per-barrier MFMA work
//     is what sets matrix-pipe utilisation (PMC: 35 % at 24 MFMAs per barrier).
//   * LDS rows are padded 128 -> 144 bytes: a half-wave's 32 lanes then read 32 distinct rows with
//     ds_read_b128 conflict-free (row stride 9 x 16 B, 9 odd).
//   * fragment reads are software-pipelined one K=16 step ahead of the MFMAs that consume them.
//   * epilogue: the tile goes through LDS (the slabs are dead by then) so that every lane stores
//     16 bytes and a wave writes whole 512-byte row pieces; GroupNorm partial sums (fp32 per row
//     unit -> fp64 per sample, fixed order) or bias / GELU / residual for Linear layers.
//   * blockIdx is remapped so that the n-tiles of one m-tile run on the same XCD (shared L2).
#include <algorithm>
#include <cstdlib>

#include "device_utils.h"

namespace spdm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// PREC_F32   : operands stay fp32, v_mfma_f32_32x32x2_f32 (bit-exact fp32 FMA chain).
// PREC_SPLIT : every fp32 operand x is split into two fp16 numbers  hi = fp16(x'), lo = fp16(x' - hi)
//              of the pre-scaled value x' = x * 2^s (activations s = 4, weights s = 7; powers of two, so
//              the scaling is exact and only moves the numbers into fp16's normal range: lo stays a
//              normal fp16 for |x| >= 2^-7 resp. 2^-10, hi overflows only for |x| > 4094 resp. 511).
//              A product a*b is evaluated as  ah*bh + ah*bl + al*bh  with three v_mfma_f32_32x32x16_f16
//              into ONE fp32 accumulator (fp16 x fp16 products are exact in fp32), and the accumulator is
//              multiplied by 2^-11 once in the epilogue.  The dropped al*bl term is 2^-22 relative (4 fp32
//              ulps per product, random sign); measured end-to-end error is BELOW the fp32-MFMA path's,
//              because the fp16 pipe rounds once per 16 products instead of once per product.
//              The fp16 matrix pipe runs 16x the fp32 one per clock, so 3 MFMAs per K=16 are 5.3x the
//              fp32-MFMA rate.  Weights are split once at load time (host); activations at LDS-staging.
constexpr float SPLIT_ACT_SCALE = 16.0f;      // 2^4
// (weights: 2^7, applied on the host -- spdm_api.hip upload_split)
constexpr float SPLIT_DESCALE = 1.0f / 2048.0f;

// {packed fp16 hi(a,b), packed fp16 lo(a,b)} of the pre-scaled pair, as raw bits in two floats.
// (Written per PAIR on purpose: bit-casting a 4 x fp16 vector to 2 x u32 is miscompiled by hipcc 7.2 --
// elements 2,3 are replaced by 0,1.)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 split_pair(float a, float b) {
    unsigned h, l;
    split_pair_f16(a * SPLIT_ACT_SCALE, b * SPLIT_ACT_SCALE, h, l);
    return f32x2{__builtin_bit_cast(float, h), __builtin_bit_cast(float, l)};
}

constexpr int CK = 32;    // channels per K chunk
constexpr int LDK = 36;   // padded LDS row length (floats)
enum { PREC_F32 = 0, PREC_SPLIT = 1 };

// DEEP (small grids: 4-wave split configurations with TPI = 3): the global loads of an iteration are issued TWO iterations
// ahead into a second set of staging registers, and the A slab is double-buffered in LDS, so that a workgroup alone on its
// CU (which is what a small grid is) is no longer paced by one weight round trip per iteration.
template <bool HALO, int PREC, int WM, int WN, int MT, int NT, int TPI, bool W2, bool DEEP>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_gemm_kernel(const GemmArgs a, const int epi_slots) {
    constexpr int NTHR = WM * WN * 64;
    constexpr int RP = NTHR / 8;                                     // slab rows staged per pass
    constexpr int M_T = WM * MT * 32, N_T = WN * NT * 32;
    constexpr int APASS = HALO ? (M_T + 18 + RP - 1) / RP : M_T / RP;   // halo <= 9 rows each side (W <= 8)
    constexpr int WPASS = N_T / RP;                                  // per tap
    constexpr int NBA = (TPI == 1 || DEEP) ? 2 : 1;                  // A slab buffers
    static_assert(!DEEP || (NTHR == 256 && TPI == 3 && PREC == PREC_SPLIT && HALO && !W2), "DEEP: the 4-wave small-grid configurations");
    constexpr bool PP = (NTHR == 512);                               // ping-pong schedule (see main loop)
    // WDMA (experiment, OFF): weight slabs global -> LDS by LDS-DMA (global_load_lds_dwordx4; un-padded rows,
    // XOR swizzle of the 16-byte chunk with (row>>1)&7 on the per-lane source address and on the fragment
    // reads), issued by waves 0-3 only.  Measured on MI355X: numerically identical, 18 fewer VGPRs, SAME kernel
    // time -- in-kernel stamps show each LDS-DMA wave instruction costing 250-430 cycles of issue while the
    // MFMA/LDS pipes are busy, i.e. the CU's vector-memory path (48 KB of weights per 4608 MFMA cycles), not the
    // instruction mix of the staging, is the co-limiter.  Left in for tile-shape experiments (-DSPDM_WDMA).
#ifdef SPDM_WDMA
    constexpr bool WDMA = (NTHR == 512) && (PREC == PREC_SPLIT);
#else
    constexpr bool WDMA = false;
#endif
    constexpr int WROW = WDMA ? 32 : LDK;                            // floats per W row in LDS
    static_assert(!PP || NBA == 1, "ping-pong uses the single-slab hand-over");
    // W2: image width 2 (level-2 maps).  A third of the (position, tap) pairs of a 3x3 kernel then hit zero
    // padding: for w = 0 the dw = -1 column, for w = 1 the dw = +1 column.  Each wave's 64 rows are PERMUTED so
    // that MFMA tile 0 holds the w = 0 positions and tile 1 the w = 1 positions (row 2i + mt instead of
    // 32 mt + i); the two side columns then fuse into ONE virtual tap per kernel row -- tile 0 multiplies
    // w[dh][+1], tile 1 w[dh][-1] -- and a kernel row costs 2 taps of MFMAs instead of 3.  Same products,
    // same sums: only all-zero terms are dropped.
    static_assert(!W2 || (HALO && TPI == 3 && MT == 2 && PREC == PREC_SPLIT), "W2 needs the 3-tap split configuration");
#define SPDM_ROW(mt_, i_) (wm * MT * 32 + (W2 ? 2 * (i_) + (mt_) : (mt_) * 32 + (i_)))
    static_assert(N_T % RP == 0 && M_T % RP == 0, "tile / thread-count mismatch");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, kh = lane >> 5;
    const int HW = a.HW, W = a.W, H = a.H, M = a.M, K = a.K, N = a.N, taps = a.taps;
    const int halo = HALO ? (W + 1) : 0;
    const int QA = M_T + 2 * halo;
    const int QZ = QA + 1;                    // + one all-zero row: what a masked (out-of-image) tap reads
    const int NSP = (QA + 4) & ~3;

    // ---- tile of this workgroup (XCD-aware, bijective remap) ----
    const int n_ntiles = N / N_T;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    // split-K: workgroup (tile, ks) walks the chunks [kc0, kc1) only and leaves a raw partial tile (kernels.h)
    const int ksp = a.ksplit > 1 ? a.ksplit : 1;
    const int tiles_mn = nblk / ksp;
    const int ks = logical / tiles_mn, tl = logical - ks * tiles_mn;
    const int mtile = tl / n_ntiles, ntile = tl - mtile * n_ntiles;
    const int m0 = mtile * M_T, n0 = ntile * N_T;
    const int kc0 = (int)((long long)ks * (K / CK) / ksp), kc1 = (int)((long long)(ks + 1) * (K / CK) / ksp);

    float* Abuf = smem;                                   // [NBA][QZ][LDK]
    float* Wbuf = Abuf + NBA * QZ * LDK;                  // [2][TPI][N_T][WROW]
    float* smean = Wbuf + 2 * TPI * N_T * WROW;           // [NSP]
    float* srstd = smean + NSP;                           // [NSP]

    // diagnostic stamps (DBG_STAMP): thread 0 of each half of the workgroup that owns logical tile 37
#ifdef SPDM_DIAG
    const bool stamping = (a.debug & DBG_STAMP) && a.stamps != nullptr && logical == 37 && (tid & 255) == 0;
    int nstamp = 0;
#endif
#ifdef SPDM_DIAG
#define SPDM_STAMP()                                                                                 \
    if (stamping && nstamp < 126) a.stamps[(tid >> 8) * 128 + nstamp++] = (unsigned long long)clock64();
#else
#define SPDM_STAMP()
#endif
    SPDM_STAMP()

    if (tid < NBA * LDK) Abuf[(tid / LDK) * QZ * LDK + QA * LDK + tid % LDK] = 0.f;

    // ---- prologue statistics of the samples this slab touches ----
    const bool pro = (a.pro != PRO_NONE);
    const int dbg = a.debug;
    const bool pro_gelu = (a.pro == PRO_GN_GELU) && !(dbg & DBG_NO_GELU);
    int bh_first = 0;
    if (pro) {
        const int lo = max(m0 - halo, 0), hi = min(m0 + M_T + halo, M) - 1;
        bh_first = lo / HW;
        const int bh_last = hi / HW;
        for (int t = tid; t <= bh_last - bh_first; t += NTHR) {
            float mean, rstd;
            sample_mean_rstd(a.pro_stats, bh_first + t, mean, rstd);
            smean[t] = mean;
            srstd[t] = rstd;
        }
        __syncthreads();
    }

    // ---- per-thread staging assignment: 8 threads x 16 bytes cover one 32-channel row ----
    const int srow_t = tid >> 3, c4 = tid & 7;
    bool aval[APASS];
    const float* aptr[APASS];
    float amean[APASS], arstd[APASS];
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
        const int q = p * RP + srow_t;
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
    const float* wptr = a.wgt + (size_t)(n0 + srow_t) * K + c4 * 4;   // + (tap*N + p*RP)*K + chunk*32

    // ---- per-lane fragment rows and tap masks ----
    int aoff[MT], boff[NT];
    unsigned amask[MT];
    const int koff = kh * 4;    // floats: 16 B per lane half in both slab formats
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = SPDM_ROW(mt, li);
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
    for (int nt = 0; nt < NT; ++nt) boff[nt] = (wn * NT * 32 + nt * 32 + li) * WROW + (WDMA ? 0 : koff);
    const int swz = (li >> 1) & 7;            // WDMA: chunk swizzle of this lane's W rows ((row>>1)&7; rows are 32-aligned + li)
    // float offset of the 16-byte fragment (k-step s2, hi/lo) inside a W row
#define SPDM_WOFF(s2_, lo_) (WDMA ? (((((lo_) ? 4 : 0) + 2 * (s2_) + kh) ^ swz) << 2) : ((s2_) * 8 + ((lo_) ? 16 : 0)))

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // Staging registers.  Everything below is written with compile-time indices and UNCONDITIONAL
    // loads (rows outside the tensor are clamped to a valid address and zeroed at LDS-write time):
    // a predicated load makes hipcc wait for it right where it is issued, and HIP's float4 struct
    // arrays end up in scratch -- both defeat the overlap of the loads with the MFMA block.
    f32x4 areg[APASS], wreg[WDMA ? 1 : TPI * WPASS];
    f32x4 g4r = {1.f, 1.f, 1.f, 1.f}, b4r = {0.f, 0.f, 0.f, 0.f};   // GroupNorm gain / offset of this thread's 4 channels
    f32x4 areg2[DEEP ? APASS : 1], wreg2[DEEP ? TPI * WPASS : 1];    // DEEP: the second staging set
    f32x4 g4r2 = {1.f, 1.f, 1.f, 1.f}, b4r2 = {0.f, 0.f, 0.f, 0.f};

#define SPDM_LOAD_A_X(AR_, G_, B_, chunk_)                                                          \
    {                                                                                               \
        _Pragma("unroll") for (int p_ = 0; p_ < APASS; ++p_)                                       \
            AR_[p_] = *reinterpret_cast<const f32x4*>(aptr[p_] + (kc0 + (chunk_)) * CK);            \
        if (pro) {                                                                                  \
            G_ = *reinterpret_cast<const f32x4*>(a.pro_gamma + (kc0 + (chunk_)) * CK + c4 * 4);     \
            B_ = *reinterpret_cast<const f32x4*>(a.pro_beta + (kc0 + (chunk_)) * CK + c4 * 4);      \
        }                                                                                           \
    }
#define SPDM_LOAD_A(chunk_) SPDM_LOAD_A_X(areg, g4r, b4r, chunk_)
    const int gw = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave index, provably uniform
    // per-lane part of the DMA source address: row (lane>>3) of the 8-row block, swizzled 16-byte chunk.  The
    // swizzle key (row>>1)&7 of row = 8 blk + (lane>>3) is (4 (blk&1) + (lane>>4)) & 7, and blk&1 == gw&1 for
    // every block a wave handles -> one offset register per lane.
    const int dma_loff = (lane >> 3) * K + (((lane & 7) ^ ((4 * (gw & 1) + (lane >> 4)) & 7)) << 2);
    // LDS-DMA of one W group slab (TPI taps x N_T rows x 128 B), by waves 0-3: each wave instruction moves
    // 8 rows = 1 KiB (lane -> row lane>>3, physical chunk lane&7, source chunk = physical ^ ((row>>1)&7))
#define SPDM_DMA_W(chunk_, tg_, buf_)                                                               \
    if (gw < 4) {                                                                                   \
        _Pragma("unroll") for (int blk_ = 0; blk_ < TPI * N_T / 32; ++blk_) {                      \
            const int row0_ = (blk_ * 4 + gw) * 8;                 /* wave-uniform */               \
            const int tp_ = row0_ / N_T, n_ = row0_ % N_T;                                          \
            const float* sb_ = a.wgt + (size_t)(((tg_) * TPI + tp_) * N + n0 + n_) * K + (kc0 + (chunk_)) * CK; \
            float* dst_ = Wbuf + (buf_) * TPI * N_T * WROW + row0_ * 32;                            \
            __builtin_amdgcn_global_load_lds(                                                       \
                (const __attribute__((address_space(1))) void*)(sb_ + dma_loff),                    \
                (__attribute__((address_space(3))) void*)(dst_), 16, 0, 0);                         \
        }                                                                                           \
    }
#define SPDM_LOAD_W_X(WR_, chunk_, tg_)                                                             \
    if (!WDMA) {                                                                                    \
        _Pragma("unroll") for (int tp_ = 0; tp_ < TPI; ++tp_) {                                    \
            const float* wb_ = wptr + (size_t)((tg_) * TPI + tp_) * N * K + (kc0 + (chunk_)) * CK;  \
            _Pragma("unroll") for (int p_ = 0; p_ < WPASS; ++p_)                                   \
                WR_[tp_ * WPASS + p_] = *reinterpret_cast<const f32x4*>(wb_ + (size_t)p_ * RP * K); \
        }                                                                                           \
    }
#define SPDM_LOAD_W(chunk_, tg_) SPDM_LOAD_W_X(wreg, chunk_, tg_)
#define SPDM_STORE_W_X(WR_, buf_)                                                                   \
    if (!WDMA) {                                                                                    \
        float* wd_ = Wbuf + (buf_) * TPI * N_T * LDK + srow_t * LDK + c4 * 4;                       \
        _Pragma("unroll") for (int tp_ = 0; tp_ < TPI; ++tp_)                                      \
            _Pragma("unroll") for (int p_ = 0; p_ < WPASS; ++p_)                                   \
                *reinterpret_cast<f32x4*>(wd_ + (tp_ * N_T + p_ * RP) * LDK) = WR_[tp_ * WPASS + p_]; \
    }
#define SPDM_STORE_W(buf_) SPDM_STORE_W_X(wreg, buf_)
    // fp32 slab row: 32 floats.  split slab row: [32 x fp16 hi | 32 x fp16 lo] (same 128 bytes).
    // TRANSFORM turns the raw loaded values into what the slab holds (in the same registers; the split
    // form packs {hi0..3} into .xy and {lo0..3} into .zw); WRITE_A puts them into LDS.
#define SPDM_TRANSFORM_A_X(AR_, G_, B_)                                                             \
    {                                                                                               \
        _Pragma("unroll") for (int p_ = 0; p_ < APASS; ++p_) {                                     \
            f32x4 v_ = AR_[p_];                                                                     \
            if (pro) {                                                                              \
                const float rs_ = arstd[p_], mu_ = amean[p_];                                       \
                v_.x = (v_.x - mu_) * (rs_ * G_.x) + B_.x;                                          \
                v_.y = (v_.y - mu_) * (rs_ * G_.y) + B_.y;                                          \
                v_.z = (v_.z - mu_) * (rs_ * G_.z) + B_.z;                                          \
                v_.w = (v_.w - mu_) * (rs_ * G_.w) + B_.w;                                          \
                if (pro_gelu) {                                                                     \
                    v_.x = gelu_erf(v_.x); v_.y = gelu_erf(v_.y);                                   \
                    v_.z = gelu_erf(v_.z); v_.w = gelu_erf(v_.w);                                   \
                }                                                                                   \
            }                                                                                       \
            if (!aval[p_]) v_ = f32x4{0.f, 0.f, 0.f, 0.f};                                          \
            if (PREC == PREC_SPLIT) {                                                               \
                const f32x2 p0_ = split_pair(v_.x, v_.y), p1_ = split_pair(v_.z, v_.w);             \
                v_ = f32x4{p0_.x, p1_.x, p0_.y, p1_.y};                                             \
            }                                                                                       \
            AR_[p_] = v_;                                                                           \
        }                                                                                           \
    }
#define SPDM_TRANSFORM_A() SPDM_TRANSFORM_A_X(areg, g4r, b4r)
#define SPDM_WRITE_A(buf_) SPDM_WRITE_A_X(areg, buf_)
#define SPDM_WRITE_A_X(AR_, buf_)                                                                   \
    {                                                                                               \
        float* ad_ = Abuf + (buf_) * QZ * LDK;                                                      \
        _Pragma("unroll") for (int p_ = 0; p_ < APASS; ++p_) {                                     \
            if (p_ * RP + srow_t < QA) {                                                            \
                float* row_ = ad_ + (p_ * RP + srow_t) * LDK;                                       \
                if (PREC == PREC_SPLIT) {   /* hi: 8 bytes at c4*8, lo: 8 bytes at 64 + c4*8 */     \
                    *reinterpret_cast<f32x2*>(row_ + c4 * 2) = f32x2{AR_[p_].x, AR_[p_].y};         \
                    *reinterpret_cast<f32x2*>(row_ + 16 + c4 * 2) = f32x2{AR_[p_].z, AR_[p_].w};    \
                } else {                                                                            \
                    *reinterpret_cast<f32x4*>(row_ + c4 * 4) = AR_[p_];                             \
                }                                                                                   \
            }                                                                                       \
        }                                                                                           \
    }

    const int nchunks = kc1 - kc0;            // chunk indices below are relative to kc0
    const int ngroups = taps / TPI;
    const int niter = nchunks * ngroups;

    SPDM_LOAD_A(0)
    if (WDMA) SPDM_DMA_W(0, 0, 0)
    SPDM_LOAD_W(0, 0)
    SPDM_TRANSFORM_A()
    SPDM_WRITE_A(0)
    SPDM_STAMP()
    SPDM_STORE_W(0)
    __syncthreads();
    SPDM_STAMP()

    // The MFMA block of one iteration (TPI taps on the A slab of `chunk_` and W group buffer `wbuf_`).
#define SPDM_MFMA_BLOCK(chunk_, tg_, wbuf_)                                                         \
    {                                                                                               \
        const float* Ab = Abuf + (NBA == 2 ? ((chunk_) & 1) : 0) * QZ * LDK;                        \
        const float* Wb = Wbuf + (wbuf_) * TPI * N_T * WROW;                                        \
        const float* ap[TPI][MT];                                                                   \
        _Pragma("unroll") for (int tp = 0; tp < TPI; ++tp) {                                       \
            const int tap = (tg_) * TPI + tp;                                                       \
            int shift = 0;                                                                          \
            if (HALO) {                                                                             \
                const int dh = (taps == 9) ? tap / 3 - 1 : tap - 1;                                 \
                const int dw = (taps == 9) ? tap % 3 - 1 : 0;                                       \
                shift = (dh * W + dw) * LDK;                                                        \
            }                                                                                       \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
                ap[tp][mt] = Ab + (((amask[mt] >> tap) & 1u) ? aoff[mt] + shift : zoff);            \
        }                                                                                           \
        if (dbg & DBG_NO_MFMA) {                                                                    \
        } else if (W2) {                                                                            \
            /* 8 half-steps: {centre, side} x {k-step 0, 1} x {tile 0, tile 1}; reads of half-step hs  \
               are in flight while the MFMAs of half-step hs-1 run */                                \
            f16x8 fa[2][2], fb[2][NT][2];                                                           \
            _Pragma("unroll") for (int hs = 0; hs <= 8; ++hs) {                                    \
                if (hs < 8) {                                                                       \
                    const int vt = hs >> 2, s2 = (hs >> 1) & 1, mt = hs & 1, set = hs & 1;          \
                    const int tp = (vt == 0) ? 1 : (mt == 0 ? 2 : 0);                               \
                    fa[set][0] = *reinterpret_cast<const f16x8*>(ap[tp][mt] + s2 * 8);              \
                    fa[set][1] = *reinterpret_cast<const f16x8*>(ap[tp][mt] + 16 + s2 * 8);         \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                            \
                        fb[set][nt][0] = *reinterpret_cast<const f16x8*>(Wb + tp * N_T * WROW + boff[nt] + SPDM_WOFF(s2, 0)); \
                        fb[set][nt][1] = *reinterpret_cast<const f16x8*>(Wb + tp * N_T * WROW + boff[nt] + SPDM_WOFF(s2, 1)); \
                    }                                                                               \
                }                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                if (hs > 0) {                                                                       \
                    const int mt = (hs - 1) & 1, set = (hs - 1) & 1;                                \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                            \
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][0], fb[set][nt][0], acc[mt][nt], 0, 0, 0); \
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][0], fb[set][nt][1], acc[mt][nt], 0, 0, 0); \
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][1], fb[set][nt][0], acc[mt][nt], 0, 0, 0); \
                    }                                                                               \
                }                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                  \
            }                                                                                       \
        } else if (PREC == PREC_SPLIT) {                                                            \
            /* row = [hi: 32 fp16 | lo: 32 fp16]; K=16 step s2 of tap tp, lane half kh: 8 fp16 at  \
               byte 32 s2 + 16 kh.  Two fragment sets: the reads of step st are in flight while    \
               the MFMAs of step st-1 run. */                                                       \
            constexpr int NSTEP = 2 * TPI;                                                          \
            f16x8 fa[2][MT][2], fb[2][NT][2];                                                       \
            _Pragma("unroll") for (int st = 0; st <= NSTEP; ++st) {                                \
                if (st < NSTEP) {                                                                   \
                    const int tp = st >> 1, s2 = st & 1, set = st & 1;                              \
                    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                            \
                        fa[set][mt][0] = *reinterpret_cast<const f16x8*>(ap[tp][mt] + s2 * 8);      \
                        fa[set][mt][1] = *reinterpret_cast<const f16x8*>(ap[tp][mt] + 16 + s2 * 8); \
                    }                                                                               \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                            \
                        fb[set][nt][0] = *reinterpret_cast<const f16x8*>(Wb + tp * N_T * WROW + boff[nt] + SPDM_WOFF(s2, 0)); \
                        fb[set][nt][1] = *reinterpret_cast<const f16x8*>(Wb + tp * N_T * WROW + boff[nt] + SPDM_WOFF(s2, 1)); \
                    }                                                                               \
                }                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                if (st > 0) {                                                                       \
                    const int set = (st - 1) & 1;                                                   \
                    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                              \
                        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                        \
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][mt][0], fb[set][nt][0], acc[mt][nt], 0, 0, 0); \
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][mt][0], fb[set][nt][1], acc[mt][nt], 0, 0, 0); \
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][mt][1], fb[set][nt][0], acc[mt][nt], 0, 0, 0); \
                        }                                                                           \
                }                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                  \
            }                                                                                       \
        } else {                                                                                    \
            _Pragma("unroll") for (int tp = 0; tp < TPI; ++tp)                                     \
                _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                    \
                    f32x4 av[MT], bv[NT];                                                           \
                    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                              \
                        av[mt] = *reinterpret_cast<const f32x4*>(ap[tp][mt] + g * 8);               \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                              \
                        bv[nt] = *reinterpret_cast<const f32x4*>(Wb + tp * N_T * WROW + boff[nt] + g * 8); \
                    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                              \
                        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                        \
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].x, bv[nt].x, acc[mt][nt], 0, 0, 0); \
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].y, bv[nt].y, acc[mt][nt], 0, 0, 0); \
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].z, bv[nt].z, acc[mt][nt], 0, 0, 0); \
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].w, bv[nt].w, acc[mt][nt], 0, 0, 0); \
                        }                                                                           \
                }                                                                                   \
        }                                                                                           \
    }

    int chunk = 0, tg = 0;
#ifdef SPDM_DIAG
    constexpr bool PP_COMPILED = PP;
#else
    constexpr bool PP_COMPILED = false;      // product build: no second loop body, no extra register pressure
#endif
    if constexpr (DEEP) {
        // iteration j = (chunk j / ngroups, tap group j % ngroups).  Data of iteration j + 2 is loaded during iteration j into
        // staging set j & 1; data of iteration j + 1 (loaded during j - 1, set (j + 1) & 1) goes to LDS after the MFMA block
        // of j: W -> buffer (j + 1) & 1, and -- when j + 1 opens a chunk -- the A slab -> buffer chunk & 1 (both were last
        // read two barriers ago).  One barrier per iteration.  The prologue above staged iteration 0 from set 0.
        {   // data of iteration 1 -> set 1
            const int j1 = min(1, niter - 1);
            const int c1 = j1 / ngroups, t1 = j1 - c1 * ngroups;
            if (t1 == 0) SPDM_LOAD_A_X(areg2, g4r2, b4r2, c1)
            SPDM_LOAD_W_X(wreg2, c1, t1)
        }
        int c_cur = 0, t_cur = 0;                       // (chunk, tap group) of iteration j
        for (int it = 0; it < niter; it += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int j = it + half;
                if (j < niter) {
                    int t_n1 = t_cur + 1, c_n1 = c_cur;
                    if (t_n1 == ngroups) { t_n1 = 0; c_n1 = c_cur + 1; }
                    int t_n2 = t_n1 + 1, c_n2 = c_n1;
                    if (t_n2 == ngroups) { t_n2 = 0; c_n2 = c_n1 + 1; }
                    const bool have1 = (j + 1 < niter), have2 = (j + 2 < niter);
                    // loads of iteration j + 2 (past the end: the last iteration's data again, never stored)
                    const int cl = have2 ? c_n2 : c_cur, tl = have2 ? t_n2 : t_cur;
                    if (half == 0) {
                        if (have2 && t_n2 == 0) SPDM_LOAD_A_X(areg, g4r, b4r, cl)
                        SPDM_LOAD_W_X(wreg, cl, tl)
                    } else {
                        if (have2 && t_n2 == 0) SPDM_LOAD_A_X(areg2, g4r2, b4r2, cl)
                        SPDM_LOAD_W_X(wreg2, cl, tl)
                    }
                    SPDM_MFMA_BLOCK(c_cur, t_cur, j & 1)
                    if (have1) {
                        if (half == 0) {
                            SPDM_STORE_W_X(wreg2, (j + 1) & 1)
                            if (t_n1 == 0) { SPDM_TRANSFORM_A_X(areg2, g4r2, b4r2) SPDM_WRITE_A_X(areg2, c_n1 & 1) }
                        } else {
                            SPDM_STORE_W_X(wreg, (j + 1) & 1)
                            if (t_n1 == 0) { SPDM_TRANSFORM_A_X(areg, g4r, b4r) SPDM_WRITE_A_X(areg, c_n1 & 1) }
                        }
                    }
                    __syncthreads();
                    t_cur = t_n1;
                    c_cur = c_n1;
                }
            }
        }
    } else if (PP_COMPILED && (dbg & DBG_PP)) {
        // Ping-pong schedule (measured SLOWER than the plain schedule on MI355X; kept for experiments:
        // enable with DBG_PP) (8 waves): waves 0-3 (one per SIMD) and waves 4-7 alternate between
        // "run the MFMA block" and "stage the next slabs" (LDS writes of W, GroupNorm/GELU/split
        // transform of A), separated by workgroup barriers -- the matrix pipe always has exactly one
        // MFMA-issuing wave per SIMD and the staging work of one half hides behind the MFMAs of the other.
        const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);
        for (int it = 0; it < niter; ++it) {
            int ntg = tg + 1, nchunk = chunk;
            if (ntg == ngroups) { ntg = 0; nchunk = chunk + 1; }
            const bool have_next = (it + 1 < niter);
            const bool next_A = have_next && (ntg == 0);
            if (next_A && !(dbg & DBG_NO_ALOAD)) SPDM_LOAD_A(nchunk)
            if (have_next && !(dbg & DBG_NO_WLOAD)) SPDM_LOAD_W(nchunk, ntg)
            for (int ph = 0; ph < 2; ++ph) {
                SPDM_STAMP()
                if (grp == ph) {
                    SPDM_MFMA_BLOCK(chunk, tg, it & 1)
                } else {
                    if (have_next) SPDM_STORE_W((it + 1) & 1)
                    if (next_A) SPDM_TRANSFORM_A()
                }
                SPDM_STAMP()
                __syncthreads();
            }
            if (next_A) {               // both halves are done with the single A slab
                SPDM_WRITE_A(0)
                __syncthreads();
            }
            tg = ntg;
            chunk = nchunk;
        }
    } else {
        // 8-wave workgroups stagger the A-slab transform (GroupNorm affine, erf-GELU, fp16 split -- ~30 VALU
        // instructions per element) between their two halves: waves 0-3 load the next slab one tap group
        // EARLIER and transform it BEFORE their last MFMA block of the chunk, waves 4-7 transform AFTER theirs,
        // so one half's VALU work runs while the other half keeps the matrix pipe busy.
        const bool early = (NTHR == 512) && (ngroups >= 2) && (__builtin_amdgcn_readfirstlane(tid >> 8) == 0);
        for (int it = 0; it < niter; ++it) {
            int ntg = tg + 1, nchunk = chunk;
            if (ntg == ngroups) { ntg = 0; nchunk = chunk + 1; }
            const bool have_next = (it + 1 < niter);
            const bool next_A = have_next && (ntg == 0);
            const bool more_chunks = (chunk + 1 < nchunks);
            // A before W: hipcc guards the re-use of the A staging registers with a vmcnt wait that would
            // otherwise also wait for the W loads issued just before it
            if (!(dbg & DBG_NO_ALOAD)) {
                if (early) {
                    if (more_chunks && tg == ngroups - 2) SPDM_LOAD_A(chunk + 1)
                } else {
                    if (next_A) SPDM_LOAD_A(nchunk)
                }
            }
            if (have_next && !(dbg & DBG_NO_WLOAD)) {
                if (WDMA) SPDM_DMA_W(nchunk, ntg, (it + 1) & 1)
                SPDM_LOAD_W(nchunk, ntg)
            }
            if (early && next_A) SPDM_TRANSFORM_A()
            SPDM_STAMP()
            SPDM_MFMA_BLOCK(chunk, tg, it & 1)
            SPDM_STAMP()
            if (have_next) SPDM_STORE_W((it + 1) & 1)
            if (next_A) {
                if (!early) SPDM_TRANSFORM_A()
                if (NBA == 1) __syncthreads();      // every wave is done reading the single A slab
                SPDM_WRITE_A(NBA == 2 ? (nchunk & 1) : 0)
            }
            SPDM_STAMP()
            __syncthreads();
            SPDM_STAMP()
            tg = ntg;
            chunk = nchunk;
        }
    }
#undef SPDM_MFMA_BLOCK
#undef SPDM_LOAD_A_X
#undef SPDM_LOAD_W_X
#undef SPDM_STORE_W_X
#undef SPDM_TRANSFORM_A_X
#undef SPDM_WRITE_A_X
#undef SPDM_DMA_W
#undef SPDM_WOFF
#undef SPDM_LOAD_A
#undef SPDM_LOAD_W
#undef SPDM_STORE_W
#undef SPDM_TRANSFORM_A
#undef SPDM_WRITE_A

    if (PREC == PREC_SPLIT) {   // undo the operand pre-scaling (exact: power of two)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= SPLIT_DESCALE;
    }

    // ---- epilogue ----
    // The accumulator layout has channels on lanes and rows on registers, i.e. one dword per lane per
    // store.  Going through LDS (the A/W slabs are dead now) turns the tile into whole rows so that
    // every lane stores 16 bytes and a wave instruction writes whole 512-byte row pieces: 4x fewer
    // store instructions (the store tail is issue-bound, not bandwidth-bound).
    SPDM_STAMP()
    float* srow = smem;                                  // [M_T / unit][WN][2] GroupNorm partials
    constexpr int SROW_FLOATS = M_T * WN * 2;
    float* otile = smem + SROW_FLOATS;                   // [M_T][N_T] fp32 output tile
    const bool unit4 = !W2 && (HW & 3) == 0;      // W2: a register quad is 4 rows 2 apart -> per-row partials
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col_l = wn * NT * 32 + nt * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row_l = SPDM_ROW(mt, (r & 3) + 8 * (r >> 2) + 4 * kh);
                otile[row_l * N_T + col_l] = acc[mt][nt][r];
            }
        }

    const int epi = (ksp > 1) ? EPI_PLAIN : a.epi;                       // split-K: raw partial tile only
    float* const dstp = (ksp > 1) ? a.partial + (size_t)ks * M * N : a.dst;
    const int dst_ld = (ksp > 1) ? N : a.dst_ld;
    if (epi == EPI_STATS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r0 = wm * MT * 32 + mt * 32 + 8 * g + 4 * kh;     // tile-local row of register 4g (unpermuted)
                if (unit4) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = acc[mt][nt][4 * g + j];
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
                            const float v = acc[mt][nt][4 * g + j];
                            s1 += v;
                            s2 += v * v;
                        }
                        s1 = half_sum(s1);
                        s2 = half_sum(s2);
                        if (li == 0) {
                            const int rr = SPDM_ROW(mt, 8 * g + 4 * kh + j);
                            srow[(rr * WN + wn) * 2] = s1;
                            srow[(rr * WN + wn) * 2 + 1] = s2;
                        }
                    }
                }
            }
    }
    SPDM_STAMP()
    __syncthreads();
    SPDM_STAMP()

    if (epi == EPI_STATS) {
        const int t_lo = m0, t_hi = min(m0 + M_T, M);
        if (t_hi > t_lo) {
            const int b_first = t_lo / HW, b_last = (t_hi - 1) / HW;
            const int ush = unit4 ? 2 : 0;
            for (int t = tid; t <= b_last - b_first; t += NTHR) {
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
        constexpr int TPR = N_T / 4, RPP = NTHR / TPR;    // threads per row, rows per pass
        const int c4o = tid % TPR, r0 = tid / TPR;
        const bool has_bias = (epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_BIAS_RESID);
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (has_bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + n0 + c4o * 4);
#pragma unroll 4
        for (int p = 0; p < M_T / RPP; ++p) {
            const int row_l = p * RPP + r0;
            const int row = m0 + row_l;
            if (row < M && !(dbg & DBG_NO_STORE)) {
                f32x4 v = *reinterpret_cast<const f32x4*>(otile + row_l * N_T + c4o * 4);
                v += bias4;
                if (epi == EPI_BIAS_GELU) {
                    v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
                }
                if (epi == EPI_BIAS_RESID)
                    v += *reinterpret_cast<const f32x4*>(a.resid + (size_t)row * a.resid_ld + n0 + c4o * 4);
                *reinterpret_cast<f32x4*>(dstp + (size_t)row * dst_ld + n0 + c4o * 4) = v;
                if (a.row_stats != nullptr) {   // LayerNorm partials of the consumer: the row sits on TPR consecutive lanes
                    float s1 = (v.x + v.y) + (v.z + v.w);
                    float s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
                    for (int o2 = TPR >> 1; o2 > 0; o2 >>= 1) {
                        s1 += __shfl_xor(s1, o2, 64);
                        s2 += __shfl_xor(s2, o2, 64);
                    }
                    if (c4o == 0) {
                        double* rs = a.row_stats + ((size_t)row * n_ntiles + ntile) * 2;
                        rs[0] = (double)s1;
                        rs[1] = (double)s2;
                    }
                }
            }
        }
    }
    SPDM_STAMP()
#ifdef SPDM_DIAG
    if (stamping) a.stamps[(tid >> 8) * 128 + 127] = (unsigned long long)nstamp;
#endif
#undef SPDM_STAMP
#undef SPDM_ROW
}

// -------------------------------------------------------------------------------------------------
// Launch configurations.  cfg 0: 4 waves, 128 x {64,128}, one tap per barrier (exact-fp32 path and
// 1-tap GEMMs).  cfg 1: 8 waves, 256 x {64,128}, three taps per barrier (split path, 3x3 / 3x1 convs).
// Tile choice is occupancy-aware: the 8-wave 256-row configuration only when it still yields enough
// workgroups to cover the 256 CUs; otherwise 128-row tiles, and 64-wide instead of 128-wide tiles when even
// those would leave CUs idle (small batches / coarse levels).
// tuning knobs for experiments (SPDM_TUNE0.. read once per process; defaults = the measured choice)
int spdm_tune(int idx, int dflt) {
    constexpr int NT = 24;
    static int val[NT];
    static bool have[NT], init = false;
    if (!init) {
        for (int i = 0; i < NT; ++i) {
            char name[20];
            snprintf(name, sizeof(name), "SPDM_TUNE%d", i);
            const char* e = getenv(name);
            have[i] = e != nullptr;
            val[i] = e ? atoi(e) : 0;
        }
        init = true;
    }
    return (idx >= 0 && idx < NT && have[idx]) ? val[idx] : dflt;
}

GemmGeom gemm_geometry(int M, int N, int K, int HW, int W, int taps, int split, unsigned sw, bool stats_epi) {
    GemmGeom g;
    g.skinny = 0;
    g.reg = 0;
    if (stats_epi && conv_reg_geometry(M, N, K, HW, W, taps, split, sw)) {
        g.reg = 1;
        g.m_tile = 64; g.n_tile = 64; g.n_tiles = 1; g.ksplit = 1;
        g.st_m_tile = 64; g.st_n_tiles = 1;
        g.slots = stats_slots(HW, 64, 1);
        return g;
    }
    if (stats_epi && taps != 1 && conv_skinny_geometry(M, N, K, HW, W, taps, split, sw, &g.m_tile, &g.n_tile)) {
        g.skinny = 1;
        g.ksplit = 1;
        g.n_tiles = N / g.n_tile;
        g.st_m_tile = g.m_tile;
        g.st_n_tiles = g.n_tiles;
        g.slots = stats_slots(HW, g.st_m_tile, g.st_n_tiles);
        return g;
    }
    // tuning override (tools/autotune_convs.py): SPDM_TUNE5 / 6 / 7 = forced m_tile / n_tile / ksplit of the statistics-epilogue convs
    if (stats_epi && split && taps != 1 && spdm_tune(5, 0) > 0) {
        g.m_tile = spdm_tune(5, 0);
        g.n_tile = spdm_tune(6, 64);
        g.ksplit = std::max(1, spdm_tune(7, 1));
        if (N % g.n_tile != 0 || (g.m_tile != 128 && g.m_tile != 256)) { g.n_tile = 64; g.m_tile = 128; }
        g.n_tiles = N / g.n_tile;
        if (g.ksplit > K / CK) g.ksplit = K / CK;
        while (g.ksplit > 1 && (size_t)g.ksplit * M * N * sizeof(float) > SPLITK_WORKSPACE_BYTES) --g.ksplit;
        if (g.ksplit > 1) { g.st_m_tile = combine_rows(HW, N); g.st_n_tiles = 1; }
        else { g.st_m_tile = g.m_tile; g.st_n_tiles = g.n_tiles; }
        g.slots = stats_slots(HW, g.st_m_tile, g.st_n_tiles);
        return g;
    }
    const int nt128 = (N % 128 == 0) ? N / 128 : 0;
    const int nt_pref = nt128 ? nt128 : N / 64;
    bool big = split && taps != 1 && M >= 256;
    // 128-wide layers: a 256-row tiling of <= 256 workgroups (at most one per CU) loses to twice as many 128-row tiles, two per
    // CU (up2.dc1 at B = 512: 115 -> 103, 127 -> 114 us; down1.dc2 at B = 1024: 42 -> 36, 72 -> 63); 64-wide layers do not
    // (down1.dc1 at 1024: 25 -> 28): tools/bench_convs.py with SPDM_TUNE4
    if (big && (long long)((M + 255) / 256) * nt_pref < spdm_tune(4, nt128 ? 257 : 192)) big = false;
    // 3-tap convs (W == 1 level): 128-row tiles (conv_wide's 4 x (64 x 64) variant, two workgroups per CU) until the
    // 256-row tiling would give every CU two workgroups
    if (big && taps == 3 && HW % 4 == 0 && (long long)((M + 255) / 256) * nt_pref < 512 && !(sw & SW_T3_BIG)) big = false;
    g.m_tile = big ? 256 : 128;
    g.n_tile = nt128 ? 128 : 64;
    if (!big && nt128 && (long long)((M + 127) / 128) * nt128 < 192) g.n_tile = 64;
    // 64-wide outputs: 512-row tiles keep 72 MFMAs per wave between barriers (8 waves x 64x64)
    // (superseded for HW % 4 == 0 by conv_wide.hip's 256 x 64 configuration, two workgroups per CU)
    if (big && g.n_tile == 64 && taps == 9 && (long long)((M + 511) / 512) * (N / 64) >= 192 &&
        !(sw & SW_NO_T512) && ((HW & 3) != 0 || (sw & SW_T512)))
        g.m_tile = 512;
    g.n_tiles = N / g.n_tile;
    g.ksplit = 1;
    // Split-K: a grid that leaves most CUs idle (small batches; the coarse levels at medium batches) is a few long serial
    // K loops, each paced by one weight round trip per iteration.  Split the 32-channel chunks over ksplit x as many
    // workgroups (partial slabs + launch_splitk_combine, kernels.h) until the grid reaches ~one workgroup per CU.
    const bool may_split = stats_epi && split && taps != 1 && K % CK == 0 && !(sw & SW_NO_SPLITK);
    // Width-2 maps (level 2): the 256-row kernels skip the zero-padding taps (a third of the MFMAs) and read half the weight
    // bytes per MFMA; keep that tiling and let split-K restore the grid (measured at B = 256 / 512 / 1024 on up1.dc1:
    // 83 -> 63, 126 -> 100 us; on width-4 maps the same trade LOSES 5-10 us per layer, so only here)
    // (K >= 256 only: with four chunks the combine pass costs more than the skipped taps save -- down2.dc1 at B = 2048: 48 vs 39 us)
    if (may_split && !big && split && taps == 9 && W == 2 && (HW & 7) == 0 && M >= 256 && K % 64 == 0 && K >= 256 && spdm_tune(2, 1) != 0) {
        const long long t256 = (long long)((M + 255) / 256) * nt_pref;
        const int target = spdm_tune(0, 256);
        int S = (int)std::min<long long>(K / 64, (target + t256 - 1) / t256);
        while (S > 1 && (size_t)S * M * N * sizeof(float) > SPLITK_WORKSPACE_BYTES) --S;
        if (t256 * S >= target * 3 / 4 && S <= spdm_tune(3, 8)) {
            g.m_tile = 256; g.n_tile = nt128 ? 128 : 64; g.n_tiles = N / g.n_tile; g.ksplit = S;
            if (S > 1) { g.st_m_tile = combine_rows(HW, N); g.st_n_tiles = 1; }      // the combine kernel writes the statistics
            else { g.st_m_tile = g.m_tile; g.st_n_tiles = g.n_tiles; }               // (S == 1: the plain 256-row launch)
            g.slots = stats_slots(HW, g.st_m_tile, g.st_n_tiles);
            return g;
        }
    }
    if (may_split && !big) {
        const int nchunks = K / CK;
        const int target = spdm_tune(0, 256);
        // with split-K available, 128-wide tiles (conv_wide.hip's 128-row variant: weights straight from L2 into registers)
        // no longer need to be narrowed to fill the chip
        // (measured the other way round: 64-wide conv_gemm tiles + split-K beat 128-wide conv_wide tiles + split-K at every
        //  batch from 1 to 512 -- 540 vs 629 us per step at B = 1 -- so this preference stays an experiment, SPDM_TUNE1=1)
        const bool wide128 = nt128 && (HW & 3) == 0 && K % 64 == 0 && !(sw & (SW_NO_WIDE | SW_NO_WIDE128)) && spdm_tune(1, 0) != 0;
        const long long mt = (M + 127) / 128;
        if (wide128 && mt * nt128 < target) g.n_tile = 128, g.n_tiles = nt128;
        const long long tiles = mt * g.n_tiles;
        if (tiles < target) {
            // chunks per split step: a 128 x 128 tiling that conv_wide.hip takes (conv_wide_supported: rows per sample a
            // multiple of 4, K a multiple of 64) walks its chunks in PAIRS -- kc0 = 2 (ks (K / 64) / ksplit) -- so splitting
            // finer than K / 64 would give some workgroups an empty chunk range (an all-zero slab and a wasted combine pass)
            const bool pairs = g.n_tile == 128 && (HW & 3) == 0 && K % 64 == 0 && !(sw & (SW_NO_WIDE | SW_NO_WIDE128));
            const int unit = pairs ? 2 : 1;
            int S = (int)std::min<long long>(nchunks / unit, (target + tiles - 1) / tiles);
            while (S > 1 && (size_t)S * M * N * sizeof(float) > SPLITK_WORKSPACE_BYTES) --S;
            if (S > 1) g.ksplit = S;
        }
    }
    if (g.ksplit > 1) {          // the combine kernel writes the statistics
        g.st_m_tile = combine_rows(HW, N);
        g.st_n_tiles = 1;
    } else {
        g.st_m_tile = g.m_tile;
        g.st_n_tiles = g.n_tiles;
    }
    g.slots = stats_slots(HW, g.st_m_tile, g.st_n_tiles);
    return g;
}

// Which launches take a fused source (GemmArgs: PRO_POOL / PRO_UPCAT): the small-grid kernel (conv_skinny.hip) for both modes.
// The plan (spdm_api.hip) asks before it decides whether to materialise the pooled / concatenated tensor.
bool gemm_takes_fused_source(const GemmArgs& a) {
    if (!(a.pro == PRO_POOL || a.pro == PRO_UPCAT) || !a.split || a.wgt_frag == nullptr || (a.sw & SW_NO_FUSED_SRC) || a.epi != EPI_STATS) return false;
    if (a.pro == PRO_UPCAT && (a.up_C <= 0 || a.up_C >= a.K || a.up_C % CK != 0 || (a.H & 1) || (a.W & 1))) return false;
    const GemmGeom g = gemm_geometry(a.geom_M > 0 ? a.geom_M : a.M, a.N, a.K, a.HW, a.W, a.taps, a.split, a.sw, a.epi == EPI_STATS && a.partial != nullptr);
    return g.skinny != 0;
}

// Two-source input (GemmArgs::skip with pro = PRO_NONE / PRO_GN: channels [0, up_C) from src, the rest from skip): conv_wide.hip's
// 128-wide configurations.  Same question as above, for the plan's "upsample only, never concatenate" path.
bool gemm_takes_two_sources(const GemmArgs& a0) {
    if (a0.skip == nullptr || !a0.split || a0.wgt_frag == nullptr || (a0.sw & SW_NO_FUSED_SRC) || a0.epi != EPI_STATS) return false;
    GemmArgs a = a0;
    const GemmGeom g = gemm_geometry(a.geom_M > 0 ? a.geom_M : a.M, a.N, a.K, a.HW, a.W, a.taps, a.split, a.sw, a.epi == EPI_STATS && a.partial != nullptr);
    if (g.skinny || g.reg || g.m_tile == 512) return false;
    a.ksplit = g.ksplit;
    return conv_wide_supported(a, g);
}

// 2 x MACs the launch actually evaluates (the W = 2 zero-tap skipping runs 6 of the 9 taps)
static bool uses_w2(const GemmArgs& a, const GemmGeom& g) {
    return a.split && g.m_tile == 256 && a.taps == 9 && a.W == 2 && a.HW % 2 == 0 && !(a.sw & SW_NO_W2);
}
// width-4 maps on conv_wide.hip's row-permuted variants: 20 of 24 tile steps per kernel row
static bool uses_wp4(const GemmArgs& a0, const GemmGeom& g) {
    if (!(a0.split && a0.taps == 9 && a0.W == 4 && g.n_tile == 128 && (g.m_tile == 128 || g.m_tile == 256) && a0.HW % 16 == 0 &&
          a0.K % 64 == 0 && !(a0.sw & SW_NO_WP4)) || g.skinny) return false;
    GemmArgs a = a0;
    a.ksplit = g.ksplit;
    return conv_wide_supported(a, g);
}
// ... width-8 maps on the class-major variants (256-row tiles): 44 of 48
static bool uses_wp8(const GemmArgs& a0, const GemmGeom& g) {
    if (!(a0.split && a0.taps == 9 && a0.W == 8 && g.n_tile == 128 && g.m_tile == 256 && a0.HW % 32 == 0 && a0.K % 64 == 0 &&
          !(a0.sw & SW_NO_WP8)) || g.skinny) return false;
    GemmArgs a = a0;
    a.ksplit = g.ksplit;
    return conv_wide_supported(a, g);
}
double gemm_flops(const GemmArgs& a) {
    const GemmGeom g = gemm_geometry(a.geom_M > 0 ? a.geom_M : a.M, a.N, a.K, a.HW, a.W, a.taps, a.split, a.sw, a.epi == EPI_STATS);
    const double taps = uses_w2(a, g) ? 6.0 : uses_wp4(a, g) ? 7.5 : uses_wp8(a, g) ? 8.25 : (double)a.taps;
    return 2.0 * (double)a.M * (double)a.N * (double)a.K * taps;
}

template <bool HALO, int PREC, int WM, int WN, int MT, int NT, int TPI, bool W2 = false, bool DEEP = false>
static hipError_t launch_cfg(const GemmArgs& a, const GemmGeom& g, hipStream_t s) {
    constexpr int M_T = WM * MT * 32, N_T = WN * NT * 32, NTHR = WM * WN * 64;
    constexpr int NBA = (TPI == 1 || DEEP) ? 2 : 1;
    const int halo = HALO ? a.W + 1 : 0;
    const int QA = M_T + 2 * halo;
    const int NSP = (QA + 4) & ~3;
#ifdef SPDM_WDMA
    constexpr bool WDMA = (NTHR == 512) && (PREC == PREC_SPLIT);
#else
    constexpr bool WDMA = false;
#endif
    constexpr int WROW = WDMA ? 32 : LDK;
    size_t lds = (size_t)(NBA * (QA + 1) * LDK + 2 * TPI * N_T * WROW + 2 * NSP) * sizeof(float);
    lds = std::max(lds, (size_t)(M_T * WN * 2 + M_T * N_T) * sizeof(float));     // epilogue staging: srow + output tile
    if (lds > 160 * 1024 || g.m_tile != M_T || g.n_tile != N_T) return hipErrorInvalidValue;
    auto kern = conv_gemm_kernel<HALO, PREC, WM, WN, MT, NT, TPI, W2, DEEP>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return e;
    const int n_mtiles = (a.M + M_T - 1) / M_T;
    const int grid = n_mtiles * g.n_tiles * std::max(a.ksplit, 1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, s, a, g.slots);
    return hipGetLastError();
}

static hipError_t launch_gemm_kernel(const GemmArgs& a, const GemmGeom& g, hipStream_t s);

hipError_t launch_gemm(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.ksplit = 1;
    const GemmGeom g = gemm_geometry(a.geom_M > 0 ? a.geom_M : a.M, a.N, a.K, a.HW, a.W, a.taps, a.split, a.sw, a.epi == EPI_STATS && a.partial != nullptr);
    const bool fused_src = a.pro == PRO_POOL || a.pro == PRO_UPCAT;
    if (fused_src && !g.skinny) return hipErrorInvalidValue;                            // (the plan asks gemm_takes_fused_source first)
    if (!fused_src && a.skip != nullptr && !gemm_takes_two_sources(a)) return hipErrorInvalidValue;   // (... gemm_takes_two_sources)
    if (g.skinny) return launch_conv_skinny(a, g, s);
    if (g.reg) return launch_conv_reg64(a, g, s);
    if (g.ksplit > 1) {
        if ((size_t)g.ksplit * a.M * a.N * sizeof(float) > SPLITK_WORKSPACE_BYTES || a.dst_ld != a.N) return hipErrorInvalidValue;
        a.ksplit = g.ksplit;
        if (hipError_t e = launch_gemm_kernel(a, g, s); e != hipSuccess) return e;
        return launch_splitk_combine(a.partial, g.ksplit, a.dst, a.M, a.N, a.HW, a.epi_stats, s);
    }
    return launch_gemm_kernel(a, g, s);
}

static hipError_t launch_gemm_kernel(const GemmArgs& a, const GemmGeom& g, hipStream_t s) {
    // shape contract of the kernel -- checked on the host so that a bad plan can never fault the GPU
    if (a.M <= 0 || a.K <= 0 || a.N <= 0) return hipErrorInvalidValue;
    if (a.K % CK != 0 || a.N % 64 != 0) return hipErrorInvalidValue;
    if (a.src_ld % 4 != 0 || a.src_ld < (a.skip != nullptr ? a.up_C : a.K) || a.dst_ld % 4 != 0) return hipErrorInvalidValue;     // (two-source input: src holds the first up_C channels)
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
    if (a.split) {
        if (a.taps == 1) {
            if (g.n_tile == 128) return launch_cfg<false, PREC_SPLIT, 2, 2, 2, 2, 1>(a, g, s);
            return launch_cfg<false, PREC_SPLIT, 2, 2, 2, 1, 1>(a, g, s);
        }
        if (g.m_tile == 512) return launch_cfg<true, PREC_SPLIT, 8, 1, 2, 2, 3>(a, g, s);
        // width-2 maps whose 256-row tiling is at most one workgroup per CU: conv_gemm's 8-wave form of the same tile (two waves per
        // SIMD) instead of conv_wide's 4-wave one (one) -- whole step, same box, alternating: -16 us at B = 1024, -22 us at 512
        // (SPDM_TUNE20 bit 2 off: the old routing; the same trade for the 128 x 128 tiles of the width-4 level LOSES 14-16 us)
        // (rows as the geometry was chosen for: a pinned shard must run the kernels of the whole batch)
        const int gM = a.geom_M > 0 ? a.geom_M : a.M;
        if ((spdm_tune(20, 7) & 4) && uses_w2(a, g) && g.m_tile == 256 && a.skip == nullptr &&
            (long long)((gM + 255) / 256) * g.n_tiles * std::max(a.ksplit, 1) <= 256) {
            if (g.n_tile == 128) return launch_cfg<true, PREC_SPLIT, 4, 2, 2, 2, 3, true>(a, g, s);
            return launch_cfg<true, PREC_SPLIT, 4, 2, 2, 1, 3, true>(a, g, s);
        }
        if (conv_wide_supported(a, g)) return launch_conv_wide(a, g, s);          // conv_wide.hip (256- or 128-row tiles)
        if (g.m_tile == 256) {
            if (uses_w2(a, g)) {   // level-2 maps: zero-tap skipping
                if (g.n_tile == 128) return launch_cfg<true, PREC_SPLIT, 4, 2, 2, 2, 3, true>(a, g, s);
                return launch_cfg<true, PREC_SPLIT, 4, 2, 2, 1, 3, true>(a, g, s);
            }
            if (g.n_tile == 128) return launch_cfg<true, PREC_SPLIT, 4, 2, 2, 2, 3>(a, g, s);
            return launch_cfg<true, PREC_SPLIT, 4, 2, 2, 1, 3>(a, g, s);
        }
        // few workgroups (small batches): the K loop is paced by one weight fetch per iteration, so walk three taps per
        // iteration (3 x the bytes in flight per round trip)
        const long long wgs = (long long)((gM + 127) / 128) * g.n_tiles;
        if (wgs <= 512 && !(a.sw & SW_NO_SMALL_TPI3)) {
            // DEEP (loads two iterations ahead, double-buffered A slab) is an opt-in experiment (SPDM_DEEP=1): measured over all
            // 31 layers at B = 1 / 256 / 512 it changes NOTHING (369 vs 364, 1018 vs 1014, 1440 vs 1441 us of convs per step) --
            // these launches are not paced by their weight round trips but by the serial phases of a workgroup that is alone on
            // its CU: ~1150 MFMA cycles + ~1600 cycles of GroupNorm/GELU/split staging + LDS writes and fragment reads per iteration.
            // (64-wide tiles only: with 128-wide tiles two staging sets do not fit 256 registers.)
            if ((a.sw & SW_DEEP) && g.n_tile == 64) return launch_cfg<true, PREC_SPLIT, 2, 2, 2, 1, 3, false, true>(a, g, s);
            // ... on EIGHT waves (32 x 32 per wave) rather than four (64 x 32 / 64 x 64): these grids are at most two workgroups per
            // CU, so a 4-wave workgroup leaves a SIMD with one or two waves and every round trip exposed.  Whole step, same box,
            // alternating (tools/probes/ab_switch.sh SPDM_TUNE20=0): -38 us at B = 1024, -41 us at 512, -18 us at 256, -16 us at 64.
            if (g.n_tile == 128 && (spdm_tune(20, 7) & 2)) return launch_cfg<true, PREC_SPLIT, 4, 2, 1, 2, 3>(a, g, s);
            if (g.n_tile == 128) return launch_cfg<true, PREC_SPLIT, 2, 2, 2, 2, 3>(a, g, s);
            if (spdm_tune(20, 7) & 1) return launch_cfg<true, PREC_SPLIT, 4, 2, 1, 1, 3>(a, g, s);
            return launch_cfg<true, PREC_SPLIT, 2, 2, 2, 1, 3>(a, g, s);
        }
        if (g.n_tile == 128) return launch_cfg<true, PREC_SPLIT, 2, 2, 2, 2, 1>(a, g, s);
        return launch_cfg<true, PREC_SPLIT, 2, 2, 2, 1, 1>(a, g, s);
    }
    if (a.taps == 1) {
        if (g.n_tile == 128) return launch_cfg<false, PREC_F32, 2, 2, 2, 2, 1>(a, g, s);
        return launch_cfg<false, PREC_F32, 2, 2, 2, 1, 1>(a, g, s);
    }
    if (g.n_tile == 128) return launch_cfg<true, PREC_F32, 2, 2, 2, 2, 1>(a, g, s);
    return launch_cfg<true, PREC_F32, 2, 2, 2, 1, 1>(a, g, s);
}

}  // namespace spdm
