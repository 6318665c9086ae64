// conv_wide.hip -- 3x3 convolution as an implicit GEMM, the configuration for the LARGE layers
// (128-wide output tiles at levels 0 and 1: up3.dc1, up2.dc1, down1 -- 55 % of the path's FLOPs).
//
// Same math, operand formats and LDS slab layout as conv_gemm.hip (split-fp16 operands, three
// v_mfma_f32_32x32x16_f16 per K = 16 into one fp32 accumulator, halo'd input slab with the
// GroupNorm(1,C) -> GELU prologue applied at staging, GroupNorm partial sums as the epilogue;
// replaces nn.Conv2d(k=3, padding=1, bias=False) + the GroupNorm/GELU around it,
// models/Unet_FiLmLayer.py:101-115).  What differs is how the work is laid on the CU:
//
//   * one workgroup = 4 waves (one per SIMD), each owning a 128 x 64 accumulator tile
//     (8 MFMA tiles = 128 accumulator registers): 256 x 128 outputs per workgroup with HALF the
//     weight-fragment LDS reads per MFMA of the 8-wave 64 x 64-per-wave configuration;
//   * the workgroup needs 77 KiB of LDS and <= 256 registers, so TWO workgroups share a CU and run
//     out of phase: one workgroup's prologue, slab hand-over, barrier waits and store tail overlap
//     the other's MFMA blocks (the 8-wave configuration runs one workgroup per CU, all of whose
//     waves stall together: measured 44-50 % matrix-pipe occupancy);
//   * the WEIGHT fragments never touch LDS: the host stores a fragment-order copy of the split weights
//     (frag_order_weights, kernels.h: one 1-KiB block per MFMA B operand, lane-contiguous), and each wave
//     loads its B operands straight into registers with coalesced global_load_dwordx4, one K = 16
//     half-tap (24 MFMAs) ahead; the waves of a workgroup then only meet at the slab hand-over (one
//     barrier pair per 32-channel chunk = per 432 MFMAs, instead of one barrier per tap), and the MFMA
//     stream of a chunk is unbroken;
//   * the input slab for the next 32-channel chunk is loaded at the hand-over (the fragment
//     registers are dead there) instead of being carried in registers across an MFMA block;
//   * A fragments are software-pipelined per 6-MFMA group across taps (2 x 2 register sets);
//   * the epilogue goes through LDS in two 128-row halves.
#include <algorithm>
#include <cstdlib>

#include "device_utils.h"

namespace spdm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr float ACT_SCALE = 16.0f;            // same scales as conv_gemm.hip (2^4 activations, 2^7 weights)
constexpr float DESCALE = 1.0f / 2048.0f;
constexpr int CK = 32;
constexpr int LDK = 36;

__device__ __forceinline__ f32x2 split2(float a, float b) {
    const float xa = a * ACT_SCALE, xb = b * ACT_SCALE;
    const _Float16 ha = (_Float16)xa, hb = (_Float16)xb;
    const f16x2 h = {ha, hb};
    const f16x2 l = {(_Float16)(xa - (float)ha), (_Float16)(xb - (float)hb)};
    return f32x2{__builtin_bit_cast(float, h), __builtin_bit_cast(float, l)};
}

// Sum over the 32 lanes of a half-wave with DPP adds only (no LDS crossbar): quad swaps, row half-mirror,
// row mirror, then lane 15 of rows 0 / 2 broadcast into rows 1 / 3.  The total is valid in lanes 16-31 and 48-63.
__device__ __forceinline__ float half_sum_dpp(float v) {
#define DPP_ADD(ctrl_, rmask_)                                                                                  \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl_, rmask_, 0xf, false));
    DPP_ADD(0xB1, 0xf)      // quad_perm [1,0,3,2]
    DPP_ADD(0x4E, 0xf)      // quad_perm [2,3,0,1]
    DPP_ADD(0x141, 0xf)     // row_half_mirror
    DPP_ADD(0x140, 0xf)     // row_mirror
    DPP_ADD(0x142, 0xa)     // row_bcast:15 into rows 1 and 3
#undef DPP_ADD
    return v;
}

template <int NT, int PRO, bool W2>
__global__ __launch_bounds__(256, 2) void conv3x3_wide_kernel(const GemmArgs a, const int epi_slots, const int NS, const int stagger) {
    constexpr int WM = 2, WN = 2, MT = 4;               // NT = 2: 256 x 128 tile (wave 128 x 64); NT = 1: 256 x 64 (wave 128 x 32)
    constexpr int NTHR = WM * WN * 64;
    constexpr int RP = NTHR / 8;
    constexpr int M_T = WM * MT * 32, N_T = WN * NT * 32;
    constexpr int APASS = (M_T + 18 + RP - 1) / RP;
    static_assert(APASS <= 12, "sample-index packing: 5 bits per pass in 64");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, kh = lane >> 5;
    const int HW = a.HW, W = a.W, H = a.H, M = a.M, K = a.K, N = a.N;
    const int halo = W + 1;
    const int QA = M_T + 2 * halo;
    const int QZ = QA + 2;                    // + the all-zero row (masked taps read it) + a dump row

    // ---- tile of this workgroup (XCD-aware, bijective remap: the n-tiles of an m-tile share an XCD's L2) ----
    const int n_ntiles = N / N_T;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int mtile = logical / n_ntiles, ntile = logical - mtile * n_ntiles;
    const int m0 = mtile * M_T, n0 = ntile * N_T;

    float* Abuf = smem;                       // [QZ][LDK]
    float* smean = Abuf + QZ * LDK;           // [NS]
    float* srstd = smean + NS;                // [NS]

    // diagnostic builds: per-workgroup timeline {memrealtime, memtime x5, HW_ID, XCC_ID} (tools/bench_gemm.py --stamp)
#ifdef SPDM_DIAG
    const bool stamping = (a.debug & DBG_STAMP) && a.stamps != nullptr && tid == 0;
#define WIDE_STAMP(k_) if (stamping) a.stamps[(size_t)bid * 8 + (k_)] = (unsigned long long)__builtin_amdgcn_s_memtime();
    if (stamping) {
        a.stamps[(size_t)bid * 8 + 0] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
        a.stamps[(size_t)bid * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg(63492);   // HW_ID
        a.stamps[(size_t)bid * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg(63508);   // XCC_ID
    }
    const bool fine = (a.debug & DBG_STAMP) && a.stamps != nullptr && tid == 0 && bid == 700;
    int nfine = 0;
#define WIDE_FINE() if (fine && nfine < 120) a.stamps[(size_t)40000 * 8 + nfine++] = (unsigned long long)__builtin_amdgcn_s_memtime();
    const bool dbg_no_mfma = (a.debug & DBG_NO_MFMA) != 0, dbg_no_wload = (a.debug & DBG_NO_WLOAD) != 0;
#else
#define WIDE_STAMP(k_)
#define WIDE_FINE()
    constexpr bool dbg_no_mfma = false, dbg_no_wload = false;     // ablation knobs exist in diagnostic builds only
#endif
    WIDE_STAMP(1)
    // Two workgroups share a CU.  Dispatched together, they would run in lock-step -- both in their MFMA loops,
    // then both in their store tails with the matrix pipe idle.  The workgroup that landed in the CU's second
    // wave slot therefore starts `stagger` cycles late, once (first generation only); from then on one
    // workgroup's prologue / store tail overlaps the other's MFMA loop.  Purely a scheduling hint: results do
    // not depend on it.
    if (stagger > 0 && bid < 2 * 256 && (__builtin_amdgcn_s_getreg(63492) & 1u)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)stagger) __builtin_amdgcn_s_sleep(64);
    }
    if (tid < LDK) Abuf[QA * LDK + tid] = 0.f;

    constexpr bool pro = (PRO != PRO_NONE);
    constexpr bool pro_gelu = (PRO == PRO_GN_GELU);
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

    // ---- staging assignment: 8 threads x 16 bytes cover one 32-channel row; per pass p the thread's row is
    //      q = p RP + (tid >> 3).  Validity and the sample index of each row are packed into two scalars. ----
    const int srow_t = tid >> 3, c4 = tid & 7;
    unsigned avalid = 0u;
    unsigned long long abidx = 0ull;
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
        const int q = p * RP + srow_t;
        const int m = m0 - halo + q;
        const bool v = (q < QA) && (m >= 0) && (m < M);
        if (v) {
            avalid |= 1u << p;
            if (pro) abidx |= (unsigned long long)(m / HW - bh_first) << (5 * p);
        }
    }
    const float* abase = a.src + c4 * 4;

    // ---- per-lane fragment rows and tap masks ----
    int aoff[MT];
    unsigned amask[MT];
    const int koff = kh * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        // W2 (image width 2): the wave's rows are permuted so that even tiles hold the w = 0 positions and odd tiles
        // the w = 1 positions; a side column of the kernel then only concerns the tiles of one parity (below)
        const int r = W2 ? wm * MT * 32 + (mt >> 1) * 64 + 2 * li + (mt & 1) : wm * MT * 32 + mt * 32 + li;
        aoff[mt] = (r + halo) * LDK + koff;
        unsigned mask = 0u;
        const int m = m0 + r;
        if (m < M) {
            const int p = m % HW;
            const int h = p / W, w = p - h * W;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dh = t / 3 - 1, dw = t % 3 - 1;
                const bool ok = (h + dh >= 0) && (h + dh < H) && (w + dw >= 0) && (w + dw < W);
                mask |= (ok ? 1u : 0u) << t;
            }
        }
        amask[mt] = mask;
    }
    const int zoff = QA * LDK + koff;
    // B operands: fragment-order weights, block ((tap nchunks + chunk) N/32 + nb) x {s2} x {hi, lo} of 256 floats
    const int nchunks = K / CK;
    const float* wfl = a.wgt_frag + lane * 4;
    const int nb0 = (n0 >> 5) + wn * NT;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    f32x4 areg[APASS];
    f32x4 g4r = {1.f, 1.f, 1.f, 1.f}, b4r = {0.f, 0.f, 0.f, 0.f};

#define WIDE_LOAD_A(chunk_)                                                                          \
    {                                                                                                \
        _Pragma("unroll") for (int p_ = 0; p_ < APASS; ++p_) {                                      \
            const int mc_ = min(max(m0 - halo + p_ * RP + srow_t, 0), M - 1);                        \
            areg[p_] = *reinterpret_cast<const f32x4*>(abase + (size_t)mc_ * a.src_ld + (chunk_) * CK); \
        }                                                                                            \
        if (pro) {                                                                                   \
            g4r = *reinterpret_cast<const f32x4*>(a.pro_gamma + (chunk_) * CK + c4 * 4);             \
            b4r = *reinterpret_cast<const f32x4*>(a.pro_beta + (chunk_) * CK + c4 * 4);              \
        }                                                                                            \
    }
#define WIDE_STAGE_A()                                                                               \
    {                                                                                                \
        _Pragma("unroll") for (int p_ = 0; p_ < APASS; ++p_) {                                      \
            f32x4 v_ = areg[p_];                                                                     \
            if (pro) {                                                                               \
                const int bi_ = (int)((abidx >> (5 * p_)) & 31ull);                                  \
                const float rs_ = srstd[bi_], mu_ = smean[bi_];                                      \
                v_.x = (v_.x - mu_) * (rs_ * g4r.x) + b4r.x;                                         \
                v_.y = (v_.y - mu_) * (rs_ * g4r.y) + b4r.y;                                         \
                v_.z = (v_.z - mu_) * (rs_ * g4r.z) + b4r.z;                                         \
                v_.w = (v_.w - mu_) * (rs_ * g4r.w) + b4r.w;                                         \
                if (pro_gelu) {                                                                      \
                    v_.x = gelu_erf(v_.x); v_.y = gelu_erf(v_.y);                                    \
                    v_.z = gelu_erf(v_.z); v_.w = gelu_erf(v_.w);                                    \
                }                                                                                    \
            }                                                                                        \
            if (!((avalid >> p_) & 1u)) v_ = f32x4{0.f, 0.f, 0.f, 0.f};                              \
            const f32x2 p0_ = split2(v_.x, v_.y), p1_ = split2(v_.z, v_.w);                          \
            {   /* rows past the slab go to a dump row: no branch, so no conditional vmcnt wait that the   \
                   compiler would have to repeat (as vmcnt(0), behind the W loads) at the loop head */      \
                float* row_ = Abuf + min(p_ * RP + srow_t, QA + 1) * LDK;                            \
                *reinterpret_cast<f32x2*>(row_ + c4 * 2) = f32x2{p0_.x, p1_.x};       /* hi */       \
                *reinterpret_cast<f32x2*>(row_ + 16 + c4 * 2) = f32x2{p0_.y, p1_.y};  /* lo */       \
            }                                                                                        \
            if (pro) __builtin_amdgcn_sched_barrier(0);        /* one pass at a time: register pressure */ \
        }                                                                                            \
    }
#define WIDE_LOAD_B(set_, chunk_, tap_, s2_)                                                          \
    if (!dbg_no_wload) {                                                                             \
        const float* p_ = wfl + ((size_t)(((tap_) * nchunks + (chunk_)) * (N >> 5) + nb0) * 4 + (s2_) * 2) * 256; \
        _Pragma("unroll") for (int nt_ = 0; nt_ < NT; ++nt_) {                                      \
            fb[set_][nt_][0] = *reinterpret_cast<const f16x8*>(p_ + nt_ * 1024);                     \
            fb[set_][nt_][1] = *reinterpret_cast<const f16x8*>(p_ + nt_ * 1024 + 256);               \
        }                                                                                            \
    }
#define WIDE_LOAD_FA(set_, ao_, s2_)                                                                 \
    {                                                                                                \
        fa[set_][0] = *reinterpret_cast<const f16x8*>(Abuf + (ao_) + (s2_) * 8);                     \
        fa[set_][1] = *reinterpret_cast<const f16x8*>(Abuf + (ao_) + 16 + (s2_) * 8);                \
    }
#define WIDE_TAP_OFFSETS(dst_, tap_)                                                                 \
    {                                                                                                \
        const int dh_ = (tap_) / 3 - 1, dw_ = (tap_) - ((tap_) / 3) * 3 - 1;                         \
        const int shift_ = (dh_ * W + dw_) * LDK;                                                    \
        _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_)                                        \
            dst_[mt_] = ((amask[mt_] >> (tap_)) & 1u) ? aoff[mt_] + shift_ : zoff;                   \
    }
#define WIDE_GROUP(set_, fbs_, mt_)                                                                  \
    if (!dbg_no_mfma) {                                                                              \
        _Pragma("unroll") for (int nt_ = 0; nt_ < NT; ++nt_) {                                      \
            acc[mt_][nt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set_][0], fb[fbs_][nt_][0], acc[mt_][nt_], 0, 0, 0); \
            acc[mt_][nt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set_][0], fb[fbs_][nt_][1], acc[mt_][nt_], 0, 0, 0); \
            acc[mt_][nt_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set_][1], fb[fbs_][nt_][0], acc[mt_][nt_], 0, 0, 0); \
        }                                                                                            \
    }

    f16x8 fa[2][2], fb[2][NT][2];
    if constexpr (W2) {
        // Image width 2: for w = 0 the dw = -1 column of the kernel only sees zero padding, for w = 1 the dw = +1
        // column.  With the rows permuted by parity (above), a kernel row costs: the centre tap on all four tiles,
        // the dw = +1 tap on the even tiles, the dw = -1 tap on the odd tiles -- 16 tile steps instead of 24, only
        // all-zero products dropped.  Loop body = one kernel row = six K=16 half-taps (centre, +1, -1) x (s2 0, 1).
        const int nrows = nchunks * 3;
        int aoc[MT], aos[MT];
#define W2_ROW_OFFSETS(kr_)                                                                          \
        {                                                                                            \
            const int sh_ = ((kr_) - 1) * W * LDK;                                                   \
            _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) {                                  \
                aoc[mt_] = ((amask[mt_] >> ((kr_) * 3 + 1)) & 1u) ? aoff[mt_] + sh_ : zoff;          \
                const int ts_ = (mt_ & 1) ? (kr_) * 3 : (kr_) * 3 + 2;                               \
                aos[mt_] = ((amask[mt_] >> ts_) & 1u) ? aoff[mt_] + sh_ + ((mt_ & 1) ? -LDK : LDK) : zoff; \
            }                                                                                        \
        }
        WIDE_LOAD_B(0, 0, 1, 0)
        WIDE_LOAD_A(0)
        WIDE_STAGE_A()
        __syncthreads();
        WIDE_STAMP(2)
        W2_ROW_OFFSETS(0)
        WIDE_LOAD_FA(0, aoc[0], 0)
        int chunk = 0, kr = 0;
        for (int r = 0; r < nrows; ++r) {
            int nkr = kr + 1, nchunk = chunk;
            if (nkr == 3) { nkr = 0; nchunk = chunk + 1; }
            const bool have_next = (r + 1 < nrows);
            const bool next_A = have_next && (nkr == 0);
            const int bkr = have_next ? nkr : kr, bchunk = have_next ? nchunk : chunk;   // B prefetch past this row (last row: re-read)
            const int tc = kr * 3 + 1, tp = kr * 3 + 2, tm = kr * 3;
            W2_ROW_OFFSETS(kr)
            const int aoc_n0 = ((amask[0] >> (nkr * 3 + 1)) & 1u) ? aoff[0] + (nkr - 1) * W * LDK : zoff;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = q < 8 ? q / 4 : 2 + (q - 8) / 2;                          // half-tap 0..5
                const int mt = q < 8 ? q % 4 : (k < 4 ? 0 : 1) + 2 * ((q - 8) % 2);     // tile of this step
                if (q == 0) { WIDE_LOAD_B(1, chunk, tc, 1) }
                if (q == 4) { WIDE_LOAD_B(0, chunk, tp, 0) }
                if (q == 8) { WIDE_LOAD_B(1, chunk, tp, 1) }
                if (q == 10) { WIDE_LOAD_B(0, chunk, tm, 0) }
                if (q == 12) { WIDE_LOAD_B(1, chunk, tm, 1) }
                if (q == 14) { WIDE_LOAD_B(0, bchunk, bkr * 3 + 1, 0) }
                if (q + 1 < 16) {
                    const int qn = q + 1;
                    const int kn = qn < 8 ? qn / 4 : 2 + (qn - 8) / 2;
                    const int mtn = qn < 8 ? qn % 4 : (kn < 4 ? 0 : 1) + 2 * ((qn - 8) % 2);
                    WIDE_LOAD_FA(qn & 1, (kn < 2 ? aoc[mtn] : aos[mtn]), kn & 1)
                } else if (have_next && !next_A) {
                    WIDE_LOAD_FA(0, aoc_n0, 0)
                }
                __builtin_amdgcn_sched_barrier(0);
                WIDE_GROUP(q & 1, k & 1, mt)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (next_A) {
                WIDE_LOAD_A(nchunk)
                __syncthreads();
                WIDE_STAGE_A()
                __syncthreads();
                WIDE_LOAD_FA(0, aoc_n0, 0)
            }
            kr = nkr;
            chunk = nchunk;
        }
#undef W2_ROW_OFFSETS
    } else {
    const int niter = nchunks * 9;

    int ao[MT], aon[MT];
    WIDE_LOAD_B(0, 0, 0, 0)
    WIDE_LOAD_A(0)
    WIDE_STAGE_A()
    __syncthreads();
    WIDE_STAMP(2)
    WIDE_TAP_OFFSETS(ao, 0)
    WIDE_LOAD_FA(0, ao[0], 0)

    int chunk = 0, tap = 0;
    for (int it = 0; it < niter; ++it) {
        int ntap = tap + 1, nchunk = chunk;
        if (ntap == 9) { ntap = 0; nchunk = chunk + 1; }
        const bool have_next = (it + 1 < niter);
        const bool next_A = have_next && (ntap == 0);
        WIDE_FINE()
        // ---- K = 16 half-tap 0 (B set 0; the loads of set 1 fly meanwhile) ----
        WIDE_LOAD_B(1, chunk, tap, 1)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt < MT - 1) WIDE_LOAD_FA((mt + 1) & 1, ao[mt + 1], 0) else WIDE_LOAD_FA(0, ao[0], 1)
            __builtin_amdgcn_sched_barrier(0);
            WIDE_GROUP(mt & 1, 0, mt)
            __builtin_amdgcn_sched_barrier(0);
        }
        WIDE_FINE()
        // ---- half-tap 1 (B set 1; set 0 is refilled for the next tap) ----
        // unconditional (the last iteration re-reads its own block): a branch here would make the compiler wait
        // with vmcnt(0), i.e. for these loads too, before the MFMAs of this half-tap
        WIDE_LOAD_B(0, (have_next ? nchunk : chunk), (have_next ? ntap : tap), 0)
        WIDE_TAP_OFFSETS(aon, ntap)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt < MT - 1) WIDE_LOAD_FA((mt + 1) & 1, ao[mt + 1], 1) else if (have_next && !next_A) WIDE_LOAD_FA(0, aon[0], 0)
            __builtin_amdgcn_sched_barrier(0);
            WIDE_GROUP(mt & 1, 1, mt)
            __builtin_amdgcn_sched_barrier(0);
        }
        WIDE_FINE()
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ao[mt] = aon[mt];
        if (next_A) {
            WIDE_LOAD_A(nchunk)             // in flight across the barrier
            __syncthreads();                // every wave is done reading the slab of this chunk
            WIDE_STAGE_A()
            __syncthreads();
            WIDE_LOAD_FA(0, ao[0], 0)
        }
        WIDE_FINE()
        tap = ntap;
        chunk = nchunk;
    }
    }
#undef WIDE_LOAD_A
#undef WIDE_STAGE_A
#undef WIDE_LOAD_B
#undef WIDE_LOAD_FA
#undef WIDE_TAP_OFFSETS
#undef WIDE_GROUP
    WIDE_STAMP(3)
    __syncthreads();          // the epilogue re-uses the slab's LDS: every wave must be done reading A fragments

#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= DESCALE;

    // ---- epilogue: GroupNorm partial sums (fp32 per 4-row unit -> fp64 per sample, fixed order), then the
    //      tile through LDS in two halves so that every lane stores 16 bytes ----
    constexpr int HROWS = WM * 64;                       // rows per half
    float* otile = smem;                                 // [HROWS][N_T]
    float* srow = smem + HROWS * N_T;                    // [M_T / 4][WN][2]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // 4-row unit of this register quad (W2: the quad's rows are 2 apart inside one 8-row block -> slot unit
            // 2 (block) + parity; a sample is still a contiguous run of HW / 4 slot units because HW % 8 == 0)
            const int r0 = W2 ? 4 * (wm * 32 + (mt >> 1) * 16 + 4 * g + 2 * kh + (mt & 1)) : wm * MT * 32 + mt * 32 + 8 * g + 4 * kh;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = acc[mt][nt][4 * g + j];
                    s1 += v;
                    s2 += v * v;
                }
            s1 = half_sum_dpp(s1);
            s2 = half_sum_dpp(s2);
            if (li == 31) {
                srow[((r0 >> 2) * WN + wn) * 2] = s1;
                srow[((r0 >> 2) * WN + wn) * 2 + 1] = s2;
            }
        }

#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();
#pragma unroll
        for (int mq = 0; mq < 2; ++mq)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col_l = wn * NT * 32 + nt * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ri = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int row_l = W2 ? wm * 64 + 2 * ri + mq : wm * 64 + mq * 32 + ri;
                    otile[row_l * N_T + col_l] = acc[2 * h + mq][nt][r];
                }
            }
        __syncthreads();
        if (h == 0) WIDE_STAMP(4)
        if (h == 0) {
            const int t_lo = m0, t_hi = min(m0 + M_T, M);
            const int ups = HW >> 2;                                   // 4-row units per sample
            const bool whole = (HW % M_T == 0);                        // the tile lies inside one sample
            const bool aligned = whole || (M_T % HW == 0 && (ups & (ups - 1)) == 0);
            if (M_T == 256 && aligned) {
                // one lane per unit, fp64 butterfly inside each sample's (aligned, power-of-two) lane segment:
                // fixed order, position-independent -> deterministic and identical for every sample
                if (wave == 0) {
                    const int u = lane;
                    double d1 = (double)srow[(u * WN) * 2] + (double)srow[(u * WN + 1) * 2];
                    double d2 = (double)srow[(u * WN) * 2 + 1] + (double)srow[(u * WN + 1) * 2 + 1];
                    const int seg = whole ? 64 : ups;
                    for (int o = 1; o < seg; o <<= 1) {
                        d1 += __shfl_xor(d1, o, 64);
                        d2 += __shfl_xor(d2, o, 64);
                    }
                    const int row = t_lo + 4 * u;
                    if ((u & (seg - 1)) == 0 && row < t_hi) {
                        const int b = row / HW;
                        const int slot = (mtile - (b * HW) / M_T) * n_ntiles + ntile;
                        double* o = a.epi_stats + ((size_t)b * epi_slots + slot) * 2;
                        o[0] = d1;
                        o[1] = d2;
                    }
                }
            } else if (t_hi > t_lo) {
                const int b_first = t_lo / HW, b_last = (t_hi - 1) / HW;
                for (int t = tid; t <= b_last - b_first; t += NTHR) {
                    const int b = b_first + t;
                    const int r_lo = max(b * HW, t_lo) - m0, r_hi = min((b + 1) * HW, t_hi) - m0;
                    double s1 = 0.0, s2 = 0.0;
                    for (int u = r_lo >> 2; u < ((r_hi + 3) >> 2); ++u)
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
        constexpr int TPR = N_T / 4, RPP = NTHR / TPR;
        const int c4o = tid % TPR, r0 = tid / TPR;
#pragma unroll 4
        for (int p = 0; p < HROWS / RPP; ++p) {
            const int lr = p * RPP + r0;
            const int row = m0 + (lr >> 6) * (MT * 32) + h * 64 + (lr & 63);
            if (row < M) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(otile + lr * N_T + c4o * 4);
                *reinterpret_cast<f32x4*>(a.dst + (size_t)row * a.dst_ld + n0 + c4o * 4) = v;
            }
        }
    }
    WIDE_STAMP(5)
#undef WIDE_STAMP
#undef WIDE_FINE
}

template <int NT, int PRO, bool W2 = false>
hipError_t launch_wide_cfg(const GemmArgs& a, const GemmGeom& g, hipStream_t s) {
    constexpr int M_T = 256, N_T = 64 * NT, NTHR = 256;
    const int halo = a.W + 1;
    const int QA = M_T + 2 * halo;
    const int NS = (((QA - 1) / a.HW + 2) + 3) & ~3;
    if (NS > 32 || g.m_tile != M_T || g.n_tile != N_T) return hipErrorInvalidValue;
    size_t lds = (size_t)((QA + 2) * LDK + 2 * NS) * sizeof(float);
    lds = std::max(lds, (size_t)(128 * N_T + (M_T / 4) * 2 * 2) * sizeof(float));
    if (const char* pad = getenv("SPDM_WIDE_LDSPAD")) lds += (size_t)atoi(pad);      // experiment: force one workgroup per CU
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = conv3x3_wide_kernel<NT, PRO, W2>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int n_mtiles = (a.M + M_T - 1) / M_T;
        static const int stagger = getenv("SPDM_WIDE_STAGGER") ? atoi(getenv("SPDM_WIDE_STAGGER")) : 0;
    hipLaunchKernelGGL(kern, dim3(n_mtiles * g.n_tiles), dim3(NTHR), lds, s, a, g.slots, NS, stagger);
    return hipGetLastError();
}

}  // namespace

static bool wide_w2(const GemmArgs& a) { return a.W == 2 && getenv("SPDM_NO_W2") == nullptr; }

bool conv_wide_supported(const GemmArgs& a, const GemmGeom& g) {
    const bool ok = a.split && a.wgt_frag != nullptr && a.taps == 9 && g.m_tile == 256 && (g.n_tile == 128 || g.n_tile == 64) &&
                    a.W >= 1 && a.W <= 8 && (a.HW & 3) == 0 && a.M % a.HW == 0 && a.epi == EPI_STATS && a.row_stats == nullptr &&
                    (a.debug & ~(DBG_STAMP | DBG_NO_MFMA | DBG_NO_WLOAD)) == 0 && a.K % CK == 0 &&
                    (256 + 2 * (a.W + 1) - 1) / a.HW + 2 <= 32 && getenv("SPDM_NO_WIDE") == nullptr;
    if (!ok) return false;
    // width-2 maps: only with the zero-tap skipping variant (otherwise conv_gemm.hip's W2 configuration does less work)
    if (wide_w2(a)) return g.n_tile == 128 && (a.HW & 7) == 0;
    return true;
}

hipError_t launch_conv_wide(const GemmArgs& a, const GemmGeom& g, hipStream_t s) {
    if (!conv_wide_supported(a, g)) return hipErrorInvalidValue;
    if (wide_w2(a)) {
        if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, true>(a, g, s);
        if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN, true>(a, g, s);
        return launch_wide_cfg<2, PRO_GN_GELU, true>(a, g, s);
    }
    if (g.n_tile == 128) {
        if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE>(a, g, s);
        if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN>(a, g, s);
        return launch_wide_cfg<2, PRO_GN_GELU>(a, g, s);
    }
    if (a.pro == PRO_NONE) return launch_wide_cfg<1, PRO_NONE>(a, g, s);
    if (a.pro == PRO_GN) return launch_wide_cfg<1, PRO_GN>(a, g, s);
    return launch_wide_cfg<1, PRO_GN_GELU>(a, g, s);
}

}  // namespace spdm
