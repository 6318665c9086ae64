// conv_wide.hip -- 3x3 convolution as an implicit GEMM: the kernel for every 3x3 layer of levels 0-2 at large
// batch (256-row output tiles; 80 % of the path's FLOPs).
//
// Same math, operand formats and LDS slab layout as conv_gemm.hip (split-fp16 operands, hi*hi + hi*lo + lo*hi into
// one fp32 accumulator, halo'd input slab with the GroupNorm(1,C) -> GELU prologue applied at staging, GroupNorm
// partial sums as the epilogue; replaces nn.Conv2d(k=3, padding=1, bias=False) + the GroupNorm/GELU around it,
// models/Unet_FiLmLayer.py:101-115).  What differs is how the work is laid on the CU -- each point below answers a
// measurement of the first kernel (PMC counters and in-kernel s_memtime timelines):
//
//   * v_mfma_f32_16x16x32_f16 instead of 32x32x16: same cycles per FLOP, but this path is clock-throttled by its own
//     MFMA load and the 16x16x32 shape holds a ~10 % higher clock on live data (measured in place: -9..-18 % time);
//     one instruction covers the whole 32-channel chunk of a tap;
//   * one workgroup = 4 waves (one per SIMD), each owning a 128 x 64 (NT = 2) or 128 x 32 (NT = 1) accumulator
//     strip (8 x 4 or 8 x 2 tiles of 16 x 16 = 128 / 64 accumulator registers); <= 256 registers and <= 77 KiB of
//     LDS, so two workgroups share a CU;
//   * the WEIGHT fragments never touch LDS: the host stores a fragment-order copy of the split weights
//     (frag_order_weights, kernels.h: one 1-KiB block per MFMA B operand, lane-contiguous) and each wave loads its B
//     operands straight into registers with coalesced global_load_dwordx4, one phase (48 MFMAs) ahead.  The waves of
//     a workgroup then only meet at the slab hand-over -- one barrier pair per 32-channel chunk instead of one
//     barrier per tap -- and the MFMA stream of a chunk is unbroken;
//   * a "phase" is (tap, pair of 16-column tiles): 8 row tiles x 2 column tiles x 3 MFMAs.  A fragments live in a
//     ring of three (four for W2) row tiles, read two tile steps ahead of use, across taps; B fragments in a ring of
//     two (NT = 2) or three (NT = 1) phases.  The loop body is one kernel ROW, so every ring slot is a compile-time
//     register set;
//   * every load in the MFMA loop is unconditional (clamped addresses, a dump row in LDS): a branch around a load
//     makes hipcc's waitcnt pass repeat the wait as vmcnt(0) at the next use, i.e. behind the weight loads;
//   * width-2 maps (W2): rows are permuted by parity so that even row tiles hold w = 0 and odd ones w = 1 positions;
//     a side column of the kernel then only concerns the tiles of one parity (32 instead of 48 tile steps per row);
//   * width-4 maps (WP, round 3): the same idea one level up -- row tile rt of a wave holds the positions w = rt % 4 of a 64-row
//     block (rows 4 l16 + w), so the left kernel column is skipped on the w = 0 tiles and the right one on the w = 3 tiles:
//     20 instead of 24 tile steps per kernel row, only all-zero products dropped;
//   * epilogue: 16-lane reductions by DPP adds, per-sample fp64 totals by a one-wave butterfly, tile through LDS in
//     two 128-row halves so that every lane stores 16 bytes.
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
// LDS bank conflicts of the A-fragment reads.  A ds_read_b128 is serviced in four groups of 16 lanes that are NOT the four
// kg groups: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, ...}, {36-43, ...} (MI355X_MICROARCH.md, LDS) -- each holds
// the rows l16 in {0-3, 12-15} of one kg and the rows {4-11} of the next.  With the 144-byte row pitch (conflict-free for
// 32 consecutive rows of ONE kg) seven of those eight row pairs share a bank: every A-fragment read took two LDS cycles
// (PMC round 1: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.52-0.57).  No pitch fixes that for rows in natural order (the
// rows {4-11} would have to be a set invariant under a shift).  WIDE_LDS_PERM: pitch 160 bytes (rows r and r + 8 share a
// 16-byte bank slot, even slots for one kg, odd for the next) and lane l16 reads tile row rho(l16) = {0-3 -> 0-3, 4-11 -> 8-15,
// 12-15 -> 4-7}, so each hardware group sees rows 0-7 of one kg and rows 8-15 of the next: conflict-free.  The accumulator
// rows follow: lane group kg holds tile rows {0, 8, 12, 4}[kg] + j.  (Width-2 maps read every other row: they keep the old pitch.)
#ifndef WIDE_LDS_PERM
#define WIDE_LDS_PERM 1
#endif
template <bool W2, bool PIPE = false, bool WP = false> struct WidePitch { static constexpr int value = (W2 || PIPE || WP || !WIDE_LDS_PERM) ? 36 : 40; };
// WP (width-4 maps): compile-time schedule of the row tiles a tap visits.  Tap t = 3 (dh + 1) + (dw + 1); the tiles of class
// w = rt % 4 == 0 never see a valid dw = -1 tap, those of class 3 never a valid dw = +1 tap.
constexpr bool wp_active(int t, int rt) { return !((t % 3 == 0 && (rt & 3) == 0) || (t % 3 == 2 && (rt & 3) == 3)); }
constexpr int wp_first(int t) { return (t % 3 == 0) ? 1 : 0; }
constexpr int wp_next(int t, int rt, int RT) {        // the next tile tap t visits after rt, or -1
    for (int r = rt + 1; r < RT; ++r)
        if (wp_active(t, r)) return r;
    return -1;
}
constexpr int wp_order(int t, int rt, int RT) {       // position of (t, rt) in the chunk's sequence of tile steps
    int n = 0;
    for (int tt = 0; tt < t; ++tt)
        for (int r = 0; r < RT; ++r) n += wp_active(tt, r) ? 1 : 0;
    for (int r = 0; r < rt; ++r) n += wp_active(t, r) ? 1 : 0;
    return n;
}
// (PIPE keeps two slabs: 2 x 276 rows only fit two workgroups per CU at the 144-byte pitch, and the conflict is harmless)
#ifndef WIDE_STAGE_GROUP
#define WIDE_STAGE_GROUP 1
#endif
// Hand-over experiments of the tap-pair loop (measured at B = 4096, tools/bench_gemm.py, alternating builds on one box):
//   WIDE_DB    two slab buffers, the next chunk staged into the idle one, ONE barrier per hand-over: no gain (923 -> 939-949 us
//              on up3.dc1a) -- the first barrier was never the cost;
//   WIDE_EARLY its global loads issued a tap before the hand-over: the 36 staging registers then live through a tap of the
//              256-register loop -> 180-212 bytes of scratch per lane, 923 -> 1088 us;
//   (a third variant -- one lane per slab row "touching" the next chunk's row a tap early through an inline-asm load into a
//   dead register, to turn the hand-over's HBM misses into L2 hits -- is unsound: the compiler re-uses the destination register
//   while the load is still in flight; it faulted on the GPU and was removed.  An LDS-DMA prefetch of the raw slab is the
//   register-free way to do this.)
#ifndef WIDE_DB
#define WIDE_DB 0
#endif
#ifndef WIDE_EARLY
#define WIDE_EARLY 0
#endif
#ifndef WIDE_NH
#define WIDE_NH 2          // parts the output tile leaves in (epilogue LDS = M_T / WIDE_NH rows).  4 -> 40 KB per workgroup, i.e.
                           // FOUR workgroups per CU instead of two: measured 0.06 ms per step SLOWER (9.72 / 9.53 vs 9.66 / 9.47 ms,
                           // same box, alternating) -- the matrix pipe is clock-limited, more residents only add contention
#endif

__device__ __forceinline__ f32x2 split2(float a, float b) {
    unsigned h, l;
    split_pair_f16(a * ACT_SCALE, b * ACT_SCALE, h, l);
    return f32x2{__builtin_bit_cast(float, h), __builtin_bit_cast(float, l)};
}

// TWO: two-source input (the first convolution of an UpSample block, models/Unet_FiLmLayer.py:217-219): the 32-channel chunks
// [0, up_C / 32) come from a.src -- the upsampled tensor, finished -- and the rest from a.skip, the skip connection, which is
// what the prologue PRO (its pending GroupNorm) applies to.  torch.cat is then never materialised: upcat_kernel shrinks to the
// upsample alone (no copy of the skip half: at up3 that copy was 268 MB in + 268 MB out per step at B = 4096).
// G2 (experiment, opt-in SPDM_G2=1; see launch_conv_wide for the measurement): two 32-channel chunks per slab hand-over on the
// 64-row-per-wave variants of the tap loop -- the loads of both chunks in flight together, one round trip and one barrier pair
// per 64 channels.  Neutral to slower: the round trip is not what the hand-over costs.
template <int NT, int PRO, bool W2, int RT, int WN, bool PIPE, bool TWO = false, int WP = 0, bool G2 = false>
// launch bound (256, 2): a 256-register budget.  The 8 x 2 variants with a prologue then carry 88-128 bytes of scratch per lane
// (stats finalisation state parked across the main loop); with (256, 1) the compiler allocates 203-250 registers and no scratch,
// but schedules the loop worse: measured 10.5 vs 9.6-9.9 ms per step (same box, alternating).  Keep 2.
#ifndef WIDE_MINB
#define WIDE_MINB 2
#endif
__global__ __launch_bounds__(256, WIDE_MINB) void conv3x3_wide_kernel(const GemmArgs a, const int epi_slots, const int NS) {
    constexpr int WM = 4 / WN;                          // 4 waves: 2 x 2, or 4 x 1 for 64-wide outputs (each wave 64 rows x 64 columns)
    constexpr int CT = 2 * NT;                          // RT 16-row x CT 16-column tiles per wave (RT = 8: 128-row strip;
                                                        // RT = 4: 64 rows -- 128-row workgroup tiles for small / coarse layers)
    constexpr int RW = RT * 16;                         // rows per wave
    constexpr int NTHR = 256;
    constexpr int RP = NTHR / 8;
    constexpr int M_T = WM * RW, N_T = WN * NT * 32;
    constexpr int APASS = (M_T + 18 + RP - 1) / RP;
    constexpr int FAR = (W2 || WP) ? 4 : 3;             // A-fragment ring (row tiles)
    constexpr int FBR = (NT == 2) ? 2 : 3;              // B-fragment ring (phases)
    constexpr int PH = 3 * NT;                          // phases per kernel row (generic layout)
    constexpr int BB = (APASS <= 9) ? 7 : (APASS <= 10) ? 6 : 5;   // bits per staging pass of the packed sample index
    static_assert(APASS * BB <= 64, "sample-index packing");
    static_assert(!W2 || NT == 2, "the width-2 variant exists for 128-wide tiles");
    static_assert(RT == 8 || (RT == 4 && NT == 2 && !W2), "64-row waves only in the tap-pair loop");
    static_assert(WN == 2 || (WN == 1 && RT == 4), "the 4 x 1 wave arrangement uses 64-row waves");
    static_assert(!TWO || (!PIPE && PRO != PRO_GN_GELU), "two-source input: first conv of a block, plain hand-over");
    static_assert(WP == 0 || WP == 4 || WP == 8, "row permutation for width-4 or width-8 maps");
    static_assert(!WP || (NT == 2 && WN == 2 && !W2 && !PIPE), "row permutation: 128-wide tiles");
    static_assert((16 + 4 * (WP == 8 ? 7 : 6)) % 4 == 0, "A ring phase");
    static_assert(WP != 8 || RT == 8, "width-8 maps: one 128-row block per wave");
    constexpr bool WP8 = (WP == 8);
    // WP8: CLASS-MAJOR slab.  Slab row q (image position m0 - halo + q, halo = 9) has class c = (q + 7) % 8 = its w and index
    // i = (q + 7) / 8; it is stored at LDS row c NI8 + i.  A row tile (one class, 16 consecutive indices) then reads 16
    // CONSECUTIVE LDS rows, and a tap (dh, dw) is the uniform shift dw NI8 + dh -- in the natural layout the stride-8 fragment
    // read would be 4-way bank-conflicted at any 16-byte-aligned pitch and the LDS array would become the bound.
    constexpr int NI8 = (M_T + 18 + 6) / 8 + 1;
    static_assert(!G2 || (NT == 2 && RT == 4 && !W2 && !PIPE && !WP && !WIDE_DB), "two chunks per hand-over: the plain tap loop, 64 rows per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, l16 = lane & 15, kg = lane >> 4;
    const int HW = a.HW, W = a.W, H = a.H, M = a.M, K = a.K, N = a.N;
    const int halo = W + 1;
    const int QA = M_T + 2 * halo;
    const int QZ = WP8 ? 8 * NI8 + 2 : QA + 2;  // + the all-zero row (masked taps read it) + a dump row

    // ---- tile of this workgroup (XCD-aware, bijective remap: the n-tiles of an m-tile share an XCD's L2) ----
    const int n_ntiles = N / N_T;
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    // split-K: workgroup (tile, ks) walks the chunks [kc0, kc1) (host: K % 64 == 0)
    const int ksp = a.ksplit > 1 ? a.ksplit : 1;
    const int tiles_mn = nblk / ksp;
    const int ks = logical / tiles_mn, tl = logical - ks * tiles_mn;
    const int mtile = tl / n_ntiles, ntile = tl - mtile * n_ntiles;
    const int m0 = mtile * M_T, n0 = ntile * N_T;
    // even chunk ranges: the flat (chunk, tap) sequence of the tap-pair loop must have even length
    const int kc0 = 2 * (int)((long long)ks * (K / (2 * CK)) / ksp), kc1 = (ksp > 1) ? 2 * (int)((long long)(ks + 1) * (K / (2 * CK)) / ksp) : K / CK;

    static_assert(!PIPE || (NT == 2 && !W2), "the pipelined hand-over exists in the tap loop of 128-wide tiles");
    constexpr int LDK = WidePitch<W2, PIPE, (WP != 0) || G2>::value; // (shadows the namespace constant: every macro below uses it)
    constexpr bool PERM = WIDE_LDS_PERM && !W2 && !PIPE && !WP && !G2;
    constexpr bool DB = PIPE || (WIDE_DB && (NT == 2) && !W2);       // (the other loops keep the single slab)
    float* Abuf = smem;                       // [QZ][LDK]: the slab the MFMA loop reads
    float* Awr = smem + (DB ? QZ * LDK : 0);  // the slab being staged (DB: the idle one of two)
    float* smean = smem + ((DB || G2) ? 2 : 1) * QZ * LDK;     // [NS]  (G2: the slab holds two chunks)
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
    const bool dbg_no_mfma = (a.debug & DBG_NO_MFMA) != 0, dbg_no_wload = (a.debug & DBG_NO_WLOAD) != 0;
    const bool dbg_no_aload = (a.debug & DBG_NO_ALOAD) != 0;       // skip every slab hand-over (timing only: wrong results)
#else
#define WIDE_STAMP(k_)
    constexpr bool dbg_no_mfma = false, dbg_no_wload = false, dbg_no_aload = false;     // ablation knobs exist in diagnostic builds only
#endif
    WIDE_STAMP(1)
    const int zrow = WP8 ? 8 * NI8 : QA;        // the all-zero row; zrow + 1: the dump row
    if (tid < LDK) { Abuf[zrow * LDK + tid] = 0.f; Awr[zrow * LDK + tid] = 0.f; }
    if (G2 && tid < LDK) Abuf[(QZ + QA) * LDK + tid] = 0.f;         // the all-zero row of the second chunk's sub-slab

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
            if (pro) abidx |= (unsigned long long)(m / HW - bh_first) << (BB * p);
        }
    }
    const float* abase = a.src + c4 * 4;
    const float* abase1 = TWO ? a.skip + c4 * 4 : nullptr;        // chunks >= nc0 (two-source input)
    const int nc0 = TWO ? a.up_C / CK : 0;
    bool ident = false;                                           // the chunk being staged comes from a.src: no prologue on it

    // ---- A operand of row tile rt: lane (l16, kg) reads 16 bytes (8 fp16 of k = 8 kg ..) of slab row
    //      halo + wm 128 + ROWOFF(rt) + rowlane.  W2: rows permuted by parity (even tiles w = 0, odd tiles w = 1).
    //      The 9 tap-validity bits of each of the 8 tiles are packed three tiles to a register. ----
#define WIDE_ROWOFF(rt_) (W2 ? (((rt_) >> 1) * 32 + ((rt_) & 1)) : WP8 ? (rt_) : WP ? (((rt_) >> 2) * 64 + ((rt_) & 3)) : (rt_) * 16)
#define WIDE_LDSOFF(rt_) (WP8 ? (rt_) * NI8 : WIDE_ROWOFF(rt_))      /* LDS row offset of row tile rt_ from the lane's base row */
    const int rowlane = W2 ? 2 * l16 : WP ? WP * l16 : PERM ? (l16 < 4 ? l16 : l16 < 12 ? l16 + 4 : l16 - 8) : l16;
    const int aoff0 = WP8 ? (2 + wm * 16 + l16) * LDK + kg * 4 : (wm * RW + rowlane + halo) * LDK + kg * 4;
    const int zoff = zrow * LDK + kg * 4;
    unsigned am[(RT + 2) / 3] = {};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int m = m0 + wm * RW + WIDE_ROWOFF(rt) + rowlane;
        unsigned mask = 0u;
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
        am[rt / 3] |= mask << (9 * (rt % 3));
    }
    // B operand: fragment-order weights, block ((tap nchunks + chunk) N/16 + nb16) x {hi, lo} of 256 floats
    const int nchunks = K / CK;
    const float* wfl = a.wgt_frag + ((size_t)((n0 >> 4) + wn * CT) * 2) * 256 + lane * 4;
    const size_t wtap = (size_t)(N >> 4) * 2 * 256;          // floats per (tap, chunk)

    f32x4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 areg[APASS];
    f32x4 g4r = {1.f, 1.f, 1.f, 1.f}, b4r = {0.f, 0.f, 0.f, 0.f};

    // The per-pass row addresses are loop-invariant; hoisted out of the main loop they cost ~24 registers, which the
    // 128-row-per-wave variants with a prologue do not have: they came back as 16 scratch reloads per tap (88-128 bytes of
    // scratch per lane).  There an opaque copy of the thread's row makes the compiler recompute the addresses (a few
    // integer ops per pass) at each hand-over instead: no scratch, -3 % on up3.dc1.second / up1.dc1.second.  Everywhere
    // else the hoisted form measured 2-5 % FASTER (tools/bench_gemm.py, alternating builds), so it stays.
    int srow_o = srow_t;
#ifdef WIDE_NO_OPAQUE
#define WIDE_OPAQUE_ROW
#else
#define WIDE_OPAQUE_ROW if constexpr (PRO != PRO_NONE && RT == 8) asm volatile("" : "+v"(srow_o));
#endif
#define WIDE_LOAD_A(chunk_) WIDE_LOAD_A_(chunk_, areg, g4r, b4r, ident)
#define WIDE_STAGE_A() WIDE_STAGE_A_(areg, g4r, b4r, ident, Awr)
#define WIDE_LOAD_A_(chunk_, areg_, g4r_, b4r_, ident_)                                              \
    {                                                                                                \
        WIDE_OPAQUE_ROW                                                                              \
        const bool first_ = !TWO || (chunk_) < nc0;                                                  \
        const int cc_ = (TWO && !first_) ? (chunk_) - nc0 : (chunk_);      /* chunk inside its tensor */ \
        const float* ab_ = (first_ ? abase : abase1) + cc_ * CK;                                     \
        const int ld_ = first_ ? a.src_ld : a.skip_ld;                                               \
        _Pragma("unroll") for (int p_ = 0; p_ < APASS; ++p_) {                                      \
            const int mc_ = min(max(m0 - halo + p_ * RP + srow_o, 0), M - 1);                        \
            areg_[p_] = *reinterpret_cast<const f32x4*>(ab_ + (size_t)mc_ * ld_);                     \
        }                                                                                            \
        if (pro) {                                                                                   \
            ident_ = TWO && first_;                                                                   \
            g4r_ = *reinterpret_cast<const f32x4*>(a.pro_gamma + cc_ * CK + c4 * 4);                  \
            b4r_ = *reinterpret_cast<const f32x4*>(a.pro_beta + cc_ * CK + c4 * 4);                   \
            if (TWO && first_) { g4r_ = f32x4{1.f, 1.f, 1.f, 1.f}; b4r_ = f32x4{0.f, 0.f, 0.f, 0.f}; } \
        }                                                                                            \
    }
#define WIDE_STAGE_A_(areg_, g4r_, b4r_, ident_, dst_)                                                                             \
    {                                                                                                \
        _Pragma("unroll") for (int p_ = 0; p_ < APASS; ++p_) {                                      \
            f32x4 v_ = areg_[p_];                                                                     \
            if (pro) {                                                                               \
                const int bi_ = (int)((abidx >> (BB * p_)) & ((1ull << BB) - 1));                                  \
                float rs_ = srstd[bi_], mu_ = smean[bi_];                                            \
                if (TWO && ident_) { rs_ = 1.f; mu_ = 0.f; }       /* (v - 0) (1 x 1) + 0 == v, bit for bit */ \
                v_.x = (v_.x - mu_) * (rs_ * g4r_.x) + b4r_.x;                                         \
                v_.y = (v_.y - mu_) * (rs_ * g4r_.y) + b4r_.y;                                         \
                v_.z = (v_.z - mu_) * (rs_ * g4r_.z) + b4r_.z;                                         \
                v_.w = (v_.w - mu_) * (rs_ * g4r_.w) + b4r_.w;                                         \
                if (pro_gelu) {                                                                      \
                    v_.x = gelu_erf(v_.x); v_.y = gelu_erf(v_.y);                                    \
                    v_.z = gelu_erf(v_.z); v_.w = gelu_erf(v_.w);                                    \
                }                                                                                    \
            }                                                                                        \
            if (!((avalid >> p_) & 1u)) v_ = f32x4{0.f, 0.f, 0.f, 0.f};                              \
            const f32x2 p0_ = split2(v_.x, v_.y), p1_ = split2(v_.z, v_.w);                          \
            {   /* rows past the slab go to a dump row: no branch (see the header) */                 \
                const int q_ = p_ * RP + srow_o;                                                     \
                float* row_ = (dst_) + (WP8 ? (q_ < QA ? ((q_ + 7) & 7) * NI8 + ((q_ + 7) >> 3) : 8 * NI8 + 1) : min(q_, QA + 1)) * LDK; \
                *reinterpret_cast<f32x2*>(row_ + c4 * 2) = f32x2{p0_.x, p1_.x};       /* hi */       \
                *reinterpret_cast<f32x2*>(row_ + 16 + c4 * 2) = f32x2{p0_.y, p1_.y};  /* lo */       \
            }                                                                                        \
            if (pro && (p_ % WIDE_STAGE_GROUP) == WIDE_STAGE_GROUP - 1) __builtin_amdgcn_sched_barrier(0);   /* WIDE_STAGE_GROUP passes at a time: register pressure */ \
        }                                                                                            \
    }
    // B ring slot <- the two 16-column tiles (cp_ 2, cp_ 2 + 1) of tap tap_ in chunk chunk_
#define WIDE_LOAD_B(slot_, chunk_, tap_, cp_)                                                        \
    if (!dbg_no_wload) {                                                                             \
        const float* p_ = wfl + (size_t)((tap_) * nchunks + (chunk_)) * wtap + (cp_) * 1024;         \
        _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) {                                          \
            fb[slot_][c_][0] = *reinterpret_cast<const f16x8*>(p_ + c_ * 512);                       \
            fb[slot_][c_][1] = *reinterpret_cast<const f16x8*>(p_ + c_ * 512 + 256);                 \
        }                                                                                            \
    }
    // A ring slot <- row tile rt_ seen through tap tap_ (slab row shift shift_ floats)
#define WIDE_LOAD_FA(slot_, tap_, shift_, rt_)                                                       \
    {                                                                                                \
        const unsigned mb_ = (am[(rt_) / 3] >> (9 * ((rt_) % 3) + (tap_))) & 1u;                     \
        const int o_ = mb_ ? aoff0 + WIDE_LDSOFF(rt_) * LDK + (shift_) : zoff;                       \
        fa[slot_][0] = *reinterpret_cast<const f16x8*>(Abuf + o_);                                   \
        fa[slot_][1] = *reinterpret_cast<const f16x8*>(Abuf + o_ + 16);                              \
    }
    // one tile step: row tile rt_ x the phase's two column tiles ct0_, ct0_ + 1: hi*hi, hi*lo, lo*hi
#define WIDE_STEP(fas_, fbs_, rt_, ct0_)                                                             \
    if (!dbg_no_mfma) {                                                                              \
        acc[rt_][(ct0_)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[fas_][0], fb[fbs_][0][0], acc[rt_][(ct0_)], 0, 0, 0);         \
        acc[rt_][(ct0_) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[fas_][0], fb[fbs_][1][0], acc[rt_][(ct0_) + 1], 0, 0, 0); \
        acc[rt_][(ct0_)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[fas_][0], fb[fbs_][0][1], acc[rt_][(ct0_)], 0, 0, 0);         \
        acc[rt_][(ct0_) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[fas_][0], fb[fbs_][1][1], acc[rt_][(ct0_) + 1], 0, 0, 0); \
        acc[rt_][(ct0_)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[fas_][1], fb[fbs_][0][0], acc[rt_][(ct0_)], 0, 0, 0);         \
        acc[rt_][(ct0_) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[fas_][1], fb[fbs_][1][0], acc[rt_][(ct0_) + 1], 0, 0, 0); \
    }

    const int nrows = (kc1 - kc0) * 3;        // loop trips: one kernel row (3 taps) of one chunk each
    f16x8 fa[FAR][2], fb[FBR][2][2];
    // slab-row shift (floats) of tap (kernel row kr_, column index dwi_ = dw + 1)
#define WIDE_SHIFT(kr_, dwi_) (WP8 ? (((dwi_) - 1) * NI8 + ((kr_) - 1)) * LDK : (((kr_) - 1) * W + ((dwi_) - 1)) * LDK)

    if constexpr (!W2 && NT == 2 && !(WP && RT == 8)) {
        // 128-wide tiles: one tap at a time with ALL four column tiles of the tap's weights in registers (two tap
        // slots = 64 registers), so every A fragment is read from LDS once per tap and feeds 12 MFMAs; the next
        // tap's weights are loaded a whole tap (96 MFMAs) ahead.  Loop body = two taps (compile-time slots); the
        // flat tap sequence over (chunk, tap) has even length because K % 64 == 0 (checked on the host).
        f16x8 fbt[2][4][2], fat[2][2];
#define TAP_LOAD_B(slot_, chunk_, tap_)                                                              \
        if (!dbg_no_wload) {                                                                         \
            const float* p_ = wfl + (size_t)((tap_) * nchunks + (chunk_)) * wtap;                    \
            _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                      \
                fbt[slot_][c_][0] = *reinterpret_cast<const f16x8*>(p_ + c_ * 512);                  \
                fbt[slot_][c_][1] = *reinterpret_cast<const f16x8*>(p_ + c_ * 512 + 256);            \
            }                                                                                        \
        }
#define TAP_LOAD_FA(slot_, tap_, shift_, rt_)                                                        \
        {                                                                                            \
            const unsigned mb_ = (am[(rt_) / 3] >> (9 * ((rt_) % 3) + (tap_))) & 1u;                 \
            const int o_ = mb_ ? aoff0 + WIDE_ROWOFF(rt_) * LDK + (shift_) : zoff;                   \
            fat[slot_][0] = *reinterpret_cast<const f16x8*>(Abuf + o_);                              \
            fat[slot_][1] = *reinterpret_cast<const f16x8*>(Abuf + o_ + 16);                         \
        }
        // taps == 9: tap = 3 (dh + 1) + (dw + 1); taps == 3 (W == 1 maps, centre column only): tap = dh + 1, mask bit 3 tap + 1
        const int TAPS = a.taps;
        const bool t3 = (TAPS == 3);
#define TAP_SHIFT(tap_) (t3 ? ((tap_) - 1) * W * LDK : (((tap_) / 3 - 1) * W + ((tap_) - ((tap_) / 3) * 3 - 1)) * LDK)
#define TAP_BIT(tap_) (t3 ? 3 * (tap_) + 1 : (tap_))
        if constexpr (WP) {
            // Width-4 maps (3x3 kernels, host-checked a.W == 4): chunk-outer loop with the 9 taps unrolled, so the tile list of a
            // tap is compile-time -- a runtime `if` around an MFMA group would cost the weight prefetch its overlap (see the
            // header).  Tile steps per chunk: 3 x (RT + 2 (RT - RT / 4)): 60 for RT = 8, 30 for RT = 4, both even, so the
            // two-slot A ring keeps its phase from chunk to chunk; tap 0 of the next chunk lands in weight slot 1 (9 is odd).
#define WP_SHIFT(t_) ((((t_) / 3 - 1) * W + ((t_) % 3 - 1)) * LDK)
            TAP_LOAD_B(0, kc0, 0)
            WIDE_LOAD_A(kc0)
            WIDE_STAGE_A()
            __syncthreads();
            WIDE_STAMP(2)
            // (the 60 per-step fragment offsets are loop-invariant; hoisted out of the chunk loop they are 60 live registers and
            //  the 128-row-per-wave variants spill ~270 bytes per lane.  An opaque copy of the lane's base offset, refreshed per
            //  chunk, makes the compiler recompute each offset -- one select + one add -- where it is used.)
            int aoff_w = aoff0;
#define WP_LOAD_FA(slot_, tap_, shift_, rt_)                                                         \
            {                                                                                        \
                const unsigned mb_ = (am[(rt_) / 3] >> (9 * ((rt_) % 3) + (tap_))) & 1u;             \
                const int o_ = mb_ ? aoff_w + WIDE_ROWOFF(rt_) * LDK + (shift_) : zoff;              \
                fat[slot_][0] = *reinterpret_cast<const f16x8*>(Abuf + o_);                          \
                fat[slot_][1] = *reinterpret_cast<const f16x8*>(Abuf + o_ + 16);                     \
            }
            WP_LOAD_FA(0, 0, WP_SHIFT(0), wp_first(0))
            for (int chunk = kc0; chunk < kc1; ++chunk) {
                const int cn1 = min(chunk + 1, kc1 - 1);        // (past the end: a harmless re-read)
                asm volatile("" : "+v"(aoff_w));
#pragma unroll
                for (int i = 0; i < (RT + 2) / 3; ++i) asm volatile("" : "+v"(am[i]));      // (... and the 60 validity bits: SGPR pairs otherwise)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t < 8) { TAP_LOAD_B((t + 1) & 1, chunk, t + 1) } else { TAP_LOAD_B(1, cn1, 0) }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        {
                            if (wp_active(t, rt)) {
                                const int g = wp_order(t, rt, RT);
                                const int nx = wp_next(t, rt, RT);
                                if (nx >= 0) { WP_LOAD_FA((g + 1) & 1, t, WP_SHIFT(t), nx) }
                                else if (t < 8) { WP_LOAD_FA((g + 1) & 1, t + 1, WP_SHIFT(t + 1), wp_first(t + 1)) }
                                __builtin_amdgcn_sched_barrier(0);
                                if (!dbg_no_mfma) {
#pragma unroll
                                    for (int c = 0; c < 4; ++c)
                                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[g & 1][0], fbt[t & 1][c][0], acc[rt][c], 0, 0, 0);
#pragma unroll
                                    for (int c = 0; c < 4; ++c)
                                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[g & 1][0], fbt[t & 1][c][1], acc[rt][c], 0, 0, 0);
#pragma unroll
                                    for (int c = 0; c < 4; ++c)
                                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[g & 1][1], fbt[t & 1][c][0], acc[rt][c], 0, 0, 0);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                }
                if (chunk + 1 < kc1 && !dbg_no_aload) {
                    WIDE_LOAD_A(chunk + 1)          // in flight across the barrier
                    __syncthreads();                // every wave is done reading the slab of this chunk
                    WIDE_STAGE_A()
                    __syncthreads();
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) { fbt[0][c][0] = fbt[1][c][0]; fbt[0][c][1] = fbt[1][c][1]; }    // tap 0's weights sit in slot 1
                WP_LOAD_FA(0, 0, WP_SHIFT(0), wp_first(0))
            }
#undef WP_SHIFT
#undef WP_LOAD_FA
        } else if constexpr (PIPE) {
            // Pipelined hand-over (3x3 kernels only: host).  Chunk-outer loop, the 9 taps unrolled, so tap, weight slot
            // and staging pass are compile-time.  While chunk c is multiplied, the slab of chunk c + 1 is built in the
            // idle buffer ONE 32-row pass per tap: pass t is transformed (GroupNorm affine, GELU, fp16 split) and written
            // at tap t from a ring of three 16-byte registers loaded three taps earlier (passes 0-2 during taps 6-8 of
            // the previous chunk) -- no hand-over stall, no 36 staging registers, ONE barrier per chunk.
            f32x4 ar[3];
            int pm0 = m0 - halo + srow_t, srow_p = srow_t, aoff_p = aoff0;
            // (the per-tap addresses below are loop-invariant; hoisted out of the chunk loop they would be ~100 live registers.
            //  Opaque copies, refreshed per chunk, make the compiler recompute them where they are used.)
#define PIPE_OPAQUE() asm volatile("" : "+v"(pm0), "+v"(srow_p), "+v"(aoff_p));
#define PIPE_LOAD_FA(slot_, tap_, shift_, rt_)                                                       \
            {                                                                                        \
                const unsigned mb_ = (am[(rt_) / 3] >> (9 * ((rt_) % 3) + (tap_))) & 1u;             \
                const int o_ = mb_ ? aoff_p + WIDE_ROWOFF(rt_) * LDK + (shift_) : zoff;              \
                fat[slot_][0] = *reinterpret_cast<const f16x8*>(Abuf + o_);                          \
                fat[slot_][1] = *reinterpret_cast<const f16x8*>(Abuf + o_ + 16);                     \
            }
#define PIPE_LOAD(slot_, chunk_, pass_)                                                              \
            {                                                                                        \
                const int mc_ = min(max(pm0 + (pass_) * RP, 0), M - 1);                              \
                ar[slot_] = *reinterpret_cast<const f32x4*>(abase + (size_t)mc_ * a.src_ld + (chunk_) * CK); \
            }
#define PIPE_GB(chunk_)                                                                              \
            if (pro) {                                                                               \
                g4r = *reinterpret_cast<const f32x4*>(a.pro_gamma + (chunk_) * CK + c4 * 4);         \
                b4r = *reinterpret_cast<const f32x4*>(a.pro_beta + (chunk_) * CK + c4 * 4);          \
            }
#define PIPE_STAGE(slot_, pass_)                                                                     \
            {                                                                                        \
                f32x4 v_ = ar[slot_];                                                                \
                if (pro) {                                                                           \
                    const int bi_ = (int)((abidx >> (BB * (pass_))) & ((1ull << BB) - 1));           \
                    const float rs_ = srstd[bi_], mu_ = smean[bi_];                                  \
                    v_.x = (v_.x - mu_) * (rs_ * g4r.x) + b4r.x;                                     \
                    v_.y = (v_.y - mu_) * (rs_ * g4r.y) + b4r.y;                                     \
                    v_.z = (v_.z - mu_) * (rs_ * g4r.z) + b4r.z;                                     \
                    v_.w = (v_.w - mu_) * (rs_ * g4r.w) + b4r.w;                                     \
                    if (pro_gelu) {                                                                  \
                        v_.x = gelu_erf(v_.x); v_.y = gelu_erf(v_.y);                                \
                        v_.z = gelu_erf(v_.z); v_.w = gelu_erf(v_.w);                                \
                    }                                                                                \
                }                                                                                    \
                if (!((avalid >> (pass_)) & 1u)) v_ = f32x4{0.f, 0.f, 0.f, 0.f};                     \
                const f32x2 p0_ = split2(v_.x, v_.y), p1_ = split2(v_.z, v_.w);                      \
                float* row_ = Awr + min((pass_) * RP + srow_p, QA + 1) * LDK;                        \
                *reinterpret_cast<f32x2*>(row_ + c4 * 2) = f32x2{p0_.x, p1_.x};                      \
                *reinterpret_cast<f32x2*>(row_ + 16 + c4 * 2) = f32x2{p0_.y, p1_.y};                 \
            }
#define PIPE_SHIFT(t_) ((((t_) / 3 - 1) * W + ((t_) % 3 - 1)) * LDK)
            TAP_LOAD_B(0, kc0, 0)
            WIDE_LOAD_A(kc0)
            WIDE_STAGE_A()
            __syncthreads();
            { float* t_ = Abuf; Abuf = Awr; Awr = t_; }
            {
                const int c1 = min(kc0 + 1, kc1 - 1);
                PIPE_LOAD(0, c1, 0)
                PIPE_LOAD(1, c1, 1)
                PIPE_LOAD(2, c1, 2)
                PIPE_GB(c1)
            }
            WIDE_STAMP(2)
            PIPE_LOAD_FA(0, 0, PIPE_SHIFT(0), 0)
            for (int chunk = kc0; chunk < kc1; ++chunk) {
                const int cn1 = min(chunk + 1, kc1 - 1), cn2 = min(chunk + 2, kc1 - 1);     // (past the end: harmless re-reads)
                PIPE_OPAQUE()
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t < 8) { TAP_LOAD_B((t + 1) & 1, chunk, t + 1) } else { TAP_LOAD_B(1, cn1, 0) }
                    if (t < APASS) { PIPE_STAGE(t % 3, t) }
                    if (t + 3 < APASS) { PIPE_LOAD(t % 3, cn1, t + 3) }
                    else if (t >= 6) { PIPE_LOAD(t - 6, cn2, t - 6) }
                    if (t == 8) { PIPE_GB(cn2) }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        if (rt + 1 < RT) { PIPE_LOAD_FA((rt + 1) & 1, t, PIPE_SHIFT(t), rt + 1) }
                        else if (t < 8) { PIPE_LOAD_FA(0, t + 1, PIPE_SHIFT(t + 1), 0) }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[rt & 1][0], fbt[t & 1][c][0], acc[rt][c], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[rt & 1][0], fbt[t & 1][c][1], acc[rt][c], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[rt & 1][1], fbt[t & 1][c][0], acc[rt][c], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();            // every wave is done with this chunk's slab; the next one is complete
                { float* t_ = Abuf; Abuf = Awr; Awr = t_; }
#pragma unroll
                for (int c = 0; c < 4; ++c) { fbt[0][c][0] = fbt[1][c][0]; fbt[0][c][1] = fbt[1][c][1]; }    // tap 0's weights sit in slot 1 (9 is odd)
                PIPE_LOAD_FA(0, 0, PIPE_SHIFT(0), 0)
            }
#undef PIPE_OPAQUE
#undef PIPE_LOAD_FA
#undef PIPE_LOAD
#undef PIPE_GB
#undef PIPE_STAGE
#undef PIPE_SHIFT
        } else if constexpr (G2) {
        // Two chunks per hand-over: the slab is [2][QZ][LDK]; chunk c of the workgroup's range lives in sub-slab (c - kc0) & 1, the
        // hand-over happens when the tap sequence enters an EVEN chunk and stages that chunk and the next one from two register
        // sets whose loads were issued together (past the end of the range: a harmless duplicate of the last chunk).
        const int ntaps = (kc1 - kc0) * TAPS;
        f32x4 areg2[APASS];
        f32x4 g4r2 = {1.f, 1.f, 1.f, 1.f}, b4r2 = {0.f, 0.f, 0.f, 0.f};
        bool ident2 = false;
        const float* Acur = Abuf;
#define G2_LOAD_FA(slot_, tap_, shift_, rt_)                                                         \
        {                                                                                            \
            const unsigned mb_ = (am[(rt_) / 3] >> (9 * ((rt_) % 3) + (tap_))) & 1u;                 \
            const int o_ = mb_ ? aoff0 + WIDE_ROWOFF(rt_) * LDK + (shift_) : zoff;                   \
            fat[slot_][0] = *reinterpret_cast<const f16x8*>(Acur + o_);                              \
            fat[slot_][1] = *reinterpret_cast<const f16x8*>(Acur + o_ + 16);                         \
        }
        TAP_LOAD_B(0, kc0, 0)
        WIDE_LOAD_A_(kc0, areg, g4r, b4r, ident)
        WIDE_LOAD_A_(min(kc0 + 1, kc1 - 1), areg2, g4r2, b4r2, ident2)
        WIDE_STAGE_A_(areg, g4r, b4r, ident, Abuf)
        WIDE_STAGE_A_(areg2, g4r2, b4r2, ident2, Abuf + QZ * LDK)
        __syncthreads();
        WIDE_STAMP(2)
        G2_LOAD_FA(0, TAP_BIT(0), TAP_SHIFT(0), 0)
        int chunk = kc0, tap = 0;
        for (int tt = 0; tt < ntaps; tt += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                int ntap = tap + 1, nchunk = chunk;
                if (ntap == TAPS) { ntap = 0; nchunk = chunk + 1; }
                const bool have_next = (tt + half + 1 < ntaps);
                const bool next_C = have_next && (ntap == 0);                       // the next tap opens a chunk ...
                const bool next_A = next_C && (((nchunk - kc0) & 1) == 0);          // ... that needs a hand-over
                TAP_LOAD_B(1 - half, (have_next ? nchunk : chunk), (have_next ? ntap : tap))
                const int sh = TAP_SHIFT(tap), nsh = TAP_SHIFT(ntap), tb = TAP_BIT(tap), ntb = TAP_BIT(ntap);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if (rt + 1 < RT) { G2_LOAD_FA((rt + 1) & 1, tb, sh, rt + 1) }
                    else if (have_next && !next_C) { G2_LOAD_FA(0, ntb, nsh, 0) }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!dbg_no_mfma) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[rt & 1][0], fbt[half][c][0], acc[rt][c], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[rt & 1][0], fbt[half][c][1], acc[rt][c], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[rt & 1][1], fbt[half][c][0], acc[rt][c], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (next_A && !dbg_no_aload) {
                    WIDE_LOAD_A_(nchunk, areg, g4r, b4r, ident)                     // both in flight across the barrier
                    WIDE_LOAD_A_(min(nchunk + 1, kc1 - 1), areg2, g4r2, b4r2, ident2)
                    __syncthreads();                // every wave is done reading both sub-slabs
                    WIDE_STAGE_A_(areg, g4r, b4r, ident, Abuf)
                    WIDE_STAGE_A_(areg2, g4r2, b4r2, ident2, Abuf + QZ * LDK)
                    __syncthreads();
                    Acur = Abuf;
                    G2_LOAD_FA(0, TAP_BIT(0), TAP_SHIFT(0), 0)
                } else if (next_C) {
                    Acur = Abuf + (dbg_no_aload ? 0 : ((nchunk - kc0) & 1) * QZ * LDK);      // the chunk staged with its predecessor
                    G2_LOAD_FA(0, TAP_BIT(0), TAP_SHIFT(0), 0)
                }
                tap = ntap;
                chunk = nchunk;
            }
        }
#undef G2_LOAD_FA
        } else {
        const int ntaps = (kc1 - kc0) * TAPS;
        TAP_LOAD_B(0, kc0, 0)
        WIDE_LOAD_A(kc0)
        WIDE_STAGE_A()
        __syncthreads();
        if (DB) { float* t_ = Abuf; Abuf = Awr; Awr = t_; }
        WIDE_STAMP(2)
        TAP_LOAD_FA(0, TAP_BIT(0), TAP_SHIFT(0), 0)
        int chunk = kc0, tap = 0;
        for (int tt = 0; tt < ntaps; tt += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                int ntap = tap + 1, nchunk = chunk;
                if (ntap == TAPS) { ntap = 0; nchunk = chunk + 1; }
                const bool have_next = (tt + half + 1 < ntaps);
                const bool next_A = have_next && (ntap == 0);
                // unconditional prefetch (the last tap re-reads its own weights): see the header
                TAP_LOAD_B(1 - half, (have_next ? nchunk : chunk), (have_next ? ntap : tap))
                const int sh = TAP_SHIFT(tap), nsh = TAP_SHIFT(ntap), tb = TAP_BIT(tap), ntb = TAP_BIT(ntap);
                if (DB && WIDE_EARLY && next_A && !dbg_no_aload) { WIDE_LOAD_A(nchunk) }      // in flight during this (last) tap of the chunk
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if (rt + 1 < RT) { TAP_LOAD_FA((rt + 1) & 1, tb, sh, rt + 1) }
                    else if (have_next && !next_A) { TAP_LOAD_FA(0, ntb, nsh, 0) }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!dbg_no_mfma) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[rt & 1][0], fbt[half][c][0], acc[rt][c], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[rt & 1][0], fbt[half][c][1], acc[rt][c], 0, 0, 0);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fat[rt & 1][1], fbt[half][c][0], acc[rt][c], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (next_A && !dbg_no_aload) {
                    if (!(DB && WIDE_EARLY)) { WIDE_LOAD_A(nchunk) }        // in flight across the barrier
                    if (!DB) __syncthreads();       // every wave is done reading the slab of this chunk
                    WIDE_STAGE_A()                  // (DB: into the idle slab, while slower waves still read the current one)
                    __syncthreads();
                    if (DB) { float* t_ = Abuf; Abuf = Awr; Awr = t_; }
                    TAP_LOAD_FA(0, TAP_BIT(0), TAP_SHIFT(0), 0)
                } else if (next_A) {
                    TAP_LOAD_FA(0, TAP_BIT(0), TAP_SHIFT(0), 0)
                }
                tap = ntap;
                chunk = nchunk;
            }
        }
        }   // !PIPE
#undef TAP_LOAD_B
#undef TAP_LOAD_FA
#undef TAP_SHIFT
#undef TAP_BIT
    } else if constexpr (WP) {
        // Width-4 maps, 128 rows per wave (the unrolled tap loop above spills at this size: its 64 weight registers; this loop
        // keeps ONE column pair of a tap's weights, 32 registers, like the width-2 loop below, of which it is the generalisation).
        // Row tile rt holds the positions w = rt % 4 of its 64-row block.  Kernel row = 6 phases: centre tap on all 8 row tiles
        // (x 2 column pairs), dw = +1 on the six tiles with w < 3, dw = -1 on the six with w > 0: 40 tile steps instead of 48.
        //   step q:  0..15 centre   16..27 (+1: tiles 0 1 2 4 5 6, pair 0 / 1)   28..39 (-1: tiles 1 2 3 5 6 7, pair 0 / 1)
#define WP_PLUS(i_) (WP8 ? (i_) : (i_) + (i_) / 3)
#define WP_MINUS(i_) (WP8 ? (i_) + 1 : (i_) + (i_) / 3 + 1)
        constexpr int WS = WP8 ? 7 : 6;                 // tiles a side column visits (W2: 4)
        constexpr int Q1 = 16 + 2 * WS, Q2 = 16 + 4 * WS;     // first step of the dw = -1 phases; steps per kernel row (40 | 44: both = 0 mod FAR)
        WIDE_LOAD_B(0, kc0, 1, 0)
        WIDE_LOAD_A(kc0)
        WIDE_STAGE_A()
        __syncthreads();
        WIDE_STAMP(2)
        WIDE_LOAD_FA(0, 1, WIDE_SHIFT(0, 1), 0)
        WIDE_LOAD_FA(1, 1, WIDE_SHIFT(0, 1), 1)
        int chunk = kc0, kr = 0;
        for (int r = 0; r < nrows; ++r) {
            int nkr = kr + 1, nchunk = chunk;
            if (nkr == 3) { nkr = 0; nchunk = chunk + 1; }
            const bool have_next = (r + 1 < nrows);
            const bool next_A = have_next && (nkr == 0);
            const int bkr = have_next ? nkr : kr, bchunk = have_next ? nchunk : chunk;
            // ---- centre tap, all row tiles: steps q = 0..15 ----
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                if (cp == 0) { WIDE_LOAD_B(1, chunk, kr * 3 + 1, 1) } else { WIDE_LOAD_B(0, chunk, kr * 3 + 2, 0) }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int q = cp * 8 + rt;
                    if (q + 2 < 16) { WIDE_LOAD_FA((q + 2) % FAR, kr * 3 + 1, WIDE_SHIFT(kr, 1), (q + 2) % 8) }
                    else { WIDE_LOAD_FA((q + 2) % FAR, kr * 3 + 2, WIDE_SHIFT(kr, 2), WP_PLUS(q + 2 - 16)) }
                    __builtin_amdgcn_sched_barrier(0);
                    WIDE_STEP(q % FAR, cp, rt, cp * 2)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- dw = +1 on the tiles with w < 3: steps 16..27 ----
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                if (cp == 0) { WIDE_LOAD_B(1, chunk, kr * 3 + 2, 1) } else { WIDE_LOAD_B(0, chunk, kr * 3, 0) }
#pragma unroll
                for (int i = 0; i < WS; ++i) {
                    const int q = 16 + cp * WS + i;
                    if (q + 2 < Q1) { WIDE_LOAD_FA((q + 2) % FAR, kr * 3 + 2, WIDE_SHIFT(kr, 2), WP_PLUS((q + 2 - 16) % WS)) }
                    else { WIDE_LOAD_FA((q + 2) % FAR, kr * 3, WIDE_SHIFT(kr, 0), WP_MINUS(q + 2 - Q1)) }
                    __builtin_amdgcn_sched_barrier(0);
                    WIDE_STEP(q % FAR, cp, WP_PLUS(i), cp * 2)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- dw = -1 on the tiles with w > 0: steps 28..39 ----
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                if (cp == 0) { WIDE_LOAD_B(1, chunk, kr * 3, 1) } else { WIDE_LOAD_B(0, bchunk, bkr * 3 + 1, 0) }
#pragma unroll
                for (int i = 0; i < WS; ++i) {
                    const int q = Q1 + cp * WS + i;
                    if (q + 2 < Q2) { WIDE_LOAD_FA((q + 2) % FAR, kr * 3, WIDE_SHIFT(kr, 0), WP_MINUS((q + 2 - Q1) % WS)) }
                    else if (have_next && !next_A) { WIDE_LOAD_FA((q + 2) % FAR, nkr * 3 + 1, WIDE_SHIFT(nkr, 1), q + 2 - Q2) }
                    __builtin_amdgcn_sched_barrier(0);
                    WIDE_STEP(q % FAR, cp, WP_MINUS(i), cp * 2)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (next_A) {
                WIDE_LOAD_A(nchunk)
                __syncthreads();
                WIDE_STAGE_A()
                __syncthreads();
                WIDE_LOAD_FA(0, 1, WIDE_SHIFT(0, 1), 0)
                WIDE_LOAD_FA(1, 1, WIDE_SHIFT(0, 1), 1)
            }
            kr = nkr;
            chunk = nchunk;
        }
#undef WP_PLUS
#undef WP_MINUS
    } else if constexpr (!W2) {
        // phase ph of a kernel row: tap column ph / NT, column-tile pair ph % NT
        WIDE_LOAD_B(0, kc0, 0, 0)
        WIDE_LOAD_A(kc0)
        WIDE_STAGE_A()
        __syncthreads();
        WIDE_STAMP(2)
        WIDE_LOAD_FA(0, 0, WIDE_SHIFT(0, 0), 0)
        WIDE_LOAD_FA(1, 0, WIDE_SHIFT(0, 0), 1)
        int chunk = kc0, kr = 0;
        for (int r = 0; r < nrows; ++r) {
            int nkr = kr + 1, nchunk = chunk;
            if (nkr == 3) { nkr = 0; nchunk = chunk + 1; }
            const bool have_next = (r + 1 < nrows);
            const bool next_A = have_next && (nkr == 0);
            const int bkr = have_next ? nkr : kr, bchunk = have_next ? nchunk : chunk;   // B prefetch past this row (last row: re-read)
#pragma unroll
            for (int ph = 0; ph < PH; ++ph) {
                if (ph + 1 < PH) {
                    WIDE_LOAD_B((ph + 1) % FBR, chunk, kr * 3 + (ph + 1) / NT, (ph + 1) % NT)
                } else {
                    WIDE_LOAD_B((ph + 1) % FBR, bchunk, bkr * 3, 0)
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int G = ph * RT + rt;           // tile step of this row; A slot G % 3, prefetch for step G + 2
                    if (G + 2 < PH * RT) {
                        const int dwi2 = ((G + 2) / RT) / NT;
                        WIDE_LOAD_FA((G + 2) % FAR, kr * 3 + dwi2, WIDE_SHIFT(kr, dwi2), (G + 2) % RT)
                    } else if (have_next && !next_A) {
                        WIDE_LOAD_FA((G + 2) % FAR, nkr * 3, WIDE_SHIFT(nkr, 0), G + 2 - PH * RT)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    WIDE_STEP(G % FAR, ph % FBR, rt, (ph % NT) * 2)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (next_A) {
                WIDE_LOAD_A(nchunk)             // in flight across the barrier
                __syncthreads();                // every wave is done reading the slab of this chunk
                WIDE_STAGE_A()
                __syncthreads();
                WIDE_LOAD_FA(0, 0, WIDE_SHIFT(0, 0), 0)
                WIDE_LOAD_FA(1, 0, WIDE_SHIFT(0, 0), 1)
            }
            kr = nkr;
            chunk = nchunk;
        }
    } else {
        // Image width 2: for w = 0 the dw = -1 column of the kernel only sees zero padding, for w = 1 the dw = +1
        // column.  Kernel row = 6 phases: centre tap on all 8 row tiles (x 2 column pairs), dw = +1 on the even
        // tiles, dw = -1 on the odd tiles: 32 tile steps instead of 48, only all-zero products dropped.
        //   step q:  0..7 (centre, pair 0)   8..15 (centre, pair 1)   16..19 / 20..23 (+1, even tiles, pair 0 / 1)
        //            24..27 / 28..31 (-1, odd tiles, pair 0 / 1)
        WIDE_LOAD_B(0, kc0, 1, 0)
        WIDE_LOAD_A(kc0)
        WIDE_STAGE_A()
        __syncthreads();
        WIDE_STAMP(2)
        WIDE_LOAD_FA(0, 1, WIDE_SHIFT(0, 1), 0)
        WIDE_LOAD_FA(1, 1, WIDE_SHIFT(0, 1), 1)
        int chunk = kc0, kr = 0;
        for (int r = 0; r < nrows; ++r) {
            int nkr = kr + 1, nchunk = chunk;
            if (nkr == 3) { nkr = 0; nchunk = chunk + 1; }
            const bool have_next = (r + 1 < nrows);
            const bool next_A = have_next && (nkr == 0);
            const int bkr = have_next ? nkr : kr, bchunk = have_next ? nchunk : chunk;
            // (three plain loop nests with affine indices rather than one 32-step loop with a step table: the
            //  accumulator array only stays in registers if every index folds in the early unroll)
            // ---- centre tap, all row tiles: steps q = 0..15 ----
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                if (cp == 0) { WIDE_LOAD_B(1, chunk, kr * 3 + 1, 1) } else { WIDE_LOAD_B(0, chunk, kr * 3 + 2, 0) }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int q = cp * 8 + rt;
                    if (q + 2 < 16) { WIDE_LOAD_FA((q + 2) % FAR, kr * 3 + 1, WIDE_SHIFT(kr, 1), (q + 2) % 8) }
                    else { WIDE_LOAD_FA((q + 2) % FAR, kr * 3 + 2, WIDE_SHIFT(kr, 2), 2 * (q + 2 - 16)) }
                    __builtin_amdgcn_sched_barrier(0);
                    WIDE_STEP(q % FAR, cp, rt, cp * 2)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- dw = +1 on the even row tiles (w = 0 positions): steps 16..23 ----
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                if (cp == 0) { WIDE_LOAD_B(1, chunk, kr * 3 + 2, 1) } else { WIDE_LOAD_B(0, chunk, kr * 3, 0) }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int q = 16 + cp * 4 + i;
                    if (q + 2 < 24) { WIDE_LOAD_FA((q + 2) % FAR, kr * 3 + 2, WIDE_SHIFT(kr, 2), 2 * ((q + 2 - 16) % 4)) }
                    else { WIDE_LOAD_FA((q + 2) % FAR, kr * 3, WIDE_SHIFT(kr, 0), 2 * (q + 2 - 24) + 1) }
                    __builtin_amdgcn_sched_barrier(0);
                    WIDE_STEP(q % FAR, cp, 2 * i, cp * 2)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- dw = -1 on the odd row tiles (w = 1 positions): steps 24..31 ----
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                if (cp == 0) { WIDE_LOAD_B(1, chunk, kr * 3, 1) } else { WIDE_LOAD_B(0, bchunk, bkr * 3 + 1, 0) }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int q = 24 + cp * 4 + i;
                    if (q + 2 < 32) { WIDE_LOAD_FA((q + 2) % FAR, kr * 3, WIDE_SHIFT(kr, 0), 2 * ((q + 2 - 24) % 4) + 1) }
                    else if (have_next && !next_A) { WIDE_LOAD_FA((q + 2) % FAR, nkr * 3 + 1, WIDE_SHIFT(nkr, 1), q + 2 - 32) }
                    __builtin_amdgcn_sched_barrier(0);
                    WIDE_STEP(q % FAR, cp, 2 * i + 1, cp * 2)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (next_A) {
                WIDE_LOAD_A(nchunk)
                __syncthreads();
                WIDE_STAGE_A()
                __syncthreads();
                WIDE_LOAD_FA(0, 1, WIDE_SHIFT(0, 1), 0)
                WIDE_LOAD_FA(1, 1, WIDE_SHIFT(0, 1), 1)
            }
            kr = nkr;
            chunk = nchunk;
        }
    }
#undef WIDE_LOAD_A
#undef WIDE_STAGE_A
#undef WIDE_LOAD_B
#undef WIDE_LOAD_FA
#undef WIDE_STEP
#undef WIDE_SHIFT
    WIDE_STAMP(3)
    __syncthreads();          // the epilogue re-uses the slab's LDS: every wave must be done reading A fragments

#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] *= DESCALE;

    // ---- epilogue: GroupNorm partial sums (fp32 per 4-row unit -> fp64 per sample, fixed order), then the
    //      tile through LDS in two halves so that every lane stores 16 bytes ----
    // accumulator layout: lane (l16, kg), register j of tile (rt, ct) = row 16 rt + 4 rb4 + j, column 16 ct + l16, where
    // rb4 = kg, or {0, 2, 3, 1}[kg] under the row permutation of the A-fragment reads (WIDE_LDS_PERM)
    const int rb4 = PERM ? ((0x1320 >> (4 * kg)) & 0xF) : kg;
    constexpr int NH = W2 ? 2 : WP ? RT / 4 : WIDE_NH;   // the tile leaves in NH parts (RW / NH rows of each wave per part; WP: one 64-row block each)
    constexpr int HROWS = M_T / NH;
    const bool partial_out = ksp > 1;                    // split-K: raw partial tile to the workspace, statistics by the combine kernel
    float* const dstp = partial_out ? a.partial + (size_t)ks * M * N : a.dst;
    const int dst_ld = partial_out ? N : a.dst_ld;
    float* otile = smem;                                 // [HROWS][N_T]
    float* srow = smem + HROWS * N_T;                    // [M_T / 4][WN][2]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        // the lane's four registers are one 4-row unit (W2: rows 2 apart inside one 8-row block -> slot unit
        // 2 (block) + parity; a sample is still a contiguous run of HW / 4 slot units because HW % 8 == 0)
        // WP: rows 4 apart inside one 16-row span -> slot unit 4 (span) + class; HW % 16 == 0 keeps a sample's units contiguous
        // (WP8: rows 8 apart inside one 32-row span -> unit 8 (span) + class; HW % 32 == 0)
        const int unit = wm * (RW / 4) + (W2 ? (rt >> 1) * 8 + 2 * kg + (rt & 1) : WP8 ? 8 * kg + rt : WP ? (rt >> 2) * 16 + 4 * kg + (rt & 3) : rt * 4 + rb4);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[rt][ct][j];
                s1 += v;
                s2 += v * v;
            }
        s1 = row16_sum_dpp(s1);
        s2 = row16_sum_dpp(s2);
        if (l16 == 0) {
            srow[(unit * WN + wn) * 2] = s1;
            srow[(unit * WN + wn) * 2 + 1] = s2;
        }
    }

#pragma unroll
    for (int h = 0; h < NH; ++h) {
        if (h) __syncthreads();
        if constexpr (WP8) {
            // every row tile spans the wave's 128 rows (rows 8 (4 kg + j) + rt): the half h = rows [64 h, 64 h + 64) is what the
            // lanes with kg >> 1 == h hold, of ALL eight tiles
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int col_l = wn * NT * 32 + ct * 16 + l16;
                    if ((kg >> 1) == h) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            otile[(wm * (RW / NH) + 8 * (4 * (kg & 1) + j) + rt) * N_T + col_l] = acc[rt][ct][j];
                    }
                }
        } else
#pragma unroll
        for (int rq = 0; rq < RT / NH; ++rq) {
            const int rt = (RT / NH) * h + rq;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int col_l = wn * NT * 32 + ct * 16 + l16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // row inside the wave's half (64 rows): W2 (rt>>1 - 2h) 32 + 2 (4 kg + j) + parity, else rq 16 + 4 kg + j
                    const int rw = W2 ? ((rt >> 1) - 2 * h) * 32 + 2 * (4 * kg + j) + (rt & 1) : WP ? 4 * (4 * kg + j) + (rt & 3) : rq * 16 + 4 * rb4 + j;
                    otile[(wm * (RW / NH) + rw) * N_T + col_l] = acc[rt][ct][j];
                }
            }
        }
        __syncthreads();
        if (h == 0) WIDE_STAMP(4)
        if (h == 0 && !partial_out) {
            const int t_lo = m0, t_hi = min(m0 + M_T, M);
            const int ups = HW >> 2;                                   // 4-row units per sample
            const bool whole = (HW % M_T == 0);                        // the tile lies inside one sample
            const bool aligned = whole || (M_T % HW == 0 && (ups & (ups - 1)) == 0);
            if (aligned) {
                // one lane per unit, fp64 butterfly inside each sample's (aligned, power-of-two) lane segment:
                // fixed order, position-independent -> deterministic and identical for every sample
                if (wave == 0) {
                    constexpr int NU = M_T / 4;                        // units of this tile: one lane each
                    const int u = lane, uc = min(lane, NU - 1);
                    double d1 = 0.0, d2 = 0.0;
#pragma unroll
                    for (int w2 = 0; w2 < WN; ++w2) {
                        d1 += (double)srow[(uc * WN + w2) * 2];
                        d2 += (double)srow[(uc * WN + w2) * 2 + 1];
                    }
                    const int seg = whole ? NU : ups;
                    d1 = seg_sum_f64(d1, seg);              // (the xor butterfly 1, 2, 4, ..., bit for bit; DPP + permlane swaps)
                    d2 = seg_sum_f64(d2, seg);
                    const int row = t_lo + 4 * u;
                    if (u < NU && (u & (seg - 1)) == 0 && row < t_hi) {
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
            const int row = m0 + (lr / (RW / NH)) * RW + h * (RW / NH) + lr % (RW / NH);
            if (row < M) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(otile + lr * N_T + c4o * 4);
                *reinterpret_cast<f32x4*>(dstp + (size_t)row * dst_ld + n0 + c4o * 4) = v;
            }
        }
    }
    WIDE_STAMP(5)
#undef WIDE_STAMP
#undef WIDE_ROWOFF
}

template <int NT, int PRO, bool W2 = false, int RT = 8, int WN = 2, bool PIPE = false, bool TWO = false, int WP = 0, bool G2 = false>
hipError_t launch_wide_cfg(const GemmArgs& a, const GemmGeom& g, hipStream_t s) {
    constexpr int M_T = (4 / WN) * RT * 16, N_T = WN * 32 * NT, NTHR = 256;
    constexpr int APASS = (M_T + 18 + 31) / 32, NSMAX = (APASS <= 9) ? 128 : (APASS <= 10) ? 64 : 32;
    const int halo = a.W + 1;
    const int QA = M_T + 2 * halo;
    const int NS = (((QA - 1) / a.HW + 2) + 3) & ~3;
    if (NS > NSMAX || g.m_tile != M_T || g.n_tile != N_T) return hipErrorInvalidValue;
    constexpr bool DB = PIPE || (WIDE_DB && (NT == 2) && !W2);
    constexpr int LDK = WidePitch<W2, PIPE, (WP != 0) || G2>::value;
    if (PIPE && a.taps != 9) return hipErrorInvalidValue;
    if (WP && (a.taps != 9 || a.W != WP || a.HW % (4 * WP) != 0)) return hipErrorInvalidValue;
    const int QZ = (WP == 8) ? 8 * ((M_T + 18 + 6) / 8 + 1) + 2 : QA + 2;          // (WP8: the class-major slab)
    size_t lds = (size_t)(((DB || G2) ? 2 : 1) * QZ * LDK + 2 * NS) * sizeof(float);
    lds = std::max(lds, (size_t)((M_T / (W2 ? 2 : WP ? RT / 4 : WIDE_NH)) * N_T + (M_T / 4) * WN * 2) * sizeof(float));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = conv3x3_wide_kernel<NT, PRO, W2, RT, WN, PIPE, TWO, WP, G2>;
    if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return e;
    const int n_mtiles = (a.M + M_T - 1) / M_T;
    if (a.ksplit > 1 && a.K % 64 != 0) return hipErrorInvalidValue;          // split-K walks even chunk ranges
    hipLaunchKernelGGL(kern, dim3(n_mtiles * g.n_tiles * std::max(a.ksplit, 1)), dim3(NTHR), lds, s, a, g.slots, NS);
    return hipGetLastError();
}

}  // namespace

static bool wide_w2(const GemmArgs& a) { return a.W == 2 && a.taps == 9 && !(a.sw & SW_NO_W2); }

// Which layers run here: split-precision 3x3 convs (and the 3x1 convs of the W == 1 level) with GroupNorm-statistics
// epilogue whose tiling (gemm_geometry) is 256 x {128, 64} -- or 128 x 128 (small batches, coarse levels), where the
// workgroup is 4 waves x (64 x 64).
bool conv_wide_supported(const GemmArgs& a, const GemmGeom& g) {
    const bool common = a.split && a.wgt_frag != nullptr && a.W >= 1 && a.W <= 8 && (a.HW & 3) == 0 && a.M % a.HW == 0 &&
                        a.epi == EPI_STATS && a.row_stats == nullptr && a.K % CK == 0 &&
                        (a.debug & ~(DBG_STAMP | DBG_NO_MFMA | DBG_NO_WLOAD | DBG_NO_ALOAD)) == 0 && !(a.sw & SW_NO_WIDE);
    if (!common || a.pro > PRO_GN_GELU) return false;
    if (a.skip != nullptr) {        // two-source input (TWO): 128-wide tilings, first conv of a block, whole chunks per tensor
        if (a.pro == PRO_GN_GELU || a.up_C <= 0 || a.up_C >= a.K || a.up_C % CK != 0 || g.n_tile != 128 || a.skip_ld % 4 != 0 ||
            a.skip_ld < a.K - a.up_C || a.src_ld < a.up_C) return false;
    }
    if (a.ksplit > 1 && a.K % 64 != 0) return false;      // split-K walks even chunk ranges
    if ((g.m_tile + 2 * (a.W + 1) - 1) / a.HW + 2 > 128) return false;       // 7-bit packed sample index per staging pass
    if (g.m_tile == 128)            // small tiles: the tap-pair loop only (128-wide, taps walked in pairs)
        return g.n_tile == 128 && (a.taps == 9 || (a.taps == 3 && a.W == 1)) && a.K % 64 == 0 && !(a.sw & SW_NO_WIDE128);
    if (g.m_tile != 256) return false;
    if (a.taps == 3) return false;       // 256-row tiles of the W == 1 level (one workgroup per CU at most): conv_gemm's 8-wave
                                         // configuration measured faster (92 vs 139 us on 512 -> 512 at B = 4096)
    if (a.taps != 9 || !(g.n_tile == 128 || g.n_tile == 64)) return false;
    // width-2 maps: only with the zero-tap skipping variant (otherwise conv_gemm.hip's W2 configuration does less work)
    if (wide_w2(a)) return g.n_tile == 128 && (a.HW & 7) == 0;
    if (g.n_tile == 128 && a.K % 64 != 0) return false;      // the 128-wide loop walks taps in pairs
    return true;
}

hipError_t launch_conv_wide(const GemmArgs& a, const GemmGeom& g, hipStream_t s) {
    if (!conv_wide_supported(a, g)) return hipErrorInvalidValue;
    // Pipelined slab hand-over (PIPE: the next chunk's slab is staged one 32-row pass per tap into a second buffer, one barrier
    // per chunk).  Measured per layer at B = 512 and 4096 (tools/bench_convs.py, SPDM_NO_WIDE_PIPE on / off): a win only on the
    // 64-wide 64-row-per-wave variant with >= 8 chunks (up2.dc2a 249 -> 218 us); neutral on 128 x 128 tiles; a loss where a
    // workgroup has few chunks (inc.b 330 -> 363) or 128 rows per wave (register spills, see below).  So: only there.
    const bool pipe = a.taps == 9 && a.K >= 256 && !(a.sw & SW_NO_WIDE_PIPE);
    // width-4 maps (level 1): the row-permuted variants skip the kernel's side columns on the w = 0 / w = 3 row tiles
    if (a.W == 4 && a.taps == 9 && g.n_tile == 128 && a.HW % 16 == 0 && a.K % 64 == 0 && !(a.sw & SW_NO_WP4)) {
        const bool two = a.skip != nullptr;
        if (g.m_tile == 128) {
            if (two) {
                if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 4, 2, false, true, 4>(a, g, s);
                return launch_wide_cfg<2, PRO_GN, false, 4, 2, false, true, 4>(a, g, s);
            }
            if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 4, 2, false, false, 4>(a, g, s);
            if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN, false, 4, 2, false, false, 4>(a, g, s);
            return launch_wide_cfg<2, PRO_GN_GELU, false, 4, 2, false, false, 4>(a, g, s);
        }
        if (two) {
            if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 8, 2, false, true, 4>(a, g, s);
            return launch_wide_cfg<2, PRO_GN, false, 8, 2, false, true, 4>(a, g, s);
        }
        if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 8, 2, false, false, 4>(a, g, s);
        if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN, false, 8, 2, false, false, 4>(a, g, s);
        return launch_wide_cfg<2, PRO_GN_GELU, false, 8, 2, false, false, 4>(a, g, s);
    }
    // width-8 maps (level 0), 256-row tiles: the same with eight classes on a class-major slab (44 of 48 tile steps per kernel row)
    if (a.W == 8 && a.taps == 9 && g.n_tile == 128 && g.m_tile == 256 && a.HW % 32 == 0 && a.K % 64 == 0 && !(a.sw & SW_NO_WP8)) {
        if (a.skip != nullptr) {
            if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 8, 2, false, true, 8>(a, g, s);
            return launch_wide_cfg<2, PRO_GN, false, 8, 2, false, true, 8>(a, g, s);
        }
        if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 8, 2, false, false, 8>(a, g, s);
        if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN, false, 8, 2, false, false, 8>(a, g, s);
        return launch_wide_cfg<2, PRO_GN_GELU, false, 8, 2, false, false, 8>(a, g, s);
    }
    if (a.skip != nullptr) {        // two-source input: the three 128-wide configurations
        if (g.m_tile == 128) {
            if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 4, 2, false, true>(a, g, s);
            return launch_wide_cfg<2, PRO_GN, false, 4, 2, false, true>(a, g, s);
        }
        if (wide_w2(a)) {
            if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, true, 8, 2, false, true>(a, g, s);
            return launch_wide_cfg<2, PRO_GN, true, 8, 2, false, true>(a, g, s);
        }
        if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 8, 2, false, true>(a, g, s);
        return launch_wide_cfg<2, PRO_GN, false, 8, 2, false, true>(a, g, s);
    }
    // Two chunks per slab hand-over (G2) on the 64-row-per-wave tap loops: measured and NOT used (opt-in: SPDM_G2=1).  Same box,
    // traced, B = 4096: the 128 x 128 tiles of levels 2-3 are unchanged within 1-2 us per launch (16 launches, 35-100 us each);
    // the 64-wide 256 x 64 tiles get SLOWER (inc.b 289 -> 311, up3.dc2a 462 -> 482, up3.dc2b 303 -> 310 us: 24-76 bytes of
    // scratch and the 144-byte pitch).  So the global-load round trip is not what a hand-over costs -- the transform (GroupNorm,
    // GELU, split), its LDS writes and the barrier pair are, and they scale with the bytes staged.
    const bool g2 = a.K >= 64 && (a.sw & SW_G2);
    if (g.m_tile == 128) {
        if (g2) {
            if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 4, 2, false, false, false, true>(a, g, s);
            if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN, false, 4, 2, false, false, false, true>(a, g, s);
            return launch_wide_cfg<2, PRO_GN_GELU, false, 4, 2, false, false, false, true>(a, g, s);
        }
        if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 4>(a, g, s);
        if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN, false, 4>(a, g, s);
        return launch_wide_cfg<2, PRO_GN_GELU, false, 4>(a, g, s);
    }
    if (wide_w2(a)) {
        if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, true>(a, g, s);
        if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN, true>(a, g, s);
        return launch_wide_cfg<2, PRO_GN_GELU, true>(a, g, s);
    }
    if (g.n_tile == 128) {
        // (not for the 128-row-per-wave variants: at 256 registers the pipelined loop spills -- 965 vs 931 us on up3.dc1a,
        //  1301 vs 999 with the GELU prologue -- and alone on a CU with 512 registers, even with a third weight slot loaded
        //  two taps ahead, it is 10 % slower than two workgroups of the old loop: profiles/r02_conv_ablation.txt)
        if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE>(a, g, s);
        if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN>(a, g, s);
        return launch_wide_cfg<2, PRO_GN_GELU>(a, g, s);
    }
    // 64-wide outputs: four waves along M, each 64 rows x all 64 columns (an A fragment feeds 12 MFMAs, as on 128-wide
    // tiles), tap-pair loop; K % 64 != 0 keeps the 2 x 2 arrangement with 128 x 32 waves
    if (a.K % 64 == 0 && !(a.sw & SW_WIDE_N64_2X2)) {
        if (pipe) {
            if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 4, 1, true>(a, g, s);
            if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN, false, 4, 1, true>(a, g, s);
            return launch_wide_cfg<2, PRO_GN_GELU, false, 4, 1, true>(a, g, s);
        }
        if (g2) {
            if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 4, 1, false, false, false, true>(a, g, s);
            if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN, false, 4, 1, false, false, false, true>(a, g, s);
            return launch_wide_cfg<2, PRO_GN_GELU, false, 4, 1, false, false, false, true>(a, g, s);
        }
        if (a.pro == PRO_NONE) return launch_wide_cfg<2, PRO_NONE, false, 4, 1>(a, g, s);
        if (a.pro == PRO_GN) return launch_wide_cfg<2, PRO_GN, false, 4, 1>(a, g, s);
        return launch_wide_cfg<2, PRO_GN_GELU, false, 4, 1>(a, g, s);
    }
    if (a.pro == PRO_NONE) return launch_wide_cfg<1, PRO_NONE>(a, g, s);
    if (a.pro == PRO_GN) return launch_wide_cfg<1, PRO_GN>(a, g, s);
    return launch_wide_cfg<1, PRO_GN_GELU>(a, g, s);
}

}  // namespace spdm
