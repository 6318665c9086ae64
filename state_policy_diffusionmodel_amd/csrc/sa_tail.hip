// sa_tail.hip -- the row-wise tail of a SelfAttention block with C = 128 (sa1, sa4) or C = 256 (sa2, sa3) channels in ONE kernel:
//
//   av  = o W_o^T + b_o + x                       (out_proj + residual,        models/Unet_FiLmLayer.py:79-80)
//   out = GELU(LayerNorm(av) W_1^T + b_1) W_2^T + b_2 + av      (ff_self + residual,               :63-68, :81)
//
// As three launches (conv_gemm_kernel<1 tap>) this chain moved 1.07 GB per step on sa1 -- every intermediate goes to
// HBM and comes back -- for 77 GFLOP; here a workgroup owns 64 token rows end to end: it reads o and x, writes out
// (0.4 GB), and the two intermediates live in LDS / registers.
//
//   * 4 waves, each a 32 x 64 accumulator tile (v_mfma_f32_16x16x32_f16, split-fp16 operands like every other
//     contraction of the path: hi*hi + hi*lo + lo*hi into fp32): 2 x 2 waves over 64 rows x 128 channels, or 1 x 4 over
//     32 rows x 256 channels -- same slab bytes, same registers per thread, same tile per wave in both shapes;
//   * the A operand of each product is a 64-row x 128-channel slab in LDS (conv_wide's row format: per 32-channel
//     chunk [32 x fp16 hi | 32 x fp16 lo], rows padded to 144 bytes); weights come straight from global memory in
//     fragment order (frag_order_weights, taps = 1), one 32-channel chunk ahead, also across the three products;
//   * between products the accumulators go through LDS once (the dead slab's space) and are picked up ROW-WISE --
//     thread (row 8 i + tid / 32, columns 4 (tid % 32) ..) for i = 0..7, the same assignment the loader uses -- so
//     bias, residual, LayerNorm (two-pass over the 128 values of a row: a half-wave holds one row), GELU and the fp16
//     split are register work, and av stays in 32 registers until the final residual.
#include <algorithm>
#include <cstdlib>

#include "device_utils.h"

namespace spdm {

namespace {

typedef float tf32x4 __attribute__((ext_vector_type(4)));
typedef float tf32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 tf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tf16x2 __attribute__((ext_vector_type(2)));

constexpr float T_ACT_SCALE = 16.0f;            // operand scales of the split scheme (activations 2^4, weights 2^7)
constexpr float T_DESCALE = 1.0f / 2048.0f;
constexpr int T_LDK = 36;                       // floats per LDS row of one 32-channel chunk (128 + 16 pad bytes)
// shape of one workgroup: TM token rows x C channels = 8192 values = 8 row pieces (4 channels each) per thread
template <int C> struct TailShape {
    static_assert(C == 128 || C == 256, "C = 128 or 256");
    static constexpr int TM = 8192 / C;             // 64 | 32 rows
    static constexpr int WM = TM / 32, WN = 4 / WM; // waves along rows / channels (each 32 x 64)
    static constexpr int NCH = C / 32;              // 32-channel chunks of a row
    static constexpr int TPR = C / 4, RPP = 256 / TPR;   // threads per row, rows per pick-up pass
    static constexpr int NP = TM / RPP;             // row pieces per thread (8)
};

struct SaTailArgs {
    const float* o;        // [M][C] attention output (heads concatenated)
    const float* x;        // [M][C] block input (residual of the out-projection)
    float* out;            // [M][C]
    int M;
    const float *wf_o, *wf_1, *wf_2;            // fragment-order split weights (C x C each)
    const float *b_o, *b_1, *b_2, *ln_g, *ln_b;
    const float* ab; int L;                     // optional [M / L][2][C]: x is y = A x + B per sample of L rows (FiLM tail folded in)
    FilmSpec fs;                                // fs.on: the coefficients are evaluated here, into LDS (ab is null then)
};

// (plain C++ form: two more instructions per pair, but fewer live registers -- sa_head_kernel sits exactly at the 128-register
//  line of four waves per SIMD, and the 4-instruction form pushed it over: 238 -> 265 us on sa1 at B = 4096)
__device__ __forceinline__ tf32x2 tsplit2_plain(float a, float b) {
    const float xa = a * T_ACT_SCALE, xb = b * T_ACT_SCALE;
    const _Float16 ha = (_Float16)xa, hb = (_Float16)xb;
    const tf16x2 h = {ha, hb};
    const tf16x2 l = {(_Float16)(xa - (float)ha), (_Float16)(xb - (float)hb)};
    return tf32x2{__builtin_bit_cast(float, h), __builtin_bit_cast(float, l)};
}
#define TAIL_SPLIT2 tsplit2
__device__ __forceinline__ tf32x2 tsplit2(float a, float b) {
    unsigned h, l;
    split_pair_f16(a * T_ACT_SCALE, b * T_ACT_SCALE, h, l);
    return tf32x2{__builtin_bit_cast(float, h), __builtin_bit_cast(float, l)};
}
// sum over the TPR consecutive lanes that hold one row (32: a half-wave; 64: the wave)
// (the first four butterfly steps inside the DPP rows -- VALU only; 16 and 32 through ds_bpermute: a LayerNorm phase of
//  8 row pieces x 2 sums x 5-6 bpermutes per thread was bound by the LDS crossbar; the row joins by v_permlane*_swap)
template <int TPR>
__device__ __forceinline__ float trow_sum(float v) {
    static_assert(TPR == 32 || TPR == 64, "rows of 32 or 64 lanes");
    v = sum_xor16(row16_sum_dpp(v));
    if (TPR == 64) v = sum_xor32(v);
    return v;
}

// FiLM tail folded into the load of the block input (film_coef_kernel): piece i (row m0 + RPP i + srow0, columns 4 cq ..)
// becomes A x + B with the coefficients of its sample; when the whole tile lies in one sample they are loaded once.
template <int C>
__device__ __forceinline__ void tail_film_fold(tf32x4 (&v)[8], const float* __restrict__ ab, int L, int m0, int M, int cq, int srow0) {
    using S = TailShape<C>;
    if (L % S::TM == 0) {
        const float* ab_ = ab + (size_t)(m0 / L) * 2 * C + cq * 4;
        const tf32x4 A = *reinterpret_cast<const tf32x4*>(ab_), B = *reinterpret_cast<const tf32x4*>(ab_ + C);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * A + B;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* ab_ = ab + (size_t)(min(m0 + S::RPP * i + srow0, M - 1) / L) * 2 * C + cq * 4;
            v[i] = v[i] * *reinterpret_cast<const tf32x4*>(ab_) + *reinterpret_cast<const tf32x4*>(ab_ + C);
        }
    }
}

// The same with the coefficients evaluated by this workgroup: abl[s] = [A | B] of sample b_first + s (LDS), filled by
// tail_film_rows below; the fold reads them after the barrier that follows.
template <int C>
__device__ __forceinline__ void tail_film_rows(const FilmSpec& fs, int L, int m0, int M, int wave, int lane, float* abl) {
    using S = TailShape<C>;
    const int b_first = m0 / L, b_last = (min(m0 + S::TM, M) - 1) / L;
    for (int sidx = wave; sidx <= b_last - b_first; sidx += 4) film_coef_row_wave(fs, b_first + sidx, lane, abl + (size_t)sidx * 2 * C);
}
template <int C>
__device__ __forceinline__ void tail_film_fold_lds(tf32x4 (&v)[8], const float* abl, int L, int m0, int M, int cq, int srow0) {
    using S = TailShape<C>;
    const int b_first = m0 / L;
    if (L % S::TM == 0) {
        const tf32x4 A = *reinterpret_cast<const tf32x4*>(abl + cq * 4), B = *reinterpret_cast<const tf32x4*>(abl + C + cq * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * A + B;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* ab_ = abl + (size_t)(min(m0 + S::RPP * i + srow0, M - 1) / L - b_first) * 2 * C + cq * 4;
            v[i] = v[i] * *reinterpret_cast<const tf32x4*>(ab_) + *reinterpret_cast<const tf32x4*>(ab_ + C);
        }
    }
}

#ifdef SPDM_DIAG_TAIL
__device__ unsigned long long g_tail_stamps[64];      // diagnostic builds: phase stamps of workgroup 0 (s_memrealtime, 10 ns ticks)
#define TAIL_STAMP(k_) if (blockIdx.x == 0 && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g_tail_stamps[(C == 256 ? 16 : 0) + (k_)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define TAIL_STAMP(k_)
#endif

template <int C>
__global__ __launch_bounds__(256, 2) void sa_tail_kernel(const SaTailArgs a) {
    TAIL_STAMP(0)
    using S = TailShape<C>;
    constexpr int T_C = C, T_M = S::TM, NCH = S::NCH, RPP = S::RPP, TPR = S::TPR;
    constexpr int RT = 2, CT = 4;                       // per wave: 32 rows x 64 columns = 2 x 4 tiles of 16 x 16
    constexpr int NP = S::NP;                           // row pieces per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / S::WN, wn = wave % S::WN, l16 = lane & 15, kg = lane >> 4;
    const int M = a.M;
    const int m0 = blockIdx.x * T_M;

    float* Abuf = smem;                                 // [NCH chunks][TM rows][T_LDK]  (36.9 KB)
    float* otile = smem;                                // [TM][C] fp32 (32 KB), aliases the slab between products

    // row-wise piece i of this thread: row RPP i + srow0, columns 4 c16 .. 4 c16 + 3
    const int c16 = tid % TPR, srow0 = tid / TPR;
    const int aoff0 = (wm * 32 + l16) * T_LDK + kg * 4;

    // B operands of 32-channel chunk kc of a C x C weight: blocks (kc C/16 + nb16) x {hi, lo} of 256 floats
    const size_t wlane = ((size_t)(wn * CT) * 2) * 256 + lane * 4;
    constexpr size_t WCHUNK = (size_t)(C / 16) * 2 * 256;
    tf16x8 fb[2][CT][2], fa[2][2];
#define TAIL_LOAD_B(slot_, w_, kc_)                                                                  \
    {                                                                                                \
        const float* p_ = (w_) + wlane + (size_t)(kc_) * WCHUNK;                                     \
        _Pragma("unroll") for (int c_ = 0; c_ < CT; ++c_) {                                         \
            fb[slot_][c_][0] = *reinterpret_cast<const tf16x8*>(p_ + c_ * 512);                      \
            fb[slot_][c_][1] = *reinterpret_cast<const tf16x8*>(p_ + c_ * 512 + 256);                \
        }                                                                                            \
    }
#define TAIL_LOAD_FA(slot_, kc_, rt_)                                                                \
    {                                                                                                \
        const float* p_ = Abuf + ((kc_) * T_M + (rt_) * 16) * T_LDK + aoff0;                         \
        fa[slot_][0] = *reinterpret_cast<const tf16x8*>(p_);                                         \
        fa[slot_][1] = *reinterpret_cast<const tf16x8*>(p_ + 16);                                    \
    }
    // slab <- the thread's 8 row pieces v_[i] (fp32), split to fp16 hi / lo
#define TAIL_WRITE_SLAB(v_)                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) {                                             \
        const tf32x2 p0_ = TAIL_SPLIT2(v_[i_].x, v_[i_].y), p1_ = TAIL_SPLIT2(v_[i_].z, v_[i_].w);   \
        float* rowp_ = Abuf + ((c16 >> 3) * T_M + RPP * i_ + srow0) * T_LDK;                         \
        *reinterpret_cast<tf32x2*>(rowp_ + (c16 & 7) * 2) = tf32x2{p0_.x, p1_.x};                    \
        *reinterpret_cast<tf32x2*>(rowp_ + 16 + (c16 & 7) * 2) = tf32x2{p0_.y, p1_.y};               \
    }
    // acc = slab . W^T over the NCH 32-channel chunks; the weights of `wnext_` chunk 0 are prefetched at the end
#define TAIL_GEMM(w_, wnext_)                                                                        \
    {                                                                                                \
        _Pragma("unroll") for (int rt_ = 0; rt_ < RT; ++rt_)                                        \
            _Pragma("unroll") for (int ct_ = 0; ct_ < CT; ++ct_) acc[rt_][ct_] = tf32x4{0.f, 0.f, 0.f, 0.f}; \
        TAIL_LOAD_FA(0, 0, 0)                                                                        \
        _Pragma("unroll") for (int kc = 0; kc < NCH; ++kc) {                                        \
            if (kc + 1 < NCH) { TAIL_LOAD_B((kc + 1) & 1, w_, kc + 1) } else { TAIL_LOAD_B(0, wnext_, 0) } \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                     \
                if (rt + 1 < RT) { TAIL_LOAD_FA((rt + 1) & 1, kc, rt + 1) }                          \
                else if (kc + 1 < NCH) { TAIL_LOAD_FA(0, kc + 1, 0) }                                \
                __builtin_amdgcn_sched_barrier(0);                                                   \
                _Pragma("unroll") for (int c = 0; c < CT; ++c)                                      \
                    acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[rt & 1][0], fb[kc & 1][c][0], acc[rt][c], 0, 0, 0); \
                _Pragma("unroll") for (int c = 0; c < CT; ++c)                                      \
                    acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[rt & 1][0], fb[kc & 1][c][1], acc[rt][c], 0, 0, 0); \
                _Pragma("unroll") for (int c = 0; c < CT; ++c)                                      \
                    acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[rt & 1][1], fb[kc & 1][c][0], acc[rt][c], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                   \
            }                                                                                        \
        }                                                                                            \
    }
    // accumulators -> LDS tile (after every wave is done reading the slab), visible to all on return
#define TAIL_ACC_TO_TILE()                                                                           \
    {                                                                                                \
        __syncthreads();                                                                             \
        _Pragma("unroll") for (int rt_ = 0; rt_ < RT; ++rt_)                                        \
            _Pragma("unroll") for (int ct_ = 0; ct_ < CT; ++ct_) {                                  \
                const int col_ = wn * 64 + ct_ * 16 + l16;                                           \
                _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                    \
                    otile[(wm * 32 + rt_ * 16 + 4 * kg + j_) * T_C + col_] = acc[rt_][ct_][j_] * T_DESCALE; \
            }                                                                                        \
        __syncthreads();                                                                             \
    }

    TAIL_LOAD_B(0, a.wf_o, 0)
    float* abl = smem + NCH * T_M * T_LDK;              // [samples of this tile][2 C]: FiLM coefficients evaluated here (fs.on)
    // the per-column parameters of every phase, fetched now: loaded where they are used, each cost its phase a memory round trip
    const tf32x4 bo = *reinterpret_cast<const tf32x4*>(a.b_o + c16 * 4);
    const tf32x4 g4 = *reinterpret_cast<const tf32x4*>(a.ln_g + c16 * 4);
    const tf32x4 b4 = *reinterpret_cast<const tf32x4*>(a.ln_b + c16 * 4);
    const tf32x4 b1 = *reinterpret_cast<const tf32x4*>(a.b_1 + c16 * 4);
    const tf32x4 b2 = *reinterpret_cast<const tf32x4*>(a.b_2 + c16 * 4);

    // ---- slab <- o (attention output), raw ----
    tf32x4 av[NP], v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = min(m0 + RPP * i + srow0, M - 1);
        v[i] = *reinterpret_cast<const tf32x4*>(a.o + (size_t)row * T_C + c16 * 4);
        if (m0 + RPP * i + srow0 >= M) v[i] = tf32x4{0.f, 0.f, 0.f, 0.f};
    }
    TAIL_WRITE_SLAB(v)
    // the residual rows x are fetched now, into the registers that will hold av: in flight during the first product
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = min(m0 + RPP * i + srow0, M - 1);
        av[i] = *reinterpret_cast<const tf32x4*>(a.x + (size_t)row * T_C + c16 * 4);
    }
    if (a.ab != nullptr) tail_film_fold<C>(av, a.ab, a.L, m0, M, c16, srow0);
    if (a.fs.on) tail_film_rows<C>(a.fs, a.L, m0, M, wave, lane, abl);          // (behind the o / x loads: one round trip, overlapped)
    __syncthreads();
    if (a.fs.on) tail_film_fold_lds<C>(av, abl, a.L, m0, M, c16, srow0);      // (av is register work until the first pick-up)
    TAIL_STAMP(1)

    tf32x4 acc[RT][CT];

    // ---- av = o W_o^T + b_o + x ;  slab <- LayerNorm(av) ----
    TAIL_GEMM(a.wf_o, a.wf_1)
    TAIL_STAMP(2)
    TAIL_ACC_TO_TILE()
    TAIL_STAMP(3)
    {
        // LayerNorm over each row's C values (two-pass, like torch): a row sits on TPR consecutive lanes.  The NP row pieces of a
        // thread go through the two cross-lane sums TOGETHER (NP independent shuffle chains in flight): one piece at a time the
        // phase was 3.4-3.7 us of a 13-16 us launch -- longer than any of the three products (tools/probes/tail_stamps.py).
        float part[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = RPP * i + srow0;
            tf32x4 t = *reinterpret_cast<const tf32x4*>(otile + r * T_C + c16 * 4);
            t += bo;
            t += av[i];
            av[i] = t;
            part[i] = (t.x + t.y) + (t.z + t.w);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = row16_sum_dpp(part[i]);
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = sum_xor16(part[i]);
        if (TPR == 64) {
#pragma unroll
            for (int i = 0; i < NP; ++i) part[i] = sum_xor32(part[i]);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float mean = part[i] * (1.0f / (float)C);
            const tf32x4 t = av[i];
            const tf32x4 d = {t.x - mean, t.y - mean, t.z - mean, t.w - mean};
            v[i] = d;
            part[i] = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = row16_sum_dpp(part[i]);
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = sum_xor16(part[i]);
        if (TPR == 64) {
#pragma unroll
            for (int i = 0; i < NP; ++i) part[i] = sum_xor32(part[i]);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float rstd = __builtin_amdgcn_rsqf(part[i] * (1.0f / (float)C) + 1e-5f);        // v_rsq_f32 (1 ulp)
            const tf32x4 d = v[i];
            v[i] = tf32x4{d.x * rstd * g4.x + b4.x, d.y * rstd * g4.y + b4.y, d.z * rstd * g4.z + b4.z, d.w * rstd * g4.w + b4.w};
        }
    }
    __syncthreads();                                    // every thread has read its pieces of the tile
    TAIL_WRITE_SLAB(v)
    __syncthreads();
    TAIL_STAMP(4)

    // ---- f1 = GELU(ln W_1^T + b_1) ;  slab <- f1 ----
    TAIL_GEMM(a.wf_1, a.wf_2)
    TAIL_STAMP(5)
    TAIL_ACC_TO_TILE()
    {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            tf32x4 t = *reinterpret_cast<const tf32x4*>(otile + (RPP * i + srow0) * T_C + c16 * 4);
            t += b1;
            v[i] = tf32x4{gelu_erf(t.x), gelu_erf(t.y), gelu_erf(t.z), gelu_erf(t.w)};      // (no scheduling fence between pieces: the
        }                                                                                   //  erf chains of the NP pieces interleave)
    }
    __syncthreads();
    TAIL_WRITE_SLAB(v)
    __syncthreads();
    TAIL_STAMP(6)

    // ---- out = f1 W_2^T + b_2 + av ----
    TAIL_GEMM(a.wf_2, a.wf_2)                           // (the trailing prefetch re-reads a valid block; unused)
    TAIL_STAMP(7)
    TAIL_ACC_TO_TILE()
    {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = RPP * i + srow0;
            if (m0 + r < M) {
                tf32x4 t = *reinterpret_cast<const tf32x4*>(otile + r * T_C + c16 * 4);
                t += b2;
                t += av[i];
                *reinterpret_cast<tf32x4*>(a.out + (size_t)(m0 + r) * T_C + c16 * 4) = t;
            }
        }
    }
    TAIL_STAMP(8)
}

// ---- the head of the same blocks:  qkv = LayerNorm(x) W_in^T + b_in   (self.ln -> mha in_proj, :76-79) -----------
// Same workgroup shape and machinery (the macros above): 64 token rows, LayerNorm computed from the row itself (two
// passes over its 128 values in a half-wave -- no statistics from the producer needed), ONE normalised slab feeding
// three 128-column products (q, k, v), each stored straight from its row-wise pick-up.  As conv_gemm_kernel<1 tap>
// this GEMM re-read the input once per 128-column tile and took its LayerNorm statistics from HBM.
struct SaQkvArgs {
    const float* x; float* qkv; int M;          // [M][C] -> [M][3 C]
    const float* wf;                            // fragment-order split in_proj weight (3 C x C)
    const float *b_in, *ln_g, *ln_b;
    const float* ab; int L;                     // optional FiLM-tail coefficients, as in SaTailArgs
    FilmSpec fs;
};

template <int C>
__global__ __launch_bounds__(256, 2) void sa_qkv_kernel(const SaQkvArgs a) {
    using S = TailShape<C>;
    constexpr int T_C = C, T_M = S::TM, NCH = S::NCH, RPP = S::RPP, TPR = S::TPR;
    constexpr int RT = 2, CT = 4;
    constexpr int NP = S::NP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / S::WN, wn = wave % S::WN, l16 = lane & 15, kg = lane >> 4;
    const int M = a.M;
    const int m0 = blockIdx.x * T_M;
    float* Abuf = smem;
    float* otile = smem;
    const int c16 = tid % TPR, srow0 = tid / TPR;
    const int aoff0 = (wm * 32 + l16) * T_LDK + kg * 4;
    // in_proj is 3 C x C: product g (q, k, v) uses rows C g .. -> 16-column blocks (C / 16) g ..; chunk stride = 3 C / 16 blocks
    const size_t wlane = ((size_t)(wn * CT) * 2) * 256 + lane * 4;
    constexpr size_t WCHUNK = (size_t)(3 * C / 16) * 2 * 256;
    constexpr size_t WPROD = (size_t)(C / 16) * 2 * 256;
    tf16x8 fb[2][CT][2], fa[2][2];
    tf32x4 acc[RT][CT], v[NP];

    // small grids: gridDim.y = 3 workgroups per row tile, one of q / k / v each (the weight stream of the launch through three
    // times as many CUs; the LayerNorm of the rows is recomputed by each)
    const int g_lo = (int)blockIdx.y * 3 / (int)gridDim.y, g_hi = ((int)blockIdx.y + 1) * 3 / (int)gridDim.y;
    TAIL_LOAD_B(0, a.wf + (size_t)g_lo * WPROD, 0)
    {
        const tf32x4 g4 = *reinterpret_cast<const tf32x4*>(a.ln_g + c16 * 4);
        const tf32x4 b4 = *reinterpret_cast<const tf32x4*>(a.ln_b + c16 * 4);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = min(m0 + RPP * i + srow0, M - 1);
            v[i] = *reinterpret_cast<const tf32x4*>(a.x + (size_t)row * T_C + c16 * 4);
        }
        if (a.ab != nullptr) tail_film_fold<C>(v, a.ab, a.L, m0, M, c16, srow0);
        if (a.fs.on) {          // coefficients evaluated here, in the (still dead) output-tile region behind the slab
            float* abl = smem + NCH * T_M * T_LDK;
            tail_film_rows<C>(a.fs, a.L, m0, M, wave, lane, abl);
            __syncthreads();
            tail_film_fold_lds<C>(v, abl, a.L, m0, M, c16, srow0);
        }
        // (the NP row pieces go through the two cross-lane sums together, as in sa_tail_kernel)
        float part[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = row16_sum_dpp(part[i]);
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = sum_xor16(part[i]);
        if (TPR == 64) {
#pragma unroll
            for (int i = 0; i < NP; ++i) part[i] = sum_xor32(part[i]);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float mean = part[i] * (1.0f / (float)C);
            const tf32x4 t = v[i];
            const tf32x4 d = {t.x - mean, t.y - mean, t.z - mean, t.w - mean};
            v[i] = d;
            part[i] = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = row16_sum_dpp(part[i]);
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = sum_xor16(part[i]);
        if (TPR == 64) {
#pragma unroll
            for (int i = 0; i < NP; ++i) part[i] = sum_xor32(part[i]);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float rstd = __builtin_amdgcn_rsqf(part[i] * (1.0f / (float)C) + 1e-5f);
            const tf32x4 d = v[i];
            v[i] = tf32x4{d.x * rstd * g4.x + b4.x, d.y * rstd * g4.y + b4.y, d.z * rstd * g4.z + b4.z, d.w * rstd * g4.w + b4.w};
        }
    }
    TAIL_WRITE_SLAB(v)
    __syncthreads();
    for (int g = g_lo; g < g_hi; ++g) {
        const float* wg = a.wf + (size_t)g * WPROD;
        const float* wnext = a.wf + (size_t)(g < 2 ? g + 1 : g) * WPROD;       // the last trailing prefetch re-reads a valid block
        const tf32x4 bi = *reinterpret_cast<const tf32x4*>(a.b_in + g * T_C + c16 * 4);     // (in flight during the product)
        TAIL_GEMM(wg, wnext)
        // accumulators -> LDS: the slab must survive for the next product, so the tile goes BEHIND it
        {
            float* ot = smem + NCH * T_M * T_LDK;
            if (g > g_lo) __syncthreads();              // every thread has picked up the previous product's rows
#pragma unroll
            for (int rt_ = 0; rt_ < RT; ++rt_)
#pragma unroll
                for (int ct_ = 0; ct_ < CT; ++ct_) {
                    const int col_ = wn * 64 + ct_ * 16 + l16;
#pragma unroll
                    for (int j_ = 0; j_ < 4; ++j_)
                        ot[(wm * 32 + rt_ * 16 + 4 * kg + j_) * T_C + col_] = acc[rt_][ct_][j_] * T_DESCALE;
                }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int r = RPP * i + srow0;
                if (m0 + r < M) {
                    tf32x4 t = *reinterpret_cast<const tf32x4*>(ot + r * T_C + c16 * 4);
                    t += bi;
                    *reinterpret_cast<tf32x4*>(a.qkv + (size_t)(m0 + r) * (3 * T_C) + g * T_C + c16 * 4) = t;
                }
            }
        }
    }
    (void)otile;
}

// ---- LayerNorm + in_proj + the attention core of the same blocks in ONE kernel (round 3) ----------------------------
// sa_qkv_kernel wrote q, k, v of every token (3 C floats per row: 403 MB at sa1, B = 4096) and the attention core read them
// back; both launches sat within 1.3 x of their own HBM floor, so the only traffic left to remove was the round trip itself.
// Here a workgroup owns TM token rows that are WHOLE samples (TM % L == 0: the host checks) and walks the four heads: per head
// three small products (TM x d each, d = C / 4) from the normalised slab, q / k / v^T of the head as split-fp16 tiles in LDS
// (24 KB), scores S = q k^T on the matrix cores over the tile's TM keys with the other samples' keys masked (block-diagonal:
// at L = 16 or 4 a tile holds 2-8 samples; the masked products are noise next to the memory traffic), softmax on the
// accumulator layout (a query row = 16 lanes of a DPP row x the key tiles), P through LDS into A-operand order, O = P v.
// Only the head outputs (C floats per row) leave the kernel.  Same split-fp16 scheme and scales as attention.hip
// (q, k, v x 16, p x 1024, hi*hi + hi*lo + lo*hi, fp32 accumulate); replaces models/Unet_FiLmLayer.py:76-79 for these blocks.
struct SaHeadArgs {
    const float* x; float* att; int M;          // [M][C] -> [M][C] (heads concatenated, before out_proj)
    const float* wf;                            // fragment-order split in_proj weight (3 C x C)
    const float *b_in, *ln_g, *ln_b;
    const float* ab; int L;                     // optional FiLM-tail coefficients, as in SaTailArgs; L = tokens per sample
    FilmSpec fs;
};

__device__ __forceinline__ float row16_max_dpp(float v) {
#define DPP_MAX(ctrl_) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl_, 0xf, 0xf, false)));
    DPP_MAX(0xB1) DPP_MAX(0x4E) DPP_MAX(0x141) DPP_MAX(0x140)
#undef DPP_MAX
    return v;
}
__device__ __forceinline__ float head_exp_neg(float x) {      // exp(x), x <= 0, through v_exp_f32 with a compensated argument
    const float t = x * 1.44269504f;
    const float tl = __fmaf_rn(x, 1.44269504f, -t) + x * 1.925963033e-8f;
    return __builtin_amdgcn_exp2f(t) * (1.0f + tl * 0.69314718f);
}

#undef TAIL_SPLIT2
#define TAIL_SPLIT2 tsplit2_plain
template <int C>
__global__ __launch_bounds__(256, 2) void sa_head_kernel(const SaHeadArgs a) {
    using S = TailShape<C>;
    constexpr int T_C = C, T_M = S::TM, NCH = S::NCH, RPP = S::RPP, TPR = S::TPR;
    constexpr int NP = S::NP;
    constexpr int D = C / 4;                            // head dimension: 32 | 64
    constexpr int QROW = 2 * D + 8;                     // halfs per row of the q / k tiles of a head: [D hi | D lo] + 16 bytes
    constexpr int VROW = 2 * T_M + 8;                   // halfs per row of v^T [D rows] and of P [TM rows]: [TM hi | TM lo] + 16 bytes
    constexpr int NKT = T_M / 16;                       // key tiles of the scores (4 | 2)
    constexpr int NDT = D / 16;                         // 16-column tiles of the head output (2 | 4)
    static_assert(T_M * VROW <= 2 * T_M * QROW, "P aliases the q / k tiles");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kg = lane >> 4;
    const int M = a.M, L = a.L;
    const int m0 = blockIdx.x * T_M;
    float* Abuf = smem;                                                     // [NCH][TM][T_LDK]: LayerNorm(x), split
    _Float16* Qs = reinterpret_cast<_Float16*>(smem + NCH * T_M * T_LDK);   // [TM][QROW]
    _Float16* Ks = Qs + T_M * QROW;                                         // [TM][QROW]
    _Float16* Vt = Ks + T_M * QROW;                                         // [D][VROW]
    _Float16* Ps = Qs;                                                      // [TM][VROW]: once the scores are in registers
    const int c16 = tid % TPR, srow0 = tid / TPR;

    // ---- slab <- LayerNorm(x) (the head of sa_qkv_kernel) ----
    {
        tf32x4 v[NP];
        const tf32x4 g4 = *reinterpret_cast<const tf32x4*>(a.ln_g + c16 * 4);
        const tf32x4 b4 = *reinterpret_cast<const tf32x4*>(a.ln_b + c16 * 4);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = min(m0 + RPP * i + srow0, M - 1);
            v[i] = *reinterpret_cast<const tf32x4*>(a.x + (size_t)row * T_C + c16 * 4);
        }
        if (a.ab != nullptr) tail_film_fold<C>(v, a.ab, a.L, m0, M, c16, srow0);
        if (a.fs.on) {
            float* abl = reinterpret_cast<float*>(Qs);           // (dead until the first head's tiles are written)
            tail_film_rows<C>(a.fs, a.L, m0, M, wave, lane, abl);
            __syncthreads();
            tail_film_fold_lds<C>(v, abl, a.L, m0, M, c16, srow0);
            __syncthreads();
        }
        float part[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = trow_sum<TPR>(part[i]);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float mean = part[i] * (1.0f / (float)C);
            const tf32x4 t = v[i];
            const tf32x4 d = {t.x - mean, t.y - mean, t.z - mean, t.w - mean};
            v[i] = d;
            part[i] = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) part[i] = trow_sum<TPR>(part[i]);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const float rstd = __builtin_amdgcn_rsqf(part[i] * (1.0f / (float)C) + 1e-5f);
            const tf32x4 d = v[i];
            v[i] = tf32x4{d.x * rstd * g4.x + b4.x, d.y * rstd * g4.y + b4.y, d.z * rstd * g4.z + b4.z, d.w * rstd * g4.w + b4.w};
        }
        TAIL_WRITE_SLAB(v)
    }
    __syncthreads();

    // product tile of this wave: 16 rows x 32 columns of a head's q, k or v  (C = 128: 4 row tiles x the head's 32 columns;
    // C = 256: 2 row tiles x 2 halves of the head's 64 columns)
    const int rtw = (C == 128) ? wave : (wave >> 1);
    const int cpw = (C == 128) ? 0 : (wave & 1);
    const bool attn_wave = wave < NKT;                  // the waves that own 16 query rows (C = 256: two of the four)
    const float qscale = (D == 32) ? 0.17677669529663687f : 0.125f;         // 1 / sqrt(d)
    constexpr size_t WCHUNK = (size_t)(3 * C / 16) * 2 * 256;               // floats per 32-channel chunk of the fragment-order in_proj

    // weight fragments of a product: two 16-column tiles x {hi, lo} per 32-channel chunk, double-buffered; the FIRST chunk of the
    // next product (also across the attention phase, into the next head) is requested during the last chunk of the current one --
    // twelve products per workgroup each starting with an exposed L2 round trip were a third of the kernel
    tf16x8 fbw[2][2][2];
#define HEAD_WPTR(h_, g_) (a.wf + ((size_t)((g_) * (C / 16) + (h_) * (D / 16) + cpw * 2) * 2) * 256 + lane * 4)
#define HEAD_LOAD_W(slot_, wp_, kc_)                                                                 \
    _Pragma("unroll") for (int ct_ = 0; ct_ < 2; ++ct_) {                                           \
        fbw[slot_][ct_][0] = *reinterpret_cast<const tf16x8*>((wp_) + (size_t)(kc_) * WCHUNK + ct_ * 512);        \
        fbw[slot_][ct_][1] = *reinterpret_cast<const tf16x8*>((wp_) + (size_t)(kc_) * WCHUNK + ct_ * 512 + 256);  \
    }
    HEAD_LOAD_W(0, HEAD_WPTR(0, 0), 0)
    for (int h = 0; h < 4; ++h) {
        // ---- q, k, v of head h: three TM x d products from the slab ----
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            tf32x4 acc[2] = {tf32x4{0.f, 0.f, 0.f, 0.f}, tf32x4{0.f, 0.f, 0.f, 0.f}};
            const float* wp = HEAD_WPTR(h, g);
            const float* wnext = (g < 2) ? HEAD_WPTR(h, g + 1) : HEAD_WPTR(min(h + 1, 3), 0);     // (after the last product: a valid re-read)
#pragma unroll
            for (int kc = 0; kc < NCH; ++kc) {
                if (kc + 1 < NCH) { HEAD_LOAD_W((kc + 1) & 1, wp, kc + 1) } else { HEAD_LOAD_W(0, wnext, 0) }
                const float* ap = Abuf + (kc * T_M + rtw * 16 + l16) * T_LDK + kg * 4;
                const tf16x8 a_h = *reinterpret_cast<const tf16x8*>(ap), a_l = *reinterpret_cast<const tf16x8*>(ap + 16);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, fbw[kc & 1][ct][0], acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, fbw[kc & 1][ct][1], acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l, fbw[kc & 1][ct][0], acc[ct], 0, 0, 0);
                }
            }
            // bias (+ 1 / sqrt d on q), x 16, split, into the head's tile: q, k row-major, v transposed
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = cpw * 32 + ct * 16 + l16;                    // column inside the head
                const float bias = a.b_in[g * T_C + h * D + col];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = rtw * 16 + 4 * kg + j;
                    float val = acc[ct][j] * T_DESCALE + bias;
                    if (g == 0) val *= qscale;
                    const float xs = val * T_ACT_SCALE;
                    const _Float16 hi = (_Float16)xs, lo = (_Float16)(xs - (float)hi);
                    if (g == 2) { Vt[col * VROW + row] = hi; Vt[col * VROW + T_M + row] = lo; }
                    else { _Float16* X = (g == 0) ? Qs : Ks; X[row * QROW + col] = hi; X[row * QROW + D + col] = lo; }
                }
            }
        }
        __syncthreads();

        // ---- S = q k^T over the tile's TM keys (this wave's 16 query rows) ----
        tf32x4 sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) sc[kt] = tf32x4{0.f, 0.f, 0.f, 0.f};
        if (attn_wave) {
#pragma unroll
            for (int ks = 0; ks < D / 32; ++ks) {
                const _Float16* qp = Qs + (wave * 16 + l16) * QROW + 32 * ks + 8 * kg;
                const tf16x8 q_h = *reinterpret_cast<const tf16x8*>(qp), q_l = *reinterpret_cast<const tf16x8*>(qp + D);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    const _Float16* kp = Ks + (kt * 16 + l16) * QROW + 32 * ks + 8 * kg;
                    const tf16x8 k_h = *reinterpret_cast<const tf16x8*>(kp), k_l = *reinterpret_cast<const tf16x8*>(kp + D);
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q_h, k_h, sc[kt], 0, 0, 0);
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q_h, k_l, sc[kt], 0, 0, 0);
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q_l, k_h, sc[kt], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                // every wave has its scores: the q / k tiles may be overwritten by P

        // ---- softmax over the keys of the query's own sample; P x 1024, split, into A-operand rows ----
        if (attn_wave) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wave * 16 + 4 * kg + j;
                const int smp = row / L;
                float s[NKT];
                float mx = -3.0e38f;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    const bool own = ((kt * 16 + l16) / L) == smp;
                    s[kt] = own ? sc[kt][j] * (1.0f / 256.0f) : -3.0e38f;
                    mx = fmaxf(mx, s[kt]);
                }
                mx = row16_max_dpp(mx);
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    s[kt] = (s[kt] > -1.0e38f) ? head_exp_neg(s[kt] - mx) : 0.f;
                    sum += s[kt];
                }
                sum = row16_sum_dpp(sum);
                const float inv = 1024.0f / sum;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    const float ps = s[kt] * inv;
                    const _Float16 hi = (_Float16)ps, lo = (_Float16)(ps - (float)hi);
                    Ps[row * VROW + kt * 16 + l16] = hi;
                    Ps[row * VROW + T_M + kt * 16 + l16] = lo;
                }
            }
        }
        __syncthreads();

        // ---- O = P v, out of the kernel ----
        if (attn_wave) {
            tf32x4 o[NDT];
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) o[dt] = tf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < T_M / 32; ++ks) {
                const _Float16* pp = Ps + (wave * 16 + l16) * VROW + 32 * ks + 8 * kg;
                const tf16x8 p_h = *reinterpret_cast<const tf16x8*>(pp), p_l = *reinterpret_cast<const tf16x8*>(pp + T_M);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const _Float16* vp = Vt + (dt * 16 + l16) * VROW + 32 * ks + 8 * kg;
                    const tf16x8 v_h = *reinterpret_cast<const tf16x8*>(vp), v_l = *reinterpret_cast<const tf16x8*>(vp + T_M);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_h, v_h, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_h, v_l, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_l, v_h, o[dt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = m0 + wave * 16 + 4 * kg + j;
                    if (row < M) a.att[(size_t)row * T_C + h * D + dt * 16 + l16] = o[dt][j] * (1.0f / 16384.0f);
                }
        }
        __syncthreads();                                // the head's tiles are free for the next head
    }
#undef TAIL_LOAD_B
#undef TAIL_LOAD_FA
#undef TAIL_WRITE_SLAB
#undef TAIL_GEMM
#undef TAIL_ACC_TO_TILE
}

}  // namespace

#ifdef SPDM_DIAG_TAIL
extern "C" int spdm_debug_tail_stamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_tail_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

bool sa_tail_supported(int C, unsigned sw) { return (C == 128 || C == 256) && !(sw & SW_NO_SA_TAIL); }

// samples a tile of TM rows can touch when samples are L rows long
static int film_rows_cap(int TM, int L) { return (TM + L - 2) / L + 1; }
// In-kernel FiLM coefficients: one [A | B] row of 2 C floats in LDS per touched sample.  sa_qkv keeps them in its output-tile
// region (32 KB, dead until the first product is done), sa_tail behind its slab (36.9 KB + rows must leave two workgroups per CU).
bool sa_tail_film_local(int C, int L) {
    if (!(C == 128 || C == 256) || L < 1) return false;
    return (size_t)film_rows_cap(8192 / C, L) * 2 * C * sizeof(float) <= (size_t)24 * 1024;
}

hipError_t launch_sa_tail(int C, const float* o, const float* x, float* out, int rows, const float* wf_o, const float* wf_1,
                          const float* wf_2, const float* b_o, const float* b_1, const float* b_2, const float* ln_g,
                          const float* ln_b, const float* ab, int L, hipStream_t s, const FilmSpec* fs) {
    if (fs && (ab || L <= 0 || !sa_tail_film_local(C, L) || fs->C != C)) return hipErrorInvalidValue;
    if (rows <= 0 || (ab && L <= 0) || !o || !x || !out || !wf_o || !wf_1 || !wf_2 || !b_o || !b_1 || !b_2 || !ln_g || !ln_b)
        return hipErrorInvalidValue;
    if (C != 128 && C != 256) return hipErrorInvalidValue;
    SaTailArgs a{};
    a.o = o; a.x = x; a.out = out; a.M = rows;
    a.wf_o = wf_o; a.wf_1 = wf_1; a.wf_2 = wf_2;
    a.b_o = b_o; a.b_1 = b_1; a.b_2 = b_2; a.ln_g = ln_g; a.ln_b = ln_b; a.ab = ab; a.L = L;
    if (fs) { a.fs = *fs; a.fs.on = 1; }
    const int TM = 8192 / C;
    const size_t lds = (size_t)(C / 32) * TM * T_LDK * sizeof(float)           // 36.9 KB
                       + (fs ? (size_t)film_rows_cap(TM, L) * 2 * C * sizeof(float) : 0);
    if (C == 128) hipLaunchKernelGGL(sa_tail_kernel<128>, dim3((rows + TM - 1) / TM), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(sa_tail_kernel<256>, dim3((rows + TM - 1) / TM), dim3(256), lds, s, a);
    return hipGetLastError();
}

// Which blocks run LayerNorm + in_proj + attention core as ONE kernel: the tile of 8192 / C rows must hold whole samples
bool sa_head_supported(int C, int L, unsigned sw) {
    return (C == 128 || C == 256) && L >= 1 && (8192 / C) % L == 0 && !(sw & (SW_NO_SA_TAIL | SW_NO_SA_HEAD));
}

hipError_t launch_sa_head(int C, const float* x, float* att, int rows, const float* wf_in, const float* b_in, const float* ln_g,
                          const float* ln_b, const float* ab, int L, hipStream_t s, const FilmSpec* fs) {
    if (rows <= 0 || !x || !att || !wf_in || !b_in || !ln_g || !ln_b || !sa_head_supported(C, L, 0) || rows % L != 0) return hipErrorInvalidValue;
    if (fs && (ab || !sa_tail_film_local(C, L) || fs->C != C)) return hipErrorInvalidValue;
    SaHeadArgs a{};
    a.x = x; a.att = att; a.M = rows; a.wf = wf_in; a.b_in = b_in; a.ln_g = ln_g; a.ln_b = ln_b; a.ab = ab; a.L = L;
    if (fs) { a.fs = *fs; a.fs.on = 1; }
    const int TM = 8192 / C, D = C / 4;
    const size_t lds = (size_t)(C / 32) * TM * T_LDK * sizeof(float) + ((size_t)2 * TM * (2 * D + 8) + (size_t)D * (2 * TM + 8)) * sizeof(_Float16);
    // (in-kernel FiLM coefficient rows sit in the q / k / v^T region, dead until the first head's tiles are written)
    if (fs && (size_t)film_rows_cap(TM, L) * 2 * C * sizeof(float) > ((size_t)2 * TM * (2 * D + 8) + (size_t)D * (2 * TM + 8)) * sizeof(_Float16)) return hipErrorInvalidValue;
    const void* kern = C == 128 ? reinterpret_cast<const void*>(sa_head_kernel<128>) : reinterpret_cast<const void*>(sa_head_kernel<256>);
    if (hipError_t e = allow_full_lds(kern); e != hipSuccess) return e;
    const int tiles = (rows + TM - 1) / TM;
    if (C == 128) hipLaunchKernelGGL(sa_head_kernel<128>, dim3(tiles), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(sa_head_kernel<256>, dim3(tiles), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_sa_qkv(int C, const float* x, float* qkv, int rows, const float* wf_in, const float* b_in, const float* ln_g,
                         const float* ln_b, const float* ab, int L, hipStream_t s, const FilmSpec* fs) {
    if (fs && (ab || L <= 0 || !sa_tail_film_local(C, L) || fs->C != C)) return hipErrorInvalidValue;
    if (rows <= 0 || (ab && L <= 0) || !x || !qkv || !wf_in || !b_in || !ln_g || !ln_b) return hipErrorInvalidValue;
    if (C != 128 && C != 256) return hipErrorInvalidValue;
    SaQkvArgs a{};
    a.x = x; a.qkv = qkv; a.M = rows; a.wf = wf_in; a.b_in = b_in; a.ln_g = ln_g; a.ln_b = ln_b; a.ab = ab; a.L = L;
    if (fs) { a.fs = *fs; a.fs.on = 1; }
    const int TM = 8192 / C;
    const size_t lds = (size_t)((C / 32) * TM * T_LDK + TM * C) * sizeof(float);     // slab 36.9 KB + output tile 32 KB
    const void* kern = C == 128 ? reinterpret_cast<const void*>(sa_qkv_kernel<128>) : reinterpret_cast<const void*>(sa_qkv_kernel<256>);
    if (hipError_t e = allow_full_lds(kern); e != hipSuccess) return e;
    const int tiles = (rows + TM - 1) / TM;
    const int gy = (tiles < spdm_tune(15, 512)) ? 3 : 1;      // q, k, v on a workgroup each while the grid is small
    if (C == 128) hipLaunchKernelGGL(sa_qkv_kernel<128>, dim3(tiles, gy), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(sa_qkv_kernel<256>, dim3(tiles, gy), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace spdm
