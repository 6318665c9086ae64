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

__device__ __forceinline__ tf32x2 tsplit2(float a, float b) {
    const float xa = a * T_ACT_SCALE, xb = b * T_ACT_SCALE;
    const _Float16 ha = (_Float16)xa, hb = (_Float16)xb;
    const tf16x2 h = {ha, hb};
    const tf16x2 l = {(_Float16)(xa - (float)ha), (_Float16)(xb - (float)hb)};
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
        const tf32x2 p0_ = tsplit2(v_[i_].x, v_[i_].y), p1_ = tsplit2(v_[i_].z, v_[i_].w);           \
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
