// sa_fused.hip -- the whole SelfAttention block (models/Unet_FiLmLayer.py:71-82) of one trajectory in
// ONE kernel, for the C = 64 levels (sa5: L = H/2 * 4, sa6: L = H * 8 tokens; L <= 512, two workgroups per trajectory
// above 256):
//
//   ln  = LayerNorm(x)                     qkv = ln W_in^T + b_in
//   o_h = softmax(q_h k_h^T / sqrt d) v_h  (4 heads, d = 16)
//   av  = [o_0..o_3] W_o^T + b_o + x       out = GELU(LayerNorm(av) W_1^T + b_1) W_2^T + b_2 + av
//
// Everything is computed TRANSPOSED -- features on the MFMA rows (accumulator registers), tokens on the
// MFMA columns (lanes): Z^T = W Y^T.  One wave owns 32 tokens; lane l and lane l^32 hold the same token
// and complementary feature rows ((reg&3) + 8 (reg>>2) + 4 (l>>5) inside each 32-row tile).  With that
// orientation
//   * a 32x32 accumulator tile IS the B operand of the next product (its rows are the summed index):
//     registers 8s..8s+7 are the fragment of k-step s, so LN -> QKV -> out-proj -> LN -> FF1 -> GELU -> FF2
//     chain through registers with no LDS or HBM round trip; residuals stay in registers too;
//   * LayerNorm and softmax are per-lane register reductions plus one cross-half shuffle;
//   * the weights are the A operands, read as 16-byte fragments straight from L2 (98 KB per layer).
// Element j of lane half h in k-step s is row 16 s + 8 (j>>2) + 4 h + (j&3) of the producing tile; the host
// stores every weight matrix with its input-feature axis permuted by that map inside each group of 16
// (perm16 = 0 1 2 3 8 9 10 11 | 4 5 6 7 12 13 14 15) so a fragment is 16 contiguous bytes.
// Only K and V^T of the current head go through LDS (shared by the waves of the trajectory):
// 2 workgroup barriers per head.  All products use the split-fp16 scheme of conv_gemm.hip (fp16 hi + lo of
// the pre-scaled operand, three v_mfma_f32_32x32x16_f16, fp32 accumulate): activations x16, weights x128,
// probabilities x1024.
#include <algorithm>
#include <cstdlib>

#include "device_utils.h"

namespace spdm {

typedef float s_f32x16 __attribute__((ext_vector_type(16)));
typedef float s_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 s_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s_f16x4 __attribute__((ext_vector_type(4)));

constexpr int SA_C = 64;          // 4 heads of d = 16
// 1: issue the S product of key block kb + 1 before the softmax of block kb (software pipelining inside the wave).
// Measured 9.32 vs 9.28 ms per step without it (three alternating runs): the partner wave on the SIMD already covers
// the softmax with its own MFMAs, so the default stays 0.
#ifndef SA_SCORES_AHEAD
#define SA_SCORES_AHEAD 0
#endif
constexpr int SA_KROW = 24;          // halfs per K row in LDS (16 + 8 pad: conflict-free ds_read_b128)
constexpr float SA_DESCALE = 1.0f / 2048.0f;      // act x16, weight x128

struct SaFusedArgs {
    const float* x; float* out; int L;
    int nb;                // trajectories (WLDS launches: persistent workgroups walk b = blockIdx.x, + gridDim.x, ...)
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const _Float16 *wqkv_h, *wqkv_l, *wo_h, *wo_l, *w1_h, *w1_l, *w2_h, *w2_l;   // [rows][64] permuted, x128, in fragment order
    const float *bqkv, *bo, *b1, *b2;
    const float* ab;       // optional [B][2][64]: the block input is y = A x + B per sample (FiLM tail folded into the load)
    FilmSpec fs;           // fs.on: ... with the coefficients evaluated here (wave 0, film_coef_row_wave) instead of read from ab
};

typedef unsigned s_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 s_f16x2 __attribute__((ext_vector_type(2)));
// hi / lo fragments of 8 values f(0..7): four split_pair_f16 (device_utils.h: 4 instructions per pair)
template <typename F>
__device__ __forceinline__ void sa_split8(s_f16x8& h, s_f16x8& l, F f) {
    s_u32x4 hu, lu;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned a, b;
        split_pair_f16(f(2 * q), f(2 * q + 1), a, b);
        hu[q] = a;
        lu[q] = b;
    }
    h = __builtin_bit_cast(s_f16x8, hu);
    l = __builtin_bit_cast(s_f16x8, lu);
}
__device__ __forceinline__ float sa_exp_neg(float x) {
    const float t = x * 1.44269504f;
    const float tl = __fmaf_rn(x, 1.44269504f, -t) + x * 1.925963033e-8f;
    return __builtin_amdgcn_exp2f(t) * (1.0f + tl * 0.69314718f);
}
// B fragments (k-steps 2T, 2T+1) of a 2-tile activation set z[2] (C = 64 features), pre-scale 16
__device__ __forceinline__ void sa_make_frags(const s_f32x16 (&z)[2], s_f16x8 (&bh)[4], s_f16x8 (&bl)[4]) {
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int s = 0; s < 2; ++s) sa_split8(bh[2 * T + s], bl[2 * T + s], [&](int j) { return z[T][8 * s + j] * 16.0f; });
}
// acc (+)= W[row0 + li][0..63] . B   (one 32-row output tile, K = 64 = 4 k-steps)
__device__ __forceinline__ s_f32x16 sa_gemm_tile(const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl, int row0,
                                                  int li, int kh, const s_f16x8 (&bh)[4], const s_f16x8 (&bl)[4], s_f32x16 acc) {
    // fragment order (spdm_api.hip perm_split): block (row0 / 32) * 4 + ks holds, for lane kh * 32 + li, the 8 halfs
    // W[row0 + li][16 ks + 8 kh ..] -- one contiguous 1 KiB per wave load
    const _Float16* ph = Wh + ((size_t)(row0 >> 5) * 256 + 32 * kh + li) * 8;
    const _Float16* pl = Wl + ((size_t)(row0 >> 5) * 256 + 32 * kh + li) * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const s_f16x8 ah = *reinterpret_cast<const s_f16x8*>(ph + 512 * ks);
        const s_f16x8 al = *reinterpret_cast<const s_f16x8*>(pl + 512 * ks);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc, 0, 0, 0);
    }
    return acc;
}
// same product with the weight rows staged in LDS (rows padded 128 -> 144 bytes: 32 rows x 16 bytes conflict-free)
constexpr int SA_WROW = 72;          // halfs per staged weight row
__device__ __forceinline__ s_f32x16 sa_gemm_tile_lds(const _Float16* Wh, const _Float16* Wl, int row0, int li, int kh,
                                                      const s_f16x8 (&bh)[4], const s_f16x8 (&bl)[4], s_f32x16 acc) {
    const _Float16* ph = Wh + (row0 + li) * SA_WROW + 8 * kh;
    const _Float16* pl = Wl + (row0 + li) * SA_WROW + 8 * kh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const s_f16x8 ah = *reinterpret_cast<const s_f16x8*>(ph + 16 * ks);
        const s_f16x8 al = *reinterpret_cast<const s_f16x8*>(pl + 16 * ks);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], acc, 0, 0, 0);
    }
    return acc;
}
// cooperative copy of nrows weight rows (64 halfs each) global -> LDS rows dst_row0.. (16 bytes per thread per step)
__device__ __forceinline__ void sa_stage_rows(_Float16* dst, int dst_row0, const _Float16* __restrict__ src, int nrows, int tid, int nthr) {
    // (the source is in fragment order: unit i = block (rt, ks), lane (kh, li) -> row 32 rt + li, halfs 16 ks + 8 kh ..)
    for (int i = tid; i < nrows * 8; i += nthr) {
        const int blk = i >> 6, ln = i & 63;
        const int r = 32 * (blk >> 2) + (ln & 31), c = 2 * (blk & 3) + (ln >> 5);
        *reinterpret_cast<s_f16x8*>(dst + (dst_row0 + r) * SA_WROW + 8 * c) = *reinterpret_cast<const s_f16x8*>(src + (size_t)i * 8);
    }
}
// z[T][r] -> z * DESCALE + bias[feature]   (feature of register r in tile T: 32 T + (r&3) + 8 (r>>2) + 4 kh)
__device__ __forceinline__ void sa_bias(s_f32x16& z, const float* __restrict__ bias, int T, int kh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const s_f32x4 b = *reinterpret_cast<const s_f32x4*>(bias + 32 * T + 8 * g + 4 * kh);
#pragma unroll
        for (int j = 0; j < 4; ++j) z[4 * g + j] = z[4 * g + j] * SA_DESCALE + b[j];
    }
}
// LayerNorm over the 64 features of this lane's token (two-pass, like torch): z -> (z - mean) rstd gamma + beta
__device__ __forceinline__ void sa_layernorm(const s_f32x16 (&z)[2], s_f32x16 (&y)[2], const float* __restrict__ gamma,
                                             const float* __restrict__ beta, int kh) {
    float s = 0.f;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += z[T][r];
    s = sum_xor32(s);
    const float mean = s * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = z[T][r] - mean;
            q += d * d;
        }
    q = sum_xor32(q);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const s_f32x4 ga = *reinterpret_cast<const s_f32x4*>(gamma + 32 * T + 8 * g + 4 * kh);
            const s_f32x4 be = *reinterpret_cast<const s_f32x4*>(beta + 32 * T + 8 * g + 4 * kh);
#pragma unroll
            for (int j = 0; j < 4; ++j) y[T][4 * g + j] = (z[T][4 * g + j] - mean) * rstd * ga[j] + be[j];
        }
}

// WLDS: the workgroup stages the weight matrices in LDS once (256 + 128 rows, two phases) and every wave reads its A
// fragments from there -- for the long-sequence block (8 waves per trajectory) this replaces 8 x 98 KB of per-wave
// L2 reads, whose latency the 2 waves per SIMD could not hide (40 % of wave cycles parked), by one 98 KB copy.
// PAIR (256 < L <= 512, horizons up to 64): TWO workgroups per trajectory, each owning 256 query tokens.  A block needs
// K and V of all tokens, so every wave also projects K and V of its 32 partner tokens (same lane, other half) -- 2 extra
// 32-row tile products per head pair, no exchange between workgroups.
#ifdef SPDM_DIAG_SAF
// diagnostic builds: phase stamps (s_memrealtime, 10-ns ticks) of workgroup 0, thread 0; FULL (L = 256) at 0.., the other at 32..
__device__ unsigned long long g_saf_stamps[64];
#define SAF_STAMP() if (blockIdx.x == 0 && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        if (saf_n < 30) g_saf_stamps[(FULL ? 0 : 32) + (saf_n++)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define SAF_STAMP()
#endif

template <bool FULL, bool WLDS, bool PAIR>
__global__ __launch_bounds__(512, 2) void sa_fused64_kernel(const SaFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sa_smem[];
#ifdef SPDM_DIAG_SAF
    int saf_n = 0;
#endif
    SAF_STAMP()
    const int L = a.L;
    const int nwave = blockDim.x >> 6;
    const int Lp = PAIR ? 2 * nwave * 32 : nwave * 32;
    const int VROW = Lp + 8;                       // halfs per V^T row (16-byte aligned rows, 4-bank skew between rows)
    // LDS layout.  Without staged weights: K hi | K lo | V^T hi | V^T lo | FiLM row.  WLDS: V^T hi | V^T lo | W hi | W lo |
    // FiLM row | K hi | K lo | spare -- the qkv / out-proj weights ([256][SA_WROW] hi and lo) stay resident for every trajectory
    // the workgroup walks; ff1 / ff2 ([128][SA_WROW] hi and lo) are staged per trajectory over the K region + spare, which is
    // dead once the last head's attention loop is done.
    _Float16* const lds0 = reinterpret_cast<_Float16*>(sa_smem);
    _Float16* Vhi = WLDS ? lds0 : lds0 + (size_t)2 * Lp * SA_KROW;      // [32][VROW], rows 16..31 constant
    _Float16* Vlo = Vhi + (size_t)32 * VROW;
    _Float16* Wsh = Vlo + (size_t)32 * VROW;       // WLDS: [256][SA_WROW] hi rows: qkv 0..191, out-proj 192..255
    _Float16* Wsl = Wsh + (size_t)256 * SA_WROW;   //        [256][SA_WROW] lo rows
    _Float16* Khi = WLDS ? Wsl + (size_t)256 * SA_WROW + 4 * SA_C : lds0;             // (4 SA_C halfs = the FiLM row's 2 SA_C floats)
    _Float16* Klo = Khi + (size_t)Lp * SA_KROW;
    _Float16* Fsh = Khi;                           // WLDS: [128][SA_WROW] hi rows of ff1 (0..63) and ff2 (64..127)
    _Float16* Fsl = Fsh + (size_t)128 * SA_WROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int b_first = PAIR ? blockIdx.x >> 1 : blockIdx.x;
    const int hf = PAIR ? (blockIdx.x & 1) : 0;
    const int t = hf * 256 + wave * 32 + li;       // this lane's token
    const int t2 = (1 - hf) * 256 + wave * 32 + li;        // PAIR: the partner token whose K / V this lane also projects
    const int tp2 = (t2 & ~12) | ((t2 & 4) << 1) | ((t2 & 8) >> 1);
    // position of token t in a V^T row: inside each group of 16 keys the order is [0-3, 8-11, 4-7, 12-15] (bits 2 and 3
    // swapped), which makes the 8 keys a lane half feeds to one P.V MFMA (4 kh + 0..3 and 8 + 4 kh + 0..3) contiguous
    const int tp = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1);
    const int tc = min(t, L - 1);

    // rows 16..31 of V^T are written once (never overwritten): the d = 16 heads fill only half of a 32-row A tile.  Rows 16
    // and 20 are all ONES (hi part), the rest zero: the P.V product then leaves sum_keys p in accumulator register 8 of both
    // lane halves (row (r&3) + 8 (r>>2) + 4 kh) -- the softmax denominator comes out of the matrix pipe, which idles in this
    // VALU-bound loop, instead of 16 adds per key block; the running rescale by alpha covers it like every other row.
    for (int i = tid; i < 16 * VROW; i += blockDim.x) {
        const int row = i / VROW;
        Vhi[16 * VROW + i] = (row == 0 || row == 4) ? (_Float16)1.0f : (_Float16)0.f;
        Vlo[16 * VROW + i] = (_Float16)0.f;
    }

    if (WLDS) {
        sa_stage_rows(Wsh, 0, a.wqkv_h, 192, tid, blockDim.x);
        sa_stage_rows(Wsl, 0, a.wqkv_l, 192, tid, blockDim.x);
        sa_stage_rows(Wsh, 192, a.wo_h, 64, tid, blockDim.x);
        sa_stage_rows(Wsl, 192, a.wo_l, 64, tid, blockDim.x);
        __syncthreads();
    }

    // WLDS: a persistent workgroup (the launch has one per CU) walks trajectories b_first, b_first + gridDim.x, ... with the
    // weights above staged ONCE -- re-staging the same 74 KB per trajectory was 6 of the 52 us a trajectory took at B = 4096
    // (phase stamps, profiles/r02_sa_fused_phases.txt).  Otherwise one trajectory per workgroup.
    const int bstep = WLDS ? (int)gridDim.x : a.nb;
    for (int b = b_first; b < a.nb; b += bstep) {
    if (b != b_first) __syncthreads();              // every wave is done with the previous trajectory (FiLM row, ff weights)
    const float* xrow = a.x + ((size_t)b * L + tc) * SA_C;

    // x^T tiles of a token: register r of tile T = feature 32 T + (r&3) + 8 (r>>2) + 4 kh.  They are (re)loaded where they
    // are used (twice for LayerNorm 1, once for the residual: L2 hits) instead of living in 32 registers throughout.
#define SA_LOAD_X(dst_, row_)                                                                            \
    _Pragma("unroll") for (int T_ = 0; T_ < 2; ++T_)                                                    \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                              \
            s_f32x4 v_ = *reinterpret_cast<const s_f32x4*>((row_) + 32 * T_ + 8 * g_ + 4 * kh);          \
            if (fold) {                                                                                  \
                const s_f32x4 A_ = *reinterpret_cast<const s_f32x4*>(ab_s + 32 * T_ + 8 * g_ + 4 * kh); \
                const s_f32x4 B_ = *reinterpret_cast<const s_f32x4*>(ab_s + SA_C + 32 * T_ + 8 * g_ + 4 * kh); \
                v_ = v_ * A_ + B_;                                                                       \
            }                                                                                            \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) dst_[T_][4 * g_ + j_] = v_[j_];           \
        }
    // FiLM tail folded into the load (film_coef_kernel): the sample's [A | B] sits in LDS, every x load applies it.
    // (not in PAIR mode: its register budget is full, the plan keeps film_apply there)
    float* ab_s = reinterpret_cast<float*>(WLDS ? Wsl + (size_t)256 * SA_WROW : Wsh);     // the FiLM row (see the layout)
    const bool fold = !PAIR && (a.ab != nullptr || a.fs.on);
    if (fold) {
        if (a.fs.on) {
            if (tid < 64) film_coef_row_wave(a.fs, b, tid, ab_s);
        } else if (tid < 2 * SA_C / 4) {
            *reinterpret_cast<s_f32x4*>(ab_s + 4 * tid) = *reinterpret_cast<const s_f32x4*>(a.ab + (size_t)b * 2 * SA_C + 4 * tid);
        }
        __syncthreads();
    }

    SAF_STAMP()
    // (LayerNorm 1 and its B fragments are re-made per head pair inside the loop: 32 registers that need not live
    //  through the attention phase)
    s_f16x8 bh[4], bl[4];
    s_f32x16 av[2];                                // out-proj accumulator (scaled by 2048 until the end)
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) av[T][r] = 0.f;

    // ---- two head pairs: tile p of Q, K, V rows holds heads 2p (rows 0..15) and 2p+1 (rows 16..31) ----
    for (int p = 0; p < 2; ++p) {
        s_f32x16 qt, kt, vt;
#pragma unroll
        for (int r = 0; r < 16; ++r) { qt[r] = 0.f; kt[r] = 0.f; vt[r] = 0.f; }
        {
            s_f32x16 x1[2], ln[2];
            SA_LOAD_X(x1, xrow)
            sa_layernorm(x1, ln, a.ln1_g, a.ln1_b, kh);
            sa_make_frags(ln, bh, bl);
        }
        if (WLDS) {
            qt = sa_gemm_tile_lds(Wsh, Wsl, 32 * p, li, kh, bh, bl, qt);
            kt = sa_gemm_tile_lds(Wsh, Wsl, 64 + 32 * p, li, kh, bh, bl, kt);
            vt = sa_gemm_tile_lds(Wsh, Wsl, 128 + 32 * p, li, kh, bh, bl, vt);
        } else {
            qt = sa_gemm_tile(a.wqkv_h, a.wqkv_l, 32 * p, li, kh, bh, bl, qt);
            kt = sa_gemm_tile(a.wqkv_h, a.wqkv_l, 64 + 32 * p, li, kh, bh, bl, kt);
            vt = sa_gemm_tile(a.wqkv_h, a.wqkv_l, 128 + 32 * p, li, kh, bh, bl, vt);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const s_f32x4 bq = *reinterpret_cast<const s_f32x4*>(a.bqkv + 32 * p + 8 * g + 4 * kh);
            const s_f32x4 bk = *reinterpret_cast<const s_f32x4*>(a.bqkv + 64 + 32 * p + 8 * g + 4 * kh);
            const s_f32x4 bv = *reinterpret_cast<const s_f32x4*>(a.bqkv + 128 + 32 * p + 8 * g + 4 * kh);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qt[4 * g + j] = (qt[4 * g + j] * SA_DESCALE + bq[j]) * 0.25f;      // 1/sqrt(16), applied to q like torch
                kt[4 * g + j] = kt[4 * g + j] * SA_DESCALE + bk[j];
                vt[4 * g + j] = vt[4 * g + j] * SA_DESCALE + bv[j];
            }
        }

        SAF_STAMP()
        s_f32x16 kt2, vt2;
        if constexpr (PAIR) {
            {   // LayerNorm-1 fragments of the partner token (re-made per head pair, into the same registers)
                const float* xrow2 = a.x + ((size_t)b * L + min(t2, L - 1)) * SA_C;
                s_f32x16 x2[2], ln2[2];
                SA_LOAD_X(x2, xrow2)
                sa_layernorm(x2, ln2, a.ln1_g, a.ln1_b, kh);
                sa_make_frags(ln2, bh, bl);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { kt2[r] = 0.f; vt2[r] = 0.f; }
            kt2 = sa_gemm_tile(a.wqkv_h, a.wqkv_l, 64 + 32 * p, li, kh, bh, bl, kt2);
            vt2 = sa_gemm_tile(a.wqkv_h, a.wqkv_l, 128 + 32 * p, li, kh, bh, bl, vt2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const s_f32x4 bk = *reinterpret_cast<const s_f32x4*>(a.bqkv + 64 + 32 * p + 8 * g + 4 * kh);
                const s_f32x4 bv = *reinterpret_cast<const s_f32x4*>(a.bqkv + 128 + 32 * p + 8 * g + 4 * kh);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    kt2[4 * g + j] = kt2[4 * g + j] * SA_DESCALE + bk[j];
                    vt2[4 * g + j] = vt2[4 * g + j] * SA_DESCALE + bv[j];
                }
            }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int head = 2 * p + sub;
            // registers 8 sub .. 8 sub + 7 are this head's 16 features (8 per lane half, in fragment order)
            s_f16x8 q_h, q_l, k_h, k_l;
            sa_split8(q_h, q_l, [&](int j) { return qt[8 * sub + j] * 16.0f; });
            sa_split8(k_h, k_l, [&](int j) { return kt[8 * sub + j] * 16.0f; });
#ifndef SA_ABLATE_NOKV
            __syncthreads();                        // every wave is done with the previous head's K / V^T
            *reinterpret_cast<s_f16x8*>(Khi + t * SA_KROW + 8 * kh) = k_h;
            *reinterpret_cast<s_f16x8*>(Klo + t * SA_KROW + 8 * kh) = k_l;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {        // V^T[d][token], d = 8 (j>>2) + 4 kh + (j&3)
                unsigned hu, lu;
                split_pair_f16(vt[8 * sub + j] * 16.0f, vt[8 * sub + j + 1] * 16.0f, hu, lu);
                const s_f16x2 h2 = __builtin_bit_cast(s_f16x2, hu), l2 = __builtin_bit_cast(s_f16x2, lu);
                const int dd = 8 * (j >> 2) + 4 * kh + (j & 3);
                Vhi[dd * VROW + tp] = h2[0];
                Vlo[dd * VROW + tp] = l2[0];
                Vhi[(dd + 1) * VROW + tp] = h2[1];
                Vlo[(dd + 1) * VROW + tp] = l2[1];
            }
            if constexpr (PAIR) {
                s_f16x8 k2_h, k2_l;
                sa_split8(k2_h, k2_l, [&](int j) { return kt2[8 * sub + j] * 16.0f; });
                *reinterpret_cast<s_f16x8*>(Khi + t2 * SA_KROW + 8 * kh) = k2_h;
                *reinterpret_cast<s_f16x8*>(Klo + t2 * SA_KROW + 8 * kh) = k2_l;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    unsigned hu, lu;
                    split_pair_f16(vt2[8 * sub + j] * 16.0f, vt2[8 * sub + j + 1] * 16.0f, hu, lu);
                    const s_f16x2 h2 = __builtin_bit_cast(s_f16x2, hu), l2 = __builtin_bit_cast(s_f16x2, lu);
                    const int dd = 8 * (j >> 2) + 4 * kh + (j & 3);
                    Vhi[dd * VROW + tp2] = h2[0];
                    Vlo[dd * VROW + tp2] = l2[0];
                    Vhi[(dd + 1) * VROW + tp2] = h2[1];
                    Vlo[(dd + 1) * VROW + tp2] = l2[1];
                }
            }
            __syncthreads();

#endif
            SAF_STAMP()
            // ---- flash attention of this wave's 32 queries over all key blocks ----
            s_f32x16 acc_o;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
            float m = -1e30f;
            // S^T of key block kb_ (3 MFMAs).  With SA_SCORES_AHEAD (off by default, see the top of the file) the product of
            // block kb + 1 is issued BEFORE the softmax of block kb (16 more live registers).
#define SA_SCORES(dst_, kb_)                                                                                        \
            {                                                                                                           \
                _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) dst_[r_] = 0.f;                                       \
                const s_f16x8 ka_h_ = *reinterpret_cast<const s_f16x8*>(Khi + ((kb_) * 32 + li) * SA_KROW + 8 * kh);    \
                const s_f16x8 ka_l_ = *reinterpret_cast<const s_f16x8*>(Klo + ((kb_) * 32 + li) * SA_KROW + 8 * kh);    \
                dst_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka_h_, q_h, dst_, 0, 0, 0);                                \
                dst_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka_h_, q_l, dst_, 0, 0, 0);                                \
                dst_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka_l_, q_h, dst_, 0, 0, 0);                                \
            }
            constexpr bool AHEAD = !PAIR && SA_SCORES_AHEAD;
#ifdef SA_ABLATE_NOLOOP
            const int nkb = (a.L < 0) ? 1 : 0;                  // timing experiment: no attention loop at all (wrong results)
#else
            const int nkb = (L + 31) / 32;                      // key blocks that hold at least one real token
#endif
            s_f32x16 acc_n;
            if constexpr (AHEAD) SA_SCORES(acc_n, 0)
            for (int kb = 0; kb < nkb; ++kb) {
                s_f32x16 acc_s;
                if constexpr (AHEAD) {
                    acc_s = acc_n;
                    if (kb + 1 < nkb) SA_SCORES(acc_n, kb + 1)
                } else {
                    SA_SCORES(acc_s, kb)
                }
                // softmax through v_exp_f32 directly: p x 1024 = exp2(acc_s c + (10 - m)), c = log2(e) / 256 (undoes the
                // 16 x 16 operand pre-scale), running max m kept in log2 units -- one max, one fma, one exp2 per score
                constexpr float SC = 1.44269504088896340736f / 256.0f;
                float sc[16];
                float mraw = -3.0e38f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sc[r] = acc_s[r];
                    if (!FULL) {
                        const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        if (key >= L) sc[r] = -3.0e38f;
                    }
                    mraw = fmaxf(mraw, sc[r]);
                }
                mraw = max_xor32(mraw);                 // (v_permlane32_swap: no LDS round trip on the per-block chain)
                const float m_new = fmaxf(m, mraw * SC);
                // rescale only when some query of the wave saw a new maximum (after the first blocks it rarely moves): a
                // wave-uniform branch that saves the exp2 and the 16 multiplies of the common case; alpha would be exactly 1
                if (__any(m_new > m)) {
                    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_o[r] *= alpha;
                }
                const float off = 10.0f - m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(__fmaf_rn(sc[r], SC, off));
                m = m_new;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    s_f16x8 p_h, p_l;
                    sa_split8(p_h, p_l, [&](int j) { return sc[8 * s2 + j]; });
                    const s_f16x8 v_h = *reinterpret_cast<const s_f16x8*>(Vhi + li * VROW + kb * 32 + 16 * s2 + 8 * kh);
                    const s_f16x8 v_l = *reinterpret_cast<const s_f16x8*>(Vlo + li * VROW + kb * 32 + 16 * s2 + 8 * kh);
                    acc_o = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_h, p_h, acc_o, 0, 0, 0);
                    acc_o = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_h, p_l, acc_o, 0, 0, 0);
                    acc_o = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_l, p_h, acc_o, 0, 0, 0);
                }
            }
            SAF_STAMP()
            // accumulator row 16 / 20 (register 8): sum over ALL keys of 1.0 x p x 1024 -- the denominator, complete in both halves
            const float inv = 1.0f / (acc_o[8] * 16.0f);       // p carries x1024, v x16

            // ---- out-proj: av^T += W_o[:, 16 head .. 16 head + 15] . o_head^T  (one k-step; o rows = registers 0..7) ----
            s_f16x8 o_h, o_l;
            sa_split8(o_h, o_l, [&](int j) { return (acc_o[j] * inv) * 16.0f; });
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                const s_f16x8 ah = WLDS ? *reinterpret_cast<const s_f16x8*>(Wsh + (192 + 32 * T + li) * SA_WROW + 16 * head + 8 * kh)
                                        : *reinterpret_cast<const s_f16x8*>(a.wo_h + ((size_t)(4 * T + head) * 64 + 32 * kh + li) * 8);
                const s_f16x8 al = WLDS ? *reinterpret_cast<const s_f16x8*>(Wsl + (192 + 32 * T + li) * SA_WROW + 16 * head + 8 * kh)
                                        : *reinterpret_cast<const s_f16x8*>(a.wo_l + ((size_t)(4 * T + head) * 64 + 32 * kh + li) * 8);
                av[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, o_h, av[T], 0, 0, 0);
                av[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, o_l, av[T], 0, 0, 0);
                av[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, o_h, av[T], 0, 0, 0);
            }
            SAF_STAMP()
        }
    }

    // ---- av = out_proj + b_o + x ----
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        sa_bias(av[T], a.bo, T, kh);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            s_f32x4 v = *reinterpret_cast<const s_f32x4*>(xrow + 32 * T + 8 * g + 4 * kh);
            if (fold)
                v = v * *reinterpret_cast<const s_f32x4*>(ab_s + 32 * T + 8 * g + 4 * kh) +
                    *reinterpret_cast<const s_f32x4*>(ab_s + SA_C + 32 * T + 8 * g + 4 * kh);
#pragma unroll
            for (int j = 0; j < 4; ++j) av[T][4 * g + j] += v[j];
        }
    }
    SAF_STAMP()
#ifdef SA_ABLATE_NOFF
    if (a.L < 0) {
#endif
    // ---- feed-forward: LN -> W1 -> GELU -> W2 -> + av ----
    if (WLDS) {                                     // phase B of the staged weights (every wave is past the out-proj reads)
        __syncthreads();
        sa_stage_rows(Fsh, 0, a.w1_h, 64, tid, blockDim.x);
        sa_stage_rows(Fsl, 0, a.w1_l, 64, tid, blockDim.x);
        sa_stage_rows(Fsh, 64, a.w2_h, 64, tid, blockDim.x);
        sa_stage_rows(Fsl, 64, a.w2_l, 64, tid, blockDim.x);
        __syncthreads();
    }
    SAF_STAMP()
    {
        s_f32x16 ln[2];
        sa_layernorm(av, ln, a.ln2_g, a.ln2_b, kh);
        sa_make_frags(ln, bh, bl);
    }
    s_f32x16 f[2];
#pragma unroll
    for (int T = 0; T < 2; ++T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) f[T][r] = 0.f;
        f[T] = WLDS ? sa_gemm_tile_lds(Fsh, Fsl, 32 * T, li, kh, bh, bl, f[T]) : sa_gemm_tile(a.w1_h, a.w1_l, 32 * T, li, kh, bh, bl, f[T]);
        sa_bias(f[T], a.b1, T, kh);
#pragma unroll
        for (int r = 0; r < 16; ++r) f[T][r] = gelu_erf(f[T][r]);
    }
    sa_make_frags(f, bh, bl);
    SAF_STAMP()
#pragma unroll
    for (int T = 0; T < 2; ++T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) f[T][r] = 0.f;
        f[T] = WLDS ? sa_gemm_tile_lds(Fsh, Fsl, 64 + 32 * T, li, kh, bh, bl, f[T]) : sa_gemm_tile(a.w2_h, a.w2_l, 32 * T, li, kh, bh, bl, f[T]);
        sa_bias(f[T], a.b2, T, kh);
    }
#ifdef SA_ABLATE_NOFF
    }
    s_f32x16 f[2];
    for (int T = 0; T < 2; ++T) for (int r = 0; r < 16; ++r) f[T][r] = 0.f;
#endif
    SAF_STAMP()
    if (t < L) {
        float* orow = a.out + ((size_t)b * L + t) * SA_C;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const s_f32x4 v = {f[T][4 * g] + av[T][4 * g], f[T][4 * g + 1] + av[T][4 * g + 1],
                                   f[T][4 * g + 2] + av[T][4 * g + 2], f[T][4 * g + 3] + av[T][4 * g + 3]};
                *reinterpret_cast<s_f32x4*>(orow + 32 * T + 8 * g + 4 * kh) = v;
            }
    }
    SAF_STAMP()
    }   // trajectories of this workgroup
}

#ifdef SPDM_DIAG_SAF
extern "C" int spdm_debug_saf_stamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_saf_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

bool sa_fused_supported(int L, int C) { return C == SA_C && L >= 1 && L <= 512; }

hipError_t launch_sa_fused64(const float* x, float* out, int B, int L, const float* ln1_g, const float* ln1_b,
                             const float* ln2_g, const float* ln2_b, const void* const w_hl[8], const float* bqkv,
                             const float* bo, const float* b1, const float* b2, const float* ab, unsigned sw, hipStream_t s,
                             const FilmSpec* fs) {
    if (!sa_fused_supported(L, SA_C) || B <= 0) return hipErrorInvalidValue;
    if (fs && (ab != nullptr || fs->C != SA_C)) return hipErrorInvalidValue;
    SaFusedArgs a{};
    a.x = x; a.out = out; a.L = L; a.nb = B;
    a.ln1_g = ln1_g; a.ln1_b = ln1_b; a.ln2_g = ln2_g; a.ln2_b = ln2_b;
    a.wqkv_h = (const _Float16*)w_hl[0]; a.wqkv_l = (const _Float16*)w_hl[1];
    a.wo_h = (const _Float16*)w_hl[2]; a.wo_l = (const _Float16*)w_hl[3];
    a.w1_h = (const _Float16*)w_hl[4]; a.w1_l = (const _Float16*)w_hl[5];
    a.w2_h = (const _Float16*)w_hl[6]; a.w2_l = (const _Float16*)w_hl[7];
    a.bqkv = bqkv; a.bo = bo; a.b1 = b1; a.b2 = b2; a.ab = ab;
    if (fs) { a.fs = *fs; a.fs.on = 1; }
    const bool pair = L > 256;                                                    // two workgroups per trajectory
    if (pair && (ab != nullptr || fs != nullptr)) return hipErrorInvalidValue;    // (the plan keeps film_apply there)
    const int nwave = pair ? 8 : (L + 31) / 32;
    const int Lp = pair ? 512 : nwave * 32;
    const bool wlds = !pair && (nwave >= 4) && !(sw & SW_SA_NO_WLDS);      // long sequences: weights staged in LDS
    const size_t kreg = (size_t)2 * Lp * SA_KROW, freg = (size_t)2 * 128 * SA_WROW;           // halfs: K hi + lo; ff1 / ff2 hi + lo
    const size_t lds = ((wlds ? std::max(kreg, freg) : kreg) + (size_t)2 * 32 * (Lp + 8) + (wlds ? (size_t)2 * 256 * SA_WROW : 0)) * sizeof(_Float16) +
                       2 * SA_C * sizeof(float);                                               // + the FiLM-coefficient row
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const bool full = (L % 32 == 0);
    void (*kern)(const SaFusedArgs) =
        pair ? (full ? sa_fused64_kernel<true, false, true> : sa_fused64_kernel<false, false, true>)
             : full ? (wlds ? sa_fused64_kernel<true, true, false> : sa_fused64_kernel<true, false, false>)
                    : (wlds ? sa_fused64_kernel<false, true, false> : sa_fused64_kernel<false, false, false>);
    if (lds > 64 * 1024)
        if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return e;
    int grid = pair ? 2 * B : B;
    if (wlds) {                                    // persistent workgroups: one per CU (LDS-bound), each walks B / grid trajectories
        static int ncu = 0;
        if (ncu == 0) {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            ncu = n;
        }
        grid = std::min(B, ncu);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nwave), lds, s, a);
    return hipGetLastError();
}

}  // namespace spdm
