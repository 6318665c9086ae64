// conv_reg.hip -- 3x3 convolution 64 -> 64 channels on the width-8 and width-4 levels (inc's second conv, the last conv of up3,
// down1's first DoubleConvolution and the last conv of up2: models/Unet_FiLmLayer.py:101-115, 256-266) with the ACTIVATIONS IN
// REGISTERS and the WEIGHTS IN LDS -- the opposite placement of conv_wide.hip.
//
// Why: a 64-wide layer gives a wave a 64 x 64 accumulator strip, so each 16 KB of weight fragments is good for 96 MFMAs only
// and every wave streams the layer's whole 144 KB per 64 rows -- two thirds of a CU's L2 bandwidth at full MFMA rate -- while the
// slab hand-over (transform -> LDS -> barrier -> fragment reads) is exposed twice per tile: conv_wide's 256 x 64 variant keeps the
// matrix pipe 35 % busy on these layers (PMC, profiles/r03_conv_pmc_summary_b4096_h32d3.json), against 55-64 % on the 128-wide
// ones.  Here
//   * the whole layer's split-fp16 weights (9 taps x 64 x 64 x {hi, lo} = 144 KB) are staged in LDS ONCE per workgroup, which is
//     persistent (one per CU, 8 waves); B operands are conflict-free ds_read_b128 of 1-KiB lane-contiguous blocks;
//   * a wave owns 64 output positions = 8 image rows x 8 columns (16 x 4 on the width-4 level) of one sample and loads them (+ one
//     image row above and below) straight from global memory in MFMA A-operand layout: lane (kg, l16) = position l16 of a row
//     tile, channels 8 kg .. 8 kg + 7 of the 32-channel k-step; the GroupNorm -> GELU prologue and the hi / lo split run on registers;
//   * row tile t holds image rows h0 + t, h0 + t + 4 (, + 8, + 12) interleaved lane by lane, so the VERTICAL neighbours of a tile
//     are another tile's registers (t - 1 / t + 1), except at the block edge, where one composite tile is put together from the
//     halo row and a tile shifted by one lane; the HORIZONTAL neighbours are ONE DPP row shift per register whose zero fill at the
//     end of the 16-lane row is exactly the image edge.  No activation ever touches LDS, the waves never meet after the weight
//     staging (no barrier, no hand-over); the next tile's rows are requested before this tile's stores (vmcnt is in order);
//   * the A operand of unit (tap, row tile) u + 1 is put together in the shadow of unit u's 12 MFMAs -- but VALU work and the matrix
//     pipe of a SIMD overlap only partly on this chip (tools/probes/coexec_probe.hip: another wave's plain VALU stream hides 65 %
//     under MFMAs, v_pk_* FP32 not at all), so the launch is close to MFMA time + prologue time: 241-254 us for 110 us of MFMAs
//     at B = 4096 against 333-350 us on conv_wide's 256 x 64 variant;
//   * output channels are permuted among the B tiles (tile nb' holds channels 4 l16 + nb') so that a lane ends up with four
//     consecutive channels of a position: 16-byte stores straight from the accumulators, whole 256-byte rows per instruction.
// Same math and contracts as the other convolution kernels: split-fp16 operands (activations x16, weights x128), hi*hi + hi*lo +
// lo*hi into fp32, GroupNorm(1,C) partial sums of the STORED values as the epilogue (one fp64 slot per wave tile:
// StatsRef{m_tile = 64, n_tiles = 1}).
#include <algorithm>

#include "device_utils.h"

namespace spdm {

namespace {

typedef float r_f32x4 __attribute__((ext_vector_type(4)));
typedef int r_i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 r_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 r_f16x2 __attribute__((ext_vector_type(2)));

constexpr float R_ACT_SCALE = 16.0f;
constexpr float R_DESCALE = 1.0f / 2048.0f;
constexpr int R_WBYTES = 9 * 2 * 4 * 2 * 1024;       // weights in LDS: [tap][k-step][B tile][hi | lo] x 1 KiB
constexpr int R_NTHR = 512, R_WAVES = 8;
constexpr int R_MAXT = 192;                          // tiles one wave may walk (the LDS left beside the weights holds its statistics table)

struct RTile { r_i32x4 h, l; };                       // one A operand pair: 8 fp16 hi | 8 fp16 lo of this lane's position

__device__ __forceinline__ void r_split2(float a, float b, int& h, int& l) {
    unsigned hu, lu;
    split_pair_f16(a * R_ACT_SCALE, b * R_ACT_SCALE, hu, lu);
    h = (int)hu;
    l = (int)lu;
}

// Lane map of a row tile (RPT = 16 / W image rows per tile): lane l16 of a 16-lane DPP row = column w = l16 / RPT of image row
// h0 + t + 4 (l16 % RPT).  Horizontal neighbour: lane i <- lane i + RPT DX of its DPP row; bound_ctrl writes zero where the source
// falls off the row, which is exactly the image edge (w = W - 1 for DX = +1, w = 0 for DX = -1): one instruction per register, no mask
template <int DX, int RPT>
__device__ __forceinline__ RTile r_shift(const RTile& t) {
    if constexpr (DX == 0) return t;
    RTile c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        c.h[r] = __builtin_amdgcn_update_dpp(0, t.h[r], (DX > 0 ? 0x100 : 0x110) + RPT, 0xf, 0xf, true);     // row_shl:RPT / row_shr:RPT
        c.l[r] = __builtin_amdgcn_update_dpp(0, t.l[r], (DX > 0 ? 0x100 : 0x110) + RPT, 0xf, 0xf, true);
    }
    return c;
}
// composite tile at the block edge: the lanes of one image-row slot keep `halo`, the others take the neighbouring lane of `t`
//   UP  : rows (h0 - 1 | h0 + 3 | ..): slot 0 keeps the halo row, slot s >= 1 <- slot s - 1 of tile 3 (lane i - 1)
//   DOWN: rows (h0 + 4 | .. | h0 + 4 RPT): the last slot keeps the halo row, slot s <- slot s + 1 of tile 0 (lane i + 1)
template <bool UP>
__device__ __forceinline__ RTile r_composite(const RTile& halo, const RTile& t, bool keep_halo) {
    RTile c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int mh = __builtin_amdgcn_update_dpp(0, t.h[r], UP ? 0x111 : 0x101, 0xf, 0xf, true);     // row_shr:1 / row_shl:1
        const int ml = __builtin_amdgcn_update_dpp(0, t.l[r], UP ? 0x111 : 0x101, 0xf, 0xf, true);
        c.h[r] = keep_halo ? halo.h[r] : mh;
        c.l[r] = keep_halo ? halo.l[r] : ml;
    }
    return c;
}

}  // namespace

#ifdef SPDM_DIAG_REG
// diagnostic builds: phase stamps (s_memrealtime, 10-ns ticks) of workgroup 0, lane 0 of waves 0 and 4, first 3 tiles x 8 stamps
__device__ unsigned long long g_reg_stamps[2 * 3 * 8];
#define REG_STAMP(k_) if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && tile_no < 3) { \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g_reg_stamps[((wave >> 2) * 3 + tile_no) * 8 + (k_)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define REG_STAMP(k_)
#endif

template <int PRO, int WD>
__global__ __launch_bounds__(R_NTHR, 1) void conv_reg64_kernel(const GemmArgs a, const int epi_slots, const int nw, const int n_wt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char r_smem[];
    r_i32x4* const Wl = reinterpret_cast<r_i32x4*>(r_smem);
    float* const gb = reinterpret_cast<float*>(r_smem + R_WBYTES);       // gamma[64] | beta[64]
    float* const mr = gb + 128;                                          // [8 waves][R_MAXT tiles] x {mean, rstd}
    constexpr bool pro = (PRO != PRO_NONE), pro_gelu = (PRO == PRO_GN_GELU);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, kg = lane >> 4;

    const int HW = a.HW, H = a.H;
    constexpr int RPT = 16 / WD, TH = 4 * RPT;      // image rows per row tile / per wave tile (W = 8: 2, 8; W = 4: 4, 16)
    const int wcol = l16 / RPT, hsel = l16 % RPT;   // lane map of a row tile: column, image-row slot

    // raw rows of wave tile wt_: tiles 0..3 = image rows (h0 + t | h0 + t + 4 | ..), tile 4 = halo rows (h0 - 1 in slot 0, h0 + TH in
    // the last slot; the slots between, W = 4 only, are never used)
#ifdef REG_ABL_NOLOAD
#define R_ROWPTR(sb_, hrow_) (a.src + lane * 4 + (size_t)((hrow_) & 1) * 256)       /* (the same few lines for every tile: cache hits) */
#else
#define R_ROWPTR(sb_, hrow_) ((sb_) + (size_t)((hrow_) * WD + wcol) * a.src_ld)
#endif
#define R_LOAD_RAW(wt_, ks_)                                                                            \
    {                                                                                                  \
        const int m0_ = (wt_) * 64;                                                                    \
        const int b_ = m0_ / HW;                                                                       \
        const int h0_ = (m0_ - b_ * HW) / WD;                                                          \
        const float* sb_ = a.src + (size_t)b_ * HW * a.src_ld + kg * 8;                                \
        const int hh_ = hsel ? h0_ + TH : h0_ - 1;                                                     \
        _Pragma("unroll") for (int t_ = 0; t_ < 5; ++t_) {                                             \
            const int hrow_ = (t_ < 4) ? h0_ + t_ + 4 * hsel : min(max(hh_, 0), H - 1);                \
            const float* p_ = R_ROWPTR(sb_, hrow_);                                                    \
            raw[ks_][t_][0] = *reinterpret_cast<const r_f32x4*>(p_ + 32 * (ks_));                      \
            raw[ks_][t_][1] = *reinterpret_cast<const r_f32x4*>(p_ + 32 * (ks_) + 4);                  \
        }                                                                                              \
    }
    r_f32x4 raw[2][5][2];
    const int wt_first = (int)blockIdx.x * nw + wave, wt_step = (int)gridDim.x * nw;
    R_LOAD_RAW(min(wt_first, n_wt - 1), 0)          // (the first tile's rows fly under the weight staging)

    // ---- weights -> LDS, once: block ((tap 2 + ks) 4 + nb') x {hi, lo}, lane (kg, l16) <- row n = 4 l16 + nb' of the fragment-order
    //      copy (frag_order_weights, kernels.h: block ((tap 2 + ks) 4 + n / 16) x {hi, lo}, lane (kg, n % 16)) ----
    {
        // (all 18 loads of a thread in flight before the first LDS write: the weights are cold in L2 inside a real step, and a
        //  load -> store loop is 18 dependent round trips -- measured +5 us per launch against the warm micro-benchmark)
        const r_i32x4* wsrc = reinterpret_cast<const r_i32x4*>(a.wgt_frag);
        r_i32x4 wreg[R_WBYTES / 16 / R_NTHR];
#pragma unroll
        for (int j = 0; j < R_WBYTES / 16 / R_NTHR; ++j) {
            const int i = tid + j * R_NTHR;
            const int ln = i & 63, blk = i >> 6;
            const int part = blk & 1, nbp = (blk >> 1) & 3, tk = blk >> 3;
            const int n = 4 * (ln & 15) + nbp;
            wreg[j] = wsrc[(((tk * 4 + (n >> 4)) * 2 + part) << 6) + (ln & 48) + (n & 15)];
        }
#pragma unroll
        for (int j = 0; j < R_WBYTES / 16 / R_NTHR; ++j) Wl[tid + j * R_NTHR] = wreg[j];
        if (pro && tid < 128) gb[tid] = (tid < 64) ? a.pro_gamma[tid] : a.pro_beta[tid - 64];
        // mean / rstd of the sample of every tile this wave will walk: lane i takes the wave's i-th tile, so the whole table is ONE
        // dependent round trip, under the weight staging, instead of one per tile in front of its prologue
        if (pro && wave < nw) {
            const int stride = (int)gridDim.x * nw;
            for (int i = lane; (int)blockIdx.x * nw + wave + i * stride < n_wt; i += 64) {
                const int wt = (int)blockIdx.x * nw + wave + i * stride;
                float mean, rstd;
                sample_mean_rstd(a.pro_stats, (wt * 64) / a.HW, mean, rstd);
                mr[(wave * R_MAXT + i) * 2] = mean;
                mr[(wave * R_MAXT + i) * 2 + 1] = rstd;
            }
        }
    }
    __syncthreads();
    if (wave >= nw) return;                 // (no barrier below: the waves are independent from here on)
#ifdef SPDM_DIAG_REG
    int tile_no = -1;
#endif
    int tile_i = 0;
    for (int wt = wt_first; wt < n_wt; wt += wt_step, ++tile_i) {
#ifdef SPDM_DIAG_REG
        ++tile_no;
#endif
        REG_STAMP(0)
        R_LOAD_RAW(wt, 1)            // (k-step 1's half of the rows: behind the previous tile's stores in vmcnt order, not needed before the first MFMA phase is over)
        const int m0 = wt * 64;
        const int b = m0 / HW;
        const int h0 = (m0 - b * HW) / WD;
        const int hh = hsel ? h0 + TH : h0 - 1;
        const bool halo_ok = hh >= 0 && hh < H;
        REG_STAMP(1)
        float mean = 0.f, rstd = 1.f;
        if (pro) {
            mean = mr[(wave * R_MAXT + tile_i) * 2];
            rstd = mr[(wave * R_MAXT + tile_i) * 2 + 1];
        }
        REG_STAMP(2)

        r_f32x4 acc[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[t][nb] = r_f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // ---- prologue on registers: GroupNorm affine, GELU, x16, hi / lo split ----
            float sc[8], sh[8];
            if (pro) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const r_f32x4 g4 = *reinterpret_cast<const r_f32x4*>(gb + 32 * ks + 8 * kg + 4 * q);
                    const r_f32x4 b4 = *reinterpret_cast<const r_f32x4*>(gb + 64 + 32 * ks + 8 * kg + 4 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        sc[4 * q + j] = rstd * g4[j];
                        sh[4 * q + j] = b4[j] - mean * sc[4 * q + j];
                    }
                }
            }
            RTile T[6];                      // 0..3: the tiles; 4: rows (h0 - 1 | h0 + 3); 5: rows (h0 + 4 | h0 + 8)
#pragma unroll
            for (int t = 0; t < 5; ++t) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float x[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = raw[ks][t][q][j];
                        if (pro) v = __fmaf_rn(v, sc[4 * q + j], sh[4 * q + j]);
#ifndef REG_ABL_NOGELU
                        if (pro_gelu) v = gelu_erf(v);
#endif
                        if (t == 4 && !halo_ok) v = 0.f;
                        x[j] = v;
                    }
                    int h0_, l0_, h1_, l1_;
                    r_split2(x[0], x[1], h0_, l0_);
                    r_split2(x[2], x[3], h1_, l1_);
                    T[t].h[2 * q] = h0_; T[t].l[2 * q] = l0_;
                    T[t].h[2 * q + 1] = h1_; T[t].l[2 * q + 1] = l1_;
                }
            }
            if (ks == 0) { REG_STAMP(3) } else { REG_STAMP(5) }
            T[5] = r_composite<false>(T[4], T[0], hsel == RPT - 1);      // slot s <- image row h0 + 4 (s + 1) (slot s + 1 of tile 0); the last keeps h0 + TH
            T[4] = r_composite<true>(T[4], T[3], hsel == 0);             // slot s <- image row h0 + 4 s - 1 (slot s - 1 of tile 3); slot 0 keeps h0 - 1

            // ---- 36 units (tap, row tile) x 4 B tiles x 3 MFMAs.  The A operand of unit u + 1 is put together (8 DPP moves, or
            //      nothing when dx = 0) in the shadow of unit u's 12 MFMAs; the B tiles of tap + 1 replace those of tap pair by pair
            //      inside the tap's last unit, as soon as their last MFMA has issued (32 registers of weights, not 64). ----
#ifdef REG_ABL_NOSHIFT
#define REG_NOSHIFT 1
#else
#define REG_NOSHIFT 0
#endif
#ifdef REG_ABL_NOMFMA      /* timing experiments (wrong results): SPDM_EXTRA_FLAGS=-DREG_ABL_... */
#define R_MFMA(acc_, a_, b_) { const r_f16x8 a__ = (a_), b__ = (b_); acc_.x += (float)a__[0] * (float)b__[0]; }
#else
#define R_MFMA(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, acc_, 0, 0, 0);
#endif
#define R_SRC(u_) (((u_) & 3) + ((u_) >> 2) / 3 - 1)
#define R_MAKE_S(dst_, u_)                                                                                    \
            {                                                                                                  \
                const int src_ = R_SRC(u_), dx_ = ((u_) >> 2) % 3 - 1;                                         \
                const RTile& base_ = (src_ < 0) ? T[4] : (src_ > 3) ? T[5] : T[src_ < 0 ? 0 : src_ > 3 ? 3 : src_]; \
                dst_ = (dx_ == 0 || REG_NOSHIFT) ? base_ : (dx_ > 0) ? r_shift<1, RPT>(base_) : r_shift<-1, RPT>(base_);  \
            }
#define R_LOAD_B(p_, tap_)                                                                                     \
            {                                                                                                  \
                const r_i32x4* wb_ = Wl + (((((tap_) * 2 + ks) * 4 + 2 * (p_)) * 2) << 6) + lane;               \
                Bf[2 * (p_)][0] = __builtin_bit_cast(r_f16x8, wb_[0]);                                         \
                Bf[2 * (p_)][1] = __builtin_bit_cast(r_f16x8, wb_[64]);                                        \
                Bf[2 * (p_) + 1][0] = __builtin_bit_cast(r_f16x8, wb_[128]);                                   \
                Bf[2 * (p_) + 1][1] = __builtin_bit_cast(r_f16x8, wb_[192]);                                   \
            }
            r_f16x8 Bf[4][2];
            R_LOAD_B(0, 0)
            R_LOAD_B(1, 0)
            RTile Sc;
            R_MAKE_S(Sc, 0)
#pragma unroll
            for (int u = 0; u < 36; ++u) {
                const int tap = u >> 2, t = u & 3;
                __builtin_amdgcn_sched_barrier(0);
                RTile Sn = Sc;
                if (u + 1 < 36) R_MAKE_S(Sn, (u + 1 < 36 ? u + 1 : 35))
                const r_f16x8 ah = __builtin_bit_cast(r_f16x8, Sc.h), al = __builtin_bit_cast(r_f16x8, Sc.l);
                if (t < 3 || tap == 8) {
                    // three passes over the unit's 4 accumulators: a dependent MFMA is 4 instructions behind its producer
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb)
                            R_MFMA(acc[t][nb], term == 2 ? al : ah, Bf[nb][term == 1 ? 1 : 0])
                } else {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
#pragma unroll
                        for (int term = 0; term < 3; ++term)
#pragma unroll
                            for (int nb = 2 * p; nb < 2 * p + 2; ++nb)
                                R_MFMA(acc[t][nb], term == 2 ? al : ah, Bf[nb][term == 1 ? 1 : 0])
                        __builtin_amdgcn_sched_barrier(0);
                        R_LOAD_B(p, tap + 1)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                Sc = Sn;
            }
#undef R_LOAD_B
#undef R_MAKE_S
#undef R_SRC
            if (ks == 0) { REG_STAMP(4) } else { REG_STAMP(6) }
        }

        // ---- the next tile's rows (k-step 0's half) are requested BEFORE this tile's stores: vmcnt counts both in order, so loads
        //      behind the stores would wait for 16 write acknowledgements; this way they fly under the epilogue and the stores
        //      drain under the next tile's MFMAs ----
        __builtin_amdgcn_sched_barrier(0);
        R_LOAD_RAW(min(wt + wt_step, n_wt - 1), 0)         // (unconditional, clamped: no control flow around loads)
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue: accumulator register j of tile (t, nb') = position row 4 kg + j of tile t, channel 4 l16 + nb' ----
        float s1 = 0.f, s2 = 0.f;
        float* dbase = a.dst + (size_t)b * HW * a.dst_ld + 4 * l16;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            __builtin_amdgcn_sched_barrier(0);          // (one row tile at a time: the next tile's raw rows are live here)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * kg + j;
                const int hrow = h0 + t + 4 * (r % RPT);
                const r_f32x4 v = {acc[t][0][j] * R_DESCALE, acc[t][1][j] * R_DESCALE, acc[t][2][j] * R_DESCALE, acc[t][3][j] * R_DESCALE};
#ifdef REG_ABL_NOSTORE
                if (v.x == 1234.5f)
#endif
                *reinterpret_cast<r_f32x4*>(dbase + (size_t)(hrow * WD + r / RPT) * a.dst_ld) = v;
                s1 += (v.x + v.y) + (v.z + v.w);
                s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
        }
        const double t1 = wave_sum_f64((double)s1), t2 = wave_sum_f64((double)s2);
        if (lane == 0) {
            double* o = a.epi_stats + ((size_t)b * epi_slots + (wt - (b * HW) / 64)) * 2;
            o[0] = t1;
            o[1] = t2;
        }
        REG_STAMP(7)
    }
#undef R_LOAD_RAW
#undef R_ROWPTR
}

#ifdef SPDM_DIAG_REG
extern "C" int spdm_debug_reg_stamps(unsigned long long* out48) {
    return hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_reg_stamps), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : -1;
}
#endif

// shape rule (gemm_geometry asks before the plan is made; M = the rows the geometry is chosen for)
bool conv_reg_geometry(int M, int N, int K, int HW, int W, int taps, int split, unsigned sw) {
    if (!split || taps != 9 || N != 64 || K != 64 || !(W == 8 || W == 4) || HW % 64 != 0 || (sw & SW_NO_REG64)) return false;
    // from one wave tile per CU up (tools/bench_convs.py, old / new us per launch: 256 tiles 20 / 16.6 on both levels, 128 tiles 10 / 15.8
    // -- a launch costs ~15 us whatever its size: 144 KB of weights into LDS per workgroup, then at least one whole tile per wave)
    return M / 64 >= spdm_tune(17, 256);
}

hipError_t launch_conv_reg64(const GemmArgs& a, const GemmGeom& g, hipStream_t s) {
    // shape contract of the kernel -- checked on the host so that a bad plan can never fault the GPU
    if (!a.split || a.wgt_frag == nullptr || a.taps != 9 || a.N != 64 || a.K != 64 || !(a.W == 8 || a.W == 4) || a.H < 64 / a.W || a.HW != a.H * a.W ||
        a.HW % 64 != 0 || a.M <= 0 || a.M % a.HW != 0 || a.src_ld % 4 != 0 || a.src_ld < 64 || a.dst_ld % 4 != 0 || a.dst_ld < 64 ||
        a.epi != EPI_STATS || a.epi_stats == nullptr || a.row_stats != nullptr || a.skip != nullptr || a.ksplit > 1 || a.debug != 0 ||
        a.pro < PRO_NONE || a.pro > PRO_GN_GELU || g.m_tile != 64 || g.n_tile != 64 || g.n_tiles != 1)
        return hipErrorInvalidValue;
    if (a.pro != PRO_NONE && (a.pro_stats.p == nullptr || a.pro_gamma == nullptr || a.pro_beta == nullptr || a.pro_stats.HW != a.HW))
        return hipErrorInvalidValue;
    const int n_wt = a.M / 64;
    const int nw = std::min(R_WAVES, (n_wt + 255) / 256);
    const int grid = std::min(256, (n_wt + nw - 1) / nw);
    if ((n_wt + grid * nw - 1) / (grid * nw) > R_MAXT) return hipErrorInvalidValue;          // (batch > ~390 000 at horizon 32)
    const size_t lds = R_WBYTES + (128 + R_WAVES * R_MAXT * 2) * sizeof(float);
    auto launch = [&](auto kern) -> hipError_t {
        if (hipError_t e = allow_full_lds(reinterpret_cast<const void*>(kern)); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(R_NTHR), lds, s, a, g.slots, nw, n_wt);
        return hipGetLastError();
    };
    if (a.W == 8) {
        if (a.pro == PRO_NONE) return launch(conv_reg64_kernel<PRO_NONE, 8>);
        if (a.pro == PRO_GN) return launch(conv_reg64_kernel<PRO_GN, 8>);
        return launch(conv_reg64_kernel<PRO_GN_GELU, 8>);
    }
    if (a.pro == PRO_NONE) return launch(conv_reg64_kernel<PRO_NONE, 4>);
    if (a.pro == PRO_GN) return launch(conv_reg64_kernel<PRO_GN, 4>);
    return launch(conv_reg64_kernel<PRO_GN_GELU, 4>);
}

}  // namespace spdm
