// kernels.h -- launch-side declarations of the hand-written gfx950 kernels.
//
// Activation layout everywhere: CHANNELS-LAST  act[b][p][c]  with p = h*W_l + w
// (the reference's NCHW "(B,C,H,W) image", models/Unet_FiLmLayer.py:286, viewed
// as tokens x channels -- which is also the (B,L,C) view its attention blocks
// take at :74).  A 3x3 convolution is then an implicit GEMM
//     out[m][n] = sum_{tap,ci} in[m + shift(tap)][ci] * w[tap][n][ci],   m = b*HW + p
// and a Linear layer is the same GEMM with one tap.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

namespace spdm {

// Kernels that declare more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, a PER-DEVICE
// function attribute: set it once per (device, kernel) -- a process may hold handles on several GPUs.
inline hipError_t allow_full_lds(const void* kern) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, kern})) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    if (e == hipSuccess) done.insert({dev, kern});
    return e;
}

// GroupNorm(1,C) statistics are produced by the kernel that writes a tensor, as
// per-(sample, tile) partial sums in fp64:  stats[(b*slots + slot)*2 + {0,1}] =
// {sum x, sum x^2}.  Slot of the partial that tile (mt, nt) contributes to
// sample b:  (mt - (b*HW)/m_tile) * n_tiles + nt.  Consumers add the valid
// slots in a fixed order, so results are run-to-run deterministic.
struct StatsRef {
    const double* p;     // nullptr: no normalisation
    int slots;           // slots per sample
    int m_tile;          // rows per producer tile
    int n_tiles;         // producer tiles along channels
    int HW;              // rows per sample of the normalised tensor
    double inv_count;    // 1 / (C * HW)
};

__host__ __device__ inline int stats_slots(int HW, int m_tile, int n_tiles) {
    return ((HW + m_tile - 2) / m_tile + 1) * n_tiles;
}

// Kernel-selection switches.  Read from the environment ONCE (spdm_create / spdm_bench_gemm: switches_from_env) and
// carried in the handle; every launch decision and the step-graph key derive from this one word, so a mid-session
// change of the environment cannot change which kernels run.  spdm_set_switch flips one on a live handle (tests).
enum : unsigned {
    SW_NO_WIDE = 1u << 0, SW_NO_WIDE128 = 1u << 1, SW_NO_W2 = 1u << 2, SW_NO_T512 = 1u << 3, SW_T512 = 1u << 4,
    SW_T3_BIG = 1u << 5, SW_NO_SMALL_TPI3 = 1u << 6, SW_WIDE_N64_2X2 = 1u << 7, SW_NO_SA_FUSED = 1u << 8,
    SW_NO_SA_TAIL = 1u << 9, SW_ATTN_VALU = 1u << 10, SW_SA_NO_WLDS = 1u << 11, SW_NO_FILM_FOLD = 1u << 12,
    SW_NO_GRAPH = 1u << 13, SW_NO_SPLITK = 1u << 14, SW_ARENA_TRACE = 1u << 15, SW_NO_WIDE_PIPE = 1u << 16, SW_NO_SKINNY = 1u << 17, SW_DEEP = 1u << 18,
    SW_NO_FILM_LOCAL = 1u << 19, SW_NO_FUSED_SRC = 1u << 20, SW_FILM_LOCAL = 1u << 21, SW_PIN_GEOMETRY = 1u << 22, SW_NO_WP4 = 1u << 23, SW_G2 = 1u << 24, SW_NO_WP8 = 1u << 25, SW_NO_SA_HEAD = 1u << 26, SW_SA_HEAD = 1u << 27, SW_NO_REG64 = 1u << 28,
};
struct SwitchName { const char* env; unsigned bit; };
inline const SwitchName* switch_table(int* n) {
    static const SwitchName t[] = {
        {"SPDM_NO_WIDE", SW_NO_WIDE}, {"SPDM_NO_WIDE128", SW_NO_WIDE128}, {"SPDM_NO_W2", SW_NO_W2}, {"SPDM_NO_T512", SW_NO_T512},
        {"SPDM_T512", SW_T512}, {"SPDM_T3_BIG", SW_T3_BIG}, {"SPDM_NO_SMALL_TPI3", SW_NO_SMALL_TPI3},
        {"SPDM_WIDE_N64_2X2", SW_WIDE_N64_2X2}, {"SPDM_NO_SA_FUSED", SW_NO_SA_FUSED}, {"SPDM_NO_SA_TAIL", SW_NO_SA_TAIL},
        {"SPDM_ATTN_VALU", SW_ATTN_VALU}, {"SPDM_SA_NO_WLDS", SW_SA_NO_WLDS}, {"SPDM_NO_FILM_FOLD", SW_NO_FILM_FOLD},
        {"SPDM_NO_GRAPH", SW_NO_GRAPH}, {"SPDM_NO_SPLITK", SW_NO_SPLITK}, {"SPDM_ARENA_TRACE", SW_ARENA_TRACE}, {"SPDM_NO_WIDE_PIPE", SW_NO_WIDE_PIPE}, {"SPDM_NO_SKINNY", SW_NO_SKINNY}, {"SPDM_DEEP", SW_DEEP},
        {"SPDM_NO_FILM_LOCAL", SW_NO_FILM_LOCAL}, {"SPDM_NO_FUSED_SRC", SW_NO_FUSED_SRC},
        {"SPDM_FILM_LOCAL", SW_FILM_LOCAL}, {"SPDM_PIN_GEOMETRY", SW_PIN_GEOMETRY},
        {"SPDM_NO_WP4", SW_NO_WP4}, {"SPDM_G2", SW_G2}, {"SPDM_NO_WP8", SW_NO_WP8},
        {"SPDM_NO_SA_HEAD", SW_NO_SA_HEAD}, {"SPDM_SA_HEAD", SW_SA_HEAD}, {"SPDM_NO_REG64", SW_NO_REG64}};
    *n = (int)(sizeof(t) / sizeof(t[0]));
    return t;
}
unsigned switches_from_env();          // spdm_api.hip
int spdm_tune(int idx, int dflt);      // conv_gemm.hip: SPDM_TUNE<idx> (read once per process) or dflt -- tuning experiments only

// Load prologue of a convolution.  PRO_POOL / PRO_UPCAT: the FIRST convolution of a Down / UpSample block reads the block's input
// through the resampling op itself (GemmArgs: "fused sources") instead of a materialised pooled / concatenated tensor.
enum { PRO_NONE = 0, PRO_GN = 1, PRO_GN_GELU = 2, PRO_POOL = 3, PRO_UPCAT = 4 };
enum { EPI_STATS = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_BIAS_RESID = 3, EPI_PLAIN = 4 };

struct GemmArgs {
    const float* src;  int src_ld;     // [M][src_ld], K valid channels
    const float* wgt;                  // [taps][N][K]  (k contiguous); split: per 32-k chunk [32 fp16 hi | 32 fp16 lo]
    int split;                         // 0: fp32 MFMA (exact); 1: split-fp16 MFMA (wgt in the split format)
    const float* wgt_frag;             // optional: fragment-order copy of the split weights (frag_order_weights) -> conv_wide.hip
    float* dst;        int dst_ld;
    int M, K, N;
    int geom_M;                        // > 0: choose tiles / split-K / kernel as for THIS many rows (SW_PIN_GEOMETRY: a shard of a larger batch
                                       // then runs exactly the kernels the whole batch would, and reproduces it bit for bit)
    int taps;                          // 1, 3 (vertical taps, W == 1) or 9
    int H, W, HW;                      // spatial dims of this level
    int pro;  StatsRef pro_stats;  const float* pro_gamma;  const float* pro_beta;
    // Fused sources (conv_skinny.hip, conv_wide.hip; gemm_takes_fused_source says which launches):
    //   PRO_POOL : the conv input is MaxPool2d(2) (models/Unet_FiLmLayer.py:132,159) of src, src being the FINER level's tensor
    //              [B][4 HW][K] -- a (2 H) x (2 W) map per sample -- with its pending GroupNorm in pro_stats / pro_gamma / pro_beta
    //              (pro_stats.p == nullptr: none; the affine is applied per tap: max does not commute with a negative gain);
    //   PRO_UPCAT: torch.cat([Upsample(x2, bilinear, align_corners=True)(x), x_res], dim=1) (:191,217-218): input channels
    //              [0, up_C) are the upsample of src, the COARSER level's tensor [B][HW / 4][up_C] (pending GroupNorm in pro_*),
    //              channels [up_C, K) are skip [B][HW][K - up_C] (pending GroupNorm in skip_*; skip_stats.p == nullptr: none).
    int up_C;  const float* skip;  int skip_ld;  StatsRef skip_stats;  const float* skip_gamma;  const float* skip_beta;
    int epi;  double* epi_stats;   const float* bias;  const float* resid;  int resid_ld;
    double* row_stats;                 // optional: per-row {sum, sum^2} of the STORED values, [row][n_tiles][2] (LayerNorm of the consumer)
    // Split-K (small grids: few rows, long K).  ksplit > 1: the launch has ksplit x as many workgroups; workgroup (tile, ks)
    // walks the 32-channel chunks [ks nchunks / ksplit, (ks + 1) nchunks / ksplit) of every tap and stores its raw fp32
    // partial tile to partial[ks][M][N]; no statistics, no dst.  launch_splitk_combine then adds the slabs in the fixed
    // order ks = 0, 1, ... (deterministic: no atomics), writes dst and the GroupNorm partial sums.
    int ksplit;  float* partial;
    unsigned sw;                       // kernel-selection switches (SW_*) of the owning handle
    int debug;                         // ablation knobs for spdm_bench_gemm only (0 in the product path)
    unsigned long long* stamps;        // DBG_STAMP: [2][128] s_memtime stamps of one workgroup (diagnostic builds of the bench)
};
enum { DBG_NO_MFMA = 1, DBG_NO_WLOAD = 2, DBG_NO_GELU = 4, DBG_NO_STORE = 8, DBG_NO_ALOAD = 16, DBG_PP = 64, DBG_STAMP = 128 };

// geometry of the stats the GEMM writes (EPI_STATS)
struct GemmGeom {
    int m_tile, n_tile, n_tiles;     // output tile of the GEMM kernel
    int slots;                       // statistics slots per sample of whoever writes them
    int ksplit;                      // > 1: split-K launch + combine
    int skinny;                      // 1: conv_skinny.hip (a handful of rows: whole-K slab in LDS, the waves of a workgroup split K)
    int reg;                         // 1: conv_reg.hip (64 -> 64 channels on the width-8 level: activations in registers, weights in LDS)
    int st_m_tile, st_n_tiles;       // StatsRef geometry of the statistics (the GEMM's own tiling, or the combine kernel's)
};
// K = input channels (per tap); ksplit is only ever > 1 for split-precision 3x3 / 3x1 convolutions with the statistics
// epilogue (the callers that pass stats_epi = true)
GemmGeom gemm_geometry(int M, int N, int K, int HW, int W, int taps, int split, unsigned sw, bool stats_epi = false);
constexpr size_t SPLITK_WORKSPACE_BYTES = (size_t)48 << 20;     // partial slabs of one launch (handle-owned buffer)
// rows of one sample a combine workgroup owns (a power-of-two fraction of HW; <= 2048 values = two 16-byte pieces per thread
// where HW allows: at batch 1 the combine is a handful of workgroups, so each must be short)
__host__ __device__ inline int combine_rows(int HW, int N) {
    int r = HW;
    while ((r & 1) == 0 && (long long)r * N > 2048) r >>= 1;
    return r;
}
// dst[m][n] = sum_ks partial[ks][m][n] (ks ascending) + GroupNorm partial sums in the layout StatsRef{slots = HW / rows + 1,
// m_tile = rows, n_tiles = 1} with rows = combine_rows(HW, N)
// conv_skinny.hip: 3x3 / 3x1 convolutions of <= 256 rows (batch 1-4).  conv_skinny_geometry is the shape rule (false: not here)
bool conv_skinny_geometry(int M, int N, int K, int HW, int W, int taps, int split, unsigned sw, int* m_tile, int* n_tile);
hipError_t launch_conv_skinny(const GemmArgs& a, const GemmGeom& g, hipStream_t s);
// conv_reg.hip: 3x3 convolutions 64 -> 64 on width-8 maps at large batch (64-row wave tiles; one statistics slot per wave tile)
bool conv_reg_geometry(int M, int N, int K, int HW, int W, int taps, int split, unsigned sw);
hipError_t launch_conv_reg64(const GemmArgs& a, const GemmGeom& g, hipStream_t s);
hipError_t launch_splitk_combine(const float* partial, int ksplit, float* dst, int M, int N, int HW, double* stats,
                                 hipStream_t s);
hipError_t launch_gemm(const GemmArgs& a, hipStream_t s);
// would launch_gemm run this statistics-epilogue convolution on a kernel that takes a fused source (pro = PRO_POOL / PRO_UPCAT)?
// (a: the complete launch arguments, as launch_gemm would get them)
bool gemm_takes_fused_source(const GemmArgs& a);
// ... or a two-source input (pro = PRO_NONE / PRO_GN with `skip` set: channels [0, up_C) from src, [up_C, K) from skip, the
// prologue applying to the skip part only)?  conv_wide.hip's 128-wide configurations.
bool gemm_takes_two_sources(const GemmArgs& a);
double gemm_flops(const GemmArgs& a);
// conv_wide.hip: the 4-wave / 128x64-per-wave configuration of the 3x3 implicit GEMM (256 x 128 tiles, two
// workgroups per CU); launch_gemm routes to it when conv_wide_supported
bool conv_wide_supported(const GemmArgs& a, const GemmGeom& g);
hipError_t launch_conv_wide(const GemmArgs& a, const GemmGeom& g, hipStream_t s);

// Fragment-order copy of split-format weights [taps][N][K] for conv_wide.hip (v_mfma_f32_16x16x32_f16 B operands):
//   block ((tap K/32 + chunk) N/16 + nb16) x {hi, lo} of 1 KiB; lane (kg 16 + l16) -> 16 bytes = 8 fp16 of row
//   nb16 16 + l16, k = chunk 32 + kg 8 .. -- so a wave's operand load is one coalesced global_load_dwordx4.
//   Same bytes as the split array, permuted.  (N % 16 == 0, K % 32 == 0.)
inline std::vector<float> frag_order_weights(const std::vector<float>& split, int taps, int N, int K) {
    std::vector<float> out(split.size());
    const int nch = K / 32, nbn = N / 16;
    for (int t = 0; t < taps; ++t)
        for (int c = 0; c < nch; ++c)
            for (int nb = 0; nb < nbn; ++nb)
                for (int part = 0; part < 2; ++part)
                    for (int kg = 0; kg < 4; ++kg)
                        for (int l16 = 0; l16 < 16; ++l16) {
                            const size_t dst = ((((size_t)(t * nch + c) * nbn + nb) * 2 + part) * 64 + kg * 16 + l16) * 4;
                            const size_t src = ((size_t)t * N + nb * 16 + l16) * K + c * 32 + part * 16 + kg * 4;
                            for (int j = 0; j < 4; ++j) out[dst + j] = split[src + j];
                        }
    return out;
}

// ---- streaming / small kernels (elementwise.hip) -------------------------------------------
// first conv, Cin = 1, fused zero-padding of the (H0, D) trajectory to (Hp, Wp)
// ... and, as the first kernel of a denoise step, the loop bookkeeping (adv: -2 none, -1 advance by one, >= 0 set the step)
int conv_in_parts(int Hp, int Wp, int B);        // row parts per sample = statistics slots it writes (m_tile = HW / parts)
hipError_t launch_conv_in(const float* x, const float* w /*[9][64]*/, float* dst, double* stats,
                          int B, int H0, int D, int Hp, int Wp, int lh, int lw, int* step_dev, int* t_dev,
                          const int* timesteps_dev, int n_steps, int adv, hipStream_t s, int B_geom = 0);     // B_geom > 0: row parts as for that batch

struct AffineSrc {               // a tensor + the per-(sample,channel) affine that finishes it
    const float* x; int C;       // channels-last [B][HW][C]
    StatsRef st; const float* gamma; const float* beta;   // GroupNorm part (st.p may be null)
};
// MaxPool2d(2) of affine(src):  (B, H*W, C) -> (B, H/2*W/2, C)
hipError_t launch_pool(const AffineSrc& src, float* dst, int B, int H, int W, hipStream_t s);
// cat([bilinear_x2_align_corners(affine(up)), affine(skip)], channel)
hipError_t launch_upcat(const AffineSrc& up, const AffineSrc& skip, float* dst, int B, int Hin, int Win,
                        hipStream_t s);
// block tail: y = scale * (GN(x) + temb[t]) + bias   (FiLM; film == nullptr: y = GN(x) + temb)
// row_stats (optional): per-token {sum, sum^2} over C of y as fp64 -- the LayerNorm statistics the
// attention block's first GEMM applies in its load prologue
hipError_t launch_film_apply(const AffineSrc& src, const float* temb_table /*[T][C]*/, const int* t_dev,
                             int t_count, const float* film /*[B][2C] or null*/, float* dst, double* row_stats,
                             int B, int HW, hipStream_t s);
// The same tail as a recipe: the attention kernels that read the block input themselves (sa_qkv / sa_tail / sa_fused64) evaluate
// the coefficients of the samples their tile touches at kernel start (device_utils.h film_coef_row_wave: statistics slots, one
// time-embedding row, the FiLM row) instead of reading them from a film_coef_kernel launch -- one launch per block less, which is
// what a small-batch step is made of (64 launches of ~5-8 us at batch 1).
struct FilmSpec {
    StatsRef st; const float* gamma; const float* beta;   // GroupNorm(1,C) still pending on the raw tensor (st.p may be null)
    const float* temb; const int* t_dev; int t_count;     // time-embedding table [T][C] and the device timestep(s); temb may be null
    const float* film;                                    // [B][2C] = [scale | bias] of the FiLM encoder, or null
    int C;
    int on;                                               // 0: unused (the consumer takes `ab` from memory, or nothing)
};
// the same tail as per-(sample, channel) coefficients ab[b] = [A (C) | B (C)], y = A x + B, for consumers that apply it on load
hipError_t launch_film_coef(const AffineSrc& src, const float* temb_table, const int* t_dev, int t_count, const float* film,
                            float* ab, int B, hipStream_t s);
// encoder.hip: the three stride-2 2x2 convolutions (+ReLU) of the observation autoencoder's encoder
// (models/encoder/autoencoder.py:11-17), (n,3,96,96) -> flattened (n, 64*12*12) rows for its Linear layer
hipError_t launch_encoder_convs(const float* img, const float* w1, const float* b1, const float* w2, const float* b2,
                                const float* w3, const float* b3, float* feat, int n_images, hipStream_t s);
// plain GN apply (materialise): y = GN(x)
hipError_t launch_gn_apply(const AffineSrc& src, float* dst, int B, int HW, hipStream_t s);
hipError_t launch_layernorm(const float* x, const float* g, const float* b, float* y, int rows, int C,
                            hipStream_t s);
hipError_t launch_mish_pad(const float* cond, float* dst, int B, int cond_dim, int Kp, hipStream_t s);
hipError_t launch_silu(const float* x, float* y, size_t n, hipStream_t s);
hipError_t launch_gelu(const float* x, float* y, size_t n, hipStream_t s);

struct StepArgs {
    const float* feat;            // (B, Hp*Wp, 64) channels-last
    const float* w; float bias;   // outc 1x1 conv
    float* x;                     // (B, H0, D) current iterate, updated in place
    float* eps_out;               // non-null: only write eps (spdm_unet_forward)
    const float* coef;            // device [n_steps][6]
    const int* step_dev;          // device scalar: loop iteration
    int kind;
    const float* noise;           // (n_steps, B, H0, D) or null
    const unsigned long long* rng_dev;   // device {seed, first global trajectory index} of the Philox noise stream
    int* flag_dev;                // device word, set to 1 when an updated iterate (or eps) is not finite
    const float* inpaint; int inp_h; int inpaint_per_sample;
    float* history;               // (n_steps+1, B, H0, D) or null
    // Sampling loop: the three caller-owned buffers above are read from this device block {inpaint, noise, history} instead
    // (written by spdm_sample_begin), so the launch arguments -- and with them the captured step graph -- do not depend on
    // where a caller's tensors happen to live.  Null: the fields above are used as given (spdm_unet_forward).
    const void* const* ptrs_dev;
    int B, H0, D, Hp, Wp, lh, lw;
};
hipError_t launch_out_step(const StepArgs& a, hipStream_t s);
hipError_t launch_advance(int* step_dev, int* t_dev, const int* timesteps_dev, int n_steps, hipStream_t s);
hipError_t launch_set_step(int* step_dev, int* t_dev, const int* timesteps_dev, int n_steps, int i, hipStream_t s);

// ---- attention core (attention.hip): softmax(q k^T / sqrt d) v per (sample, head) -----------
hipError_t launch_attention(const float* qkv /*[B*L][3C]*/, float* out /*[B*L][C]*/, int B, int L, int C,
                            int heads, hipStream_t s);        // VALU kernel (small L)
hipError_t launch_attention_auto(const float* qkv, float* out, int B, int L, int C, int heads, unsigned sw, hipStream_t s);  // MFMA kernel for L >= 32

// ---- fused SelfAttention block for the C = 64 levels (sa_fused.hip) -------------------------------------
bool sa_fused_supported(int L, int C);
// w_hl: {Wqkv hi, lo, Wo hi, lo, W1 hi, lo, W2 hi, lo} as fp16 [rows][64], input axis permuted by perm16 inside
// each group of 16, pre-scaled by 128 (spdm_api.hip: Loader::perm_split)
// ab (optional): the block input is ab-affine of x per sample (film_coef_kernel), applied on load
hipError_t launch_sa_fused64(const float* x, float* out, int B, int L, const float* ln1_g, const float* ln1_b,
                             const float* ln2_g, const float* ln2_b, const void* const w_hl[8], const float* bqkv,
                             const float* bo, const float* b1, const float* b2, const float* ab, unsigned sw, hipStream_t s,
                             const FilmSpec* fs = nullptr);

// ---- row-wise tail of a C = 128 / 256 SelfAttention block (sa_tail.hip): out_proj + x -> LayerNorm -> ff1 -> GELU -> ff2 + av ----
bool sa_tail_supported(int C, unsigned sw);
hipError_t launch_sa_tail(int C, const float* o, const float* x, float* out, int rows, const float* wf_o, const float* wf_1,
                          const float* wf_2, const float* b_o, const float* b_1, const float* b_2, const float* ln_g,
                          const float* ln_b, const float* ab, int L, hipStream_t s, const FilmSpec* fs = nullptr);
// may the two kernels evaluate the FiLM coefficients themselves for samples of L rows?  (their LDS row per touched sample)
bool sa_tail_film_local(int C, int L);
// LayerNorm + in_proj + attention core in one kernel (att = softmax(q k^T / sqrt d) v per sample and head, heads concatenated):
// blocks whose 8192 / C-row tile holds whole samples of L tokens
bool sa_head_supported(int C, int L, unsigned sw);
hipError_t launch_sa_head(int C, const float* x, float* att, int rows, const float* wf_in, const float* b_in, const float* ln_g,
                          const float* ln_b, const float* ab, int L, hipStream_t s, const FilmSpec* fs = nullptr);
// qkv = LayerNorm(x) W_in^T + b_in of the same blocks (LayerNorm from the row itself)
hipError_t launch_sa_qkv(int C, const float* x, float* qkv, int rows, const float* wf_in, const float* b_in, const float* ln_g,
                         const float* ln_b, const float* ab, int L, hipStream_t s, const FilmSpec* fs = nullptr);

}  // namespace spdm
