// spdm_api.hip -- host side of libspdm_hip.so: the C ABI of include/spdm.h, the weight re-layout,
// the workspace arena and the launch plan of one U-Net evaluation / one denoise step.
//
// The plan follows UNet_Film.forward (models/Unet_FiLmLayer.py:277-312) block by block; every
// launch_* call names the kernel that replaces the torch ops of that line range (kernels.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/spdm.h"
#include "kernels.h"

using namespace spdm;

// -------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(SPDM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define SPDM_TRY(expr)              \
    do {                            \
        int _r = (expr);            \
        if (_r != SPDM_OK) return _r; \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// -------------------------------------------------------------------------------------------------
// Workspace arena: first-fit free list over one hipMalloc'd slab.  The plan allocates and releases
// in a deterministic order, so a dry run with max_batch at create() sizes the slab exactly.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, peak = 0;
    bool dry = false, keep = false;
    struct Blk { size_t off, size; bool free; };
    std::vector<Blk> blks;
    void reset() {
        blks.clear();
        blks.push_back({0, (size_t)1 << 60, true});
    }
    bool alloc(size_t bytes, size_t* off) {
        bytes = align_up(std::max<size_t>(bytes, 256), 256);
        for (size_t i = 0; i < blks.size(); ++i) {
            if (blks[i].free && blks[i].size >= bytes) {
                const size_t o = blks[i].off, rest = blks[i].size - bytes;
                blks[i].size = bytes;
                blks[i].free = false;
                if (rest) blks.insert(blks.begin() + i + 1, Blk{o + bytes, rest, true});
                peak = std::max(peak, o + bytes);
                if (!dry && o + bytes > cap) return false;
                *off = o;
                return true;
            }
        }
        return false;
    }
    void release(size_t off) {
        if (keep) return;
        for (size_t i = 0; i < blks.size(); ++i)
            if (blks[i].off == off && !blks[i].free) {
                blks[i].free = true;
                if (i + 1 < blks.size() && blks[i + 1].free) {
                    blks[i].size += blks[i + 1].size;
                    blks.erase(blks.begin() + i + 1);
                }
                if (i > 0 && blks[i - 1].free) {
                    blks[i - 1].size += blks[i].size;
                    blks.erase(blks.begin() + i);
                }
                return;
            }
    }
};

struct Tensor {            // channels-last activation [B][HW_level][C] living in the arena
    float* p = nullptr;
    size_t off = 0;
    int C = 0, level = 0;
    bool valid = false;
};
struct StatsBuf {          // GroupNorm partial sums of a raw conv output
    double* p = nullptr;
    size_t off = 0;
    StatsRef ref{};
    bool valid = false;
};
struct Value {             // a tensor plus the GroupNorm affine still pending on it (if any)
    Tensor t;
    StatsBuf st;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    bool pending_gn() const { return st.valid; }
};

struct ConvW { float* w = nullptr; float* ws = nullptr; float* wf = nullptr; int taps = 0, cin = 0, cout = 0; };   // ws: split-fp16 copy; wf: its fragment-order copy (conv_wide.hip)
struct DoubleConvW { ConvW first, second; float* gamma = nullptr; float* beta = nullptr; };
struct LinW { float* w = nullptr; float* ws = nullptr; float* b = nullptr; int in = 0, out = 0; };
struct ResampleW { DoubleConvW dc1, dc2; LinW emb, film; float* temb_table = nullptr; int cout = 0; };
struct AttnW {
    LinW in_proj, out_proj, ff1, ff2;
    float* ln_g = nullptr; float* ln_b = nullptr; float* ff_ln_g = nullptr; float* ff_ln_b = nullptr;
    int C = 0;
    void* fw[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // fused-kernel weights (C == 64)
    float* tail_wf[3] = {nullptr, nullptr, nullptr};   // fragment-order split copies of out_proj / ff1 / ff2 (C == 128: sa_tail.hip)
    float* qkv_wf = nullptr;                           // ... and of in_proj
};

struct ProfEvt { hipEvent_t a, b; double flops; int launches; };

struct spdm_handle {
    spdm_config cfg{};
    int Hp = 0, Wp = 0, lh = 0, uh = 0, lw = 0, uw = 0;
    int film_kp = 0;
    // weights
    std::vector<void*> owned;             // every hipMalloc to free
    float* w_inc_first = nullptr;         // [9][64]
    DoubleConvW inc, bot[3];
    ResampleW down[3], up[3];
    AttnW sa[6];
    float* outc_w = nullptr;
    float outc_b = 0.f;
    bool weights_loaded = false, temb_ready = false;
    bool split = true;                    // split-fp16 MFMA path (default); SPDM_PREC=f32 selects the exact fp32 MFMA path
    unsigned sw = 0;                      // kernel-selection switches (SW_*, kernels.h): environment read once at create
    int demoted = 0;                      // tensors outside the split format's range, kept on the exact fp32 kernels
    std::vector<float> time_table;        // host (T, time_dim)
    float* d_time_silu = nullptr;         // device SiLU(pos_encoding) (T, time_dim)
    // schedule
    int sched_kind = -1, n_steps = 0;
    std::vector<int> timesteps;
    int* d_timesteps = nullptr;
    float* d_coef = nullptr;
    // persistent per-call state (sized for max_batch)
    int* d_t = nullptr;                   // [max_batch] timestep(s) of the current evaluation
    int* d_step = nullptr;                // [0] loop iteration, [2] non-finite flag (set by out_step_kernel)
    unsigned long long* d_rng = nullptr;  // {seed, first global trajectory index} of the device noise stream
    const void** d_ptrs = nullptr;        // {inpaint, noise, history} of the session: read by out_step_kernel (StepArgs::ptrs_dev)
    float* d_condm = nullptr;             // Mish(cond), K padded
    float* d_film[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    float* d_x = nullptr;                 // current iterate (B,H0,D)
    float* d_partial = nullptr;           // split-K partial slabs of the launch in flight (SPLITK_WORKSPACE_BYTES)
    bool have_film = false;
    // sampling session
    int sB = 0, s_inp_h = 0, s_inp_per_sample = 0;
    const float* s_inpaint = nullptr;
    const float* s_noise = nullptr;
    float* s_history = nullptr;
    unsigned long long s_seed = 0, s_offset = 0;
    bool session = false;
    // arena
    Arena arena;
    size_t persistent_bytes = 0;
    // debug
    std::map<std::string, Tensor> taps;
    int tapB = 0;
    // hipGraph of one denoise step (advance -> U-Net -> scheduler update), replayed by spdm_sample_run
    struct StepGraphKey {
        int B = 0, inp_h = 0, per_sample = 0, have_film = 0, sched_kind = 0, n_steps = 0;
        unsigned env = 0;                 // the handle's kernel-selection switches when the step was captured
        // NOT part of the key: seed and trajectory offset (out_step_kernel reads them from d_rng) and the ADDRESSES of the
        // caller's inpaint / noise / history buffers (read from d_ptrs) -- a fresh seed, or freshly allocated tensors of the
        // same shapes, replay the same graph.  Presence or absence of a buffer needs no entry either: the kernel tests the
        // pointer it loads.
        bool operator==(const StepGraphKey& o) const {
            return env == o.env && B == o.B && inp_h == o.inp_h && per_sample == o.per_sample && have_film == o.have_film &&
                   sched_kind == o.sched_kind && n_steps == o.n_steps;
        }
    } graph_key;
    long long graph_captures = 0;         // step graphs built so far (spdm_graph_captures)
    int dry_fuse_mask = 0;                // dry runs only: bit k set = resampling op k (pool 0-2, upsample+concat 3-5) is read through
                                          // by its consumer (Ctx::conv_fused), bits 6-8 = up block k - 6 takes the two-source
                                          // conv (Ctx::conv_two) -- the workspace is sized for every combination
    hipGraph_t step_graph = nullptr;
    hipGraphExec_t step_exec = nullptr;
    hipStream_t gstream = nullptr;        // blocking stream the loop runs on when the caller passes the NULL stream
    // profiling of the dominant kernel class
    bool prof = false;
    std::vector<ProfEvt> prof_evts;       // event pairs in use since the last reset
    std::vector<ProfEvt> prof_pool;       // created ahead of time (spdm_profile_enable(h, 2)) / recycled: no hipEventCreate while timing
    int prof_open = -1;                   // index of the event pair that brackets the current run of consecutive conv launches
    long long prof_launches = 0;
    double prof_ms = 0.0, prof_flops = 0.0;
};

static int dev_alloc(spdm_handle* h, void** p, size_t bytes) {
    HIP_TRY(hipMalloc(p, std::max<size_t>(bytes, 256)));
    h->owned.push_back(*p);
    h->persistent_bytes += bytes;
    return SPDM_OK;
}

// -------------------------------------------------------------------------------------------------
// schedule tables (host, fp32, operation order of diffusers 0.17.1 -- see oracle/scheduler_ref.py)
#pragma clang fp contract(off)
static int build_schedule(int kind, int T, int n, float beta_start, float beta_end, int* ts, float* coef) {
    if (T < 1 || n < 1 || n > T) return fail(SPDM_ERR_INVALID, "schedule: need 1 <= n (%d) <= T (%d)", n, T);
    if (kind != SPDM_DDPM && kind != SPDM_DDIM) return fail(SPDM_ERR_INVALID, "schedule: unknown kind %d", kind);
    std::vector<float> acp(T);
    {   // torch.linspace(fp32): symmetric evaluation from both ends; alphas = 1 - betas; cumprod
        const float step = (T > 1) ? (beta_end - beta_start) / (float)(T - 1) : 0.f;
        float run = 1.0f;
        for (int i = 0; i < T; ++i) {
            const float beta = (i < T / 2) ? beta_start + step * (float)i : beta_end - step * (float)(T - 1 - i);
            const float alpha = 1.0f - beta;
            run = (i == 0) ? alpha : run * alpha;
            acp[i] = run;
        }
    }
    const int ratio = T / n;
    for (int i = 0; i < n; ++i) {
        const int t = (n - 1 - i) * ratio;
        const int prev_t = t - ratio;
        ts[i] = t;
        const float a_t = acp[t];
        const float a_prev = (prev_t >= 0) ? acp[prev_t] : 1.0f;
        const float b_t = 1.0f - a_t;
        float* c = coef + (size_t)i * 6;
        c[0] = sqrtf(b_t);
        c[1] = sqrtf(a_t);
        c[2] = c[3] = c[4] = c[5] = 0.f;
        if (kind == SPDM_DDPM) {
            const float b_prev = 1.0f - a_prev;
            const float cur_a = a_t / a_prev;
            const float cur_b = 1.0f - cur_a;
            c[2] = (sqrtf(a_prev) * cur_b) / b_t;
            c[3] = sqrtf(cur_a) * b_prev / b_t;
            if (t > 0) {
                float var = (1.0f - a_prev) / (1.0f - a_t) * cur_b;
                if (var < 1e-20f) var = 1e-20f;
                c[5] = sqrtf(var);
            }
        } else {
            c[2] = sqrtf(a_prev);
            c[4] = sqrtf(1.0f - a_prev - 0.0f);
        }
    }
    return SPDM_OK;
}

// pos_encoding(t) for t = 0..T-1 (models/Unet_FiLmLayer.py:266-274), default host computation
static void default_time_table(std::vector<float>& tab, int T, int dim) {
    tab.resize((size_t)T * dim);
    const int half = dim / 2;
    for (int i = 0; i < half; ++i) {
        const float expo = (float)(2 * i) / (float)dim;
        const float inv_freq = 1.0f / powf(10000.0f, expo);
        for (int t = 0; t < T; ++t) {
            const float arg = (float)t * inv_freq;
            tab[(size_t)t * dim + i] = sinf(arg);
            tab[(size_t)t * dim + half + i] = cosf(arg);
        }
    }
}

// -------------------------------------------------------------------------------------------------
unsigned spdm::switches_from_env() {
    int n = 0;
    const SwitchName* t = switch_table(&n);
    unsigned sw = 0;
    for (int i = 0; i < n; ++i)
        if (getenv(t[i].env) != nullptr) sw |= t[i].bit;
    return sw;
}

extern "C" int spdm_abi_version(void) { return SPDM_ABI_VERSION; }
extern "C" const char* spdm_last_error(void) { return g_err; }

extern "C" int spdm_schedule_tables(int32_t kind, int32_t T, int32_t n, float beta_start, float beta_end,
                                    int32_t* ts, float* coef) {
    if (!ts || !coef) return fail(SPDM_ERR_INVALID, "null output");
    return build_schedule(kind, T, n, beta_start, beta_end, ts, coef);
}

static int plan_forward(spdm_handle* h, int B, bool use_cond, hipStream_t s, Tensor* feat_out);
// dry runs of the plan at max_batch for every fused / materialised combination of the six resampling ops (Ctx::conv_fused):
// arena.peak ends as the largest of them (arena.dry must be set by the caller)
static int dry_plan_all(spdm_handle* h) {
    size_t peak = 0;
    for (int mask = 0; mask < 512; ++mask) {      // bits 0-2 pool read through, 3-5 upsample + concat read through, 6-8 two-source conv
        if (((mask >> 3) & (mask >> 6) & 7) != 0) continue;       // (a block is one or the other)
        h->dry_fuse_mask = mask;
        h->arena.peak = 0;
        h->arena.reset();
        Tensor feat;
        const int rc = plan_forward(h, h->cfg.max_batch, true, nullptr, &feat);
        if (rc != SPDM_OK) { h->dry_fuse_mask = 0; return rc; }
        peak = std::max(peak, h->arena.peak);
    }
    h->dry_fuse_mask = 0;
    h->arena.peak = peak;
    return SPDM_OK;
}

// channel / tap geometry of every layer (UNet_Film.__init__, models/Unet_FiLmLayer.py:246-264); known
// before any weight is loaded, so that create() can size the workspace with a dry run of the plan
static void init_arch(spdm_handle* h) {
    // level-3 maps are (Hp/8) x 1: a 3x3 kernel only ever multiplies its centre column there
    const int t3 = (h->Wp >> 3) == 1 ? 3 : 9;
    auto dc = [](DoubleConvW& d, int cin, int cout, int taps) {
        d.first.cin = cin; d.first.cout = cout; d.first.taps = taps;
        d.second.cin = cout; d.second.cout = cout; d.second.taps = taps;
    };
    auto rs = [&](ResampleW& r, int cin, int cout, int taps) {
        dc(r.dc1, cin, cin, taps);
        dc(r.dc2, cin, cout, taps);
        r.cout = cout;
    };
    dc(h->inc, 1, 64, 9);
    rs(h->down[0], 64, 128, 9);
    rs(h->down[1], 128, 256, 9);
    rs(h->down[2], 256, 256, t3);
    dc(h->bot[0], 256, 512, t3);
    dc(h->bot[1], 512, 512, t3);
    dc(h->bot[2], 512, 256, t3);
    rs(h->up[0], 512, 128, 9);
    rs(h->up[1], 256, 64, 9);
    rs(h->up[2], 128, 64, 9);
    static const int sc[6] = {128, 256, 256, 128, 64, 64};
    for (int i = 0; i < 6; ++i) h->sa[i].C = sc[i];
}

extern "C" int spdm_create(const spdm_config* cfg, spdm_handle** out) {
    if (!cfg || !out) return fail(SPDM_ERR_INVALID, "null argument");
    if (cfg->horizon < 1 || cfg->state_dim < 1 || cfg->state_dim > 8)
        return fail(SPDM_ERR_INVALID, "horizon must be >= 1 and 1 <= state_dim <= 8 (got %d, %d)", cfg->horizon, cfg->state_dim);
    if (cfg->time_dim < 32 || cfg->time_dim % 32 != 0) return fail(SPDM_ERR_INVALID, "time_dim must be a multiple of 32");
    if (cfg->cond_dim < 0 || cfg->max_batch < 1 || cfg->num_train_timesteps < 1)
        return fail(SPDM_ERR_INVALID, "cond_dim >= 0, max_batch >= 1, num_train_timesteps >= 1 required");
    HIP_TRY(hipSetDevice(cfg->device));
    spdm_handle* h = new spdm_handle();
    h->cfg = *cfg;
    h->sw = switches_from_env();          // the ONLY place the product path reads SPDM_* switches
    if (const char* pe = getenv("SPDM_PREC")) h->split = !(strcmp(pe, "f32") == 0 || strcmp(pe, "fp32") == 0);
    if (cfg->flags & SPDM_FLAG_EXACT_FP32) h->split = false;
    // pad_to(x, 8): models/Unet_FiLmLayer.py:15-28
    const int H0 = cfg->horizon, D = cfg->state_dim;
    h->Hp = (H0 % 8) ? H0 + 8 - H0 % 8 : H0;
    h->Wp = (D % 8) ? D + 8 - D % 8 : D;
    h->lh = (h->Hp - H0) / 2;  h->uh = (h->Hp - H0) - h->lh;
    h->lw = (h->Wp - D) / 2;   h->uw = (h->Wp - D) - h->lw;
    h->film_kp = (int)align_up((size_t)std::max(cfg->cond_dim, 1), 32);
    const int mb = cfg->max_batch;
    init_arch(h);
    int rc = SPDM_OK;
    do {
        if ((rc = dev_alloc(h, (void**)&h->d_t, sizeof(int) * mb))) break;
        if ((rc = dev_alloc(h, (void**)&h->d_step, sizeof(int) * 4))) break;
        if ((rc = dev_alloc(h, (void**)&h->d_rng, sizeof(unsigned long long) * 2))) break;
        if ((rc = dev_alloc(h, (void**)&h->d_ptrs, sizeof(void*) * 4))) break;
        if (hipMemset(h->d_step, 0, sizeof(int) * 4) != hipSuccess) { rc = fail(SPDM_ERR_HIP, "memset failed"); break; }
        if ((rc = dev_alloc(h, (void**)&h->d_condm, sizeof(float) * (size_t)mb * h->film_kp))) break;
        static const int film_c[6] = {128, 256, 256, 128, 64, 64};
        for (int i = 0; i < 6 && !rc; ++i) rc = dev_alloc(h, (void**)&h->d_film[i], sizeof(float) * (size_t)mb * 2 * film_c[i]);
        if (rc) break;
        if ((rc = dev_alloc(h, (void**)&h->d_x, sizeof(float) * (size_t)mb * H0 * D))) break;
        if (h->split && !(h->sw & SW_NO_SPLITK) && (rc = dev_alloc(h, (void**)&h->d_partial, SPLITK_WORKSPACE_BYTES))) break;
        // size the arena with a dry run of the plan at max_batch
        h->arena.dry = true;
        h->arena.keep = (cfg->flags & SPDM_FLAG_DEBUG_KEEP) != 0;
        h->arena.reset();
        if ((rc = dry_plan_all(h))) break;
        h->arena.dry = false;
        h->arena.cap = align_up(h->arena.peak, 4096);
        hipError_t e = hipMalloc((void**)&h->arena.base, h->arena.cap);
        if (e != hipSuccess) { rc = fail(SPDM_ERR_NOMEM, "workspace of %zu bytes: %s", h->arena.cap, hipGetErrorString(e)); break; }
        h->owned.push_back(h->arena.base);
        if (h->sw & SW_ARENA_TRACE) fprintf(stderr, "[spdm] arena %p, %zu MiB\n", (void*)h->arena.base, h->arena.cap >> 20);
    } while (0);
    if (rc) { spdm_destroy(h); return rc; }
    default_time_table(h->time_table, cfg->num_train_timesteps, cfg->time_dim);
    *out = h;
    return SPDM_OK;
}

extern "C" void spdm_destroy(spdm_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    if (h->step_exec) (void)hipGraphExecDestroy(h->step_exec);
    if (h->step_graph) (void)hipGraphDestroy(h->step_graph);
    if (h->gstream) (void)hipStreamDestroy(h->gstream);
    for (void* p : h->owned) (void)hipFree(p);
    for (auto& e : h->prof_evts) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : h->prof_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    delete h;
}

extern "C" int32_t spdm_uses_split_precision(const spdm_handle* h) { return (h && h->split) ? 1 : 0; }

extern "C" size_t spdm_device_bytes(const spdm_handle* h) { return h ? h->persistent_bytes + h->arena.cap : 0; }

// -------------------------------------------------------------------------------------------------
// weights
struct Loader {
    spdm_handle* h;
    const float* blob;
    size_t n;
    std::map<std::string, const spdm_tensor_index*> idx;
    int err = SPDM_OK;
    std::set<std::string> demoted_names;   // tensors outside the split format's range (counted once each, however many copies they lose)
    void demote(const std::string& name) { demoted_names.insert(name); if (h) h->demoted = (int)demoted_names.size(); }
    const float* find(const std::string& name, std::initializer_list<int> shape) {
        auto it = idx.find(name);
        if (it == idx.end()) { err = fail(SPDM_ERR_MISSING, "tensor '%s' not in the index", name.c_str()); return nullptr; }
        const spdm_tensor_index* e = it->second;
        size_t numel = 1;
        int d = 0;
        for (int sdim : shape) {
            if (d >= e->ndim || e->shape[d] != sdim) { err = fail(SPDM_ERR_INVALID, "tensor '%s': unexpected shape", name.c_str()); return nullptr; }
            numel *= (size_t)sdim;
            ++d;
        }
        if (d != e->ndim || numel != e->numel || e->offset + e->numel > n) { err = fail(SPDM_ERR_INVALID, "tensor '%s': bad extent", name.c_str()); return nullptr; }
        return blob + e->offset;
    }
    // fp32 [rows][K] -> per 32-k chunk [32 x fp16 hi | 32 x fp16 lo] of x' = 128 x (conv_gemm.hip, PREC_SPLIT):
    // hi = fp16(x'), lo = fp16(x' - hi); same byte size as the fp32 array
    // The split format scales weights by 2^7 before the fp16 cast: |w| >= 511.75 would become inf.  Such a tensor gets no
    // split copy and its layer runs on the exact fp32 kernel instead (same results, 5x slower for that layer).
    static bool split_range_ok(const std::vector<float>& v) {
        float mx = 0.f;
        for (float x : v) { const float ax = std::fabs(x); if (!(ax <= mx)) mx = ax; }    // NaN-propagating max
        return mx < 511.0f;
    }
    static std::vector<float> split_format(const std::vector<float>& v) {
        std::vector<float> out(v.size());
        for (size_t base = 0; base < v.size(); base += 32) {
            _Float16* hp = reinterpret_cast<_Float16*>(&out[base]);
            for (int j = 0; j < 32; ++j) {
                const float x = v[base + j] * 128.0f;
                const _Float16 hi = (_Float16)x;
                hp[j] = hi;
                hp[32 + j] = (_Float16)(x - (float)hi);
            }
        }
        return out;
    }
    float* upload_split(const std::vector<float>& v, size_t K, const std::string& name) {
        if (K % 32 != 0) { err = fail(SPDM_ERR_INVALID, "split weights need K %% 32 == 0"); return nullptr; }
        if (!split_range_ok(v)) { demote(name); return nullptr; }
        return upload(split_format(v));
    }
    float* upload(const std::vector<float>& v) {
        void* p = nullptr;
        if (dev_alloc(h, &p, v.size() * sizeof(float)) != SPDM_OK) { err = SPDM_ERR_HIP; return nullptr; }
        if (hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            err = fail(SPDM_ERR_HIP, "weight upload failed");
            return nullptr;
        }
        return (float*)p;
    }
    // (Cout,Cin,3,3) -> [taps][Cout][Cin]; taps == 3 keeps only the centre column (W == 1 levels, where
    // the left/right taps only ever see zero padding)
    ConvW conv(const std::string& name, int cout, int cin, int taps) {
        ConvW c;
        const float* src = find(name, {cout, cin, 3, 3});
        if (!src) return c;
        std::vector<float> v((size_t)taps * cout * cin);
        for (int t = 0; t < taps; ++t) {
            const int kh = (taps == 9) ? t / 3 : t, kw = (taps == 9) ? t % 3 : 1;
            for (int o = 0; o < cout; ++o)
                for (int i = 0; i < cin; ++i)
                    v[((size_t)t * cout + o) * cin + i] = src[(((size_t)o * cin + i) * 3 + kh) * 3 + kw];
        }
        c.w = upload(v);
        c.ws = (cin % 32 == 0) ? upload_split(v, cin, name) : nullptr;
        if (c.ws && cout % 64 == 0) c.wf = upload(frag_order_weights(split_format(v), taps, cout, cin));
        c.taps = taps; c.cin = cin; c.cout = cout;
        return c;
    }
    float* vec(const std::string& name, int nelem) {
        const float* src = find(name, {nelem});
        if (!src) return nullptr;
        return upload(std::vector<float>(src, src + nelem));
    }
    LinW linear(const std::string& wname, const std::string& bname, int out, int in, int in_pad) {
        LinW l;
        const float* w = find(wname, {out, in});
        if (!w) return l;
        std::vector<float> v((size_t)out * in_pad, 0.f);
        for (int o = 0; o < out; ++o) memcpy(&v[(size_t)o * in_pad], w + (size_t)o * in, sizeof(float) * in);
        l.w = upload(v);
        l.ws = (in_pad % 32 == 0) ? upload_split(v, in_pad, wname) : nullptr;
        l.b = vec(bname, out);
        l.in = in_pad; l.out = out;
        return l;
    }
    DoubleConvW dconv(const std::string& p, int cin, int cout, int taps) {
        DoubleConvW d;
        d.first = conv(p + ".first.weight", cout, cin, taps);
        d.second = conv(p + ".second.weight", cout, cout, taps);
        d.gamma = vec(p + ".norm.weight", cout);
        d.beta = vec(p + ".norm.bias", cout);
        return d;
    }
    ResampleW resample(const std::string& p, int cin, int cout, int taps) {
        ResampleW r;
        r.dc1 = dconv(p + ".doubleConv1", cin, cin, taps);
        r.dc2 = dconv(p + ".doubleConv2", cin, cout, taps);
        r.emb = linear(p + ".emb_layer.1.weight", p + ".emb_layer.1.bias", cout, h->cfg.time_dim, h->cfg.time_dim);
        if (h->cfg.cond_dim > 0)
            r.film = linear(p + ".cond_encoder.2.weight", p + ".cond_encoder.2.bias", 2 * cout, h->cfg.cond_dim, h->film_kp);
        r.cout = cout;
        return r;
    }
    // (out, 64) fp32 -> two fp16 arrays (hi, lo of 128 x) of [out][64] in fragment order, input axis permuted inside each group of 16 by
    // perm16 = 0 1 2 3 8 9 10 11 | 4 5 6 7 12 13 14 15: the k-slot order of an accumulator tile used as B operand
    // (sa_fused.hip)
    void perm_split(const std::string& wname, int out, void** hi_dev, void** lo_dev) {
        static const int perm16[16] = {0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15};
        const float* w = find(wname, {out, 64});
        if (!w) return;
        if (!split_range_ok(std::vector<float>(w, w + (size_t)out * 64))) { demote(wname); return; }   // block falls back to the GEMM chain
        std::vector<_Float16> hi((size_t)out * 64), lo((size_t)out * 64);
        for (int o = 0; o < out; ++o)
            for (int g = 0; g < 4; ++g)
                for (int pos = 0; pos < 16; ++pos) {
                    const float x = w[(size_t)o * 64 + 16 * g + perm16[pos]] * 128.0f;
                    const _Float16 h16 = (_Float16)x;
                    hi[(size_t)o * 64 + 16 * g + pos] = h16;
                    lo[(size_t)o * 64 + 16 * g + pos] = (_Float16)(x - (float)h16);
                }
        // device layout = MFMA A-fragment order: block (o / 32) * 4 + ks, lane kh * 32 + o % 32 -> halfs [o][16 ks + 8 kh ..]
        auto to_frag = [&](const std::vector<_Float16>& rm) {
            std::vector<_Float16> f(rm.size());
            for (int o = 0; o < out; ++o)
                for (int ks = 0; ks < 4; ++ks)
                    for (int kh = 0; kh < 2; ++kh)
                        for (int j = 0; j < 8; ++j)
                            f[((((size_t)(o / 32) * 4 + ks) * 64) + kh * 32 + o % 32) * 8 + j] = rm[(size_t)o * 64 + 16 * ks + 8 * kh + j];
            return f;
        };
        hi = to_frag(hi);
        lo = to_frag(lo);
        for (int which = 0; which < 2; ++which) {
            void* d = nullptr;
            const std::vector<_Float16>& src = which ? lo : hi;
            if (dev_alloc(h, &d, src.size() * sizeof(_Float16)) != SPDM_OK) { err = SPDM_ERR_HIP; return; }
            if (hipMemcpy(d, src.data(), src.size() * sizeof(_Float16), hipMemcpyHostToDevice) != hipSuccess) {
                err = fail(SPDM_ERR_HIP, "weight upload failed");
                return;
            }
            *(which ? lo_dev : hi_dev) = d;
        }
    }
    // fragment-order split copy of a (out, in) Linear weight (sa_tail.hip reads its B operands straight from it)
    float* linear_frag(const std::string& wname, int out, int in) {
        const float* w = find(wname, {out, in});
        if (!w) return nullptr;
        if (!split_range_ok(std::vector<float>(w, w + (size_t)out * in))) { demote(wname); return nullptr; }
        return upload(frag_order_weights(split_format(std::vector<float>(w, w + (size_t)out * in)), 1, out, in));
    }
    AttnW attn(const std::string& p, int C) {
        AttnW a;
        a.C = C;
        if (C == 128 || C == 256) {
            a.tail_wf[0] = linear_frag(p + ".attention.out_proj.weight", C, C);
            a.tail_wf[1] = linear_frag(p + ".ff_self.1.weight", C, C);
            a.tail_wf[2] = linear_frag(p + ".ff_self.3.weight", C, C);
            a.qkv_wf = linear_frag(p + ".attention.in_proj_weight", 3 * C, C);
        }
        if (C == 64) {
            perm_split(p + ".attention.in_proj_weight", 192, &a.fw[0], &a.fw[1]);
            perm_split(p + ".attention.out_proj.weight", 64, &a.fw[2], &a.fw[3]);
            perm_split(p + ".ff_self.1.weight", 64, &a.fw[4], &a.fw[5]);
            perm_split(p + ".ff_self.3.weight", 64, &a.fw[6], &a.fw[7]);
            for (int k = 0; k < 8; ++k)
                if (!a.fw[k]) { a.fw[0] = nullptr; break; }        // the fused kernel needs all four matrices
        }
        if ((C == 128 || C == 256) && !(a.tail_wf[0] && a.tail_wf[1] && a.tail_wf[2])) a.tail_wf[0] = nullptr;
        a.in_proj = linear(p + ".attention.in_proj_weight", p + ".attention.in_proj_bias", 3 * C, C, C);
        a.out_proj = linear(p + ".attention.out_proj.weight", p + ".attention.out_proj.bias", C, C, C);
        a.ln_g = vec(p + ".ln.weight", C);
        a.ln_b = vec(p + ".ln.bias", C);
        a.ff_ln_g = vec(p + ".ff_self.0.weight", C);
        a.ff_ln_b = vec(p + ".ff_self.0.bias", C);
        a.ff1 = linear(p + ".ff_self.1.weight", p + ".ff_self.1.bias", C, C, C);
        a.ff2 = linear(p + ".ff_self.3.weight", p + ".ff_self.3.bias", C, C, C);
        return a;
    }
};

// dry run of the plan at max_batch with the handle's current kernel selection; grows the slab if this plan needs more
static int replan_arena(spdm_handle* h) {
    (void)hipDeviceSynchronize();         // nothing may still be running in the slab about to be re-planned
    const bool keep = h->arena.keep;
    h->arena.dry = true;
    const int rc = dry_plan_all(h);
    h->arena.dry = false;
    h->arena.keep = keep;
    if (rc != SPDM_OK) return rc;
    if (align_up(h->arena.peak, 4096) > h->arena.cap) {
        char* nb = nullptr;
        const size_t cap = align_up(h->arena.peak, 4096);
        hipError_t e = hipMalloc((void**)&nb, cap);
        if (e != hipSuccess) return fail(SPDM_ERR_NOMEM, "workspace of %zu bytes: %s", cap, hipGetErrorString(e));
        h->owned.push_back(nb);        // (the smaller slab stays owned until destroy)
        h->arena.base = nb;
        h->arena.cap = cap;
    }
    return SPDM_OK;
}

extern "C" int spdm_load_weights(spdm_handle* h, const float* blob, size_t n, const spdm_tensor_index* index,
                                 int32_t n_index) {
    if (!h || !blob || !index || n_index <= 0) return fail(SPDM_ERR_INVALID, "null argument");
    if (h->weights_loaded) return fail(SPDM_ERR_STATE, "weights already loaded on this handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    Loader L{h, blob, n};
    for (int i = 0; i < n_index; ++i) {
        char name[SPDM_NAME_MAX + 1];
        memcpy(name, index[i].name, SPDM_NAME_MAX);
        name[SPDM_NAME_MAX] = 0;
        L.idx[name] = &index[i];
    }
    // level-3 maps are (Hp/8) x 1: a 3x3 kernel only ever multiplies its centre column there
    const int t3 = (h->Wp >> 3) == 1 ? 3 : 9;
    {   // inc.first (64,1,3,3) -> [9][64]
        const float* src = L.find("inc.first.weight", {64, 1, 3, 3});
        if (src) {
            std::vector<float> v(9 * 64);
            for (int t = 0; t < 9; ++t)
                for (int o = 0; o < 64; ++o) v[t * 64 + o] = src[o * 9 + t];
            h->w_inc_first = L.upload(v);
        }
    }
    h->inc.second = L.conv("inc.second.weight", 64, 64, 9);
    h->inc.gamma = L.vec("inc.norm.weight", 64);
    h->inc.beta = L.vec("inc.norm.bias", 64);
    h->down[0] = L.resample("down1", 64, 128, 9);
    h->down[1] = L.resample("down2", 128, 256, 9);
    h->down[2] = L.resample("down3", 256, 256, t3);
    h->bot[0] = L.dconv("bot1", 256, 512, t3);
    h->bot[1] = L.dconv("bot2", 512, 512, t3);
    h->bot[2] = L.dconv("bot3", 512, 256, t3);
    h->up[0] = L.resample("up1", 512, 128, 9);
    h->up[1] = L.resample("up2", 256, 64, 9);
    h->up[2] = L.resample("up3", 128, 64, 9);
    if (h->cfg.attention) {
        static const int sc[6] = {128, 256, 256, 128, 64, 64};
        for (int i = 0; i < 6; ++i) h->sa[i] = L.attn("sa" + std::to_string(i + 1), sc[i]);
    }
    h->outc_w = nullptr;
    {
        const float* w = L.find("outc.weight", {1, 64, 1, 1});
        const float* b = L.find("outc.bias", {1});
        if (w && b) {
            h->outc_w = L.upload(std::vector<float>(w, w + 64));
            h->outc_b = b[0];
        }
    }
    if (L.err != SPDM_OK) return L.err;
    // time-embedding tables (one per resample block), filled lazily on the first evaluation
    ResampleW* blocks[6] = {&h->down[0], &h->down[1], &h->down[2], &h->up[0], &h->up[1], &h->up[2]};
    for (ResampleW* r : blocks)
        SPDM_TRY(dev_alloc(h, (void**)&r->temb_table, sizeof(float) * (size_t)h->cfg.num_train_timesteps * r->cout));
    SPDM_TRY(dev_alloc(h, (void**)&h->d_time_silu, sizeof(float) * (size_t)h->cfg.num_train_timesteps * h->cfg.time_dim));
    h->weights_loaded = true;
    h->temb_ready = false;
    if (h->demoted > 0) {
        // Some layers left the kernels the workspace was sized for (create() planned by shape only): plan again with the
        // weights known and grow the slab if this plan needs more.
        SPDM_TRY(replan_arena(h));
    }
    return SPDM_OK;
}

extern "C" int32_t spdm_demoted_tensors(const spdm_handle* h) { return h ? h->demoted : 0; }

extern "C" int spdm_set_switch(spdm_handle* h, const char* name, int32_t on) {
    if (!h || !name) return fail(SPDM_ERR_INVALID, "null argument");
    int n = 0;
    const SwitchName* t = switch_table(&n);
    for (int i = 0; i < n; ++i)
        if (strcmp(t[i].env, name) == 0) {
            h->sw = on ? (h->sw | t[i].bit) : (h->sw & ~t[i].bit);
            HIP_TRY(hipSetDevice(h->cfg.device));
            return replan_arena(h);      // the arena was sized for the previous selection (the unfused fallbacks allocate more)
        }
    return fail(SPDM_ERR_INVALID, "unknown switch '%s'", name);
}

extern "C" int spdm_set_time_table(spdm_handle* h, const float* tab, int32_t T) {
    if (!h || !tab) return fail(SPDM_ERR_INVALID, "null argument");
    if (T != h->cfg.num_train_timesteps) return fail(SPDM_ERR_INVALID, "time table has %d rows, handle was created for %d", T, h->cfg.num_train_timesteps);
    h->time_table.assign(tab, tab + (size_t)T * h->cfg.time_dim);
    h->temb_ready = false;
    return SPDM_OK;
}

static int install_schedule(spdm_handle* h, int kind, int n, const int* ts, const float* coef) {
    HIP_TRY(hipSetDevice(h->cfg.device));
    for (int i = 0; i < n; ++i)
        if (ts[i] < 0 || ts[i] >= h->cfg.num_train_timesteps)
            return fail(SPDM_ERR_INVALID, "timestep %d outside the handle's time table [0,%d)", ts[i], h->cfg.num_train_timesteps);
    if (n > h->n_steps || !h->d_coef) {
        SPDM_TRY(dev_alloc(h, (void**)&h->d_timesteps, sizeof(int) * n));
        SPDM_TRY(dev_alloc(h, (void**)&h->d_coef, sizeof(float) * 6 * n));
    }
    HIP_TRY(hipMemcpy(h->d_timesteps, ts, sizeof(int) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_coef, coef, sizeof(float) * 6 * n, hipMemcpyHostToDevice));
    h->timesteps.assign(ts, ts + n);
    h->n_steps = n;
    h->sched_kind = kind;
    h->session = false;
    return SPDM_OK;
}

extern "C" int spdm_set_schedule(spdm_handle* h, int32_t kind, int32_t T, int32_t n, float b0, float b1) {
    if (!h) return fail(SPDM_ERR_INVALID, "null handle");
    if (T > h->cfg.num_train_timesteps) return fail(SPDM_ERR_INVALID, "T = %d exceeds the handle's time table (%d)", T, h->cfg.num_train_timesteps);
    std::vector<int> ts(std::max(n, 1));
    std::vector<float> coef((size_t)std::max(n, 1) * 6);
    SPDM_TRY(build_schedule(kind, T, n, b0, b1, ts.data(), coef.data()));
    return install_schedule(h, kind, n, ts.data(), coef.data());
}

extern "C" int spdm_set_schedule_tables(spdm_handle* h, int32_t kind, int32_t n, const int32_t* ts, const float* coef) {
    if (!h || !ts || !coef || n < 1) return fail(SPDM_ERR_INVALID, "bad argument");
    if (kind != SPDM_DDPM && kind != SPDM_DDIM) return fail(SPDM_ERR_INVALID, "unknown scheduler kind %d", kind);
    return install_schedule(h, kind, n, ts, coef);
}

// -------------------------------------------------------------------------------------------------
// plan helpers
struct Ctx {
    spdm_handle* h;
    int B;
    hipStream_t s;
    bool dry;
    int err = SPDM_OK;
    // batch the launch geometry is chosen for: the call's, or -- SW_PIN_GEOMETRY -- always the handle's max_batch, so that a shard
    // of a larger batch selects exactly the kernels (tiles, split-K, small-grid kernel, fused sources) the whole batch would
    int Bg() const { return (h->sw & SW_PIN_GEOMETRY) ? h->cfg.max_batch : B; }
    int HWl(int l) const { return (h->Hp >> l) * (h->Wp >> l); }
    int Hl(int l) const { return h->Hp >> l; }
    int Wl(int l) const { return h->Wp >> l; }

    Tensor talloc(int C, int level) {
        Tensor t;
        t.C = C; t.level = level;
        size_t off = 0;
        if (!h->arena.alloc(sizeof(float) * (size_t)B * HWl(level) * C, &off)) {
            if (!err) err = fail(SPDM_ERR_NOMEM, "workspace exhausted (batch %d)", B);
            return t;
        }
        t.off = off; t.p = (float*)(h->arena.base + off); t.valid = true;
        if (!dry && (h->sw & SW_ARENA_TRACE)) fprintf(stderr, "[spdm] level %d C %3d at %8.2f MiB (%.1f MiB)\n", level, C, off / 1048576.0, (double)B * HWl(level) * C * 4 / 1048576.0);
        return t;
    }
    Tensor ralloc(int rows, int C) {       // [rows][C] scratch (attention path)
        Tensor t;
        t.C = C; t.level = -1;
        size_t off = 0;
        if (!h->arena.alloc(sizeof(float) * (size_t)rows * C, &off)) {
            if (!err) err = fail(SPDM_ERR_NOMEM, "workspace exhausted (batch %d)", B);
            return t;
        }
        t.off = off; t.p = (float*)(h->arena.base + off); t.valid = true;
        return t;
    }
    StatsBuf salloc(int HW, int C, int m_tile, int n_tiles) {
        StatsBuf sb;
        const int slots = stats_slots(HW, m_tile, n_tiles);
        // reserve for the finest tiling any batch size can select (gemm_geometry is batch-dependent, the
        // dry run that sizes the arena is not): 128-row x 64-wide tiles, or the split-K combine kernel's row groups
        const int slots_max = std::max(std::max(std::max(slots, stats_slots(HW, 128, std::max(1, C / 64))),   // (coarser tilings need fewer)
                                                stats_slots(HW, combine_rows(HW, C), 1)),
                                       std::max(stats_slots(HW, 16, std::max(1, C / 16)),              // conv_skinny's finest tiling
                                                stats_slots(HW, std::max(HW / 4, 1), 1)));             // conv_in_kernel's four row parts
        size_t off = 0;
        if (!h->arena.alloc(sizeof(double) * 2 * (size_t)B * slots_max, &off)) {
            if (!err) err = fail(SPDM_ERR_NOMEM, "workspace exhausted (batch %d)", B);
            return sb;
        }
        sb.off = off; sb.p = (double*)(h->arena.base + off); sb.valid = true;
        sb.ref.p = sb.p; sb.ref.slots = slots; sb.ref.m_tile = m_tile; sb.ref.n_tiles = n_tiles; sb.ref.HW = HW;
        sb.ref.inv_count = 1.0 / ((double)C * (double)HW);
        return sb;
    }
    void free(Tensor& t) { if (t.valid) { h->arena.release(t.off); t.valid = false; } }
    void free(StatsBuf& s2) { if (s2.valid) { h->arena.release(s2.off); s2.valid = false; } }
    void free(Value& v) { free(v.t); free(v.st); }
    void check(hipError_t e, const char* what) {
        if (e != hipSuccess && !err) err = fail(SPDM_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    }
    AffineSrc asrc(const Value& v) const {
        AffineSrc a{};
        a.x = v.t.p; a.C = v.t.C;
        if (v.pending_gn()) { a.st = v.st.ref; a.gamma = v.gamma; a.beta = v.beta; }
        else { a.st = StatsRef{}; a.st.p = nullptr; a.gamma = nullptr; a.beta = nullptr; }
        return a;
    }
    void tap(const char* name, const Tensor& t) {
        if (h->arena.keep && !dry) { h->taps[name] = t; h->tapB = B; }
    }

    // one 3x3 conv: reads `in` (finishing its pending GroupNorm, + GELU if asked, in the load
    // prologue), writes the raw output and its GroupNorm partial sums
    Value conv(const Value& in, const ConvW& w, int level, bool gelu, const float* gamma, const float* beta) {
        Value out;
        const int HW = HWl(level), M = B * HW;
        // by shape before the weights are known (the dry run at create); afterwards a tensor outside the split format's
        // range has no split copy and stays on the exact fp32 kernel (Loader::conv)
        const int split = (h->split && w.cin % 32 == 0 && (!h->weights_loaded || w.ws)) ? 1 : 0;
        const GemmGeom g = gemm_geometry(Bg() * HW, w.cout, w.cin, HW, Wl(level), w.taps, split, h->sw, /*stats_epi=*/h->d_partial != nullptr);
        out.t = talloc(w.cout, level);
        out.st = salloc(HW, w.cout, g.st_m_tile, g.st_n_tiles);
        out.gamma = gamma; out.beta = beta;
        if (err || dry) return out;
        GemmArgs a{};
        a.sw = h->sw;
        a.split = split;
        a.src = in.t.p; a.src_ld = in.t.C; a.wgt = a.split ? w.ws : w.w; a.dst = out.t.p; a.dst_ld = w.cout;
        a.wgt_frag = a.split ? w.wf : nullptr;
        if (a.wgt == nullptr) { if (!err) err = fail(SPDM_ERR_STATE, "plan: conv weights missing"); return out; }
        a.M = M; a.K = w.cin; a.N = w.cout; a.taps = w.taps; a.geom_M = Bg() * HW;
        a.H = Hl(level); a.W = Wl(level); a.HW = HW;
        a.pro = in.pending_gn() ? (gelu ? PRO_GN_GELU : PRO_GN) : PRO_NONE;
        if (in.pending_gn()) { a.pro_stats = in.st.ref; a.pro_gamma = in.gamma; a.pro_beta = in.beta; }
        a.epi = EPI_STATS; a.epi_stats = out.st.p;
        a.partial = h->d_partial;            // non-null: launch_gemm may split K (small grids)
        if (in.t.C != w.cin) { if (!err) err = fail(SPDM_ERR_INVALID, "plan: conv input has %d channels, weight expects %d", in.t.C, w.cin); return out; }
        // Profiling (spdm_profile_*): HIP events on the launch stream.  Runs of CONSECUTIVE conv launches (the two
        // DoubleConvolutions of a block: nothing else is launched in between) share one event pair -- every event
        // record is a queue barrier that costs the neighbouring kernels their overlap (62 records per step were
        // worth 0.25 ms), and the class average only needs total time / launches.
        const bool solo = h->prof && h->prof_open < 0;
        if (solo) prof_begin();
        if (h->prof && h->prof_open >= 0) {
            h->prof_evts[h->prof_open].flops += gemm_flops(a);
            h->prof_evts[h->prof_open].launches += 1;
        }
        check(launch_gemm(a, s), "conv3x3 implicit GEMM");
        if (solo) prof_end();
        return out;
    }
    void prof_begin() {
        if (!h->prof || h->prof_open >= 0 || dry || err) return;
        ProfEvt e{};
        bool have = false;
        if (!h->prof_pool.empty()) {
            e = h->prof_pool.back();
            h->prof_pool.pop_back();
            e.flops = 0.0; e.launches = 0;
            have = true;
        } else {
            have = hipEventCreate(&e.a) == hipSuccess && hipEventCreate(&e.b) == hipSuccess;
        }
        if (have) {
            h->prof_evts.push_back(e);
            h->prof_open = (int)h->prof_evts.size() - 1;
            (void)hipEventRecord(e.a, s);
        }
    }
    void prof_end() {
        if (h->prof_open < 0) return;
        (void)hipEventRecord(h->prof_evts[h->prof_open].b, s);
        h->prof_open = -1;
    }
    // DoubleConvolution.forward, models/Unet_FiLmLayer.py:108-115.  Consumes `in`.
    Value double_conv(Value& in, const DoubleConvW& w, int level, bool keep_in = false) {
        Value mid = conv(in, w.first, level, /*gelu=*/false, w.gamma, w.beta);
        if (!keep_in) free(in);
        Value out = conv(mid, w.second, level, /*gelu=*/true, w.gamma, w.beta);
        free(mid);
        return out;
    }
    // The FIRST convolution of a Down / UpSample block reading its input THROUGH the resampling op (GemmArgs "fused sources":
    // PRO_POOL: MaxPool2d(2) of src0, the finer level's value; PRO_UPCAT: cat([upsample x2 of src0 (coarser level), skip])) --
    // no pooled / concatenated tensor is written or read back, and one launch less.  Returns false, with nothing allocated or
    // launched, when the kernel launch_gemm would pick for this shape does not take fused sources (the caller then materialises).
    bool conv_fused(int mode, const Value& src0, const Value* skip, const ConvW& w, int level, const float* gamma,
                    const float* beta, Value* out_v) {
        if (err || !h->split || (h->sw & SW_NO_FUSED_SRC) || h->arena.keep || h->d_partial == nullptr) return false;
        const int HW = HWl(level), M = B * HW;
        if (dry) {
            // sizing pass: the fused / materialised choice of a real run depends on its batch (launch geometry), so the dry runs walk
            // every combination (spdm_handle::dry_fuse_mask) and the slab is sized for the worst
            if (!((h->dry_fuse_mask >> fuse_slot) & 1)) return false;
            const int split = (w.cin % 32 == 0) ? 1 : 0;
            const GemmGeom g = gemm_geometry(M, w.cout, w.cin, HW, Wl(level), w.taps, split, h->sw, /*stats_epi=*/true);
            Value out;
            out.t = talloc(w.cout, level);
            out.st = salloc(HW, w.cout, g.st_m_tile, g.st_n_tiles);
            out.gamma = gamma; out.beta = beta;
            *out_v = out;
            return true;
        }
        if (!h->weights_loaded || !w.ws || !w.wf) return false;
        GemmArgs a{};
        a.sw = h->sw;
        a.split = 1;
        a.src = src0.t.p; a.src_ld = src0.t.C; a.wgt = w.ws; a.wgt_frag = w.wf; a.dst_ld = w.cout;
        a.M = M; a.K = w.cin; a.N = w.cout; a.taps = w.taps; a.geom_M = Bg() * HW;
        a.H = Hl(level); a.W = Wl(level); a.HW = HW;
        a.pro = mode;
        if (src0.pending_gn()) { a.pro_stats = src0.st.ref; a.pro_gamma = src0.gamma; a.pro_beta = src0.beta; }
        if (mode == PRO_UPCAT) {
            if (!skip || src0.t.C + skip->t.C != w.cin) return false;
            a.up_C = src0.t.C; a.skip = skip->t.p; a.skip_ld = skip->t.C;
            if (skip->pending_gn()) { a.skip_stats = skip->st.ref; a.skip_gamma = skip->gamma; a.skip_beta = skip->beta; }
        } else if (src0.t.C != w.cin) {
            return false;
        }
        a.epi = EPI_STATS;
        a.partial = h->d_partial;
        if (!gemm_takes_fused_source(a)) return false;
        const GemmGeom g = gemm_geometry(a.geom_M, w.cout, w.cin, HW, Wl(level), w.taps, 1, h->sw, /*stats_epi=*/h->d_partial != nullptr);
        Value out;
        out.t = talloc(w.cout, level);
        out.st = salloc(HW, w.cout, g.st_m_tile, g.st_n_tiles);
        out.gamma = gamma; out.beta = beta;
        if (err) { *out_v = out; return true; }
        a.dst = out.t.p; a.epi_stats = out.st.p;
        const bool solo = h->prof && h->prof_open < 0;
        if (solo) prof_begin();
        if (h->prof && h->prof_open >= 0) {
            h->prof_evts[h->prof_open].flops += gemm_flops(a);
            h->prof_evts[h->prof_open].launches += 1;
        }
        check(launch_gemm(a, s), "conv3x3 implicit GEMM (fused source)");
        if (solo) prof_end();
        *out_v = out;
        return true;
    }
    // The first convolution of an UpSample block with a TWO-SOURCE input (conv_wide.hip TWO): channels [0, C_up) from `up2x`, the
    // upsampled tensor (finished), the rest from the skip connection, whose pending GroupNorm is the load prologue.  torch.cat
    // (models/Unet_FiLmLayer.py:218) is never materialised.  Returns false, nothing done, when the launch would not be a
    // two-source configuration.
    bool conv_two(const Tensor& up2x, const Value& skip, const ConvW& w, int level, const float* gamma, const float* beta, Value* out_v) {
        if (err || !h->split || (h->sw & SW_NO_FUSED_SRC) || h->arena.keep) return false;
        const int HW = HWl(level), M = B * HW;
        if (up2x.C + skip.t.C != w.cin) return false;
        if (dry) {                       // sizing pass (see conv_fused): bit 6 + block of dry_fuse_mask
            if (!((h->dry_fuse_mask >> (3 + fuse_slot)) & 1)) return false;
            const GemmGeom g = gemm_geometry(M, w.cout, w.cin, HW, Wl(level), w.taps, (w.cin % 32 == 0) ? 1 : 0, h->sw, /*stats_epi=*/true);
            Value out;
            out.t = talloc(w.cout, level);
            out.st = salloc(HW, w.cout, g.st_m_tile, g.st_n_tiles);
            out.gamma = gamma; out.beta = beta;
            *out_v = out;
            return true;
        }
        if (!h->weights_loaded || !w.ws || !w.wf) return false;
        GemmArgs a{};
        a.sw = h->sw;
        a.split = 1;
        a.src = up2x.p; a.src_ld = up2x.C; a.wgt = w.ws; a.wgt_frag = w.wf; a.dst_ld = w.cout;
        a.M = M; a.K = w.cin; a.N = w.cout; a.taps = w.taps; a.geom_M = Bg() * HW;
        a.H = Hl(level); a.W = Wl(level); a.HW = HW;
        a.up_C = up2x.C; a.skip = skip.t.p; a.skip_ld = skip.t.C;
        a.pro = skip.pending_gn() ? PRO_GN : PRO_NONE;
        if (skip.pending_gn()) { a.pro_stats = skip.st.ref; a.pro_gamma = skip.gamma; a.pro_beta = skip.beta; }
        a.epi = EPI_STATS;
        a.partial = h->d_partial;
        if (!gemm_takes_two_sources(a)) return false;
        const GemmGeom g = gemm_geometry(a.geom_M, w.cout, w.cin, HW, Wl(level), w.taps, 1, h->sw, /*stats_epi=*/h->d_partial != nullptr);
        Value out;
        out.t = talloc(w.cout, level);
        out.st = salloc(HW, w.cout, g.st_m_tile, g.st_n_tiles);
        out.gamma = gamma; out.beta = beta;
        *out_v = out;
        if (err) return true;
        a.dst = out.t.p; a.epi_stats = out.st.p;
        const bool solo = h->prof && h->prof_open < 0;
        if (solo) prof_begin();
        if (h->prof && h->prof_open >= 0) {
            h->prof_evts[h->prof_open].flops += gemm_flops(a);
            h->prof_evts[h->prof_open].launches += 1;
        }
        check(launch_gemm(a, s), "conv3x3 implicit GEMM (two-source input)");
        if (solo) prof_end();
        return true;
    }
    // would conv_two take this block?  (asked BEFORE the upsample is launched)
    bool conv_two_ok(int C_up, const Value& skip, const ConvW& w, int level) const {
        if (err || !h->split || (h->sw & SW_NO_FUSED_SRC) || h->arena.keep) return false;
        if (dry) return C_up + skip.t.C == w.cin && ((h->dry_fuse_mask >> (3 + fuse_slot)) & 1);
        if (!h->weights_loaded || !w.ws || !w.wf) return false;
        const int HW = HWl(level);
        GemmArgs a{};
        a.sw = h->sw; a.split = 1;
        a.src = skip.t.p; a.src_ld = C_up; a.wgt = w.ws; a.wgt_frag = w.wf; a.dst_ld = w.cout;     // (src: any non-null pointer; not dereferenced)
        a.M = B * HW; a.K = w.cin; a.N = w.cout; a.taps = w.taps; a.H = Hl(level); a.W = Wl(level); a.HW = HW;
        a.geom_M = Bg() * HW;
        a.up_C = C_up; a.skip = skip.t.p; a.skip_ld = skip.t.C;
        a.pro = skip.pending_gn() ? PRO_GN : PRO_NONE;
        a.epi = EPI_STATS; a.partial = h->d_partial;
        return C_up + skip.t.C == w.cin && gemm_takes_two_sources(a);
    }
    // y[rows][N] = x[rows][K] @ W^T + b  (+GELU | +resid)
    // per-token LayerNorm statistics buffer: [rows][n_tiles][2] fp64 (StatsRef with HW = 1: "sample" = row)
    StatsBuf row_stats_alloc(int rows, int C, int n_tiles) {
        StatsBuf sb;
        size_t off = 0;
        if (!h->arena.alloc(sizeof(double) * 2 * (size_t)rows * std::max(n_tiles, C / 64), &off)) {   // finest tiling
            if (!err) err = fail(SPDM_ERR_NOMEM, "workspace exhausted (batch %d)", B);
            return sb;
        }
        sb.off = off; sb.p = (double*)(h->arena.base + off); sb.valid = true;
        sb.ref.p = sb.p; sb.ref.slots = n_tiles; sb.ref.m_tile = 1 << 30; sb.ref.n_tiles = n_tiles; sb.ref.HW = 1;
        sb.ref.inv_count = 1.0 / (double)C;
        return sb;
    }
    void linear(const float* x, int ld, int rows, const LinW& w, float* y, int epi, const float* resid,
                const StatsBuf* ln = nullptr, const float* ln_g = nullptr, const float* ln_b = nullptr,
                double* row_stats_out = nullptr) {
        if (err || dry) return;
        GemmArgs a{};
        a.sw = h->sw;
        a.split = (h->split && w.ws) ? 1 : 0;
        a.src = x; a.src_ld = ld; a.wgt = a.split ? w.ws : w.w; a.dst = y; a.dst_ld = w.out;
        a.M = rows; a.K = w.in; a.N = w.out; a.taps = 1; a.H = 1; a.W = 1; a.HW = 1;
        a.geom_M = (rows / B) * Bg();
        a.pro = PRO_NONE; a.epi = epi; a.bias = w.b; a.resid = resid; a.resid_ld = w.out;
        if (ln) { a.pro = PRO_GN; a.pro_stats = ln->ref; a.pro_gamma = ln_g; a.pro_beta = ln_b; }   // LayerNorm in the load prologue
        a.row_stats = row_stats_out;
        check(launch_gemm(a, s), "linear GEMM");
    }
    // SelfAttention.forward, models/Unet_FiLmLayer.py:71-82.  Consumes x (and its per-token LayerNorm
    // statistics xs, produced by film_apply), returns the block output.  Both LayerNorms run as the load
    // prologue of the GEMM that consumes them: self.ln -> in_proj, ff_self[0] -> ff_self[1].
    bool sa_fused(const AttnW& w, int level) const {
        return h->split && sa_fused_supported(HWl(level), w.C) && !(h->sw & SW_NO_SA_FUSED) && (!h->weights_loaded || w.fw[0]);
    }
    // ab (optional): x is the RAW conv output and the block input is ab-affine of it (film_coef); consumed here.
    // fs (optional, instead of ab): the kernels evaluate those coefficients themselves (FilmSpec, kernels.h)
    Tensor attention(Tensor& x, StatsBuf& xs, const AttnW& w, int level, Tensor* ab = nullptr, const FilmSpec* fs = nullptr) {
        const int L = HWl(level), rows = B * L, C = w.C;
        const float* abp = (ab && ab->valid) ? ab->p : nullptr;
        if (sa_fused(w, level)) {            // whole block in one kernel (sa_fused.hip)
            Tensor out = talloc(C, level);
            if (!err && !dry)
                check(launch_sa_fused64(x.p, out.p, B, L, w.ln_g, w.ln_b, w.ff_ln_g, w.ff_ln_b, w.fw, w.in_proj.b,
                                        w.out_proj.b, w.ff1.b, w.ff2.b, abp, h->sw, s, fs), "fused attention block");
            free(x);
            free(xs);
            if (ab) free(*ab);
            return out;
        }
        Tensor qkv = ralloc(rows, 3 * C);
        // LayerNorm + in_proj + attention core in ONE kernel where the row tile holds whole samples (sa_head_kernel): q, k, v never
        // reach memory.  (The workspace operations stay those of the three-launch path -- qkv is reserved and released unused -- so
        // the slab sized by the dry run fits whichever path a call takes.)
        // Only on large grids: a workgroup of the fused kernel is a 25-30 us chain (LayerNorm, twelve small products, four rounds of
        // q/k/v tiles -> scores -> softmax -> P v with two to four barriers each) at two workgroups per CU, so it needs many rounds
        // of workgroups to beat three launches that each fill the chip.  Measured (traced, same box; q/k/v + core launches -> fused):
        // B = 4096: sa1 281 -> 234 us, sa2 171 -> 164, sa4 75 -> 70, sa3 50 -> 52; B = 512: 35 -> 38, 34 -> 44, 19 -> 32, 17 -> 34;
        // B = 64: 13-17 -> 26-32 us each.  Rule: at least 2048 row tiles (SPDM_SA_HEAD=1 forces the fused kernel at any size: tests; SPDM_TUNE16: the threshold).
        const int TMh = 8192 / C;
        const bool head = h->split && sa_tail_supported(C, h->sw) && sa_head_supported(C, L, h->sw) && h->weights_loaded && w.qkv_wf &&
                          (!fs || sa_tail_film_local(C, L)) && ((Bg() * L + TMh - 1) / TMh) >= ((h->sw & SW_SA_HEAD) ? 1 : spdm_tune(16, 2048));
        if (head) {
            free(xs);
            Tensor att1 = ralloc(rows, C);
            if (!err && !dry)
                check(launch_sa_head(C, x.p, att1.p, rows, w.qkv_wf, w.in_proj.b, w.ln_g, w.ln_b, abp, L, s, fs), "attention in_proj + core");
            free(qkv);
            if (h->split && sa_tail_supported(C, h->sw) && (!h->weights_loaded || w.tail_wf[0])) {
                Tensor out = talloc(C, level);
                if (!err && !dry)
                    check(launch_sa_tail(C, att1.p, x.p, out.p, rows, w.tail_wf[0], w.tail_wf[1], w.tail_wf[2], w.out_proj.b, w.ff1.b,
                                            w.ff2.b, w.ff_ln_g, w.ff_ln_b, abp, L, s, fs), "attention tail");
                free(att1);
                free(x);
                if (ab) free(*ab);
                return out;
            }
            // (no fused tail for this block: the GEMM chain below continues from att1)
            Tensor av = talloc(C, level);
            const int nt_av1 = gemm_geometry((rows / B) * Bg(), C, C, 1, 1, 1, (h->split && w.out_proj.ws) ? 1 : 0, h->sw).n_tiles;
            StatsBuf avs = row_stats_alloc(rows, C, nt_av1);
            linear(att1.p, C, rows, w.out_proj, av.p, EPI_BIAS_RESID, x.p, nullptr, nullptr, nullptr, avs.p);
            free(att1);
            free(x);
            Tensor f1 = ralloc(rows, C);
            linear(av.p, C, rows, w.ff1, f1.p, EPI_BIAS_GELU, nullptr, &avs, w.ff_ln_g, w.ff_ln_b);
            free(avs);
            Tensor out = talloc(C, level);
            linear(f1.p, C, rows, w.ff2, out.p, EPI_BIAS_RESID, av.p);
            free(f1);
            free(av);
            return out;
        }
        if (h->split && sa_tail_supported(C, h->sw) && (!h->weights_loaded || w.qkv_wf)) {     // LayerNorm + in_proj in one 64-row kernel (sa_tail.hip)
            if (!err && !dry)
                check(launch_sa_qkv(C, x.p, qkv.p, rows, w.qkv_wf, w.in_proj.b, w.ln_g, w.ln_b, abp, L, s, fs), "attention in_proj");
        } else {
            linear(x.p, C, rows, w.in_proj, qkv.p, EPI_BIAS, nullptr, &xs, w.ln_g, w.ln_b);
        }
        free(xs);
        Tensor att = ralloc(rows, C);
        if (!err && !dry) check(launch_attention_auto(qkv.p, att.p, B, L, C, 4, h->sw, s), "attention core");
        free(qkv);
        if (h->split && sa_tail_supported(C, h->sw) && (!h->weights_loaded || w.tail_wf[0])) {
            // out_proj + residual + LayerNorm + ff_self + residual in one kernel (sa_tail.hip)
            Tensor out = talloc(C, level);
            if (!err && !dry)
                check(launch_sa_tail(C, att.p, x.p, out.p, rows, w.tail_wf[0], w.tail_wf[1], w.tail_wf[2], w.out_proj.b, w.ff1.b,
                                        w.ff2.b, w.ff_ln_g, w.ff_ln_b, abp, L, s, fs), "attention tail");
            free(att);
            free(x);
            if (ab) free(*ab);
            return out;
        }
        Tensor av = talloc(C, level);
        const int nt_av = gemm_geometry((rows / B) * Bg(), C, C, 1, 1, 1, (h->split && w.out_proj.ws) ? 1 : 0, h->sw).n_tiles;   // n-tiles of the out_proj GEMM
        StatsBuf avs = row_stats_alloc(rows, C, nt_av);
        linear(att.p, C, rows, w.out_proj, av.p, EPI_BIAS_RESID, x.p, nullptr, nullptr, nullptr, avs.p);
        free(att);
        free(x);
        Tensor f1 = ralloc(rows, C);
        linear(av.p, C, rows, w.ff1, f1.p, EPI_BIAS_GELU, nullptr, &avs, w.ff_ln_g, w.ff_ln_b);
        free(avs);
        Tensor out = talloc(C, level);
        linear(f1.p, C, rows, w.ff2, out.p, EPI_BIAS_RESID, av.p);
        free(f1);
        free(av);
        return out;
    }
    // May the FiLM tail feeding this attention block be folded into the block's loads?  Only the kernels that read the
    // block input themselves take the coefficients: sa_fused64 and the C = 128 pair sa_qkv128 / sa_tail128.
    bool film_foldable(const AttnW& w, int level) const {
        if (!h->cfg.attention || h->arena.keep || !h->split || (h->sw & SW_NO_FILM_FOLD)) return false;
        if (sa_fused(w, level)) return HWl(level) <= 256;      // (the two-workgroup mode for longer sequences has no registers left)
        return sa_tail_supported(w.C, h->sw) && (!h->weights_loaded || (w.qkv_wf && w.tail_wf[0]));
    }
    // ... and may those kernels evaluate the coefficients themselves (no film_coef launch)?
    // Only at the smallest batches (every launch a single workgroup): measured on the whole step (same box, alternating,
    // graph replay), evaluating the coefficients inside the consumers saves 5 us per step at batch 1-4 (0.490 -> 0.485 ms) and
    // COSTS 8 us at batch 64, 10-30 us at 512, ~50 us at 4096 -- every workgroup of sa_qkv / sa_tail repeats the statistics
    // round trip and one barrier that the 5-us film_coef launch does once per sample.  SPDM_FILM_LOCAL=1 forces it on (tests).
    bool film_local(const AttnW& w, int level) const {
        if (!film_foldable(w, level) || (h->sw & SW_NO_FILM_LOCAL)) return false;
        if (Bg() > 4 && !(h->sw & SW_FILM_LOCAL)) return false;
        return sa_fused(w, level) || sa_tail_film_local(w.C, HWl(level));
    }
    FilmSpec film_spec(const Value& v, const ResampleW& w, int blk, bool use_cond) const {
        FilmSpec f{};
        if (v.pending_gn()) { f.st = v.st.ref; f.gamma = v.gamma; f.beta = v.beta; }
        f.temb = w.temb_table; f.t_dev = h->d_t; f.t_count = h_tcount;
        f.film = (use_cond && h->cfg.cond_dim > 0) ? h->d_film[blk] : nullptr;
        f.C = w.cout; f.on = 1;
        return f;
    }
    // the FiLM tail as coefficients (film_coef_kernel): returns the RAW conv tensor of v (its statistics are released),
    // *ab receives [B][2 C]
    Tensor film_coef(Value& v, const ResampleW& w, int blk, bool use_cond, Tensor* ab) {
        *ab = ralloc(B, 2 * w.cout);
        if (!err && !dry)
            check(launch_film_coef(asrc(v), w.temb_table, h->d_t, (h_tcount), (use_cond && h->cfg.cond_dim > 0) ? h->d_film[blk] : nullptr,
                                   ab->p, B, s), "film_coef");
        Tensor raw = v.t;
        v.t.valid = false;
        free(v);
        return raw;
    }
    // tail of DownSample/UpSample.forward: + time embedding, FiLM.  Consumes v.
    Tensor film_tail(Value& v, const ResampleW& w, int blk, int level, bool use_cond, StatsBuf* row_stats) {
        Tensor y = talloc(w.cout, level);
        if (row_stats) *row_stats = row_stats_alloc(B * HWl(level), w.cout, 1);
        if (!err && !dry)
            check(launch_film_apply(asrc(v), w.temb_table, h->d_t, (h_tcount), (use_cond && h->cfg.cond_dim > 0) ? h->d_film[blk] : nullptr,
                                    y.p, row_stats ? row_stats->p : nullptr, B, HWl(level), s), "film_apply");
        free(v);
        return y;
    }
    int h_tcount = 1;
    int fuse_slot = 0;     // which resampling op conv_fused is being asked about (dry runs: bit of dry_fuse_mask)
    int adv = -2;          // loop bookkeeping done by conv_in_kernel: -2 none, -1 advance, >= 0 set
};

// one U-Net evaluation on h->d_x... : x (B,H0,D) -> feat (B, Hp*Wp, 64)
static int plan_unet(Ctx& c, const float* x, bool use_cond, Tensor* feat_out) {
    spdm_handle* h = c.h;
    const int B = c.B;
    // ---- inc = DoubleConvolution(1, 64) on the zero-padded trajectory (:286-288) ----
    Value v0;
    v0.t = c.talloc(64, 0);
    v0.st = c.salloc(c.HWl(0), 64, c.HWl(0) / conv_in_parts(h->Hp, h->Wp, c.Bg()), 1);
    v0.gamma = h->inc.gamma; v0.beta = h->inc.beta;
    if (!c.err && !c.dry)
        c.check(launch_conv_in(x, h->w_inc_first, v0.t.p, v0.st.p, B, h->cfg.horizon, h->cfg.state_dim, h->Hp, h->Wp,
                               h->lh, h->lw, h->d_step, h->d_t, h->d_timesteps, h->n_steps, c.adv, c.s, c.Bg()), "conv_in");
    Value x1 = c.conv(v0, h->inc.second, 0, /*gelu=*/true, h->inc.gamma, h->inc.beta);   // x1 = GN(raw), pending
    c.free(v0);
    if (h->arena.keep) {
        Tensor m = c.talloc(64, 0);
        if (!c.dry && !c.err) c.check(launch_gn_apply(c.asrc(x1), m.p, B, c.HWl(0), c.s), "gn_apply");
        c.tap("x1", m);
    }

    // ---- encoder: down1..3 (+ sa1..3) ----
    Value skips[3];            // x1 (pending GN), x2, x3 (materialised)
    skips[0] = x1;
    Value cur = x1;
    static const char* dn[3] = {"d1", "d2", "d3"};
    static const char* xn[3] = {"x2", "x3", "x4"};
    for (int i = 0; i < 3; ++i) {
        const int lin = i, lout = i + 1;
        // `cur` stays alive: it is a skip connection
        Value a, mid;
        c.fuse_slot = i;
        if (c.conv_fused(PRO_POOL, cur, nullptr, h->down[i].dc1.first, lout, h->down[i].dc1.gamma, h->down[i].dc1.beta, &mid)) {
            c.prof_begin();      // MaxPool2d(2) read through by the block's first conv (small grids)
            a = c.conv(mid, h->down[i].dc1.second, lout, /*gelu=*/true, h->down[i].dc1.gamma, h->down[i].dc1.beta);
            c.free(mid);
        } else {
            Value p;
            p.t = c.talloc(cur.t.C, lout);
            if (!c.err && !c.dry)
                c.check(launch_pool(c.asrc(cur), p.t.p, B, c.Hl(lin), c.Wl(lin), c.s), "maxpool");
            c.prof_begin();
            a = c.double_conv(p, h->down[i].dc1, lout);
        }
        Value b2 = c.double_conv(a, h->down[i].dc2, lout);
        c.prof_end();
        StatsBuf ys;
        Tensor y, ab;
        if (c.film_local(h->sa[i], lout)) {           // the attention kernels finish the block tail themselves, from the raw tensor
            const FilmSpec fs = c.film_spec(b2, h->down[i], i, use_cond);
            y = b2.t;
            b2.t.valid = false;
            y = c.attention(y, ys, h->sa[i], lout, nullptr, &fs);
            c.free(b2);                               // (its statistics: released after the last launch that reads them is enqueued)
        } else {
        if (c.film_foldable(h->sa[i], lout)) {
            y = c.film_coef(b2, h->down[i], i, use_cond, &ab);
        } else {
            y = c.film_tail(b2, h->down[i], i, lout, use_cond,
                            (h->cfg.attention && !c.sa_fused(h->sa[i], lout)) ? &ys : nullptr);
            c.tap(dn[i], y);
        }
        if (h->cfg.attention) y = c.attention(y, ys, h->sa[i], lout, &ab);
        }
        c.tap(xn[i], y);
        Value nv;
        nv.t = y;
        cur = nv;
        if (i < 2) skips[i + 1] = nv;
    }
    // ---- bottleneck (:297-299) ----
    c.prof_begin();
    Value b1 = c.double_conv(cur, h->bot[0], 3);
    Value b2 = c.double_conv(b1, h->bot[1], 3);
    Value x5 = c.double_conv(b2, h->bot[2], 3);         // pending GN
    c.prof_end();
    if (h->arena.keep) {
        Tensor m = c.talloc(256, 3);
        if (!c.dry && !c.err) c.check(launch_gn_apply(c.asrc(x5), m.p, B, c.HWl(3), c.s), "gn_apply");
        c.tap("x5", m);
    }
    // ---- decoder: up1..3 (+ sa4..6) ----
    static const char* un[3] = {"u1", "u2", "u3"};
    static const char* an[3] = {"a4", "a5", "a6"};
    cur = x5;
    for (int i = 0; i < 3; ++i) {
        const int lin = 3 - i, lout = 2 - i;
        Value& skip = skips[2 - i];
        Value a, mid;
        c.fuse_slot = 3 + i;
        if (c.conv_fused(PRO_UPCAT, cur, &skip, h->up[i].dc1.first, lout, h->up[i].dc1.gamma, h->up[i].dc1.beta, &mid)) {
            c.free(cur);         // upsample + concat read through by the block's first conv: released once that launch is enqueued
            c.free(skip);
            c.prof_begin();
            a = c.conv(mid, h->up[i].dc1.second, lout, /*gelu=*/true, h->up[i].dc1.gamma, h->up[i].dc1.beta);
            c.free(mid);
        } else if (c.conv_two_ok(cur.t.C, skip, h->up[i].dc1.first, lout)) {
            // upsample only; the first conv reads [upsampled | skip] from the two tensors (no copy of the skip half)
            Tensor u2 = c.talloc(cur.t.C, lout);
            AffineSrc none{};
            if (!c.err && !c.dry) c.check(launch_upcat(c.asrc(cur), none, u2.p, B, c.Hl(lin), c.Wl(lin), c.s), "upsample");
            c.free(cur);
            c.prof_begin();
            if (!c.conv_two(u2, skip, h->up[i].dc1.first, lout, h->up[i].dc1.gamma, h->up[i].dc1.beta, &mid) && !c.err)
                c.err = fail(SPDM_ERR_STATE, "plan: two-source convolution refused after it was offered");
            c.free(u2);
            c.free(skip);
            a = c.conv(mid, h->up[i].dc1.second, lout, /*gelu=*/true, h->up[i].dc1.gamma, h->up[i].dc1.beta);
            c.free(mid);
        } else {
            Value cat;
            cat.t = c.talloc(cur.t.C + skip.t.C, lout);
            if (!c.err && !c.dry)
                c.check(launch_upcat(c.asrc(cur), c.asrc(skip), cat.t.p, B, c.Hl(lin), c.Wl(lin), c.s), "upsample+concat");
            c.free(cur);
            c.free(skip);
            c.prof_begin();
            a = c.double_conv(cat, h->up[i].dc1, lout);
        }
        Value b3 = c.double_conv(a, h->up[i].dc2, lout);
        c.prof_end();
        StatsBuf ys;
        Tensor y, ab;
        if (c.film_local(h->sa[3 + i], lout)) {
            const FilmSpec fs = c.film_spec(b3, h->up[i], 3 + i, use_cond);
            y = b3.t;
            b3.t.valid = false;
            y = c.attention(y, ys, h->sa[3 + i], lout, nullptr, &fs);
            c.free(b3);
        } else {
        if (c.film_foldable(h->sa[3 + i], lout)) {
            y = c.film_coef(b3, h->up[i], 3 + i, use_cond, &ab);
        } else {
            y = c.film_tail(b3, h->up[i], 3 + i, lout, use_cond,
                            (h->cfg.attention && !c.sa_fused(h->sa[3 + i], lout)) ? &ys : nullptr);
            c.tap(un[i], y);
        }
        if (h->cfg.attention) y = c.attention(y, ys, h->sa[3 + i], lout, &ab);
        }
        c.tap(an[i], y);
        Value nv;
        nv.t = y;
        cur = nv;
    }
    *feat_out = cur.t;
    return c.err;
}

static int plan_forward(spdm_handle* h, int B, bool use_cond, hipStream_t s, Tensor* feat_out) {
    Ctx c{h, B, s, h->arena.dry};
    h->arena.reset();
    return plan_unet(c, h->d_x, use_cond, feat_out);
}

// time-embedding tables: Linear(SiLU(pos_encoding(t))) for every t (models/Unet_FiLmLayer.py:136-142)
static int ensure_temb(spdm_handle* h, hipStream_t s) {
    if (h->temb_ready) return SPDM_OK;
    const int T = h->cfg.num_train_timesteps, dim = h->cfg.time_dim;
    float* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, sizeof(float) * (size_t)T * dim));
    hipError_t e = hipMemcpyAsync(tmp, h->time_table.data(), sizeof(float) * (size_t)T * dim, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_silu(tmp, h->d_time_silu, (size_t)T * dim, s);
    ResampleW* blocks[6] = {&h->down[0], &h->down[1], &h->down[2], &h->up[0], &h->up[1], &h->up[2]};
    for (int i = 0; i < 6 && e == hipSuccess; ++i) {
        GemmArgs a{};
        a.sw = h->sw;
        a.split = (h->split && blocks[i]->emb.ws) ? 1 : 0;
        a.src = h->d_time_silu; a.src_ld = dim; a.wgt = a.split ? blocks[i]->emb.ws : blocks[i]->emb.w; a.dst = blocks[i]->temb_table; a.dst_ld = blocks[i]->cout;
        a.M = T; a.K = dim; a.N = blocks[i]->cout; a.taps = 1; a.H = 1; a.W = 1; a.HW = 1;
        a.pro = PRO_NONE; a.epi = EPI_BIAS; a.bias = blocks[i]->emb.b;
        e = launch_gemm(a, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(SPDM_ERR_HIP, "time-embedding tables: %s", hipGetErrorString(e));
    h->temb_ready = true;
    return SPDM_OK;
}

// FiLM projections [scale | bias] = Linear(Mish(flatten(cond))) for the six resample blocks
// (models/Unet_FiLmLayer.py:149-154,171-175).  Step-invariant: hoisted out of the denoise loop.
static int compute_film(spdm_handle* h, int B, const float* d_cond, hipStream_t s) {
    h->have_film = false;
    if (!d_cond || h->cfg.cond_dim <= 0) return SPDM_OK;
    hipError_t e = launch_mish_pad(d_cond, h->d_condm, B, h->cfg.cond_dim, h->film_kp, s);
    ResampleW* blocks[6] = {&h->down[0], &h->down[1], &h->down[2], &h->up[0], &h->up[1], &h->up[2]};
    for (int i = 0; i < 6 && e == hipSuccess; ++i) {
        GemmArgs a{};
        a.sw = h->sw;
        a.split = (h->split && blocks[i]->film.ws) ? 1 : 0;
        a.src = h->d_condm; a.src_ld = h->film_kp; a.wgt = a.split ? blocks[i]->film.ws : blocks[i]->film.w; a.dst = h->d_film[i]; a.dst_ld = 2 * blocks[i]->cout;
        a.M = B; a.K = h->film_kp; a.N = 2 * blocks[i]->cout; a.taps = 1; a.H = 1; a.W = 1; a.HW = 1;
        a.geom_M = (h->sw & SW_PIN_GEOMETRY) ? h->cfg.max_batch : 0;
        a.pro = PRO_NONE; a.epi = EPI_BIAS; a.bias = blocks[i]->film.b;
        e = launch_gemm(a, s);
    }
    if (e != hipSuccess) return fail(SPDM_ERR_HIP, "FiLM projections: %s", hipGetErrorString(e));
    h->have_film = true;
    return SPDM_OK;
}

static int check_ready(spdm_handle* h, int B) {
    if (!h) return fail(SPDM_ERR_INVALID, "null handle");
    if (!h->weights_loaded) return fail(SPDM_ERR_STATE, "spdm_load_weights has not been called");
    if (B < 1 || B > h->cfg.max_batch) return fail(SPDM_ERR_INVALID, "batch %d outside [1, max_batch = %d]", B, h->cfg.max_batch);
    return SPDM_OK;
}

static StepArgs step_args(spdm_handle* h, int B, const Tensor& feat) {
    StepArgs a{};
    a.feat = feat.p; a.w = h->outc_w; a.bias = h->outc_b; a.x = h->d_x; a.eps_out = nullptr;
    a.coef = h->d_coef; a.step_dev = h->d_step; a.kind = h->sched_kind;
    a.noise = h->s_noise; a.rng_dev = h->d_rng; a.flag_dev = h->d_step + 2;
    a.inpaint = h->s_inpaint; a.inp_h = h->s_inp_h; a.inpaint_per_sample = h->s_inp_per_sample;
    a.history = h->s_history;
    a.ptrs_dev = h->d_ptrs;
    a.B = B; a.H0 = h->cfg.horizon; a.D = h->cfg.state_dim; a.Hp = h->Hp; a.Wp = h->Wp; a.lh = h->lh; a.lw = h->lw;
    return a;
}

extern "C" int spdm_unet_forward(spdm_handle* h, int32_t B, const float* d_x, const int32_t* h_t, int32_t t_count,
                                 const float* d_cond, float* d_eps, void* stream) {
    SPDM_TRY(check_ready(h, B));
    if (!d_x || !h_t || !d_eps) return fail(SPDM_ERR_INVALID, "null argument");
    if (t_count != 1 && t_count != B) return fail(SPDM_ERR_INVALID, "t_count must be 1 or B");
    for (int i = 0; i < t_count; ++i)
        if (h_t[i] < 0 || h_t[i] >= h->cfg.num_train_timesteps) return fail(SPDM_ERR_INVALID, "t = %d outside [0,%d)", h_t[i], h->cfg.num_train_timesteps);
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    SPDM_TRY(ensure_temb(h, s));
    HIP_TRY(hipMemcpyAsync(h->d_t, h_t, sizeof(int) * t_count, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(h->d_step + 2, 0, sizeof(int), s));
    HIP_TRY(hipMemcpyAsync(h->d_x, d_x, sizeof(float) * (size_t)B * h->cfg.horizon * h->cfg.state_dim, hipMemcpyDeviceToDevice, s));
    SPDM_TRY(compute_film(h, B, d_cond, s));
    h->taps.clear();
    Ctx c{h, B, s, false};
    c.h_tcount = t_count;
    h->arena.reset();
    Tensor feat;
    SPDM_TRY(plan_unet(c, h->d_x, d_cond != nullptr, &feat));
    StepArgs a = step_args(h, B, feat);
    a.eps_out = d_eps;
    a.ptrs_dev = nullptr;
    HIP_TRY(launch_out_step(a, s));
    h->session = false;
    if (!stream) HIP_TRY(hipStreamSynchronize(s));
    return SPDM_OK;
}

extern "C" int spdm_sample_begin(spdm_handle* h, int32_t B, const float* d_cond, const float* d_inpaint, int32_t inp_h,
                                 int32_t inpaint_per_sample, const float* d_xT, const float* d_noise, uint64_t seed,
                                 uint64_t sample_offset, float* d_history, void* stream) {
    SPDM_TRY(check_ready(h, B));
    if (h->sched_kind < 0) return fail(SPDM_ERR_STATE, "no schedule set (spdm_set_schedule)");
    if (!d_xT) return fail(SPDM_ERR_INVALID, "d_xT is null");
    if (inp_h < 0 || inp_h > h->cfg.horizon) return fail(SPDM_ERR_INVALID, "inpaint horizon %d outside [0,%d]", inp_h, h->cfg.horizon);
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    SPDM_TRY(ensure_temb(h, s));
    const size_t nx = (size_t)B * h->cfg.horizon * h->cfg.state_dim;
    HIP_TRY(hipMemcpyAsync(h->d_x, d_xT, sizeof(float) * nx, hipMemcpyDeviceToDevice, s));
    if (d_history) HIP_TRY(hipMemcpyAsync(d_history, d_xT, sizeof(float) * nx, hipMemcpyDeviceToDevice, s));
    SPDM_TRY(compute_film(h, B, d_cond, s));
    h->sB = B;
    h->s_inpaint = (inp_h > 0) ? d_inpaint : nullptr;
    h->s_inp_h = (d_inpaint != nullptr) ? inp_h : 0;
    h->s_inp_per_sample = inpaint_per_sample;
    h->s_noise = d_noise;
    h->s_history = d_history;
    h->s_seed = seed;
    h->s_offset = sample_offset;
    {   // the noise stream's key lives on the device (read by out_step_kernel), so a new seed does not change the step's launches
        const unsigned long long rng[2] = {seed, sample_offset};
        HIP_TRY(hipMemcpyAsync(h->d_rng, rng, sizeof(rng), hipMemcpyHostToDevice, s));
        const void* ptrs[4] = {h->s_inpaint, h->s_noise, h->s_history, nullptr};
        HIP_TRY(hipMemcpyAsync(h->d_ptrs, ptrs, sizeof(ptrs), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(h->d_step + 2, 0, sizeof(int), s));
    }
    h->session = true;
    if (!stream) HIP_TRY(hipStreamSynchronize(s));
    return SPDM_OK;
}

// one denoise iteration on stream s: loop bookkeeping (explicit index i >= 0, or "advance by one" for i < 0 --
// the form a captured graph replays), U-Net, fused 1x1 conv + scheduler update + inpainting
static int enqueue_step(spdm_handle* h, int i, hipStream_t s) {
    Ctx c{h, h->sB, s, false};
    c.h_tcount = 1;
    c.adv = (i >= 0) ? i : -1;          // the step's first kernel (conv_in_kernel) does the bookkeeping
    h->arena.reset();
    Tensor feat;
    SPDM_TRY(plan_unet(c, h->d_x, h->have_film, &feat));
    StepArgs a = step_args(h, h->sB, feat);
    HIP_TRY(launch_out_step(a, s));
    return SPDM_OK;
}

// Capture one step into a hipGraph (the step's ~70 launches take the same arguments in every iteration: the loop
// counter, the timestep and the noise offset live on the device).  Returns false -- with the stream usable and no
// error state left behind -- when the runtime refuses; the caller then stays on plain launches.
static bool build_step_graph(spdm_handle* h, hipStream_t s) {
    if (h->step_exec) { (void)hipDeviceSynchronize(); (void)hipGraphExecDestroy(h->step_exec); h->step_exec = nullptr; }
    if (h->step_graph) { (void)hipGraphDestroy(h->step_graph); h->step_graph = nullptr; }
    if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return false; }
    const int rc = enqueue_step(h, -1, s);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (rc != SPDM_OK || e != hipSuccess || g == nullptr) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return false;
    }
    hipGraphExec_t ge = nullptr;
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess || ge == nullptr) {
        (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        return false;
    }
    h->step_graph = g;
    h->step_exec = ge;
    ++h->graph_captures;
    return true;
}

extern "C" int spdm_sample_run(spdm_handle* h, int32_t step_begin, int32_t step_end, void* stream) {
    if (!h || !h->session) return fail(SPDM_ERR_STATE, "spdm_sample_begin has not been called");
    if (step_begin < 0 || step_end > h->n_steps || step_begin > step_end)
        return fail(SPDM_ERR_INVALID, "step range [%d,%d) outside [0,%d]", step_begin, step_end, h->n_steps);
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    // Graph replay (default; SPDM_NO_GRAPH=1 disables): not while the per-launch profiler or the debug taps are on.
    const bool graphs_on = !(h->sw & SW_NO_GRAPH);
    bool use_graph = graphs_on && !h->prof && !h->arena.keep && step_end - step_begin >= 3;
    if (use_graph && s == nullptr) {
        // the legacy NULL stream cannot be captured: run on a blocking stream of our own (implicitly ordered with
        // NULL-stream work on both sides) and keep the contract "NULL stream => complete on return"
        if (!h->gstream && hipStreamCreate(&h->gstream) != hipSuccess) { (void)hipGetLastError(); h->gstream = nullptr; use_graph = false; }
        if (use_graph) s = h->gstream;
    }
    int i = step_begin;
    if (use_graph) {
        SPDM_TRY(enqueue_step(h, i, s));         // first step: explicit index; also makes sure every kernel has been launched once
        ++i;
        spdm_handle::StepGraphKey key;
        key.B = h->sB; key.inp_h = h->s_inp_h; key.per_sample = h->s_inp_per_sample; key.have_film = h->have_film ? 1 : 0;
        key.sched_kind = h->sched_kind; key.n_steps = h->n_steps;
        key.env = h->sw;          // the captured launches depend on the kernel-selection switches (spdm_set_switch)
        if (!(h->step_exec && key == h->graph_key)) {
            if (build_step_graph(h, s)) h->graph_key = key;
            else use_graph = false;
        }
        if (use_graph)
            for (; i < step_end; ++i) HIP_TRY(hipGraphLaunch(h->step_exec, s));
    }
    for (; i < step_end; ++i) SPDM_TRY(enqueue_step(h, i, s));
    if (!stream) HIP_TRY(hipStreamSynchronize(s));
    return SPDM_OK;
}

extern "C" int spdm_sample_result(spdm_handle* h, float* d_out, void* stream) {
    if (!h || !h->session) return fail(SPDM_ERR_STATE, "spdm_sample_begin has not been called");
    if (!d_out) return fail(SPDM_ERR_INVALID, "d_out is null");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(d_out, h->d_x, sizeof(float) * (size_t)h->sB * h->cfg.horizon * h->cfg.state_dim,
                           hipMemcpyDeviceToDevice, s));
    if (!stream) HIP_TRY(hipStreamSynchronize(s));
    return SPDM_OK;
}

extern "C" int64_t spdm_graph_captures(const spdm_handle* h) { return h ? h->graph_captures : 0; }

extern "C" int spdm_nonfinite(spdm_handle* h, int32_t* flag_out, void* stream) {
    if (!h || !flag_out) return fail(SPDM_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    int v = 0;
    HIP_TRY(hipMemcpyAsync(&v, h->d_step + 2, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *flag_out = v;
    return SPDM_OK;
}

extern "C" int spdm_sample(spdm_handle* h, int32_t B, const float* d_cond, const float* d_inpaint, int32_t inp_h,
                           int32_t inpaint_per_sample, const float* d_xT, const float* d_noise, uint64_t seed,
                           uint64_t sample_offset, float* d_out, float* d_history, void* stream) {
    // the three pieces run asynchronously on the caller's stream (or on a private one for NULL)
    hipStream_t s = (hipStream_t)stream;
    hipStream_t own = nullptr;
    if (!s) {
        if (h) (void)hipSetDevice(h->cfg.device);
        HIP_TRY(hipStreamCreate(&own));
        s = own;
    }
    int rc = spdm_sample_begin(h, B, d_cond, d_inpaint, inp_h, inpaint_per_sample, d_xT, d_noise, seed, sample_offset, d_history, s);
    if (rc == SPDM_OK) rc = spdm_sample_run(h, 0, h->n_steps, s);
    if (rc == SPDM_OK) rc = spdm_sample_result(h, d_out, s);
    if (own) {
        hipError_t e = hipStreamSynchronize(own);
        (void)hipStreamDestroy(own);
        if (rc == SPDM_OK && e != hipSuccess) rc = fail(SPDM_ERR_HIP, "stream sync: %s", hipGetErrorString(e));
    }
    return rc;
}

extern "C" int spdm_debug_tensor(spdm_handle* h, const char* name, float* d_out, size_t cap, int32_t shape[4]) {
    if (!h || !name || !shape) return fail(SPDM_ERR_INVALID, "null argument");
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return fail(SPDM_ERR_MISSING, "no intermediate named '%s' (handle needs SPDM_FLAG_DEBUG_KEEP and a prior spdm_unet_forward)", name);
    const Tensor& t = it->second;
    const int l = t.level;
    shape[0] = h->tapB; shape[1] = h->Hp >> l; shape[2] = h->Wp >> l; shape[3] = t.C;
    const size_t n = (size_t)shape[0] * shape[1] * shape[2] * shape[3];
    if (d_out) {
        if (cap < n) return fail(SPDM_ERR_INVALID, "buffer too small: need %zu floats", n);
        HIP_TRY(hipMemcpy(d_out, t.p, n * sizeof(float), hipMemcpyDeviceToDevice));
    }
    return SPDM_OK;
}

extern "C" int spdm_profile_enable(spdm_handle* h, int32_t on) {
    if (!h) return fail(SPDM_ERR_INVALID, "null handle");
    // on = 2: prepare only -- create the events a few instrumented steps need, instrument nothing yet (bench.py does this
    // before its timed region and flips to 1 inside it)
    h->prof = on == 1;
    for (auto& e : h->prof_evts) h->prof_pool.push_back(e);
    h->prof_evts.clear();
    h->prof_open = -1;
    h->prof_launches = 0; h->prof_ms = 0.0; h->prof_flops = 0.0;
    if (on == 2) {
        HIP_TRY(hipSetDevice(h->cfg.device));
        while (h->prof_pool.size() < 256) {
            ProfEvt e{};
            if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) break;
            h->prof_pool.push_back(e);
        }
    }
    return SPDM_OK;
}

extern "C" int spdm_profile_read(spdm_handle* h, int64_t* launches, double* total_ms, double* total_flops) {
    if (!h) return fail(SPDM_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    for (auto& e : h->prof_evts) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            h->prof_ms += ms; h->prof_flops += e.flops; h->prof_launches += e.launches;
        }
        h->prof_pool.push_back(e);
    }
    h->prof_evts.clear();
    if (launches) *launches = h->prof_launches;
    if (total_ms) *total_ms = h->prof_ms;
    if (total_flops) *total_flops = h->prof_flops;
    return SPDM_OK;
}

// -------------------------------------------------------------------------------------------------
// Micro-benchmark of one implicit-GEMM launch shape on synthetic data (tools/bench_gemm.py); not on
// the product path.  Returns the average device time per launch in *ms_out (HIP events).
extern "C" int spdm_bench_gemm(int32_t device, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t taps,
                               int32_t pro, int32_t epi, int32_t split, int32_t iters, int32_t debug, double* ms_out) {
    if (!ms_out || iters < 1) return fail(SPDM_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const int HW = H * W, M = B * HW;
    const unsigned sw = switches_from_env();
    const bool may_splitk = split && epi == EPI_STATS && !(sw & SW_NO_SPLITK);
    const GemmGeom g = gemm_geometry(M, Cout, Cin, HW, W, taps, split, sw, may_splitk);
    float *src = nullptr, *wgt = nullptr, *wgt32 = nullptr, *dst = nullptr, *dst2 = nullptr, *gb = nullptr, *resid = nullptr, *wfrag = nullptr;
    double *st_in = nullptr, *st_out = nullptr, *st_out2 = nullptr;
    const size_t nsrc = (size_t)M * Cin, nw = (size_t)taps * Cout * Cin, ndst = (size_t)M * Cout;
    HIP_TRY(hipMalloc((void**)&src, nsrc * 4));
    HIP_TRY(hipMalloc((void**)&wgt, nw * 4));
    HIP_TRY(hipMalloc((void**)&wgt32, nw * 4));
    HIP_TRY(hipMalloc((void**)&dst2, ndst * 4));
    HIP_TRY(hipMalloc((void**)&dst, ndst * 4));
    HIP_TRY(hipMalloc((void**)&resid, ndst * 4));
    HIP_TRY(hipMalloc((void**)&gb, (size_t)(Cin + Cout) * 2 * 4));
    const bool row_ln = (taps == 1 && pro != 0);           // Linear with a LayerNorm prologue: statistics per ROW
    HIP_TRY(hipMalloc((void**)&st_in, (size_t)(row_ln ? M : B) * 2 * 8));
    HIP_TRY(hipMalloc((void**)&st_out, (size_t)B * g.slots * 2 * 8));
    HIP_TRY(hipMemset(st_out, 0, (size_t)B * g.slots * 2 * 8));
    const GemmGeom g2 = gemm_geometry(M, Cout, Cin, HW, W, taps, 0, sw);
    HIP_TRY(hipMalloc((void**)&st_out2, (size_t)B * g2.slots * 2 * 8));
    HIP_TRY(hipMemset(st_out2, 0, (size_t)B * g2.slots * 2 * 8));
    {   // deterministic pseudo-random fill (values ~U(-1,1)); split weights are packed as at load time
        std::vector<float> hsrc(nsrc), hw(nw), hgb((size_t)(Cin + Cout) * 2);
        unsigned x = 12345u;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
        for (auto& v : hsrc) v = rnd();
        for (auto& v : hw) v = rnd() * 0.05f;
        for (auto& v : hgb) v = 1.0f + 0.1f * rnd();
        HIP_TRY(hipMemcpy(wgt32, hw.data(), nw * 4, hipMemcpyHostToDevice));
        if (split) {
            std::vector<float> out(nw);
            for (size_t base = 0; base < nw; base += 32) {
                _Float16* hp = reinterpret_cast<_Float16*>(&out[base]);
                for (int j = 0; j < 32; ++j) {
                    const float v = hw[base + j] * 128.0f;
                    const _Float16 hi = (_Float16)v;
                    hp[j] = hi;
                    hp[32 + j] = (_Float16)(v - (float)hi);
                }
            }
            hw.swap(out);
            if ((taps == 9 || taps == 3) && Cout % 64 == 0 && Cin % 32 == 0) {
                const std::vector<float> fr = frag_order_weights(hw, taps, Cout, Cin);
                HIP_TRY(hipMalloc((void**)&wfrag, nw * 4));
                HIP_TRY(hipMemcpy(wfrag, fr.data(), nw * 4, hipMemcpyHostToDevice));
            }
        }
        std::vector<double> hst((size_t)(row_ln ? M : B) * 2);
        for (size_t b = 0; b < hst.size() / 2; ++b) { hst[2 * b] = 0.0; hst[2 * b + 1] = (double)Cin * (row_ln ? 1 : HW) / 3.0; }
        HIP_TRY(hipMemcpy(src, hsrc.data(), nsrc * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(wgt, hw.data(), nw * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(gb, hgb.data(), hgb.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(st_in, hst.data(), hst.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(resid, 0, ndst * 4));
    }
    GemmArgs a{};
    a.sw = sw;
    a.src = src; a.src_ld = Cin; a.wgt = wgt; a.wgt_frag = wfrag; a.split = split; a.dst = dst; a.dst_ld = Cout;
    a.M = M; a.K = Cin; a.N = Cout; a.taps = taps; a.H = H; a.W = W; a.HW = HW;
    if (row_ln) { a.H = 1; a.W = 1; a.HW = 1; }             // like Ctx::linear: every row is its own LayerNorm "sample"
    a.pro = pro;
    a.pro_stats.p = st_in; a.pro_stats.slots = 1; a.pro_stats.m_tile = row_ln ? (1 << 30) : HW; a.pro_stats.n_tiles = 1;
    a.pro_stats.HW = row_ln ? 1 : HW;
    a.pro_stats.inv_count = 1.0 / ((double)Cin * (row_ln ? 1 : HW));
    a.pro_gamma = gb; a.pro_beta = gb + Cin;
    a.epi = epi; a.epi_stats = st_out; a.bias = gb + 2 * Cin; a.resid = resid; a.resid_ld = Cout;
    a.debug = debug;
    float* d_part = nullptr;
    if (may_splitk) {
        HIP_TRY(hipMalloc((void**)&d_part, SPLITK_WORKSPACE_BYTES));
        a.partial = d_part;
    }
    unsigned long long* d_stamps = nullptr;
    if (debug & DBG_STAMP) {
        HIP_TRY(hipMalloc((void**)&d_stamps, (size_t)65536 * 8 * 8));      // conv_wide: 8 stamps per workgroup
        HIP_TRY(hipMemset(d_stamps, 0, (size_t)65536 * 8 * 8));
        a.stamps = d_stamps;
    }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipError_t e = launch_gemm(a, nullptr);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = launch_gemm(a, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters && e == hipSuccess; ++i) e = launch_gemm(a, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    double maxdiff = -1.0, statdiff = -1.0;
    if (e == hipSuccess && debug == 0) {      // self-check: same data through the exact fp32-MFMA configuration
        GemmArgs b2 = a;
        b2.split = 0; b2.wgt = wgt32; b2.wgt_frag = nullptr; b2.dst = dst2; b2.epi_stats = st_out2;
        e = launch_gemm(b2, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        std::vector<float> h1(ndst), h2(ndst);
        if (e == hipSuccess) e = hipMemcpy(h1.data(), dst, ndst * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(h2.data(), dst2, ndst * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) {
            maxdiff = 0.0;
            for (size_t i = 0; i < ndst; ++i) maxdiff = std::max(maxdiff, (double)std::fabs(h1[i] - h2[i]));
        }
        if (e == hipSuccess && epi == EPI_STATS) {
            // GroupNorm partials of the launch under test against totals recomputed (fp64, host) from ITS OWN output:
            // checks the epilogue's slot bookkeeping and reductions independently of kernel-vs-kernel rounding
            std::vector<double> s1((size_t)B * g.slots * 2);
            e = hipMemcpy(s1.data(), st_out, s1.size() * 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess) {
                statdiff = 0.0;
                const double n = (double)HW * Cout;
                for (int b = 0; b < B; ++b) {
                    double t1 = 0.0, t2 = 0.0, r1 = 0.0, r2 = 0.0;
                    for (int k = 0; k < g.slots; ++k) { t1 += s1[((size_t)b * g.slots + k) * 2]; t2 += s1[((size_t)b * g.slots + k) * 2 + 1]; }
                    const float* p = h1.data() + (size_t)b * HW * Cout;
                    for (size_t i = 0; i < (size_t)HW * Cout; ++i) { r1 += p[i]; r2 += (double)p[i] * p[i]; }
                    const double m1 = t1 / n, mr = r1 / n, v1 = t2 / n - m1 * m1, vr = r2 / n - mr * mr;
                    statdiff = std::max(statdiff, std::fabs(m1 - mr) / std::sqrt(std::max(vr, 1e-30)));
                    statdiff = std::max(statdiff, std::fabs(v1 - vr) / std::max(vr, 1e-30));
                }
            }
        }
    }
    ms_out[1] = maxdiff;
    ms_out[2] = statdiff;
    if (d_stamps && getenv("SPDM_STAMP_DUMP")) {      // raw per-workgroup timeline of conv_wide (analysed offline)
        std::vector<unsigned long long> hs((size_t)65536 * 8);
        if (hipMemcpy(hs.data(), d_stamps, hs.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE* f = fopen(getenv("SPDM_STAMP_DUMP"), "wb")) { fwrite(hs.data(), 8, hs.size(), f); fclose(f); }
        }
        (void)hipFree(d_stamps);
    } else if (d_stamps) {      // print the stamp deltas of the traced workgroup (cycles between consecutive stamps)
        unsigned long long hs[256];
        if (hipMemcpy(hs, d_stamps, sizeof(hs), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int g = 0; g < 2; ++g) {
                const int n = (int)hs[g * 128 + 127];
                printf("stamps half %d (%d):", g, n);
                for (int i = 1; i < n && i < 126; ++i) printf(" %llu", hs[g * 128 + i] - hs[g * 128 + i - 1]);
                printf("  total %llu\n", n > 1 ? hs[g * 128 + n - 1] - hs[g * 128] : 0ull);
            }
            fflush(stdout);
        }
        (void)hipFree(d_stamps);
    }
    (void)hipFree(wgt32); (void)hipFree(dst2); (void)hipFree(wfrag); (void)hipFree(d_part);
    (void)hipFree(src); (void)hipFree(wgt); (void)hipFree(dst); (void)hipFree(resid); (void)hipFree(gb); (void)hipFree(st_in); (void)hipFree(st_out); (void)hipFree(st_out2);
    if (e != hipSuccess) return fail(SPDM_ERR_HIP, "bench_gemm: %s", hipGetErrorString(e));
    *ms_out = ms / iters;
    return SPDM_OK;
}

// -------------------------------------------------------------------------------------------------
// Observation front end: the autoencoder's encoder, models/encoder/autoencoder.py:11-20, applied by
// prepare_obs_cond_vectors (models/diffusion_ddpm.py:317-321) to every observed frame once per sample() call.
struct spdm_encoder {
    int device = 0;
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr, *w3 = nullptr, *b3 = nullptr, *wl = nullptr, *bl = nullptr;
    float* feat = nullptr;            // [chunk][9216] flattened conv-3 maps of the chunk in flight
    int chunk = 0;
    std::vector<void*> owned;
};
static constexpr int ENC_FEAT = 64 * 12 * 12, ENC_LATENT = 128, ENC_CHUNK = 2048;

extern "C" void spdm_encoder_destroy(spdm_encoder* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    for (void* p : e->owned) (void)hipFree(p);
    delete e;
}

extern "C" int spdm_encoder_create(int32_t device, const float* blob, size_t n, const spdm_tensor_index* index, int32_t n_index,
                                   spdm_encoder** out) {
    if (!blob || !index || n_index <= 0 || !out) return fail(SPDM_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(device));
    Loader L{nullptr, blob, n};
    for (int i = 0; i < n_index; ++i) {
        char name[SPDM_NAME_MAX + 1];
        memcpy(name, index[i].name, SPDM_NAME_MAX);
        name[SPDM_NAME_MAX] = 0;
        L.idx[name] = &index[i];
    }
    // nn.Sequential indices of Autoencoder.encoder: 0, 2, 4 = Conv2d; 7 = Linear
    struct Item { const char* name; std::initializer_list<int> shape; float* spdm_encoder::*dst; };
    const Item items[8] = {{"0.weight", {16, 3, 2, 2}, &spdm_encoder::w1}, {"0.bias", {16}, &spdm_encoder::b1},
                           {"2.weight", {32, 16, 2, 2}, &spdm_encoder::w2}, {"2.bias", {32}, &spdm_encoder::b2},
                           {"4.weight", {64, 32, 2, 2}, &spdm_encoder::w3}, {"4.bias", {64}, &spdm_encoder::b3},
                           {"7.weight", {ENC_LATENT, ENC_FEAT}, &spdm_encoder::wl}, {"7.bias", {ENC_LATENT}, &spdm_encoder::bl}};
    spdm_encoder* e = new spdm_encoder();
    e->device = device;
    for (const Item& it : items) {
        const float* src = L.find(it.name, it.shape);
        if (!src) { spdm_encoder_destroy(e); return L.err; }
        size_t numel = 1;
        for (int d : it.shape) numel *= (size_t)d;
        void* p = nullptr;
        if (hipMalloc(&p, numel * sizeof(float)) != hipSuccess || hipMemcpy(p, src, numel * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            if (p) (void)hipFree(p);
            spdm_encoder_destroy(e);
            return fail(SPDM_ERR_HIP, "encoder weight upload failed");
        }
        e->owned.push_back(p);
        e->*(it.dst) = (float*)p;
    }
    *out = e;
    return SPDM_OK;
}

extern "C" int spdm_encoder_forward(spdm_encoder* e, int32_t n_images, const float* d_images, float* d_latent, void* stream) {
    if (!e || !d_images || !d_latent || n_images <= 0) return fail(SPDM_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    const int chunk = std::min<int>(n_images, ENC_CHUNK);
    if (e->chunk < chunk) {             // (grown lazily; the old buffer stays owned until destroy: at most two sizes ever exist)
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, sizeof(float) * (size_t)chunk * ENC_FEAT));
        e->owned.push_back(p);
        e->feat = (float*)p;
        e->chunk = chunk;
    }
    for (int i0 = 0; i0 < n_images; i0 += chunk) {
        const int m = std::min(chunk, n_images - i0);
        HIP_TRY(launch_encoder_convs(d_images + (size_t)i0 * 3 * 96 * 96, e->w1, e->b1, e->w2, e->b2, e->w3, e->b3, e->feat, m, s));
        GemmArgs a{};                   // Linear(9216, 128) on the exact fp32 MFMA path
        a.split = 0;
        a.src = e->feat; a.src_ld = ENC_FEAT; a.wgt = e->wl; a.dst = d_latent + (size_t)i0 * ENC_LATENT; a.dst_ld = ENC_LATENT;
        a.M = m; a.K = ENC_FEAT; a.N = ENC_LATENT; a.taps = 1; a.H = 1; a.W = 1; a.HW = 1;
        a.pro = PRO_NONE; a.epi = EPI_BIAS; a.bias = e->bl;
        HIP_TRY(launch_gemm(a, s));
    }
    if (!stream) HIP_TRY(hipStreamSynchronize(s));
    return SPDM_OK;
}

// Host-only introspection (no GPU): the launch geometry gemm_geometry picks for a statistics-epilogue convolution, plus the
// statistics-slot reservation the plan makes for it (Ctx::salloc).  tests/test_geometry.py checks the invariants between the
// two on a grid of shapes (a mismatch is a silent wrong-statistics bug on the GPU).
extern "C" int spdm_debug_geometry(int32_t M, int32_t N, int32_t K, int32_t HW, int32_t W, int32_t taps, uint32_t switches,
                                   int32_t out[10]) {
    if (!out || M <= 0 || N <= 0 || K <= 0 || HW <= 0 || W <= 0 || M % HW != 0) return fail(SPDM_ERR_INVALID, "bad argument");
    const GemmGeom g = gemm_geometry(M, N, K, HW, W, taps, /*split=*/1, switches, /*stats_epi=*/true);
    out[0] = g.m_tile; out[1] = g.n_tile; out[2] = g.n_tiles; out[3] = g.slots; out[4] = g.ksplit; out[5] = g.skinny | (g.reg << 1);
    out[6] = g.st_m_tile; out[7] = g.st_n_tiles;
    out[8] = std::max(std::max(std::max(g.slots, stats_slots(HW, 128, std::max(1, N / 64))), stats_slots(HW, combine_rows(HW, N), 1)),
                      std::max(stats_slots(HW, 16, std::max(1, N / 16)), stats_slots(HW, std::max(HW / 4, 1), 1)));   // = Ctx::salloc's reservation
    out[9] = combine_rows(HW, N);
    return SPDM_OK;
}

// op-level test hook: y = GELU(x) with the device's own erf (the one every conv prologue uses)
extern "C" int spdm_op_gelu(const float* d_x, float* d_y, size_t n, void* stream) {
    if (!d_x || !d_y || n == 0) return fail(SPDM_ERR_INVALID, "bad argument");
    HIP_TRY(launch_gelu(d_x, d_y, n, (hipStream_t)stream));
    if (!stream) HIP_TRY(hipStreamSynchronize(nullptr));
    return SPDM_OK;
}
