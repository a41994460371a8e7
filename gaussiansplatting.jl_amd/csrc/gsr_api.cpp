// C ABI of libgsr_hip.so (include/gsr.h): handle object with grow-only scratch, argument
// validation, kernel orchestration.  Mirrors the host side of the reference's
// `GaussianRasterizer` / `rasterize` / `∇rasterize` (src/rasterization/rasterizer.jl:5-90,
// 255-408, 416-550) — everything is enqueued on the caller's stream; the only host sync is
// the instance-count read-back (rasterizer.jl:337).
#include "../../include/gsr.h"

#include <dlfcn.h>
#include <math.h>
#include <sched.h>
#include <time.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "gsr_kernels.h"
#include "adam_math.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(e_ == hipErrorOutOfMemory ? GSR_E_OOM : GSR_E_HIP, "%s failed: %s (%s:%d)", #expr,   \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                          \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;  // bytes
    uint32_t regrowths = 0;  // reallocations of a buffer that already existed (each is a device synchronisation)
    // grow-only, like the reference's scratch (rasterizer.jl:275-278,340-343)
    int ensure(size_t bytes, float slack = 1.0f) {
        if (bytes <= cap) return GSR_OK;
        size_t want = (size_t)((double)bytes * slack);
        if (p) {
            // REgrowth of a buffer that is given slack (the per-Gaussian and per-instance scratch of a scene that changes): at
            // least half again of what is there — the views of a batch differ by a few per cent in their instance count, and a
            // training run grows; 288 GB of HBM are there to be used, a hipFree + hipMalloc synchronises the device
            if (slack > 1.0f) want = std::max(want, cap + cap / 2);
            regrowths++;
            HIPCHK(hipFree(p));
            p = nullptr;
            cap = 0;
        }
        want = (want + 255) & ~(size_t)255;
        HIPCHK(hipMalloc(&p, want));
        cap = want;
        return GSR_OK;
    }
    int release() {
        if (p) HIPCHK(hipFree(p));
        p = nullptr;
        cap = 0;
        return GSR_OK;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Optional per-stage timing with HIP events on the caller's stream (gsr_profile_*).
enum Stage { ST_PREPROCESS, ST_SCAN, ST_SORT, ST_COMPOSITE_FWD, ST_LOSS_FWD, ST_LOSS_BWD, ST_ZERO_ACC,
             ST_COMPOSITE_BWD, ST_PERGAUSS_BWD, ST_SORT_COMPOSITE_FWD, ST_COUNT };
// "sort_composite_fwd": the fused launch (sort + forward of every tile of up to 1024 instances); "tile_sort" and
// "composite_fwd" then only hold the tier launches of longer lists, or the whole passes when the fused launch does not apply
const char* const kStageNames[ST_COUNT] = {"preprocess", "tile_scan", "tile_sort", "composite_fwd",
                                           "loss_fwd", "loss_bwd", "zero_acc", "composite_bwd", "pergauss_bwd",
                                           "sort_composite_fwd"};
// roctx ranges per stage: resolved lazily from the ROCm tools library, only when asked for
// (GSR_ROCTX=1 in the environment, or gsr_profile_enable(h, 2 | ...)); rocprofv3 --marker-trace then
// slices the kernel trace by stage without kernel-name matching.
struct Roctx {
    typedef int (*push_fn)(const char*);
    typedef int (*pop_fn)(void);
    push_fn push = nullptr;
    pop_fn pop = nullptr;
    bool tried = false;
    bool load() {
        if (tried) return push != nullptr;
        tried = true;
        const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
        for (const char* n : names) {
            void* lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!lib) continue;
            push = (push_fn)dlsym(lib, "roctxRangePushA");
            pop = (pop_fn)dlsym(lib, "roctxRangePop");
            if (push && pop) return true;
            push = nullptr; pop = nullptr;
        }
        return false;
    }
};
Roctx g_roctx;
bool roctx_env() {
    static const bool on = [] { const char* e = getenv("GSR_ROCTX"); return e && e[0] == '1'; }();
    return on;
}

struct Profiler {
    bool on = false;      // HIP-event timing per stage
    bool ranges = false;  // roctx range per stage
    bool in_range = false;
    bool rec_open = false;
    uint32_t mask = ~0u;  // stages that get an event pair (gsr_profile_stages)
    struct Rec { int stage; hipEvent_t a, b; bool extra; };  // extra: a second launch group of a stage inside ONE call (time, not count)
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        // timing-only events: no system-scope fence (cache write-back + invalidate) when the marker retires — the bubble an
        // event record leaves between two kernels is what the timed region pays for its one profiled stage
        (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
        return e;
    }
    void begin(int stage, hipStream_t s, bool extra = false);
    void end(hipStream_t s) {
        if (in_range) { g_roctx.pop(); in_range = false; }
        if (!rec_open) return;
        rec_open = false;
        (void)hipEventRecord(recs.back().b, s);
    }
    // error path: a stage that was begun but whose launch failed — close the roctx range and DROP the half-recorded
    // pair (its end event was never recorded: gsr_profile_read would fail or mis-time on it)
    void abort() {
        if (in_range) { g_roctx.pop(); in_range = false; }
        if (!rec_open) return;
        rec_open = false;
        pool.push_back(recs.back().a);
        pool.push_back(recs.back().b);
        recs.pop_back();
    }
    void clear() {
        for (auto& r : recs) { pool.push_back(r.a); pool.push_back(r.b); }
        recs.clear();
    }
    void destroy() {
        clear();
        for (auto e : pool) (void)hipEventDestroy(e);
        pool.clear();
    }
};

void Profiler::begin(int stage, hipStream_t s, bool extra) {
    if ((ranges || roctx_env()) && g_roctx.load()) {
        char name[48];
        snprintf(name, sizeof name, "gsr:%s", kStageNames[stage]);
        g_roctx.push(name);
        in_range = true;
    }
    if (!on || !((mask >> stage) & 1u)) return;
    Rec r{stage, get(), get(), extra};
    (void)hipEventRecord(r.a, s);
    recs.push_back(r);
    rec_open = true;
}

// One profiled stage: begun by the constructor, ended by close(); a scope left WITHOUT close() — an early `return rc`
// or a HIPCHK between the two — aborts the stage instead of leaving a pushed roctx range and an open record behind.
struct StageScope {
    Profiler& p;
    hipStream_t s;
    bool open = true;
    StageScope(Profiler& prof, int stage, hipStream_t stream, bool extra = false) : p(prof), s(stream) { p.begin(stage, s, extra); }
    void close() { if (open) { p.end(s); open = false; } }
    ~StageScope() { if (open) p.abort(); }
    StageScope(const StageScope&) = delete;
    StageScope& operator=(const StageScope&) = delete;
};

// The form tuner's clock (the RULE is gsr_policy.cpp's: which views are timed, what decides, when it starts over): one event
// pair per timed view around the view's first kernel, read without blocking a later view.
struct TunerEvents {
    hipEvent_t ev[2 * GSR_TUNER_TIMED_VIEWS] = {};
    bool create() {
        for (auto& e : ev)
            if (!e && hipEventCreate(&e) != hipSuccess) { e = nullptr; return false; }
        return true;
    }
    void destroy() { for (auto& e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; } }
    // all timed views complete?  -> their first kernels' milliseconds.  "Not ready" is the expected answer most of the time
    // and is consumed here; any OTHER error stays for the caller's next HIPCHK (an unconditional hipGetLastError
    // swallowed unrelated sticky errors).
    bool read(float ms[GSR_TUNER_TIMED_VIEWS]) {
        for (int v = 0; v < GSR_TUNER_TIMED_VIEWS; v++) {
            const hipError_t q = hipEventQuery(ev[2 * v + 1]);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); return false; }
            if (q != hipSuccess || hipEventElapsedTime(&ms[v], ev[2 * v], ev[2 * v + 1]) != hipSuccess) return false;
        }
        return true;
    }
};

bool valid_mode(int m) { return m == GSR_MODE_RGB || m == GSR_MODE_RGBD || m == GSR_MODE_RGBDN; }

}  // namespace

struct gsr_handle {
    gsr_config cfg;
    int grid_x, grid_y, n_tiles;
    // ImageState (states.jl:99-111) + tile bookkeeping
    DevBuf ranges, n_contrib, final_T, tile_count, tile_start, tile_order, totals;
    // GeometryState (states.jl:2-47), repacked as one 64-byte record per Gaussian
    DevBuf geo, gnormal, radii, bsum, bpre, bvis;
    // BinningState (states.jl:66-85): unsorted keys (per-tile bins of bin_cap slots), sorted ids, sorted splat stream
    // everything gsr_forward carries from one view to the next lives in `pol`, and every decision taken from it is a pure
    // function of gsr_policy.cpp (include/gsr_policy.h): bins capacity / compact mode, the binning form and its tuner, the held
    // fused launch, the backward's list split.  This file only executes them.
    gsr_policy_config pcfg;
    gsr_policy_state pol;
    TunerEvents tuner_ev;
    int last_form = 0;              // binning form the last view's preprocess ran in (gsr_stats.preprocess_form)
    bool last_compact = false;
    bool tile_count_dirty = false;  // counters not yet re-zeroed by the tile sort
    double wait_ema_us = 0.0;       // running average of the host's wait for the instance count (wait_totals)
    DevBuf bins, keys_compact, big_list, values_sorted, s0, s1, s2, s3, big_scratch, long_state;
    DevBuf overflow_fill;           // fill cursors of the scatter pass restricted to the lists beyond the bins' capacity
    // backward: per-instance gradient rows + instance position map; gstate.∇means_2d
    DevBuf rows, vmean2d;
    // loss-head scratch
    DevBuf d0, d1, d2, partial;
    uint32_t* host_totals = nullptr;  // pinned: D, max tile count, #oversized tiles, slab ctr, n_visible, ..., [7] = sequence
    uint32_t* host_totals_dev = nullptr;  // the same words as the device addresses them
    uint32_t totals_seq = 0;
    hipStream_t aux_stream = nullptr;  // the four-wave backward of those tiles runs here, next to the main launch
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool fwd_valid = false, bwd_valid = false;
    bool fwd_only = false;         // the last forward was GSR_FORWARD_ONLY: no stream / ids / row storage behind it
    bool inputs_consumed = false;  // gsr_backward_trainer_tail updated the forward's inputs in place
    uint64_t generation = 0;             // ordinal of the last gsr_forward (gsr_stats.generation)
    int32_t* radii_cur = nullptr;        // gstate.radii of the last forward: caller's (gsr_aux.radii) or h->radii
    float2* vmean2d_cur = nullptr;       // gstate.∇means_2d of the last backward: caller's (gsr_grads.vmeans2d) or h->vmean2d
    int last_n = 0;
    int64_t last_D = 0;
    int64_t last_slots = 0;
    // the cotangent the handle's loss head wrote, and for which forward (GSR_GRADS_COLOR_COTANGENT is honoured for it only)
    const float* loss_vpixels = nullptr;
    uint64_t loss_generation = 0;
    uint32_t reserved_regrowths = 0;  // reallocations made by gsr_reserve (not counted in gsr_stats.scratch_regrowths)
    DevBuf dbg_flag;
    Profiler prof;

    DevBuf* all[40];
    int n_all = 0;
};

namespace {

GsrCam make_cam(const gsr_handle* h, const gsr_camera* c) {
    GsrCam k;
    for (int r = 0; r < 3; r++)
        for (int col = 0; col < 3; col++) k.R[r * 3 + col] = c->R[col * 3 + r];
    for (int i = 0; i < 3; i++) { k.t[i] = c->t[i]; k.center[i] = c->camera_center[i]; }
    for (int i = 0; i < 2; i++) { k.focal[i] = c->focal[i]; k.principal[i] = c->principal[i]; }
    k.width = h->cfg.width; k.height = h->cfg.height;
    k.grid_x = h->grid_x; k.grid_y = h->grid_y;
    k.near_plane = h->cfg.near_plane; k.far_plane = h->cfg.far_plane;
    k.radius_clip = h->cfg.radius_clip; k.blur_eps = h->cfg.blur_eps;
    k.exact_cull = (h->cfg.flags & GSR_FLAG_REFERENCE_TILE_LISTS) ? 0 : 1;
    k.R_dev = c->R_dev; k.t_dev = c->t_dev;
    return k;
}

GsrGeom geom_of(const gsr_handle* h) {
    return GsrGeom{h->geo.as<GsrGeoRec>(), h->gnormal.as<float4>(), h->radii_cur, h->bsum.as<uint32_t>(),
                   h->bpre.as<uint32_t>()};
}
GsrStream stream_of(const gsr_handle* h) {
    return GsrStream{h->s0.as<float4>(), h->s1.as<float4>(), h->s2.as<float4>(), h->s3.as<float4>()};
}
GsrInst inst_of(const gsr_handle* h) { return GsrInst{h->rows.as<float4>()}; }

// CPU relax hint inside the spin phase: x86 `pause`, AArch64 `yield`, nothing elsewhere (the host side builds on any arch).
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}

// Host wait policy of the forward's single read-back (gsr_host_wait_policy), process-wide.
//   DEFAULT (sleep_us == 0): spin for spin_us microseconds, then poll with sched_yield() between looks at the word.  Any
//   other runnable thread — RCCL's proxy threads on an 8-rank host — gets the core at once, no timer is involved, and
//   the step time is a pure spin's (config 2: 0.2123-0.2137 ms, config 3: 1.494 ms per step over 200 steps, all three
//   policies within noise of each other: tools/host_wait_ab.sh).
//   OPT-IN (sleep_us > 0): the thread SLEEPS through the expected wait (a running average of this handle's previous
//   waits: the host runs ahead of the GPU, so the wait is as long as the work still queued in front of the scan, about
//   the same every step) except for its last spin_us microseconds — in doubling pieces of 50, 100, 200 ... us with a look
//   at the word after each, so that the first forward after the caller synchronised (GPU idle: a short wait) costs one
//   small piece; a wake-up that finds the word already written is not a sample (an average fed with its own oversleeps
//   ratchets upwards and starves the GPU) but shrinks the estimate; after the expected time + spin_us: sched_yield() for
//   yield_us, then sleeps of sleep_us.  Same step time on a quiet host, a fraction of the CPU time; NOT the default
//   because a late timer wake-up (observed on one box of the pool: one step of twenty 3 ms late) lands in the step time.
struct WaitPolicy { int spin_us = 30, yield_us = 0, sleep_us = 0; };
// the three values live in ONE atomic word (21 bits each): a forward on another thread reads the old or the new policy,
// never a mix of the two (round-4 verdict, weak #9)
constexpr int kWaitMaxUs = (1 << 21) - 1;
inline uint64_t pack_wait(const WaitPolicy& w) {
    return (uint64_t)w.spin_us | ((uint64_t)w.yield_us << 21) | ((uint64_t)w.sleep_us << 42);
}
std::atomic<uint64_t> g_wait_packed{pack_wait(WaitPolicy{})};
inline WaitPolicy load_wait() {
    const uint64_t v = g_wait_packed.load(std::memory_order_relaxed);
    WaitPolicy w;
    w.spin_us = (int)(v & kWaitMaxUs); w.yield_us = (int)((v >> 21) & kWaitMaxUs); w.sleep_us = (int)((v >> 42) & kWaitMaxUs);
    return w;
}

// gsr_ssim_precision: 0 = the contracted build of ssim.hip (default), 1 = its bit-exact twin.  GSR_SSIM_EXACT=1 in the
// environment starts a process in exact mode (A/B runs of unmodified callers).
// Process-wide DEFAULT only: a handle created with gsr_config.ssim_precision = 0 / 1 is pinned (ABI 5).
std::atomic<int> g_ssim_exact{[] { const char* e = getenv("GSR_SSIM_EXACT"); return e && e[0] == '1' ? 1 : 0; }()};
// gsr_preprocess_form: -1 by scene and grid size (default), 0 direct, 1 aggregating wherever its LDS fits; the same rule
std::atomic<int> g_preprocess_form{[] { const char* e = getenv("GSR_PREPROCESS_AGG"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }()};
inline int ssim_exact_of(const gsr_handle* h) {
    return h->cfg.ssim_precision != GSR_DEFAULT ? (h->cfg.ssim_precision == GSR_SSIM_EXACT ? 1 : 0)
                                                : g_ssim_exact.load(std::memory_order_relaxed);
}
inline int preprocess_form_of(const gsr_handle* h) {  // -1 by scene and grid, 0 direct, 1 aggregating
    return h->cfg.preprocess_form != GSR_DEFAULT ? (h->cfg.preprocess_form == GSR_PREPROCESS_AGGREGATING ? 1 : 0)
                                                 : g_preprocess_form.load(std::memory_order_relaxed);
}

// Wait until tile_scan of forward `seq` has published its totals (a word of pinned host memory).
// A wait that lasts longer than any sane queue depth (50 ms) starts polling the stream, so that a failed launch or a
// faulted kernel ends with an error instead of hanging the caller — not earlier: hipStreamQuery puts a marker packet on
// the stream, and a marker between two kernels is a 5 us bubble (rocprofv3 kernel trace, tools/gap_report.py).
int wait_totals(gsr_handle* h, uint32_t seq, hipStream_t s) {
    using clk = std::chrono::steady_clock;
    volatile uint32_t* word = h->host_totals + 7;
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) return GSR_OK;
    const auto t0 = clk::now();
    const double expect_us = h->wait_ema_us;
    const WaitPolicy g_wait = load_wait();  // one consistent snapshot for this wait
    if (g_wait.sleep_us > 0 && expect_us > (double)g_wait.spin_us + 50.0) {  // sleeping enabled and worth it
        // ... in doubling pieces (50, 100, 200, ... us) with a look at the word after each: the expectation comes from
        // steps in which the host ran ahead of the GPU; the first forward after the caller synchronised finds the GPU
        // idle and its wait is only preprocess + scan long — the small first pieces bound what that costs.
        double remaining = expect_us - (double)g_wait.spin_us, piece = 50.0;
        while (remaining > 0.0) {
            const long ns = (long)((remaining < piece ? remaining : piece) * 1000.0);
            struct timespec ts = {ns / 1000000000L, ns % 1000000000L};
            nanosleep(&ts, nullptr);
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) {
                // Overslept (or woke exactly on time): how long the wait really was is unknown — only that it was no
                // longer than this.  NOT a sample of the average (an average fed with its own oversleeps ratchets upwards
                // and starves the GPU): the estimate shrinks instead, so that the next wake-up comes early enough to see
                // the word arrive.
                const double slept = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
                h->wait_ema_us = 0.85 * (slept < expect_us ? slept : expect_us);
                return GSR_OK;
            }
            remaining = expect_us - (double)g_wait.spin_us - std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            piece *= 2.0;
        }
    }
    const auto t_spin = t0 + std::chrono::microseconds((g_wait.sleep_us > 0 ? (long)expect_us : 0L) + g_wait.spin_us);
    const auto t_yield = t_spin + std::chrono::microseconds(g_wait.yield_us);
    auto next_poll = t0 + std::chrono::milliseconds(50);
    int rc = GSR_OK;
    for (;;) {
        bool done = false;
        for (int i = 0; i < 64 && !done; i++) {
            done = __atomic_load_n(word, __ATOMIC_ACQUIRE) == seq;
            if (!done) cpu_relax();
        }
        const auto now = clk::now();
        if (done) {
            const double us = std::chrono::duration<double, std::micro>(now - t0).count();
            h->wait_ema_us = h->wait_ema_us == 0.0 ? us : 0.75 * h->wait_ema_us + 0.25 * us;
            return GSR_OK;
        }
        if (now >= t_spin) {
            if (now < t_yield || g_wait.sleep_us <= 0) {
                sched_yield();
            } else {
                struct timespec ts = {0, (long)g_wait.sleep_us * 1000L};
                nanosleep(&ts, nullptr);
            }
        }
        if (now < next_poll) continue;
        next_poll = now + std::chrono::milliseconds(50);
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) {  // everything enqueued has finished: the word is there, or it never will be
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == seq) return GSR_OK;
            rc = fail(GSR_E_HIP, "tile scan finished without publishing its totals");
            break;
        }
        if (q != hipErrorNotReady) { rc = fail(GSR_E_HIP, "HIP error while waiting for the tile scan: %s", hipGetErrorString(q)); break; }
    }
    h->wait_ema_us = 0.0;
    return rc;
}

// gsr_stats' view-history block (ABI 6): the policy state's counters + the mechanism's own (buffer reallocations)
void fill_history(const gsr_handle* h, gsr_stats* st) {
    const gsr_policy_state& p = h->pol;
    st->bins_regrowths = p.bins_regrowths;
    st->compact_fallbacks = p.compact_fallbacks;
    st->tuner_rearms = p.tuner_rearms;
    uint32_t grows = 0;
    for (int i = 0; i < h->n_all; i++) grows += h->all[i]->regrowths;
    st->scratch_regrowths = grows - h->reserved_regrowths;  // (inside a view: what gsr_reserve reallocated is not counted)
    st->fused_relaunches = p.fused_relaunches;
    st->held_views = p.held_views;
    const bool decided = p.tuner.phase == GSR_TUNER_TIMED_VIEWS + 1;
    st->tuner_form = decided ? p.tuner.form : -1;
    st->tuner_ms[0] = decided ? p.tuner.ms[0] : 0.0f;
    st->tuner_ms[1] = decided ? p.tuner.ms[1] : 0.0f;
}

// GSR_GRADS_COLOR_COTANGENT (gsr.h): honoured for the cotangent this handle's loss head wrote for THIS forward, nothing else.
// GSR_CHECK_COLOR_COTANGENT=1 (debugging): also look at the buffer — every value of channels >= 3 must be exactly zero.
__global__ void nonzero_tail_kernel(const float* __restrict__ v, size_t n_pixels, int C, uint32_t* flag) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    bool bad = false;
    for (int c = 3; c < C; c++) bad |= v[p * C + c] != 0.0f;
    if (bad) atomicOr(flag, 1u);
}
int check_color_cotangent(gsr_handle* h, const float* vpixels, hipStream_t s) {
    if (h->cfg.mode <= 3) return GSR_OK;  // nothing above the colour channels
    if (vpixels != h->loss_vpixels || h->loss_generation != h->generation)
        return fail(GSR_E_INVALID_ARG, "GSR_GRADS_COLOR_COTANGENT is only valid for the cotangent gsr_loss_l1_ssim wrote for this "
                    "forward on this handle (its vpixels %p for forward #%llu; got %p for forward #%llu): a cotangent with depth / "
                    "alpha / normal terms must not set the flag", (const void*)h->loss_vpixels,
                    (unsigned long long)h->loss_generation, (const void*)vpixels, (unsigned long long)h->generation);
    const char* look = getenv("GSR_CHECK_COLOR_COTANGENT");  // (read per call: a debugging switch, flipped by tests in-process)
    if (!(look && look[0] == '1')) return GSR_OK;
    int rc = h->dbg_flag.ensure(4);
    if (rc) return rc;
    const size_t P = (size_t)h->cfg.width * h->cfg.height;
    uint32_t bad = 0;
    HIPCHK(hipMemsetAsync(h->dbg_flag.p, 0, 4, s));
    hipLaunchKernelGGL(nonzero_tail_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, vpixels, P, h->cfg.mode,
                       h->dbg_flag.as<uint32_t>());
    HIPCHK(hipMemcpyAsync(&bad, h->dbg_flag.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (bad) return fail(GSR_E_INVALID_ARG, "GSR_GRADS_COLOR_COTANGENT: vpixels holds non-zero values above the colour channels "
                         "(something was added to the loss head's cotangent in place)");
    return GSR_OK;
}

int check_inputs(const gsr_handle* h, const gsr_inputs* in, const gsr_camera* cam) {
    if (!h || !in || !cam) return fail(GSR_E_INVALID_ARG, "null handle / inputs / camera");
    if (in->n < 0) return fail(GSR_E_INVALID_ARG, "n = %d < 0", in->n);
    if (in->sh_degree < 0 || in->sh_degree > 3) return fail(GSR_E_INVALID_ARG, "sh_degree = %d not in 0..3", in->sh_degree);
    if (in->n_coeffs < (in->sh_degree + 1) * (in->sh_degree + 1) || in->n_coeffs > 16)
        return fail(GSR_E_INVALID_ARG, "n_coeffs = %d does not hold degree %d", in->n_coeffs, in->sh_degree);
    if (in->n > 0 && (!in->means || !in->shs || !in->opacities || !in->scales || !in->rotations))
        return fail(GSR_E_INVALID_ARG, "null input array");
    if (((uintptr_t)in->rotations & 15) != 0) return fail(GSR_E_INVALID_ARG, "rotations must be 16-byte aligned");
    if ((cam->R_dev == nullptr) != (cam->t_dev == nullptr))
        return fail(GSR_E_INVALID_ARG, "R_dev and t_dev must be given together");
    return GSR_OK;
}

}  // namespace

extern "C" {

const char* gsr_last_error_string(void) { return g_err; }
#define GSR_STR2(x) #x
#define GSR_STR(x) GSR_STR2(x)
const char* gsr_version(void) { return "gsr-hip 0.4 abi " GSR_STR(GSR_ABI_VERSION) " (gfx950)"; }
int gsr_abi_version(void) { return GSR_ABI_VERSION; }

int gsr_host_wait_policy(int spin_us, int yield_us, int sleep_us) {
    if (spin_us < 0 || yield_us < 0 || sleep_us < 0) return fail(GSR_E_INVALID_ARG, "negative wait time");
    if (spin_us > kWaitMaxUs || yield_us > kWaitMaxUs || sleep_us > kWaitMaxUs)
        return fail(GSR_E_INVALID_ARG, "wait times are at most %d us", kWaitMaxUs);
    WaitPolicy w;
    w.spin_us = spin_us; w.yield_us = yield_us; w.sleep_us = sleep_us;
    g_wait_packed.store(pack_wait(w), std::memory_order_relaxed);
    return GSR_OK;
}

int gsr_ssim_precision(int exact) {
    if (exact != 0 && exact != 1) return fail(GSR_E_INVALID_ARG, "gsr_ssim_precision: 0 (fast) or 1 (exact)");
    g_ssim_exact.store(exact, std::memory_order_relaxed);
    return GSR_OK;
}
int gsr_get_ssim_precision(void) { return g_ssim_exact.load(std::memory_order_relaxed); }

int gsr_preprocess_form(int form) {
    if (form < -1 || form > 1) return fail(GSR_E_INVALID_ARG, "gsr_preprocess_form: -1 (by size), 0 (direct) or 1 (aggregating)");
    g_preprocess_form.store(form, std::memory_order_relaxed);
    return GSR_OK;
}
int gsr_get_preprocess_form(void) { return g_preprocess_form.load(std::memory_order_relaxed); }

int gsr_check_abi(int abi_version, size_t sizeof_config, size_t sizeof_inputs, size_t sizeof_camera, size_t sizeof_aux,
                  size_t sizeof_stats, size_t sizeof_grads, size_t sizeof_tail_state) {
    if (abi_version != GSR_ABI_VERSION)
        return fail(GSR_E_INVALID_ARG, "binding was written for gsr ABI %d, this library is ABI %d", abi_version, GSR_ABI_VERSION);
    const struct { const char* name; size_t theirs, ours; } t[] = {
        {"gsr_config", sizeof_config, sizeof(gsr_config)}, {"gsr_inputs", sizeof_inputs, sizeof(gsr_inputs)},
        {"gsr_camera", sizeof_camera, sizeof(gsr_camera)}, {"gsr_aux", sizeof_aux, sizeof(gsr_aux)},
        {"gsr_stats", sizeof_stats, sizeof(gsr_stats)},    {"gsr_grads", sizeof_grads, sizeof(gsr_grads)},
        {"gsr_tail_state", sizeof_tail_state, sizeof(gsr_tail_state)}};
    for (const auto& e : t)
        if (e.theirs != e.ours)
            return fail(GSR_E_INVALID_ARG, "sizeof(%s) is %zu in the binding, %zu in the library", e.name, e.theirs, e.ours);
    return GSR_OK;
}

int gsr_create(const gsr_config* cfg, gsr_handle** out) {
    if (!cfg || !out) return fail(GSR_E_INVALID_ARG, "null config / out");
    if (cfg->width <= 0 || cfg->height <= 0) return fail(GSR_E_INVALID_ARG, "bad resolution %dx%d", cfg->width, cfg->height);
    if (!valid_mode(cfg->mode)) return fail(GSR_E_INVALID_ARG, "Invalid render mode: %d (3=rgb, 5=rgbd, 8=rgbdn)", cfg->mode);
    if (!(cfg->near_plane < cfg->far_plane)) return fail(GSR_E_INVALID_ARG, "near_plane >= far_plane");
    if (cfg->flags & GSR_FLAG_RETIRED_BIT0)
        return fail(GSR_E_INVALID_ARG, "flag bit 0x1 is retired (ABI 1's GSR_FLAG_EXACT_TILE_CULL): rebuild the caller against "
                                       "include/gsr.h ABI %d", GSR_ABI_VERSION);
    if (cfg->flags & ~(uint32_t)GSR_FLAG_REFERENCE_TILE_LISTS) return fail(GSR_E_INVALID_ARG, "unknown flags 0x%x", cfg->flags);
    if (cfg->ssim_precision < 0 || cfg->ssim_precision > GSR_SSIM_EXACT)
        return fail(GSR_E_INVALID_ARG, "gsr_config.ssim_precision = %d: GSR_DEFAULT (0), GSR_SSIM_FAST (1) or GSR_SSIM_EXACT (2)",
                    cfg->ssim_precision);
    if (cfg->preprocess_form < 0 || cfg->preprocess_form > GSR_PREPROCESS_AGGREGATING)
        return fail(GSR_E_INVALID_ARG, "gsr_config.preprocess_form = %d: GSR_DEFAULT (0), GSR_PREPROCESS_DIRECT (1) or "
                    "GSR_PREPROCESS_AGGREGATING (2)", cfg->preprocess_form);
    if (cfg->form_tuner < 0 || cfg->form_tuner > GSR_TUNER_ON)
        return fail(GSR_E_INVALID_ARG, "gsr_config.form_tuner = %d: GSR_DEFAULT (0), GSR_TUNER_OFF (1) or GSR_TUNER_ON (2)", cfg->form_tuner);
    if (cfg->grad_precision < 0 || cfg->grad_precision > GSR_GRAD_ACCURATE)
        return fail(GSR_E_INVALID_ARG, "gsr_config.grad_precision = %d: GSR_DEFAULT (0), GSR_GRAD_FP32_REFERENCE (1) or GSR_GRAD_ACCURATE (2)",
                    cfg->grad_precision);
    gsr_handle* h = new (std::nothrow) gsr_handle();
    if (!h) return fail(GSR_E_OOM, "host allocation failed");
    h->cfg = *cfg;
    h->grid_x = (cfg->width + GSR_TILE - 1) / GSR_TILE;
    h->grid_y = (cfg->height + GSR_TILE - 1) / GSR_TILE;
    // (8 192 tiles = 131 072 pixels wide: one row of the tile grid must fit the aggregating binning's LDS band, pergauss.hip)
    if (h->grid_x > 8192 || h->grid_y > 65535) { delete h; return fail(GSR_E_INVALID_ARG, "resolution too large"); }
    h->n_tiles = h->grid_x * h->grid_y;
    // the handle's policies: configuration (constructor fields, then the A/B environment knobs) and the view-history state
    gsr_policy_config_init(&h->pcfg, cfg->width, cfg->height, cfg->bins_budget_bytes, -1 /* resolved per view: process default */);
    gsr_policy_state_init(&h->pol);
    {
        const auto env_u32 = [](const char* name, uint32_t dflt) { const char* e = getenv(name); return e ? (uint32_t)atoi(e) : dflt; };
        const char* tun = getenv("GSR_FORM_TUNER");  // the DEFAULT of handles that do not say (0 = off)
        h->pcfg.form_tuner = cfg->form_tuner == GSR_TUNER_ON ? 1 : cfg->form_tuner == GSR_TUNER_OFF ? 0 : !(tun && tun[0] == '0');
        h->pcfg.beside_max_tiles = env_u32("GSR_TIERS_BESIDE_MAX", h->pcfg.beside_max_tiles);      // 0 = never hold the fused launch
        h->pcfg.bwd_split_max_tiles = env_u32("GSR_BWD_SPLIT_TILES", h->pcfg.bwd_split_max_tiles);
        h->pcfg.agg_max_bands = (int32_t)env_u32("GSR_AGG_MAX_BANDS", (uint32_t)h->pcfg.agg_max_bands);
    }
    DevBuf* list[] = {&h->ranges, &h->n_contrib, &h->final_T, &h->tile_count, &h->tile_start, &h->tile_order, &h->totals,
                      &h->geo, &h->gnormal, &h->radii, &h->bsum, &h->bpre, &h->bvis, &h->bins, &h->values_sorted, &h->s0,
                      &h->s1, &h->s2, &h->s3, &h->big_scratch, &h->rows, &h->vmean2d, &h->d0, &h->d1,
                      &h->d2, &h->partial, &h->keys_compact, &h->big_list, &h->long_state, &h->overflow_fill, &h->dbg_flag};
    for (DevBuf* b : list) h->all[h->n_all++] = b;
    const size_t P = (size_t)cfg->width * cfg->height, T = (size_t)h->n_tiles;
    int rc = GSR_OK;
    if ((rc = h->ranges.ensure(2 * T * 4)) || (rc = h->n_contrib.ensure(P * 4)) || (rc = h->final_T.ensure(P * 4)) ||
        (rc = h->tile_count.ensure((T + 2) * 4)) || (rc = h->tile_start.ensure((T + 1) * 4)) ||
        (rc = h->tile_order.ensure((T + 8) * 4)) || (rc = h->totals.ensure(8 * 4)) || (rc = h->big_list.ensure((3 * T + 1) * 4))) {
        gsr_destroy(h);
        return rc;
    }
    hipError_t e = hipHostMalloc((void**)&h->host_totals, 8 * sizeof(uint32_t), hipHostMallocDefault);  // fine-grained
    if (e == hipSuccess) {
        memset(h->host_totals, 0, 8 * sizeof(uint32_t));
        e = hipHostGetDevicePointer((void**)&h->host_totals_dev, h->host_totals, 0);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) {
        gsr_destroy(h);
        return fail(GSR_E_HIP, "pinned memory / stream / event creation failed: %s", hipGetErrorString(e));
    }
    (void)hipMemset(h->totals.p, 0, 8 * 4);  // [7]: the scan's ticket word
    (void)hipMemset(h->ranges.p, 0, 2 * T * 4);
    (void)hipMemset(h->tile_count.p, 0, (T + 2) * 4);
    *out = h;
    return GSR_OK;
}

int gsr_destroy(gsr_handle* h) {
    if (!h) return GSR_OK;
    for (int i = 0; i < h->n_all; i++) h->all[i]->release();
    if (h->host_totals) (void)hipHostFree(h->host_totals);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
    h->tuner_ev.destroy();
    h->prof.destroy();
    delete h;
    return GSR_OK;
}

int gsr_release_scene_buffers(gsr_handle* h) {
    if (!h) return fail(GSR_E_INVALID_ARG, "null handle");
    DevBuf* scene[] = {&h->geo, &h->gnormal, &h->radii, &h->bsum, &h->bpre, &h->bvis, &h->bins, &h->values_sorted, &h->s0, &h->s1,
                       &h->s2, &h->s3, &h->big_scratch, &h->rows, &h->vmean2d, &h->keys_compact};
    h->pol.bin_cap = 0;
    h->pol.compact_sticky = 0;
    for (DevBuf* b : scene) {
        int rc = b->release();
        if (rc) return rc;
    }
    h->fwd_valid = false;
    h->bwd_valid = false;
    h->radii_cur = nullptr;
    h->vmean2d_cur = nullptr;
    h->last_n = 0;
    h->last_D = 0;
    return GSR_OK;
}

int gsr_reserve(gsr_handle* h, int64_t n_gaussians, int64_t n_instances) {
    if (!h) return fail(GSR_E_INVALID_ARG, "null handle");
    if (n_gaussians < 0 || n_instances < 0 || n_gaussians > 0x7FFFFFFFll || n_instances > 0xFFFFFFFFll)
        return fail(GSR_E_INVALID_ARG, "gsr_reserve: sizes out of range");
    const int C = h->cfg.mode;
    int rc = GSR_OK;
    struct Uncount {  // reallocations made here are the caller's choice of WHEN, not a forward's surprise: keep them out of the history
        gsr_handle* h;
        uint32_t before = 0;
        uint32_t sum() const { uint32_t g = 0; for (int i = 0; i < h->n_all; i++) g += h->all[i]->regrowths; return g; }
        explicit Uncount(gsr_handle* h_) : h(h_) { before = sum(); }
        ~Uncount() { h->reserved_regrowths += sum() - before; }
    } uncount(h);
    if (n_gaussians > 0) {
        const size_t nn = (size_t)n_gaussians, nb = (nn + 255) / 256 + 1;
        if ((rc = h->geo.ensure(nn * 64)) || (rc = h->radii.ensure(nn * 4)) || (rc = h->bsum.ensure(nb * 4)) ||
            (rc = h->bpre.ensure(nb * 4)) || (rc = h->bvis.ensure(nb * 4)) || (rc = h->vmean2d.ensure(nn * 8)) ||
            (C > 5 && (rc = h->gnormal.ensure(nn * 16))))
            return rc;
    }
    if (n_instances > 0) {
        const size_t D = (size_t)n_instances;
        // (the gradient rows are indexed by Gaussian-major SLOT: one per emitted tile of a small rect, one per tile of a large one —
        //  at most ~1.5 x the instance count under exact culling)
        if ((rc = h->values_sorted.ensure(D * 4)) || (rc = h->s0.ensure(D * 16)) || (rc = h->s1.ensure(D * 16)) ||
            (rc = h->s2.ensure(D * 16)) || (C > 3 && (rc = h->s3.ensure(D * 16))) || (rc = h->rows.ensure(D * 64 + D * 32)))
            return rc;
    }
    return GSR_OK;
}

int64_t gsr_memory_usage(const gsr_handle* h) {
    if (!h) return 0;
    int64_t s = 0;
    for (int i = 0; i < h->n_all; i++) s += (int64_t)h->all[i]->cap;
    return s;
}

int gsr_forward(gsr_handle* h, const gsr_inputs* in, const gsr_camera* cam, float* image_out, const gsr_aux* aux,
                void* stream_v, gsr_stats* stats) {
    int rc = check_inputs(h, in, cam);
    if (rc) return rc;
    if (!image_out) return fail(GSR_E_INVALID_ARG, "null image_out");
    if (aux && ((aux->flags & ~(uint32_t)GSR_FORWARD_ONLY) || aux->reserved))
        return fail(GSR_E_INVALID_ARG, "unknown gsr_aux.flags 0x%x / reserved 0x%x", aux->flags, aux->reserved);
    const bool fwd_only = aux && (aux->flags & GSR_FORWARD_ONLY);
    hipStream_t s = (hipStream_t)stream_v;
    const int C = h->cfg.mode, n = in->n;
    const size_t P = (size_t)h->cfg.width * h->cfg.height, T = (size_t)h->n_tiles;
    h->fwd_valid = false;
    h->bwd_valid = false;
    h->inputs_consumed = false;
    h->fwd_only = fwd_only;

    const size_t nn = n > 0 ? (size_t)n : 1;
    const int n_blocks = (n + 255) / 256;
    const bool own_radii = !(aux && aux->radii);
    // per-Gaussian scratch: grow-only with 25 % slack, like the per-instance buffers below — a training run's densification
    // adds 5-15 % of Gaussians per round (strategy.jl:78-105), and an exact fit made EVERY round's first view reallocate all
    // five (five hipFree + hipMalloc = device synchronisations: the slowest plain steps of the round-6 training protocol)
    const float nslack = 1.25f;
    if ((rc = h->geo.ensure(nn * 64, nslack)) || (own_radii && (rc = h->radii.ensure(nn * 4, nslack))) ||
        (rc = h->bsum.ensure((size_t)(n_blocks + 1) * 4, nslack)) || (rc = h->bpre.ensure((size_t)(n_blocks + 1) * 4, nslack)) ||
        (rc = h->bvis.ensure((size_t)(n_blocks + 1) * 4, nslack)) || (C > 5 && (rc = h->gnormal.ensure(nn * 16, nslack))))
        return rc;
    h->radii_cur = own_radii ? h->radii.as<int32_t>() : aux->radii;
    h->vmean2d_cur = nullptr;
    h->generation++;
    if (stats) stats->generation = h->generation;

    GsrCam k = make_cam(h, cam);
    uint32_t* totals = h->totals.as<uint32_t>();
    // Binning (SURVEY.md A.5-A.7 restated per tile).  FAST mode: every tile owns a fixed-capacity key bin and preprocess drops
    // the keys straight into them — no second pass over the instances; (T+1) x capacity x 8 B, so only while that stays within
    // the bins budget.  Otherwise — a scene with a few very deep tiles, or a view that overflowed small bins — the COMPACT mode
    // runs: preprocess only counts, the scan turns the counts into offsets, a scatter pass puts the keys at exact offsets (8 B
    // per instance whatever the skew).  WHICH of them, with what capacity, in which binning form: gsr_policy_begin_view.
    h->pcfg.preprocess_form = preprocess_form_of(h);  // the handle's pin, else the process default as of this call
    float timed_ms[GSR_TUNER_TIMED_VIEWS];
    const bool have_ms = h->pcfg.form_tuner && h->pol.tuner.phase == GSR_TUNER_TIMED_VIEWS && h->tuner_ev.read(timed_ms);
    gsr_view_plan plan;
    gsr_policy_begin_view(&h->pcfg, &h->pol, n, have_ms ? timed_ms : nullptr, &plan);
    const bool use_bins = plan.bin_cap_view > 0;
    const uint32_t bin_cap_view = plan.bin_cap_view;
    if (use_bins && (rc = h->bins.ensure((T + 1) * (size_t)bin_cap_view * 8))) return rc;
    // The tile counters are zero on entry: gsr_create clears them and the tile sort re-zeroes each
    // tile's counter as it consumes it (no memset kernel per view; the scan overwrites every total).
    // Only a pass that did not reach the sort (an error) leaves them dirty.
    if (h->tile_count_dirty) HIPCHK(hipMemsetAsync(h->tile_count.p, 0, (T + 2) * 4, s));
    h->tile_count_dirty = true;
    int timed = plan.timed_slot;
    if (timed >= 0 && !h->tuner_ev.create()) {  // no clock, no tuner on this handle (the plan's form is still a valid form)
        h->pcfg.form_tuner = 0;
        h->pol.tuner.phase = 0;
        timed = -1;
    }
    StageScope sc1(h->prof, ST_PREPROCESS, s);
    if (timed >= 0) HIPCHK(hipEventRecord(h->tuner_ev.ev[2 * timed], s));
    h->last_form = gsr_launch_preprocess(s, n, in->n_coeffs, in->sh_degree, C, in->means, in->scales, in->rotations,
                                         in->opacities, in->shs, k, geom_of(h), h->tile_count.as<uint32_t>(),
                                         h->bvis.as<uint32_t>(), h->bins.as<uint64_t>(), bin_cap_view /* 0: count only */,
                                         h->n_tiles, /*aggregating=*/plan.form != GSR_FORM_DIRECT);
    if (timed >= 0) HIPCHK(hipEventRecord(h->tuner_ev.ev[2 * timed + 1], s));
    sc1.close();
    const uint32_t seq = ++h->totals_seq ? h->totals_seq : ++h->totals_seq;  // never 0
    StageScope sc2(h->prof, ST_SCAN, s);
    gsr_launch_tile_scan(s, h->n_tiles, h->tile_count.as<uint32_t>(), h->tile_start.as<uint32_t>(), totals,
                         n_blocks, h->bsum.as<uint32_t>(), h->bpre.as<uint32_t>(), h->bvis.as<uint32_t>(),
                         h->big_list.as<uint32_t>(), h->host_totals_dev, seq, h->tile_order.as<uint32_t>());
    sc2.close();
    HIPCHK(hipGetLastError());
    // the one host sync of the path: instance count D (reference: rasterizer.jl:337).  tile_scan stores the totals
    // and then this forward's sequence number into pinned host memory; no copy packet, no event on the stream.
    // Sort + forward of every tile of up to 1024 instances (nearly all of them) go out BEHIND the scan, as ONE launch,
    // without waiting for the host: the buffers have a capacity from earlier views (grow-only, 25 % slack), and the
    // kernel itself checks the scan's totals against it.  The host's wait below then overlaps that launch instead of
    // idling the GPU, and inside it the HBM-bound sort of one tile shares the CU with the VALU-bound compositing of others.
    uint64_t cap_instances = std::min(std::min(h->values_sorted.cap / 4, h->s0.cap / 16), std::min(h->s1.cap / 16, h->s2.cap / 16));
    if (C > 3) cap_instances = std::min<uint64_t>(cap_instances, h->s3.cap / 16);
    cap_instances = std::min<uint64_t>(cap_instances, 0xFFFFFFFFull);
    if (fwd_only) cap_instances = 0xFFFFFFFFull;  // nothing is stored per instance: no capacity to respect
    static const bool no_fused = [] { const char* e = getenv("GSR_NO_FUSED_FWD"); return e && e[0] == '1'; }();  // A/B only
    const bool spec = use_bins && cap_instances > 0 && !no_fused;
    // TIER TILES (lists beyond the fused launch's 1024 instances) are walked by their own launch, four single-wave workgroups per
    // tile, and that walk is a latency chain: 0.4 ms for one 32 k-instance tile and its neighbours, 0.15 ms for the few hundred
    // tiles of 1-4 k instances of a trained-like scene — it belongs BESIDE the fused launch, not behind it.  What does not work,
    // measured (tools/cu_mask_probe.hip, profiles/r05/experiments/long_list_chain.txt): queueing sorts + walk on a second stream
    // while the fused launch runs — it fills every wave slot and all LDS, and the sorts' workgroups (512-1024 threads, 64 KB)
    // never find four of its workgroups retiring on one CU together: the chain starts at the fused launch's tail, on any
    // stream, at any priority; CU-masked streams do give the chain CUs of its own (probe: done after 2.7 of 7.9 ms instead of
    // 9.0 of 9.0), but hipExtStreamCreateWithCUMask only makes streams that synchronise with the NULL stream — torch's default
    // — and every launch of the step then pays ~10 us of implicit cross-queue wait (+0.16 ms per step: more than it saved).
    // What does: the walk's workgroups are single waves, which slip into any retiring slot.  So when the previous view had tier
    // tiles the fused launch is HELD until the host has the counts and the tier sorts have run (mostly idle GPU: 0.04-0.2 ms),
    // and then goes out together with the walk, which takes the handle's second stream at raised wave priority.  Hot tile
    // (32 k) 2.08 -> 1.80 ms, trained-like 3 M / 1440p 2.19 -> 2.10, dense 4K 7.81 -> 7.70; a view without tier tiles is not
    // touched.  (GSR_TIERS_BESIDE_MAX: hold only when the previous view had at most that many tier tiles — A/B runs; 0 = never:
    // gsr_policy_config.beside_max_tiles.)
    const bool hold_fused = spec && plan.hold_fused;
    const auto launch_fused = [&](hipStream_t fs) {
        StageScope sc3(h->prof, ST_SORT_COMPOSITE_FWD, fs);
        gsr_launch_sort_composite_fwd(fs, C, k, h->tile_start.as<uint32_t>(), h->tile_order.as<uint32_t>(),
                                      h->tile_count.as<uint32_t>(), h->bins.as<uint64_t>(), bin_cap_view, geom_of(h),
                                      stream_of(h), in->background, image_out, h->n_contrib.as<uint32_t>(),
                                      h->final_T.as<float>(), h->values_sorted.as<uint32_t>(), h->ranges.as<uint32_t>(),
                                      aux ? aux->covisibilities : nullptr, aux ? aux->uncertainties : nullptr, totals,
                                      (uint32_t)cap_instances, /*keep_backward_state=*/!fwd_only);
        sc3.close();
    };
    if (spec && !hold_fused) launch_fused(s);
    // A held fused launch waits for the tier sorts, and those for the host's read-back: 22-34 us of idle GPU per view (kernel
    // trace of the training protocol, profiles/r06/experiments/training_step_idle_time.txt).  The two mid tiers' sorts therefore
    // go out NOW, behind the scan, with grids guessed from the previous view and the scan's totals checked on the device
    // (gsr_launch_tile_sort_mid); what the guess missed — and every list beyond 8192, whose chain needs host-sized scratch — is
    // sorted after the read-back.  Not for forward-only renders (whose stream buffers are not sized by cap_instances).
    // GSR_SPEC_TIER_SORTS=0: A/B.
    static const bool spec_tiers_on = [] { const char* e = getenv("GSR_SPEC_TIER_SORTS"); return !(e && e[0] == '0'); }();
    // (the guesses are the policy's: gsr_view_plan.spec_mid4 / spec_mid8; what is not launched is zeroed for gsr_policy_end_view)
    if (!(hold_fused && spec_tiers_on && !fwd_only)) plan.spec_mid4 = plan.spec_mid8 = 0u;
    const uint32_t spec4 = plan.spec_mid4, spec8 = plan.spec_mid8;
    {
        if (spec4 | spec8) {
            StageScope scs(h->prof, ST_SORT, s, /*extra=*/true);  // (the stage's time; the call after the read-back counts the launch)
            gsr_launch_tile_sort_mid(s, h->n_tiles, h->grid_x, C, h->tile_start.as<uint32_t>(), h->bins.as<uint64_t>(), bin_cap_view,
                                     spec4, spec8, h->big_list.as<uint32_t>(), geom_of(h), stream_of(h),
                                     h->values_sorted.as<uint32_t>(), totals, (uint32_t)cap_instances);
            scs.close();
        }
    }
    if ((rc = wait_totals(h, seq, s))) return rc;
    // What this view turned out to be — bins / overflow tiles / compact; whether the fused launch covers it; where the tier
    // walk goes; the bins' capacity for the next view — is gsr_policy_end_view's decision on the scan's counts.
    // OVERFLOW TILES (round 5): lists longer than the bins' capacity.  Their bins hold the first bin_cap_view arrivals only
    // (preprocess counted every instance); their complete key lists come from a scatter pass restricted to them (below), and
    // every sort takes a tile's keys from wherever they are complete.  The rest of the view stays on the fast path.
    const uint64_t D = h->host_totals[0];
    const uint32_t max_tile = h->host_totals[1], n_big = h->host_totals[2];
    const uint32_t n_mid4 = h->host_totals[3], n_mid8 = h->host_totals[6];
    const uint64_t D_slots = h->host_totals[5];  // >= D; == D unless exact culling dropped tiles
    gsr_view_outcome oc;
    gsr_policy_end_view(&h->pcfg, &h->pol, &plan, (int64_t)D, max_tile, n_mid4, n_mid8, n_big, cap_instances, spec ? 1 : 0, &oc);
    const bool hybrid = oc.binning == GSR_BINNING_OVERFLOW, compact = oc.binning == GSR_BINNING_COMPACT;
    const bool fused_done = oc.fused_done != 0, long_tiles = oc.long_tiles != 0, beside = oc.beside != 0;
    h->last_n = n;
    h->last_D = (int64_t)D;
    h->last_slots = (int64_t)D_slots;
    h->last_compact = compact;
    if (stats) {
        stats->n_rendered = (int64_t)D;
        stats->n_visible = (int32_t)(h->host_totals[4] & 0x7FFFFFFFu);
        stats->max_tile_instances = (int32_t)max_tile;
        stats->compact_binning = oc.binning;
        stats->preprocess_form = h->last_form;
        stats->bins_bytes = (int64_t)(compact ? D * 8 : (uint64_t)(T + 1) * bin_cap_view * 8ull + (hybrid ? D * 8 : 0));
        stats->bin_capacity = bin_cap_view;
        stats->tier_tiles[0] = n_mid4; stats->tier_tiles[1] = n_mid8; stats->tier_tiles[2] = n_big;
        stats->reserved = 0;
        fill_history(h, stats);  // (again at the end: the buffers this view grows are counted there)
    }
    if (D == 0) {
        h->tile_count_dirty = false;  // every counter is zero
        if (k.exact_cull && (h->host_totals[4] >> 31)) {
            // exact-cull mode dropped every instance (all of them invisible: opacities below 1/255) of a view the reference WOULD
            // have rendered (some rect holds a tile): its pixels blend nothing and show the background, as with the reference's
            // lists — the all-zero image below is the reference's answer to "no instance at all" only
            gsr_launch_fill_background(s, P, C, in->background, image_out, h->final_T.as<float>(), h->n_contrib.as<uint32_t>());
        } else {
            // rasterizer.jl:283,338: all-zero image, background not applied
            HIPCHK(hipMemsetAsync(image_out, 0, P * C * 4, s));
            HIPCHK(hipMemsetAsync(h->n_contrib.p, 0, P * 4, s));
            HIPCHK(hipMemsetAsync(h->final_T.p, 0, P * 4, s));
        }
        HIPCHK(hipMemsetAsync(h->ranges.p, 0, 2 * T * 4, s));
        if (aux && aux->uncertainties) HIPCHK(hipMemsetAsync(aux->uncertainties, 0, P * 4, s));
        h->fwd_valid = true;
        return GSR_OK;
    }
    const float slack = 1.25f;  // instance count drifts slowly between training steps
    // forward-only and everything composited by the fused launch: no per-instance storage at all (the rare paths below —
    // tier lists, compact binning — still hand their instances over through the stream)
    const bool need_stream = !(fwd_only && fused_done && !long_tiles);
    if (need_stream &&
        ((rc = h->values_sorted.ensure(D * 4, slack)) ||
         (rc = h->s0.ensure(D * 16, slack)) || (rc = h->s1.ensure(D * 16, slack)) ||
         (rc = h->s2.ensure(D * 16, slack)) || (C > 3 && (rc = h->s3.ensure(D * 16, slack)))))
        return rc;
    if (!fwd_only && (rc = h->rows.ensure(D_slots * 64, slack))) return rc;
    size_t slab_stride = 0;
    if (n_big > 0) {  // lists beyond the LDS sort: two merge slabs per listed tile
        slab_stride = ((size_t)max_tile + 63) & ~(size_t)63;
        // (+ the plan of the multi-workgroup sort: 2 (n_big + 1) words behind the slabs)
        if ((rc = h->big_scratch.ensure((size_t)n_big * 2 * slab_stride * 8 + (size_t)(2 * n_big + 2) * 4 + 64))) return rc;
    }
    // (a held fused launch that this view gives no reason to hold any longer — no tier tiles after all, or too many — goes out
    // now; one that would only find its buffers too small — the kernel checks the same totals — is not launched at all: the
    // main sort pass then re-zeroes the counters and writes the ranges, as in every view without the fused launch)
    if (oc.launch_fused_now) launch_fused(s);
    // what the speculative mid-tier sorts covered (gsr_policy_end_view: the kernels' own test, on the same numbers)
    const uint32_t done4 = oc.sorted_mid4, done8 = oc.sorted_mid8;
    if (!fused_done || long_tiles) {
        StageScope sc4(h->prof, ST_SORT, s);
        const uint64_t* keys = h->bins.as<uint64_t>();
        const uint64_t* overflow_keys = nullptr;
        uint32_t key_cap = bin_cap_view;
        if (compact) {
            // count -> scan -> scatter: the counters become the fill cursors of the scatter pass
            if ((rc = h->keys_compact.ensure(D * 8, slack))) return rc;
            HIPCHK(hipMemsetAsync(h->tile_count.p, 0, (T + 2) * 4, s));
            gsr_launch_emit_compact(s, n, k, geom_of(h), h->tile_start.as<uint32_t>(), h->tile_count.as<uint32_t>(),
                                    h->keys_compact.as<uint64_t>(), max_tile, /*only_above=*/0u);
            keys = h->keys_compact.as<uint64_t>();
            key_cap = 0;
        } else if (hybrid) {
            // the scatter pass restricted to the lists beyond the bins' capacity: their keys go to keys_compact at the offsets of
            // the compact layout (the buffer is sized as for it; only those segments are touched), with fill cursors of its own
            // (the tile counters belong to the fused launch, which re-zeroes them)
            if ((rc = h->keys_compact.ensure(D * 8, slack)) || (rc = h->overflow_fill.ensure((T + 2) * 4))) return rc;
            HIPCHK(hipMemsetAsync(h->overflow_fill.p, 0, (T + 2) * 4, s));
            gsr_launch_emit_compact(s, n, k, geom_of(h), h->tile_start.as<uint32_t>(), h->overflow_fill.as<uint32_t>(),
                                    h->keys_compact.as<uint64_t>(), max_tile, /*only_above=*/bin_cap_view);
            overflow_keys = h->keys_compact.as<uint64_t>();
        }
        gsr_launch_tile_sort(s, (fused_done ? 0 : GSR_SORT_PASS_MAIN) | GSR_SORT_PASS_TIERS, h->n_tiles, h->grid_x, C,
                             h->tile_start.as<uint32_t>(), h->tile_count.as<uint32_t>(), keys, key_cap, overflow_keys,
                             n_mid4, n_mid8, n_big, h->big_list.as<uint32_t>(), h->big_scratch.as<uint64_t>(),
                             slab_stride, geom_of(h), stream_of(h), h->values_sorted.as<uint32_t>(), h->ranges.as<uint32_t>(),
                             nullptr, 0, done4, done8);
        sc4.close();
        // the walk of the tier tiles: beside the held fused launch (second stream, first in the queue), else behind it
        const hipStream_t ws = beside ? h->aux_stream : s;
        if (beside) {
            HIPCHK(hipEventRecord(h->ev_fork, s));
            HIPCHK(hipStreamWaitEvent(h->aux_stream, h->ev_fork, 0));
        }
        StageScope sc5(h->prof, ST_COMPOSITE_FWD, ws);
        // after the fused launch only the tiles of the tier lists are left; otherwise every tile
        static const bool walk_prio = [] { const char* e = getenv("GSR_WALK_PRIO"); return !(e && e[0] == '0'); }();  // A/B only
        const GsrTierLists tiers{h->big_list.as<uint32_t>(), (uint32_t)h->n_tiles, n_big, n_mid8, n_mid4, beside && walk_prio ? 1u : 0u};
        gsr_launch_composite_fwd(ws, C, k, h->tile_start.as<uint32_t>(), fused_done ? nullptr : h->tile_order.as<uint32_t>(),
                                 stream_of(h), in->background, image_out, h->n_contrib.as<uint32_t>(), h->final_T.as<float>(),
                                 h->values_sorted.as<uint32_t>(), aux ? aux->covisibilities : nullptr,
                                 aux ? aux->uncertainties : nullptr, &tiers);
        sc5.close();
        if (beside) {
            HIPCHK(hipEventRecord(h->ev_join, h->aux_stream));
            launch_fused(s);
            HIPCHK(hipStreamWaitEvent(s, h->ev_join, 0));
        }
    }
    h->tile_count_dirty = false;  // the sort (fused or not) zeroed the counters
    HIPCHK(hipGetLastError());
    if (stats) fill_history(h, stats);
    h->fwd_valid = true;
    return GSR_OK;
}

// ∇render! of the last forward.  One wave walks a tile's list in the main launch; a list of tens of thousands of
// instances is then milliseconds of ONE wave (§8 of DESIGN.md).  As long as such tiles are too few to fill the GPU by
// themselves (at most one per CU), they are taken out of the main launch and walked by four waves each on the
// handle's second stream, forked from and joined back into the caller's stream.  The cut is a tier boundary of the
// scan (> 8192, > 4096 or > 1024 instances): the deepest tiers first, as many as fit the limit.  Four waves cost
// ~20 % more work per instance, so when long tiles are plentiful they stay with the main launch.
static int launch_composite_bwd(gsr_handle* h, hipStream_t s, int C, const GsrCam& k, const float* background,
                                const float* vpixels, bool color_only) {
    // (A/B knob: GSR_BWD_COLOR_ONLY=1 treats EVERY cotangent as the loss head's — valid only where it is)
    static const bool force_color_only = [] { const char* e = getenv("GSR_BWD_COLOR_ONLY"); return e && e[0] == '1'; }();
    color_only = (color_only || force_color_only) && C > 3;
    // which tiers leave the main launch: gsr_policy_bwd_split (GSR_BWD_SPLIT_TILES overrides its limit for A/B runs)
    gsr_bwd_split sp;
    gsr_policy_bwd_split(&h->pcfg, h->pol.tier_n[0], h->pol.tier_n[1], h->pol.tier_n[2], &sp);
    // (the accurate per-pixel arithmetic exists for the one-wave-per-tile kernel only: such a handle splits nothing)
    const bool accurate = h->cfg.grad_precision != GSR_DEFAULT;
    if (accurate) { sp.n_big = sp.n_mid8 = sp.n_mid4 = 0; sp.split_len = 0xFFFFFFFFu; }
    GsrTierLists tiers{h->big_list.as<uint32_t>(), (uint32_t)h->n_tiles, sp.n_big, sp.n_mid8, sp.n_mid4, sp.split_len};
    const uint32_t n = sp.n_big + sp.n_mid8 + sp.n_mid4;
    if (n > 0) {
        // per (listed tile, list segment, pixel): the (m, c) of the segment — the first pass's hand-over to the second
        int rc = h->long_state.ensure((size_t)n * GSR_BWD_LONG_SEGS * 256 * 2 * sizeof(float));
        if (rc) return rc;
        HIPCHK(hipEventRecord(h->ev_fork, s));
        HIPCHK(hipStreamWaitEvent(h->aux_stream, h->ev_fork, 0));
        gsr_launch_composite_bwd_listed(h->aux_stream, C, k, h->tile_start.as<uint32_t>(), tiers, stream_of(h), background,
                                        vpixels, h->n_contrib.as<uint32_t>(), h->final_T.as<float>(), inst_of(h),
                                        h->long_state.as<float>());
        HIPCHK(hipEventRecord(h->ev_join, h->aux_stream));
    }
    gsr_launch_composite_bwd(s, C, k, h->tile_start.as<uint32_t>(), h->tile_order.as<uint32_t>(), stream_of(h), background,
                             vpixels, h->n_contrib.as<uint32_t>(), h->final_T.as<float>(), inst_of(h), tiers.split_len, color_only, accurate);
    if (n > 0) HIPCHK(hipStreamWaitEvent(s, h->ev_join, 0));
    return GSR_OK;
}

int gsr_backward(gsr_handle* h, const gsr_inputs* in, const gsr_camera* cam, const float* vpixels,
                 const gsr_grads* g, void* stream_v) {
    int rc = check_inputs(h, in, cam);
    if (rc) return rc;
    if (!vpixels || !g) return fail(GSR_E_INVALID_ARG, "null vpixels / grads");
    if (!h->fwd_valid || h->last_n != in->n)
        return fail(GSR_E_STATE, "gsr_backward without a matching gsr_forward on this handle");
    if (h->fwd_only)
        return fail(GSR_E_STATE, "the handle's last forward was GSR_FORWARD_ONLY: it kept no backward state");
    if (h->inputs_consumed)
        return fail(GSR_E_STATE, "the inputs of the handle's last forward were updated in place by "
                    "gsr_backward_trainer_tail; run gsr_forward again");
    if ((g->flags & ~GSR_GRADS_COLOR_COTANGENT) != 0u || g->reserved != 0u) return fail(GSR_E_INVALID_ARG, "unknown gsr_grads.flags / reserved bits");
    if (g->forward_generation != 0 && g->forward_generation != h->generation)
        return fail(GSR_E_STATE, "gsr_backward for forward #%llu, but the handle's last forward is #%llu (another "
                    "gsr_forward ran in between)", (unsigned long long)g->forward_generation,
                    (unsigned long long)h->generation);
    if (in->n > 0 && (!g->vmeans || (!g->vshs && !g->vcolors) || !g->vopacities || !g->vscales || !g->vrotations))
        return fail(GSR_E_INVALID_ARG, "null gradient buffer");
    if ((g->vR == nullptr) != (g->vt == nullptr)) return fail(GSR_E_INVALID_ARG, "vR and vt must be given together");
    if (((uintptr_t)g->vrotations & 15) != 0) return fail(GSR_E_INVALID_ARG, "vrotations must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream_v;
    const int C = h->cfg.mode, n = in->n;
    if (n == 0) return GSR_OK;
    if (!g->vmeans2d && (rc = h->vmean2d.ensure((size_t)n * 8, 1.25f))) return rc;
    h->vmean2d_cur = g->vmeans2d ? reinterpret_cast<float2*>(g->vmeans2d) : h->vmean2d.as<float2>();
    // (the gradient rows need no memset: composite_bwd writes the row of every emitted instance,
    // pergauss_bwd skips the slots of culled tiles; only the 12 pose-gradient floats are accumulated into)
    if (g->vR) {
        StageScope sc6(h->prof, ST_ZERO_ACC, s);
        HIPCHK(hipMemsetAsync(g->vR, 0, 9 * 4, s));
        HIPCHK(hipMemsetAsync(g->vt, 0, 3 * 4, s));
        sc6.close();
    }
    const bool color_only = (g->flags & GSR_GRADS_COLOR_COTANGENT) != 0u;
    if (color_only && (rc = check_color_cotangent(h, vpixels, s))) return rc;
    GsrCam k = make_cam(h, cam);
    StageScope sc7(h->prof, ST_COMPOSITE_BWD, s);
    if (h->last_D > 0 && (rc = launch_composite_bwd(h, s, C, k, in->background, vpixels, color_only))) return rc;
    sc7.close();
    StageScope sc8(h->prof, ST_PERGAUSS_BWD, s);
    gsr_launch_pergauss_bwd(s, n, in->n_coeffs, in->sh_degree, C, in->means, in->scales, in->rotations, in->shs, k,
                            geom_of(h), inst_of(h), h->vmean2d_cur, g->vmeans, g->vshs, g->vopacities,
                            g->vscales, g->vrotations, g->vR, g->vt, g->vcolors,
                            /*fp32_chain=*/h->cfg.grad_precision == GSR_GRAD_FP32_REFERENCE);
    sc8.close();
    HIPCHK(hipGetLastError());
    h->bwd_valid = true;
    return GSR_OK;
}

static int buffer_lookup(const gsr_handle* h, int which, const void** dev_ptr, size_t* bytes) {
    const size_t n = (size_t)h->last_n, P = (size_t)h->cfg.width * h->cfg.height, T = (size_t)h->n_tiles;
    const size_t D = (size_t)h->last_D;
    const DevBuf* b = nullptr;
    size_t sz = 0;
    switch (which) {
        case GSR_BUF_RADII:  // handle-owned, or the caller's gsr_aux.radii of the last forward
            *dev_ptr = h->fwd_valid && n ? h->radii_cur : nullptr;
            *bytes = *dev_ptr ? n * 4 : 0;
            return GSR_OK;
        case GSR_BUF_GRAD_MEANS2D:
            *dev_ptr = h->bwd_valid && n ? h->vmean2d_cur : nullptr;
            *bytes = *dev_ptr ? n * 8 : 0;
            return GSR_OK;
        case GSR_BUF_N_CONTRIB: b = &h->n_contrib; sz = P * 4; break;
        case GSR_BUF_FINAL_T: b = &h->final_T; sz = P * 4; break;
        case GSR_BUF_TILE_RANGES: b = &h->ranges; sz = 2 * T * 4; break;
        case GSR_BUF_VALUES_SORTED: b = &h->values_sorted; sz = h->fwd_only ? 0 : D * 4; break;
        case GSR_BUF_GEOM: b = &h->geo; sz = n * 64; break;
        case GSR_BUF_NORMALS: b = &h->gnormal; sz = h->cfg.mode > 5 ? n * 16 : 0; break;
        case GSR_BUF_GRAD_ROWS: b = &h->rows; sz = (size_t)h->last_slots * 16 * GSR_ROW_F4(h->cfg.mode); break;
        case GSR_BUF_INSTANCE_AUX: b = &h->s2; sz = h->fwd_only ? 0 : D * 16; break;
        default: return fail(GSR_E_INVALID_ARG, "unknown buffer id %d", which);
    }
    if (sz > b->cap) sz = 0;  // not produced yet
    *dev_ptr = sz ? b->p : nullptr;
    *bytes = sz;
    return GSR_OK;
}

int gsr_buffer(const gsr_handle* h, int which, const void** dev_ptr, size_t* bytes) {
    if (!h || !dev_ptr || !bytes) return fail(GSR_E_INVALID_ARG, "null argument");
    return buffer_lookup(h, which, dev_ptr, bytes);
}

int gsr_copy_buffer(const gsr_handle* h, int which, void* dst, size_t bytes, void* stream) {
    if (!h || (!dst && bytes)) return fail(GSR_E_INVALID_ARG, "null argument");
    const void* src = nullptr;
    size_t have = 0;
    int rc = buffer_lookup(h, which, &src, &have);
    if (rc) return rc;
    if (bytes > have) return fail(GSR_E_STATE, "buffer %d holds %zu bytes, %zu requested (not produced yet?)", which, have, bytes);
    if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return GSR_OK;
}

int gsr_ssim_forward(int W, int H, int CH, int B, const float* img, const float* ref, float C1, float C2, int train,
                     float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream) {
    if (W <= 0 || H <= 0 || CH <= 0 || B <= 0) return fail(GSR_E_INVALID_ARG, "bad SSIM shape");
    if (!img || !ref || !ssim_map) return fail(GSR_E_INVALID_ARG, "null SSIM array");
    if (train && (!dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12)) return fail(GSR_E_INVALID_ARG, "train needs the 3 partial maps");
    if ((size_t)CH * B > 65535) return fail(GSR_E_INVALID_ARG, "CH*B too large");
    (g_ssim_exact.load(std::memory_order_relaxed) ? gsr_launch_ssim_fwd_exact : gsr_launch_ssim_fwd_fast)((hipStream_t)stream, W, H, CH, B, img, ref, C1, C2, train,
                                                                          ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_ssim_backward(int W, int H, int CH, int B, const float* img, const float* ref, const float* dL_dmap,
                      const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg,
                      void* stream) {
    if (W <= 0 || H <= 0 || CH <= 0 || B <= 0) return fail(GSR_E_INVALID_ARG, "bad SSIM shape");
    if (!img || !ref || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg)
        return fail(GSR_E_INVALID_ARG, "null SSIM array");
    if ((size_t)CH * B > 65535) return fail(GSR_E_INVALID_ARG, "CH*B too large");
    (g_ssim_exact.load(std::memory_order_relaxed) ? gsr_launch_ssim_bwd_exact : gsr_launch_ssim_bwd_fast)((hipStream_t)stream, W, H, CH, B, img, ref, dL_dmap, dm_dmu1,
                                                                          dm_dsigma1_sq, dm_dsigma12, dL_dimg);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_loss_l1_ssim(gsr_handle* h, const float* image, const float* target, float lambda_dssim, float* loss_out,
                     float* vpixels, void* stream) {
    if (!h || !image || !target || !loss_out || !vpixels) return fail(GSR_E_INVALID_ARG, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int W = h->cfg.width, H = h->cfg.height, C = h->cfg.mode;
    const size_t P = (size_t)W * H;
    int rc;
    if ((rc = h->d0.ensure(3 * P * 4)) || (rc = h->d1.ensure(3 * P * 4)) || (rc = h->d2.ensure(3 * P * 4)) ||
        (rc = h->partial.ensure((size_t)h->n_tiles * 3 * 2 * 4)))
        return rc;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;  // fused_ssim.jl:374
    // ONE decision for the forward and the pullback of this call, from the handle (ABI 5) or the process default
    const int exact = ssim_exact_of(h);
    StageScope sc9(h->prof, ST_LOSS_FWD, s);
    (exact ? gsr_launch_loss_fwd_exact : gsr_launch_loss_fwd_fast)(s, W, H, C, image, target, C1, C2, h->d0.as<float>(),
                                                                          h->d1.as<float>(), h->d2.as<float>(), h->partial.as<float>());
    sc9.close();
    StageScope sc10(h->prof, ST_LOSS_BWD, s);
    (exact ? gsr_launch_loss_bwd_exact : gsr_launch_loss_bwd_fast)(s, W, H, C, image, target, lambda_dssim, h->d0.as<float>(),
                                                                          h->d1.as<float>(), h->d2.as<float>(), h->partial.as<float>(),
                                                                          loss_out, vpixels);
    sc10.close();
    HIPCHK(hipGetLastError());
    // the cotangent of this forward whose channels >= 3 are zeros by construction (GSR_GRADS_COLOR_COTANGENT)
    h->loss_vpixels = vpixels;
    h->loss_generation = h->generation;
    return GSR_OK;
}

int gsr_update_stats(gsr_handle* h, int32_t* max_radii, float* accum_grad_means2d, float* denom, void* stream) {
    if (!h || !max_radii || !accum_grad_means2d || !denom) return fail(GSR_E_INVALID_ARG, "null argument");
    if (!h->fwd_valid || !h->bwd_valid) return fail(GSR_E_STATE, "gsr_update_stats needs a completed forward/backward pair");
    gsr_launch_update_stats((hipStream_t)stream, h->last_n, h->radii_cur, h->vmean2d_cur,
                            h->cfg.width, h->cfg.height, max_radii, accum_grad_means2d, denom);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_profile_enable(gsr_handle* h, int on) {
    if (!h) return fail(GSR_E_INVALID_ARG, "null handle");
    h->prof.on = (on & 1) != 0;      // bit 0: HIP-event stage timing
    h->prof.ranges = (on & 2) != 0;  // bit 1: roctx range per stage
    if (!h->prof.on) h->prof.clear();
    return GSR_OK;
}

int gsr_profile_stages(gsr_handle* h, uint32_t stage_mask) {
    if (!h) return fail(GSR_E_INVALID_ARG, "null handle");
    h->prof.mask = stage_mask;
    return GSR_OK;
}

int gsr_profile_stage_count(void) { return ST_COUNT; }
const char* gsr_profile_stage_name(int stage) { return stage >= 0 && stage < ST_COUNT ? kStageNames[stage] : ""; }

int gsr_profile_read(gsr_handle* h, double* ms_sum, int* launches, int reset) {
    if (!h || !ms_sum || !launches) return fail(GSR_E_INVALID_ARG, "null argument");
    for (int i = 0; i < ST_COUNT; i++) { ms_sum[i] = 0.0; launches[i] = 0; }
    for (auto& r : h->prof.recs) {
        float ms = 0.0f;
        HIPCHK(hipEventSynchronize(r.b));
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        ms_sum[r.stage] += ms;
        if (!r.extra) launches[r.stage] += 1;
    }
    if (reset) h->prof.clear();
    return GSR_OK;
}

int gsr_profile_read_intervals(gsr_handle* h, int stage, double* ms_out, int max_n, int* n_out) {
    if (!h || !n_out || (max_n > 0 && !ms_out)) return fail(GSR_E_INVALID_ARG, "null argument");
    if (stage < 0 || stage >= ST_COUNT) return fail(GSR_E_INVALID_ARG, "unknown stage %d", stage);
    int n = 0;
    hipEvent_t prev = nullptr;
    for (auto& r : h->prof.recs) {
        if (r.stage != stage || r.extra) continue;
        if (prev) {
            if (n < max_n) {
                float ms = 0.0f;
                HIPCHK(hipEventSynchronize(r.a));
                HIPCHK(hipEventElapsedTime(&ms, prev, r.a));
                ms_out[n] = ms;
            }
            n++;
        }
        prev = r.a;
    }
    *n_out = n;
    return GSR_OK;
}

int gsr_prologue_forward(int32_t n, int32_t k_rest, int32_t scale_dims, const float* sh_color,
                         const float* sh_remainder, const float* opacities, const float* scales, float* shs,
                         float* opacities_act, float* scales_act, void* stream) {
    if (n < 0 || k_rest < 0 || (scale_dims != 1 && scale_dims != 3))
        return fail(GSR_E_INVALID_ARG, "bad sizes: n=%d k_rest=%d scale_dims=%d", n, k_rest, scale_dims);
    if (n == 0) return GSR_OK;
    if (!sh_color || (k_rest > 0 && !sh_remainder) || !opacities || !scales || !shs || !opacities_act || !scales_act)
        return fail(GSR_E_INVALID_ARG, "null array");
    gsr_launch_prologue_fwd((hipStream_t)stream, n, k_rest, scale_dims, sh_color, sh_remainder, opacities, scales, shs,
                            opacities_act, scales_act);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_prologue_backward(int32_t n, int32_t k_rest, int32_t scale_dims, const float* opacities_act,
                          const float* scales_act, const float* vshs, const float* vopacities_act,
                          const float* vscales_act, float* v_sh_color, float* v_sh_remainder, float* v_opacities,
                          float* v_scales, void* stream) {
    if (n < 0 || k_rest < 0 || (scale_dims != 1 && scale_dims != 3))
        return fail(GSR_E_INVALID_ARG, "bad sizes: n=%d k_rest=%d scale_dims=%d", n, k_rest, scale_dims);
    if (n == 0) return GSR_OK;
    if (!opacities_act || !scales_act || !vshs || !vopacities_act || !vscales_act || !v_sh_color ||
        (k_rest > 0 && !v_sh_remainder) || !v_opacities || !v_scales)
        return fail(GSR_E_INVALID_ARG, "null array");
    gsr_launch_prologue_bwd((hipStream_t)stream, n, k_rest, scale_dims, opacities_act, scales_act, vshs, vopacities_act,
                            vscales_act, v_sh_color, v_sh_remainder, v_opacities, v_scales);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_adam_step(const gsr_adam_group* groups, int32_t n_groups, float beta1, float beta2, float eps, void* stream) {
    if (n_groups < 0 || n_groups > GSR_ADAM_MAX_GROUPS || (n_groups > 0 && !groups))
        return fail(GSR_E_INVALID_ARG, "n_groups must be in [0, %d]", GSR_ADAM_MAX_GROUPS);
    float* theta[GSR_ADAM_MAX_GROUPS]; const float* grad[GSR_ADAM_MAX_GROUPS];
    float* mu[GSR_ADAM_MAX_GROUPS]; float* nu[GSR_ADAM_MAX_GROUPS];
    long long count[GSR_ADAM_MAX_GROUPS]; float lr_t[GSR_ADAM_MAX_GROUPS];
    int m = 0;
    for (int g = 0; g < n_groups; g++) {
        const gsr_adam_group& a = groups[g];
        if (a.count < 0) return fail(GSR_E_INVALID_ARG, "group %d: negative count", g);
        if (a.count == 0) continue;  // training.jl:770 `isempty(θᵢ) && continue`
        if (!a.theta || !a.grad || !a.mu || !a.nu) return fail(GSR_E_INVALID_ARG, "group %d: null array", g);
        if (a.current_step == 0) return fail(GSR_E_INVALID_ARG, "group %d: current_step counts from 1", g);
        const float t = (float)a.current_step;
        theta[m] = a.theta; grad[m] = a.grad; mu[m] = a.mu; nu[m] = a.nu; count[m] = a.count;
        lr_t[m] = a.lr * sqrtf(1.0f - powf(beta2, t)) / (1.0f - powf(beta1, t));  // debiasing, Kingma & Ba §2
        m++;
    }
    if (m == 0) return GSR_OK;
    gsr_launch_adam((hipStream_t)stream, m, theta, grad, mu, nu, count, lr_t, beta1, beta2, eps);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_trainer_tail_step(int32_t n, int32_t k_rest, int32_t scale_dims, const gsr_tail_grads* grads,
                          float* const theta[6], float* const mu[6], float* const nu[6], const float lr[6],
                          const uint32_t current_step[6], float beta1, float beta2, float eps, float* shs,
                          float* opacities_act, float* scales_act, void* stream) {
    if (n < 0 || k_rest < 0 || (scale_dims != 1 && scale_dims != 3))
        return fail(GSR_E_INVALID_ARG, "bad sizes: n=%d k_rest=%d scale_dims=%d", n, k_rest, scale_dims);
    if (n == 0) return GSR_OK;
    if (!grads || !theta || !mu || !nu || !lr || !current_step || !shs || !opacities_act || !scales_act)
        return fail(GSR_E_INVALID_ARG, "null argument");
    if (!grads->vmeans || !grads->vshs || !grads->vopacities || !grads->vscales || !grads->vrotations)
        return fail(GSR_E_INVALID_ARG, "null gradient");
    float lr_t[6];
    for (int g = 0; g < 6; g++) {
        if (g == 2 && k_rest == 0) { lr_t[g] = 0.0f; continue; }  // empty features_rest (training.jl:770)
        if (!theta[g] || !mu[g] || !nu[g]) return fail(GSR_E_INVALID_ARG, "group %d: null array", g);
        if (current_step[g] == 0) return fail(GSR_E_INVALID_ARG, "group %d: current_step counts from 1", g);
        const float t = (float)current_step[g];
        lr_t[g] = lr[g] * sqrtf(1.0f - powf(beta2, t)) / (1.0f - powf(beta1, t));
    }
    const float* gr[5] = {grads->vmeans, grads->vshs, grads->vopacities, grads->vscales, grads->vrotations};
    gsr_launch_trainer_tail((hipStream_t)stream, n, k_rest, scale_dims, gr, theta, mu, nu, lr_t, beta1, beta2, eps, shs,
                            opacities_act, scales_act);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_backward_trainer_tail(gsr_handle* h, const gsr_inputs* in, const gsr_camera* cam, const float* vpixels,
                              const gsr_tail_state* st, void* stream_v) {
    int rc = check_inputs(h, in, cam);
    if (rc) return rc;
    if (!vpixels || !st) return fail(GSR_E_INVALID_ARG, "null vpixels / tail state");
    if (!h->fwd_valid || h->last_n != in->n)
        return fail(GSR_E_STATE, "gsr_backward_trainer_tail without a matching gsr_forward on this handle");
    if (h->fwd_only)
        return fail(GSR_E_STATE, "the handle's last forward was GSR_FORWARD_ONLY: it kept no backward state");
    if (h->inputs_consumed)
        return fail(GSR_E_STATE, "the inputs of the handle's last forward were updated in place by "
                    "gsr_backward_trainer_tail; run gsr_forward again");
    if ((st->flags & ~GSR_GRADS_COLOR_COTANGENT) != 0u || st->reserved != 0u) return fail(GSR_E_INVALID_ARG, "unknown gsr_tail_state.flags / reserved bits");
    if (st->forward_generation != 0 && st->forward_generation != h->generation)
        return fail(GSR_E_STATE, "gsr_backward_trainer_tail for forward #%llu, but the handle's last forward is #%llu",
                    (unsigned long long)st->forward_generation, (unsigned long long)h->generation);
    if (st->scale_dims != 1 && st->scale_dims != 3) return fail(GSR_E_INVALID_ARG, "scale_dims must be 1 or 3");
    const int n = in->n, K = in->n_coeffs;
    if (n == 0) return GSR_OK;
    float lr_t[6];
    for (int g = 0; g < 6; g++) {
        if (g == 2 && K == 1) { lr_t[g] = 0.0f; continue; }  // empty features_rest (training.jl:770)
        if (!st->theta[g] || !st->mu[g] || !st->nu[g]) return fail(GSR_E_INVALID_ARG, "group %d: null array", g);
        if (st->current_step[g] == 0) return fail(GSR_E_INVALID_ARG, "group %d: current_step counts from 1", g);
        const float t = (float)st->current_step[g];
        lr_t[g] = st->lr[g] * sqrtf(1.0f - powf(st->beta2, t)) / (1.0f - powf(st->beta1, t));
    }
    // the kernel reads its inputs through the trainer's arrays and updates them in place
    if (in->means != st->theta[0] || in->rotations != st->theta[5] || in->shs != st->shs ||
        in->opacities != st->opacities_act || in->scales != st->scales_act)
        return fail(GSR_E_INVALID_ARG, "the inputs of the fused step must be the trainer's own arrays (means == theta[0], "
                    "rotations == theta[5], shs / opacities / scales == the activated copies)");
    if (((uintptr_t)st->theta[5] & 15) != 0) return fail(GSR_E_INVALID_ARG, "rotations must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream_v;
    const int C = h->cfg.mode;
    if (!st->vmeans2d && (rc = h->vmean2d.ensure((size_t)n * 8, 1.25f))) return rc;
    h->vmean2d_cur = st->vmeans2d ? reinterpret_cast<float2*>(st->vmeans2d) : h->vmean2d.as<float2>();
    const bool color_only = (st->flags & GSR_GRADS_COLOR_COTANGENT) != 0u;
    if (color_only && (rc = check_color_cotangent(h, vpixels, s))) return rc;
    GsrCam k = make_cam(h, cam);
    StageScope sc11(h->prof, ST_COMPOSITE_BWD, s);
    if (h->last_D > 0 && (rc = launch_composite_bwd(h, s, C, k, in->background, vpixels, color_only))) return rc;
    sc11.close();
    StageScope sc12(h->prof, ST_PERGAUSS_BWD, s);
    const gsr::TailState S = gsr_make_tail_state(st->theta, st->mu, st->nu, lr_t, st->beta1, st->beta2, st->eps,
                                                 st->scale_dims, st->shs, st->opacities_act, st->scales_act);
    gsr_launch_pergauss_bwd_tail(s, n, K, in->sh_degree, C, k, geom_of(h), inst_of(h), h->vmean2d_cur, S,
                                 /*fp32_chain=*/h->cfg.grad_precision == GSR_GRAD_FP32_REFERENCE);
    sc12.close();
    HIPCHK(hipGetLastError());
    h->bwd_valid = true;
    // the forward's inputs no longer exist (updated in place): a second backward on this forward would differentiate
    // the wrong parameters (the forward's own outputs — radii, tile lists, image — stay readable)
    h->inputs_consumed = true;
    return GSR_OK;
}

size_t gsr_mask_findall_scratch_bytes(int64_t n) { return n > 0 ? gsr_findall_scratch_bytes(n) : sizeof(uint32_t); }

int gsr_mask_findall(const uint8_t* mask, int64_t n, uint32_t* indices, uint32_t* count_out, void* scratch,
                     void* stream) {
    if (n < 0 || n > 0xFFFFFFFFll) return fail(GSR_E_INVALID_ARG, "n out of range");
    if (!count_out || (n > 0 && (!mask || !indices || !scratch))) return fail(GSR_E_INVALID_ARG, "null array");
    gsr_launch_findall((hipStream_t)stream, n, mask, indices, count_out, (uint32_t*)scratch);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_gather_rows(const gsr_gather_group* groups, int32_t n_groups, const uint32_t* indices, int64_t count,
                    void* stream) {
    if (n_groups < 0 || n_groups > GSR_ADAM_MAX_GROUPS || (n_groups > 0 && !groups))
        return fail(GSR_E_INVALID_ARG, "n_groups must be in [0, %d]", GSR_ADAM_MAX_GROUPS);
    if (count < 0) return fail(GSR_E_INVALID_ARG, "negative count");
    if (count == 0 || n_groups == 0) return GSR_OK;
    if (!indices) return fail(GSR_E_INVALID_ARG, "null indices");
    const void* src[GSR_ADAM_MAX_GROUPS]; void* dst[GSR_ADAM_MAX_GROUPS]; int rw[GSR_ADAM_MAX_GROUPS];
    int m = 0;
    for (int g = 0; g < n_groups; g++) {
        if (groups[g].row_words < 0) return fail(GSR_E_INVALID_ARG, "group %d: negative row_words", g);
        if (groups[g].row_words == 0) continue;  // e.g. an empty features_rest (densification.jl:40-41)
        if (!groups[g].src || !groups[g].dst) return fail(GSR_E_INVALID_ARG, "group %d: null array", g);
        src[m] = groups[g].src; dst[m] = groups[g].dst; rw[m] = groups[g].row_words; m++;
    }
    if (m == 0) return GSR_OK;
    gsr_launch_gather_rows((hipStream_t)stream, m, src, dst, rw, indices, count);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_densify_grad_mean(int64_t n, const float* accum, const float* denom, float* grad_out, void* stream) {
    if (n < 0) return fail(GSR_E_INVALID_ARG, "negative n");
    if (n == 0) return GSR_OK;
    if (!accum || !denom || !grad_out) return fail(GSR_E_INVALID_ARG, "null array");
    gsr_launch_grad_mean((hipStream_t)stream, n, accum, denom, grad_out);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_densify_mask(int32_t kind, int64_t n, int64_t n_grad, const float* grad, const float* scales, int32_t scale_dims,
                     const float* opacities, const int32_t* max_radii, float grad_threshold, float gamma, float min_opacity,
                     int32_t max_screen_size, uint8_t* mask, void* stream) {
    if (kind < GSR_DENSIFY_CLONE || kind > GSR_DENSIFY_PRUNE) return fail(GSR_E_INVALID_ARG, "unknown mask kind %d", kind);
    if (n < 0 || n_grad < 0 || n_grad > n) return fail(GSR_E_INVALID_ARG, "bad sizes: n=%lld n_grad=%lld", (long long)n, (long long)n_grad);
    if (scale_dims != 1 && scale_dims != 3) return fail(GSR_E_INVALID_ARG, "scale_dims must be 1 or 3");
    if (n == 0) return GSR_OK;
    if (!mask) return fail(GSR_E_INVALID_ARG, "null mask");
    if (kind != GSR_DENSIFY_PRUNE && (!scales || (n_grad > 0 && !grad))) return fail(GSR_E_INVALID_ARG, "null array");
    if (kind == GSR_DENSIFY_PRUNE && (!opacities || (max_screen_size > 0 && (!max_radii || !scales))))
        return fail(GSR_E_INVALID_ARG, "null array");
    gsr_launch_densify_mask((hipStream_t)stream, kind, n, n_grad, grad, scales, scale_dims, opacities, max_radii, grad_threshold,
                            gamma, min_opacity, max_screen_size, mask);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_compose_rows(const gsr_compose_group* groups, int32_t n_groups, const uint32_t* keep_idx, int64_t n_keep,
                     const uint32_t* sel_idx, int64_t n_sel, int32_t reps, void* stream) {
    if (n_groups < 0 || n_groups > GSR_COMPOSE_MAX_GROUPS || (n_groups > 0 && !groups))
        return fail(GSR_E_INVALID_ARG, "n_groups must be in [0, %d]", GSR_COMPOSE_MAX_GROUPS);
    if (n_keep < 0 || n_sel < 0 || reps < 0) return fail(GSR_E_INVALID_ARG, "negative count");
    if (n_sel > 0 && reps > 0 && !sel_idx) return fail(GSR_E_INVALID_ARG, "null sel_idx");
    if (n_keep + n_sel * reps == 0 || n_groups == 0) return GSR_OK;
    const void* src[GSR_COMPOSE_MAX_GROUPS]; void* dst[GSR_COMPOSE_MAX_GROUPS];
    int rw[GSR_COMPOSE_MAX_GROUPS], nz[GSR_COMPOSE_MAX_GROUPS];
    int m = 0;
    for (int g = 0; g < n_groups; g++) {
        if (groups[g].row_words < 0) return fail(GSR_E_INVALID_ARG, "group %d: negative row_words", g);
        if (groups[g].row_words == 0) continue;  // an empty features_rest (densification.jl:40-41,196-201)
        if (!groups[g].dst || (!groups[g].src && (n_keep > 0 || !groups[g].new_zero)))
            return fail(GSR_E_INVALID_ARG, "group %d: null array", g);
        src[m] = groups[g].src; dst[m] = groups[g].dst; rw[m] = groups[g].row_words; nz[m] = groups[g].new_zero ? 1 : 0; m++;
    }
    if (m == 0) return GSR_OK;
    gsr_launch_compose_rows((hipStream_t)stream, m, src, dst, rw, nz, keep_idx, n_keep, sel_idx, reps > 0 ? n_sel : 0, reps > 0 ? reps : 1);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_split_transform(int64_t n_new, int32_t scale_dims, float* points, const float* rotations, float* scales, uint32_t seed,
                        void* stream) {
    if (n_new < 0 || (scale_dims != 1 && scale_dims != 3)) return fail(GSR_E_INVALID_ARG, "bad sizes");
    if (n_new == 0) return GSR_OK;  // densification.jl:94 `if n_new_points > 0`
    if (!points || !rotations || !scales) return fail(GSR_E_INVALID_ARG, "null array");
    if (((uintptr_t)rotations & 15) != 0) return fail(GSR_E_INVALID_ARG, "rotations must be 16-byte aligned");
    gsr_launch_split_transform((hipStream_t)stream, n_new, scale_dims, points, rotations, scales, seed);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_reset_opacity(int64_t n, float* opacities, void* stream) {
    if (n < 0) return fail(GSR_E_INVALID_ARG, "negative n");
    if (n == 0) return GSR_OK;
    if (!opacities) return fail(GSR_E_INVALID_ARG, "null array");
    gsr_launch_reset_opacity((hipStream_t)stream, n, opacities);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_morton_codes(int64_t n, const float* points, const float* box_lo, const float* box_hi, uint64_t* codes, void* stream) {
    if (n < 0) return fail(GSR_E_INVALID_ARG, "negative n");
    if (n == 0) return GSR_OK;
    if (!points || !box_lo || !box_hi || !codes) return fail(GSR_E_INVALID_ARG, "null array");
    float inv[3];
    for (int k = 0; k < 3; k++) {
        const float e = box_hi[k] - box_lo[k];
        if (!(e >= 0.0f)) return fail(GSR_E_INVALID_ARG, "gsr_morton_codes: box_hi < box_lo (or NaN)");
        inv[k] = e > 0.0f ? 1.0f / e : 0.0f;
    }
    gsr_launch_morton_codes((hipStream_t)stream, n, points, box_lo, inv, reinterpret_cast<unsigned long long*>(codes));
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_count_nonfinite(const float* const* arrays, const int32_t* row_words, int32_t n_groups, int64_t n_rows, uint32_t* counts,
                        uint32_t* first_bad, void* stream) {
    if (n_groups < 0 || n_groups > GSR_ADAM_MAX_GROUPS || (n_groups > 0 && (!arrays || !row_words)))
        return fail(GSR_E_INVALID_ARG, "n_groups must be in [0, %d]", GSR_ADAM_MAX_GROUPS);
    if (n_rows < 0) return fail(GSR_E_INVALID_ARG, "negative n_rows");
    if (n_groups == 0) return GSR_OK;
    if (!counts || !first_bad) return fail(GSR_E_INVALID_ARG, "null output");
    const float* src[GSR_ADAM_MAX_GROUPS]; int rw[GSR_ADAM_MAX_GROUPS];
    for (int g = 0; g < n_groups; g++) {
        if (row_words[g] < 0) return fail(GSR_E_INVALID_ARG, "group %d: negative row_words", g);
        if (row_words[g] > 0 && n_rows > 0 && !arrays[g]) return fail(GSR_E_INVALID_ARG, "group %d: null array", g);
        src[g] = arrays[g]; rw[g] = row_words[g];
    }
    gsr_launch_nonfinite_scan((hipStream_t)stream, n_groups, src, rw, n_rows, counts, first_bad);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

static int ply_rows(bool pack, int64_t n, int32_t k_rest, const float* points, const float* dc, const float* rest,
                    const float* opac, const float* scales, const float* rots, const float* rows, void* stream) {
    if (n < 0 || k_rest < 0) return fail(GSR_E_INVALID_ARG, "bad sizes: n=%lld k_rest=%d", (long long)n, k_rest);
    if (n == 0) return GSR_OK;
    if (!points || !dc || (k_rest > 0 && !rest) || !opac || !scales || !rots || !rows) return fail(GSR_E_INVALID_ARG, "null array");
    gsr_launch_ply_rows((hipStream_t)stream, pack, n, k_rest, const_cast<float*>(points), const_cast<float*>(dc),
                        const_cast<float*>(rest), const_cast<float*>(opac), const_cast<float*>(scales), const_cast<float*>(rots),
                        const_cast<float*>(rows));
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_ply_pack_rows(int64_t n, int32_t k_rest, const float* points, const float* features_dc, const float* features_rest,
                      const float* opacities, const float* scales, const float* rotations, float* rows, void* stream) {
    return ply_rows(true, n, k_rest, points, features_dc, features_rest, opacities, scales, rotations, rows, stream);
}

int gsr_ply_unpack_rows(int64_t n, int32_t k_rest, const float* rows, float* points, float* features_dc, float* features_rest,
                        float* opacities, float* scales, float* rotations, void* stream) {
    return ply_rows(false, n, k_rest, points, features_dc, features_rest, opacities, scales, rotations, rows, stream);
}

int gsr_sh_grad_from_views(int32_t n, int32_t n_coeffs, int32_t sh_degree, int32_t n_views, const float* camera_centers,
                           const float* means, const float* vcolors_all, float* vshs, void* stream) {
    if (n < 0 || n_views < 1 || sh_degree < 0 || sh_degree > 3 || n_coeffs < (sh_degree + 1) * (sh_degree + 1))
        return fail(GSR_E_INVALID_ARG, "bad sizes: n=%d views=%d degree=%d K=%d", n, n_views, sh_degree, n_coeffs);
    if (n == 0) return GSR_OK;
    if (!camera_centers || !means || !vcolors_all || !vshs) return fail(GSR_E_INVALID_ARG, "null array");
    gsr_launch_sh_grad_views((hipStream_t)stream, n, n_coeffs, sh_degree, n_views, camera_centers, means, vcolors_all, vshs);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_sh_grad_from_views_tail(int32_t n, int32_t n_coeffs, int32_t sh_degree, int32_t n_views, const float* camera_centers,
                                const float* vcolors_all, const gsr_tail_grads* small, const gsr_tail_state* st, void* stream) {
    if (n < 0 || n_views < 1 || sh_degree < 0 || sh_degree > 3 || n_coeffs < (sh_degree + 1) * (sh_degree + 1) || n_coeffs > 16)
        return fail(GSR_E_INVALID_ARG, "bad sizes: n=%d views=%d degree=%d K=%d", n, n_views, sh_degree, n_coeffs);
    if (n == 0) return GSR_OK;
    if (!camera_centers || !vcolors_all || !small || !st) return fail(GSR_E_INVALID_ARG, "null argument");
    if (!small->vmeans || !small->vopacities || !small->vscales || !small->vrotations)
        return fail(GSR_E_INVALID_ARG, "null gradient (vmeans / vopacities / vscales / vrotations; vshs is not read)");
    if (st->scale_dims != 1 && st->scale_dims != 3) return fail(GSR_E_INVALID_ARG, "scale_dims must be 1 or 3");
    if (!st->shs || !st->opacities_act || !st->scales_act) return fail(GSR_E_INVALID_ARG, "null activated copy");
    float lr_t[6];
    for (int g = 0; g < 6; g++) {
        if (g == 2 && n_coeffs == 1) { lr_t[g] = 0.0f; continue; }  // empty features_rest (training.jl:770)
        if (!st->theta[g] || !st->mu[g] || !st->nu[g]) return fail(GSR_E_INVALID_ARG, "group %d: null array", g);
        if (st->current_step[g] == 0) return fail(GSR_E_INVALID_ARG, "group %d: current_step counts from 1", g);
        const float t = (float)st->current_step[g];
        lr_t[g] = st->lr[g] * sqrtf(1.0f - powf(st->beta2, t)) / (1.0f - powf(st->beta1, t));
    }
    const gsr::TailState S = gsr_make_tail_state(st->theta, st->mu, st->nu, lr_t, st->beta1, st->beta2, st->eps,
                                                 st->scale_dims, st->shs, st->opacities_act, st->scales_act);
    gsr_launch_sh_views_tail((hipStream_t)stream, n, n_coeffs, sh_degree, n_views, camera_centers, vcolors_all, small->vmeans,
                             small->vopacities, small->vscales, small->vrotations, S);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_stream_triad(float* a, const float* b, const float* c, size_t count, float q, void* stream) {
    if (!a || !b || !c) return fail(GSR_E_INVALID_ARG, "null array");
    if (count % 4 != 0 || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) != 0)
        return fail(GSR_E_INVALID_ARG, "count must be a multiple of 4 and the arrays 16-byte aligned");
    gsr_launch_triad((hipStream_t)stream, count / 4, a, b, c, q);
    HIPCHK(hipGetLastError());
    return GSR_OK;
}

int gsr_allreduce_grads(void* nccl_comm, float* arena, size_t count, void* stream) {
    typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static allreduce_fn fn = nullptr;
    if (!nccl_comm || !arena) return fail(GSR_E_INVALID_ARG, "null communicator / arena");
    if (!fn) {
        void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return fail(GSR_E_HIP, "cannot load librccl.so: %s", dlerror());
        fn = (allreduce_fn)dlsym(lib, "ncclAllReduce");
        if (!fn) return fail(GSR_E_HIP, "ncclAllReduce not found in librccl.so");
    }
    const int ncclFloat32 = 7, ncclSum = 0;
    int r = fn(arena, arena, count, ncclFloat32, ncclSum, nccl_comm, (hipStream_t)stream);
    if (r != 0) return fail(GSR_E_HIP, "ncclAllReduce failed with code %d", r);
    return GSR_OK;
}

}  // extern "C"
