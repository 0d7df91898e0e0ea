// wrsn_api.hip -- host side of libwrsn_hip.so: the C-ABI declared in include/wrsn_hip.h.
// Owns device memory of a handle and launches the gfx950 kernels of wrsn_sim.h.  There is no CPU
// execution path: without a HIP device wrsn_create fails with WRSN_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <cstdlib>
#include <vector>

#include "../../include/wrsn_hip.h"
#include "wrsn_sim.h"
#include "wrsn_rollout.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(WRSN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Every entry point runs on the handle's device and leaves the calling thread's current device as it found it (handles of
// different GPUs may share a process and a thread with torch).
struct DeviceGuard {
    int prev; bool ok;
    explicit DeviceGuard(int dev) : prev(-1), ok(false) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
        if (prev == dev) prev = -1;                            // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define WRSN_ON_DEVICE(h_)                                                                             \
    DeviceGuard guard_((h_)->cfg.device);                                                              \
    if (!guard_.ok) return fail(WRSN_ERR_HIP, "hipSetDevice failed for the handle's device")

int npl_for(int n_node) {
    int need = (n_node + 63) / 64;
    const int choices[] = {1, 2, 4, 8, 16};
    for (int c : choices) if (c >= need) return c;
    return -1;
}

}  // namespace

// nodes-per-lane dispatch: `M_(NPL)` for the handle's register-slot count.  Diagnostic builds (tools/ab_build.sh) compile one slot count only
// (-DWRSN_ONLY_NPL=4: a quarter of the build time); the product build has all five.
#ifdef WRSN_ONLY_NPL
#define WRSN_NPL_SWITCH(npl_, M_, bad_) switch (npl_) { case WRSN_ONLY_NPL: M_(WRSN_ONLY_NPL); break; default: bad_; }
#else
#define WRSN_NPL_SWITCH(npl_, M_, bad_) switch (npl_) { case 1: M_(1); break; case 2: M_(2); break; case 4: M_(4); break; case 8: M_(8); break; case 16: M_(16); break; default: bad_; }
#endif

struct wrsn_handle {
    wrsn_cfg cfg;
    WrsnDev dev;
    hipStream_t stream;
    int npl;
    int scenario_set;
    int lds_env, lds_obs;      // LDS bytes of an environment wave, of an observation block
    int cc_bound;              // largest WrsnEnvConst.conn_bound of the scenarios set so far (-> WrsnDev.CC)
    hipStream_t stream2, stream3;   // the later stages of a pipelined step call run here, beside the first one on `stream`
    hipEvent_t ev_fork, ev_join, ev_join3; int ev2_ok;
    int cus;                   // compute units of the device
    int slots;                 // wave slots of the device for the step kernel (CUs x resident waves per CU): launch-order dependent budgets
    int waves_per_cu;          // what the occupancy query said for this handle's step kernel (diagnostic)
    long long epoch;           // counter of wrsn_step calls (its parity selects the hand-off list a budgeted call reads / writes)
    int step_budget;           // work units one wrsn_step launch may spend per environment, 0 = run every step to its end
    int deadline_ticks;        // wrsn_set_step_deadline in 100 MHz wall-clock ticks, 0 = none
    int pipe_mid_pct;          // share of the long half that stays on the caller's stream (the rest: third stream); 100 = two stages
    int pipe_short_pct;        // work cap of the short stage in per cent of the step budget (its stragglers go on in the next call)
    int pipe, pipe_long_pct;   // step calls that render as a two-stage pipeline over the launch order (WRSN_PIPE=0 disables); share of the long stage
    int lds_pad;               // extra LDS bytes per environment wave (occupancy experiments); diagnostic
    int taper;                 // packed budget taper (start << 16 | length << 24), OR-ed into the `slots` kernel argument
    int obs_reuse;             // wrsn_set_obs_reuse: the caller keeps the observation rows the library wrote
    int timing;                // record HIP events around the kernels of every wrsn_step (wrsn_set_timing)
    hipEvent_t ev[5];          // before the order kernels, after them, after the step kernel, after the continuation, after the observation
    int ev_ok, ev_obs;         // events created / the last call rendered an observation
    int ev_rec;                // a wrsn_step has recorded the events since timing was switched on
    int bp2;                   // B rounded up to a power of two when the launch order is sorted on the device (B <= 8192), else 0
    std::vector<void*> allocs;
    WrsnDev* d_dev;            // device copy of `dev`: the environment kernels read it through the constant cache
};

namespace {

template <typename T>
int dalloc(wrsn_handle* h, T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIPCHK(hipMalloc(&q, bytes));
    HIPCHK(hipMemset(q, 0, bytes));
    h->allocs.push_back(q);
    *p = (T*)q;
    return 0;
}

int alloc_node_arrays(wrsn_handle* h, WrsnNodeArrays* a) {
    const size_t B = h->dev.B, NP = h->dev.NP;
    int rc;
    if ((rc = dalloc(h, &a->E, B * NP))) return rc;
    if ((rc = dalloc(h, &a->CS, B * NP))) return rc;
    if ((rc = dalloc(h, &a->RR, B * NP))) return rc;
    if ((rc = dalloc(h, &a->d1, B * NP))) return rc;
    if ((rc = dalloc(h, &a->d2, B * NP))) return rc;
    if ((rc = dalloc(h, &a->ring, B * WRSN_RING * NP))) return rc;
    if ((rc = dalloc(h, &a->logbuf, B * NP))) return rc;
    if ((rc = dalloc(h, &a->ls, B * NP))) return rc;
    if ((rc = dalloc(h, &a->rcv, B * NP))) return rc;
    if ((rc = dalloc(h, &a->conn, B * WRSN_MAX_MC * WRSN_CONN_CAP))) return rc;
    if ((rc = dalloc(h, &a->conn_xy, B * WRSN_MAX_MC * WRSN_CONN_CAP * 2))) return rc;
    if ((rc = dalloc(h, &a->dyn, B))) return rc;
    return 0;
}

// LDS sizes, wave slots and the device copy of the descriptor; again whenever WrsnDev.CC changes
int configure_launch(wrsn_handle* h) {
    WrsnDev& d = h->dev;
    h->lds_env = wrsn_lds_bytes(d.NP, d.M, d.CC);
    {   // wave slots of the step kernel on this device (registers and LDS decide): the budget taper of a launch starts behind the blocks
        // that are resident from the first moment
        int per_cu = 0; hipError_t oe = hipErrorUnknown;
        const int lds_b = h->lds_env + h->lds_pad;
#define WRSN_OCC(NPL_) oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wrsn_step_kernel<NPL_, WRSN_KERNEL_INLINE>, 64, (size_t)lds_b)
        WRSN_NPL_SWITCH(h->npl, WRSN_OCC, oe = hipErrorUnknown)
#undef WRSN_OCC
        h->slots = h->cus * 8;
        if (oe == hipSuccess && per_cu >= 1 && per_cu <= 16) h->slots = h->cus * per_cu;
        h->waves_per_cu = (oe == hipSuccess) ? per_cu : 0;
    }
    HIPCHK(hipMemcpy(h->d_dev, &h->dev, sizeof(WrsnDev), hipMemcpyHostToDevice));
    return 0;
}

int launch_obs(wrsn_handle* h, const int32_t* agent_id, float* obs);

// `obs_pipe` (step calls that render): the observations of this call are launched from here, interleaved with the step launches (below)
int launch_env(wrsn_handle* h, int mode, int env0, int nenv, const int32_t* agent_id, const double* action,
               int auto_reset, const uint8_t* mask, const WrsnStepOutDev& out, float* obs_pipe = nullptr) {
    const int lds = h->lds_env + h->lds_pad;
    const int reset_call = (mode == WRSN_MODE_RESET) ? 1 : 0;
    const int budget = (mode == WRSN_MODE_STEP) ? h->step_budget : 0;
    const int dl = (budget > 0 && h->bp2 > 0) ? h->deadline_ticks : 0;     // the sort kernel zeroes the launch stamp
    const int taper = h->taper;
    dim3 grid(nenv), block(64);
    long long epoch = 0;
    if (mode == WRSN_MODE_STEP) epoch = ++h->epoch;            // every step call: a hand-off stamp names the one call whose heavy launch owns the environment
    const bool queue = (mode == WRSN_MODE_STEP) && h->deadline_ticks > 0;    // work-queue launch (wrsn_set_step_deadline)
    const bool timed = (mode == WRSN_MODE_STEP) && h->timing && h->ev_ok;
    if (timed) (void)hipEventRecord(h->ev[0], h->stream);
    if (queue) {
        // latch + preset kernel, then one block per environment in this launch's cyclic order: the blocks that only start when the time slice
        // is over leave their environments alone
        hipLaunchKernelGGL(wrsn_latch_kernel, dim3((nenv + 255) / 256), dim3(256), 0, h->stream, h->dev, agent_id, action, out);
        if (timed) (void)hipEventRecord(h->ev[1], h->stream);
        const int qbudget = budget > 0 ? budget : (1 << 28);   // the deadline is looked at wherever a work budget is
#define WRSN_QUEUE(NPL_) hipLaunchKernelGGL((wrsn_step_kernel<NPL_, WRSN_KERNEL_INLINE>), grid, block, lds, h->stream, (const WrsnDev*)h->d_dev, 0, agent_id, action, \
                                           auto_reset, qbudget, epoch, 0, mask, out, 3, h->deadline_ticks, 0)
        WRSN_NPL_SWITCH(h->npl, WRSN_QUEUE, return fail(WRSN_ERR_ARG, "unsupported nodes-per-lane"))
#undef WRSN_QUEUE
        if (timed) { (void)hipEventRecord(h->ev[2], h->stream); (void)hipEventRecord(h->ev[3], h->stream); h->ev_obs = 0; h->ev_rec = 1; }
        HIPCHK(hipGetLastError());
        if (obs_pipe) {
            int rc = launch_obs(h, h->dev.render_agent, obs_pipe);
            if (timed) { (void)hipEventRecord(h->ev[4], h->stream); h->ev_obs = 1; }
            return rc;
        }
        return 0;
    }
    // A step call that renders, as a PIPELINE over the two halves of the launch order (longest job first): the short half is stepped on the
    // second stream and rendered there as soon as it is done -- nearly all of its steps complete, it carries ~60 % of the observations of
    // the call -- while the long half (work-capped or slow steps, the tail of the launch) is still being stepped on the caller's stream;
    // only the observations of the long half are left for afterwards.  The step kernel is bound by instruction issue and latency and leaves
    // the HBM idle, the observation kernel is a 160 KB store stream per row: they overlap well.  Same blocks, same budgets, same results.
    // (only when the batch is at most two rounds of the wave slots: with more, the short half is the longer one and nothing overlaps --
    //  4 096 environments of 1 000 nodes on 768 slots: 0.87 M env-steps/s with the pipeline, 0.96 M without)
    const bool pipe = obs_pipe && mode == WRSN_MODE_STEP && h->pipe && h->ev2_ok && h->bp2 > 0 && nenv >= 512 && nenv == h->dev.B && nenv <= 2 * h->slots;
    if (mode == WRSN_MODE_STEP && h->bp2 > 0) {
        // launch order of this call, longest job first (wrsn_estimate_kernel / wrsn_sort_kernel, wrsn_sim.h): two tiny launches
        hipLaunchKernelGGL(wrsn_estimate_kernel, dim3((h->bp2 + WRSN_EST_THREADS - 1) / WRSN_EST_THREADS), dim3(WRSN_EST_THREADS), 0, h->stream, h->dev, agent_id, action, auto_reset, h->bp2);
        {
            const int kpt = h->bp2 / WRSN_SORT_THREADS;        // keys per thread of the sort workgroup (0, 1: plain network in LDS)
            const size_t lb = (size_t)wrsn_sort_lds_bytes();
#define WRSN_SORT(K_) hipLaunchKernelGGL((wrsn_sort_kernel<K_>), dim3(1), dim3(WRSN_SORT_THREADS), lb, h->stream, h->dev, h->bp2)
            switch (kpt) {
            case 2: WRSN_SORT(2); break;
            case 4: WRSN_SORT(4); break;
            case 8: WRSN_SORT(8); break;
            case 16: WRSN_SORT(16); break;
            case 32: WRSN_SORT(32); break;
            default: WRSN_SORT(1); break;
            }
#undef WRSN_SORT
        }
    }
    if (timed) (void)hipEventRecord(h->ev[1], h->stream);
#define WRSN_LAUNCH(NPL_)                                                                                              \
    if (mode == WRSN_MODE_WARMUP) hipLaunchKernelGGL(wrsn_warmup_kernel<NPL_>, grid, block, lds, h->stream, (const WrsnDev*)h->d_dev, env0);  \
    else if (pipe) {                                                                                                    \
        /* stages over the launch order: [0, n1) the longest jobs (caller's stream), [n1, n2) the rest of the long half (third stream), */ \
        /* [n2, B) the short half with its own work cap (second stream); n2 <= wave slots: the long half's jobs all start at once       */ \
        int n2 = (nenv * h->pipe_long_pct / 100 + 63) & ~63; if (n2 > h->slots) n2 = h->slots & ~63; if (n2 < 64) n2 = 64;     \
        int n1 = (n2 * h->pipe_mid_pct / 100 + 63) & ~63; if (n1 > n2 || !h->stream3) n1 = n2; if (n1 < 64) n1 = 64;            \
        const int b_short = budget > 0 ? (budget * h->pipe_short_pct / 100 > 64 ? budget * h->pipe_short_pct / 100 : 64) : 0; \
        (void)hipEventRecord(h->ev_fork, h->stream); (void)hipStreamWaitEvent(h->stream2, h->ev_fork, 0);              \
        if (n1 < n2) (void)hipStreamWaitEvent(h->stream3, h->ev_fork, 0);                                              \
        hipLaunchKernelGGL((wrsn_step_kernel<NPL_, WRSN_KERNEL_INLINE>), dim3(nenv - n2), block, lds, h->stream2, (const WrsnDev*)h->d_dev, reset_call, agent_id, action, \
                           auto_reset, b_short, epoch, (h->slots & 0xFFFF) | taper, mask, out, 0, dl, n2);                                     \
        hipLaunchKernelGGL(wrsn_obs_kernel, dim3(nenv - n2), dim3(256), h->lds_obs, h->stream2, h->dev, (const int32_t*)h->dev.render_agent, obs_pipe, h->obs_reuse, (const int32_t*)h->dev.order, n2); \
        (void)hipEventRecord(h->ev_join, h->stream2);                                                                 \
        if (n1 < n2) {                                                                                                 \
            hipLaunchKernelGGL((wrsn_step_kernel<NPL_, WRSN_KERNEL_INLINE>), dim3(n2 - n1), block, lds, h->stream3, (const WrsnDev*)h->d_dev, reset_call, agent_id, action, \
                               auto_reset, budget, epoch, (h->slots & 0xFFFF) | taper, mask, out, 0, dl, n1);                                  \
            hipLaunchKernelGGL(wrsn_obs_kernel, dim3(n2 - n1), dim3(256), h->lds_obs, h->stream3, h->dev, (const int32_t*)h->dev.render_agent, obs_pipe, h->obs_reuse, (const int32_t*)h->dev.order, n1); \
            (void)hipEventRecord(h->ev_join3, h->stream3);                                                             \
        }                                                                                                              \
        hipLaunchKernelGGL((wrsn_step_kernel<NPL_, WRSN_KERNEL_INLINE>), dim3(n1), block, lds, h->stream, (const WrsnDev*)h->d_dev, reset_call, agent_id, action, \
                           auto_reset, budget, epoch, (h->slots & 0xFFFF) | taper, mask, out, 0, dl, 0);                                       \
        if (timed) { (void)hipEventRecord(h->ev[2], h->stream); (void)hipEventRecord(h->ev[3], h->stream); }            \
        hipLaunchKernelGGL(wrsn_obs_kernel, dim3(n1), dim3(256), h->lds_obs, h->stream, h->dev, (const int32_t*)h->dev.render_agent, obs_pipe, h->obs_reuse, (const int32_t*)h->dev.order, 0); \
        (void)hipStreamWaitEvent(h->stream, h->ev_join, 0);                                                            \
        if (n1 < n2) (void)hipStreamWaitEvent(h->stream, h->ev_join3, 0);                                              \
    }                                                                                                                  \
    else hipLaunchKernelGGL((wrsn_step_kernel<NPL_, WRSN_KERNEL_INLINE>), grid, block, lds, h->stream, (const WrsnDev*)h->d_dev, reset_call, agent_id, action, \
                            auto_reset, budget, epoch, (h->slots & 0xFFFF) | taper, mask, out, 0, dl, 0)
    WRSN_NPL_SWITCH(h->npl, WRSN_LAUNCH, return fail(WRSN_ERR_ARG, "unsupported nodes-per-lane"))
#undef WRSN_LAUNCH
    if (pipe) {
        if (timed) { (void)hipEventRecord(h->ev[4], h->stream); h->ev_obs = 1; h->ev_rec = 1; }
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (timed) (void)hipEventRecord(h->ev[2], h->stream);
    if (timed) { (void)hipEventRecord(h->ev[3], h->stream); h->ev_obs = 0; h->ev_rec = 1; }
    HIPCHK(hipGetLastError());
    if (obs_pipe) {
        int rc = launch_obs(h, h->dev.render_agent, obs_pipe);
        if (timed) { (void)hipEventRecord(h->ev[4], h->stream); h->ev_obs = 1; }
        return rc;
    }
    return 0;
}

int launch_obs(wrsn_handle* h, const int32_t* agent_id, float* obs) {
    hipLaunchKernelGGL(wrsn_obs_kernel, dim3(h->dev.B), dim3(256), h->lds_obs, h->stream, h->dev, agent_id, obs, h->obs_reuse, (const int32_t*)nullptr, 0);
    HIPCHK(hipGetLastError());
    return 0;
}

// xoshiro256** seeded by splitmix64: the synthetic generator's own counter-free RNG
struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { uint64_t x = seed; for (int i = 0; i < 4; ++i) s[i] = splitmix(x); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double uni(double a, double b) { return a + (b - a) * uni(); }
    int below(int n) { return (int)(uni() * n) % (n > 0 ? n : 1); }
    double normal() { double u1 = uni(), u2 = uni(); if (u1 < 1e-300) u1 = 1e-300; return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2); }
};

}  // namespace

extern "C" {

const char* wrsn_last_error(void) { return g_err.c_str(); }
const char* wrsn_version(void) { return "wrsn_hip 0.1 (gfx950)"; }

int wrsn_create(const wrsn_cfg* cfg, wrsn_t** out) {
    if (!cfg || !out) return fail(WRSN_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->n_env < 1 || cfg->n_node < 1 || cfg->n_target < 1 || cfg->n_mc < 1 || cfg->n_mc > WRSN_MAX_MC || cfg->map_size < 4 ||
        cfg->map_size > 128 || cfg->n_node > 1024 || !(cfg->warm_up_time > 0.0))
        return fail(WRSN_ERR_ARG, "wrsn_cfg out of range (n_mc 1..8, n_node 1..1024, map_size 4..128, warm_up_time > 0)");
    int ndev = 0;
    {
        const hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0)
            return fail(WRSN_ERR_NO_DEVICE, std::string("no HIP device: libwrsn_hip has no CPU fallback (hipGetDeviceCount: ") + hipGetErrorString(e) + ")");
    }
    if (cfg->device < 0 || cfg->device >= ndev) return fail(WRSN_ERR_ARG, "device ordinal out of range");
    DeviceGuard guard_(cfg->device);
    if (!guard_.ok) return fail(WRSN_ERR_HIP, "hipSetDevice failed");
    wrsn_handle* h = new wrsn_handle();
    h->cfg = *cfg; h->stream = nullptr; h->scenario_set = 0; h->step_budget = 0; h->deadline_ticks = 0; h->epoch = 1; h->obs_reuse = 0; h->timing = 0; h->ev_ok = 0; h->ev_obs = 0; h->ev_rec = 0;
    { const char* e = std::getenv("WRSN_PIPE"); h->pipe = (e && *e == '0') ? 0 : 1; }
    { const char* e = std::getenv("WRSN_PIPE_MID_PCT"); h->pipe_mid_pct = e ? std::atoi(e) : 100; if (h->pipe_mid_pct < 10 || h->pipe_mid_pct > 100) h->pipe_mid_pct = 100; }
    { const char* e = std::getenv("WRSN_PIPE_SHORT_PCT"); h->pipe_short_pct = e ? std::atoi(e) : 40; if (h->pipe_short_pct < 5 || h->pipe_short_pct > 100) h->pipe_short_pct = 40; }
    { const char* e = std::getenv("WRSN_PIPE_LONG_PCT"); h->pipe_long_pct = e ? std::atoi(e) : 50; if (h->pipe_long_pct < 10 || h->pipe_long_pct > 90) h->pipe_long_pct = 50; }
    h->stream2 = nullptr; h->ev2_ok = 0; h->cc_bound = 0; h->cus = 256;
    // the second stream: high priority by default (WRSN_STREAM2_PRIO=0: normal) -- its launch is the SHORT half of a pipelined step call, whose
    // blocks should get wave slots first so that its observations can be rendered while the long half is still being stepped
    hipError_t se = hipErrorUnknown;
    { const char* e = std::getenv("WRSN_STREAM2_PRIO"); const bool hi = !(e && *e == '0');
      int lo_p = 0, hi_p = 0;
      if (hi && hipDeviceGetStreamPriorityRange(&lo_p, &hi_p) == hipSuccess && hi_p != lo_p) se = hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, hi_p);
      if (se != hipSuccess) se = hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking); }
    h->stream3 = nullptr;
    if (se == hipSuccess && hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking) == hipSuccess) {
        if (hipEventCreateWithFlags(&h->ev_join3, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(h->stream3); h->stream3 = nullptr; }
    } else h->stream3 = nullptr;
    if (se == hipSuccess && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess) {
        if (hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) == hipSuccess) h->ev2_ok = 1; else (void)hipEventDestroy(h->ev_fork);
    }
    {   // budget taper over the launch order (units of slots / 8 blocks): start 8 = after the first `slots` blocks, length 16 = down to
        // zero over two times `slots` blocks (the floor of a quarter applies first); WRSN_TAPER="start,len" overrides (diagnostic)
        int ts = 8, tl = 16; const char* e = std::getenv("WRSN_TAPER");
        if (e) { int a = 0, b = 0; if (std::sscanf(e, "%d,%d", &a, &b) == 2 && a >= 0 && a < 256 && b > 0 && b < 256) { ts = a; tl = b; } }
        h->taper = (ts << 16) | (tl << 24);
    }
    { const char* e = std::getenv("WRSN_LDS_PAD"); h->lds_pad = e ? std::atoi(e) : 0; if (h->lds_pad < 0 || h->lds_pad > 100000) h->lds_pad = 0; }
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, cfg->device) == hipSuccess && pr.multiProcessorCount > 0) h->cus = pr.multiProcessorCount; h->slots = h->cus * 8; }
    h->npl = npl_for(cfg->n_node);
    if (h->npl < 0) { delete h; return fail(WRSN_ERR_ARG, "n_node too large"); }
    WrsnDev& d = h->dev;
    std::memset(&d, 0, sizeof(d));
    d.B = cfg->n_env; d.N = cfg->n_node; d.T = cfg->n_target; d.M = cfg->n_mc; d.G = cfg->map_size;
    d.NP = h->npl * 64; d.TP = ((cfg->n_target + 63) / 64) * 64;
    d.ECAP = (cfg->max_degree > 0 ? cfg->max_degree : 24) * d.NP;
    d.CCAP = (cfg->max_cover > 0 ? cfg->max_cover : 8) * d.NP;
    // observation tile geometry must fit the register tile of wrsn_obs_kernel
    {
        int CG = (d.G + 3) / 4; int RG = 256 / CG; if (RG < 1) { delete h; return fail(WRSN_ERR_ARG, "map_size too large"); }
        int RPG = (d.G + RG - 1) / RG; if (RPG > WRSN_OBS_MAXROWS) { delete h; return fail(WRSN_ERR_ARG, "map_size too large for the observation tile"); }
    }
    d.CC = 4;                                                  // raised by wrsn_set_scenario to what the scenarios need
    h->lds_obs = wrsn_obs_lds_bytes(d.G, d.NP);
    const size_t B = d.B, NP = d.NP;
    int rc = 0;
    do {
        if ((rc = dalloc(h, &d.ec, B))) break;
        if ((rc = dalloc(h, &d.node_x, B * NP))) break;
        if ((rc = dalloc(h, &d.node_y, B * NP))) break;
        if ((rc = dalloc(h, &d.dist_bs, B * NP))) break;
        if ((rc = dalloc(h, &d.target_x, B * d.TP))) break;
        if ((rc = dalloc(h, &d.target_y, B * d.TP))) break;
        if ((rc = dalloc(h, &d.nb_off, B * (NP + 1)))) break;
        if ((rc = dalloc(h, &d.nb_idx, B * (size_t)d.ECAP))) break;
        if ((rc = dalloc(h, &d.nb_dist, B * (size_t)d.ECAP))) break;
        if ((rc = dalloc(h, &d.tc_off, B * (size_t)(d.TP + 1)))) break;
        if ((rc = dalloc(h, &d.tc_idx, B * (size_t)d.CCAP))) break;
        if ((rc = dalloc(h, &d.ncov, B * NP))) break;
        if ((rc = dalloc(h, &d.nflags, B * NP))) break;
        if ((rc = dalloc(h, &d.nbp, B * NP * 4))) break;
        if ((rc = dalloc(h, &d.nbp_es, B * NP * 8))) break;
        if ((rc = dalloc(h, &d.es_bs, B * NP))) break;
        if ((rc = dalloc(h, &d.xorder, B * NP))) break;
        if ((rc = dalloc(h, &d.adjm, B * NP * 8))) break;
        if ((rc = dalloc(h, &d.tcp, B * (size_t)d.TP * 4))) break;
        if ((rc = alloc_node_arrays(h, &d.live))) break;
        if ((rc = alloc_node_arrays(h, &d.snap))) break;
        if ((rc = dalloc(h, &d.counters, B * 25))) break;
        { int p2 = 1; while (p2 < d.B) p2 <<= 1; h->bp2 = (d.B <= 8192 && !std::getenv("WRSN_NO_ORDER")) ? p2 : 0; }
        if ((rc = dalloc(h, &d.order_key, (size_t)(h->bp2 > 0 ? h->bp2 : 1)))) break;
        if ((rc = dalloc(h, &d.order, (size_t)(h->bp2 > d.B ? h->bp2 : d.B)))) break;
        if ((rc = dalloc(h, &d.launch_t0, 1))) break;
        if ((rc = dalloc(h, &d.render_agent, B))) break;
        if ((rc = dalloc(h, &d.row_state, B))) break;
        if ((rc = dalloc(h, &d.queue, 8 + 64))) break;
        if ((rc = dalloc(h, &d.qskip, B))) break;
        if ((rc = dalloc(h, &h->d_dev, 1))) break;
    } while (0);
    if (rc) { wrsn_destroy(h); return rc; }
    {   // identity launch order: what a handle too large for the device-side sort (or WRSN_NO_ORDER) keeps
        std::vector<int32_t> ident(B); for (size_t e = 0; e < B; ++e) ident[e] = (int32_t)e;
        if (hipMemcpy(d.order, ident.data(), B * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { wrsn_destroy(h); return fail(WRSN_ERR_HIP, "hipMemcpy"); }
    }
    if (configure_launch(h) != 0) { wrsn_destroy(h); return fail(WRSN_ERR_HIP, "hipMemcpy"); }
    *out = h;
    return WRSN_OK;
}

void wrsn_destroy(wrsn_t* h) {
    if (!h) return;
    DeviceGuard guard_(h->cfg.device);
    if (h->ev_ok) for (int i = 0; i < 5; ++i) (void)hipEventDestroy(h->ev[i]);
    if (h->ev2_ok) { (void)hipEventDestroy(h->ev_fork); (void)hipEventDestroy(h->ev_join); }
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->stream3) { (void)hipStreamSynchronize(h->stream3); (void)hipEventDestroy(h->ev_join3); (void)hipStreamDestroy(h->stream3); }
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

int wrsn_set_stream(wrsn_t* h, void* hip_stream) {
    if (!h) return fail(WRSN_ERR_ARG, "null handle");
    h->stream = (hipStream_t)hip_stream;
    return WRSN_OK;
}

int wrsn_set_scenario(wrsn_t* h, int32_t env0, int32_t nenv, const double* node_xy, const double* target_xy,
                      const double* bs_xy, const int32_t* n_node_env, const int32_t* n_target_env,
                      const wrsn_node_spec* node_spec, int32_t node_spec_stride, const wrsn_mc_spec* mc_spec,
                      int32_t mc_spec_stride) {
    if (!h || !node_xy || !target_xy || !bs_xy || !node_spec || !mc_spec) return fail(WRSN_ERR_ARG, "null argument");
    const WrsnDev& d = h->dev;
    if (env0 < 0 || nenv < 1 || env0 + nenv > d.B) return fail(WRSN_ERR_ARG, "environment range out of bounds");
    WRSN_ON_DEVICE(h);
    const size_t NP = d.NP, TP = d.TP;
    std::vector<double> hx(nenv * NP, 0.0), hy(nenv * NP, 0.0), tx(nenv * TP, 0.0), ty(nenv * TP, 0.0);
    std::vector<WrsnEnvConst> ec(nenv);
    for (int e = 0; e < nenv; ++e) {
        const int n = n_node_env ? n_node_env[e] : d.N, t = n_target_env ? n_target_env[e] : d.T;
        if (n < 1 || n > d.N || t < 1 || t > d.T) return fail(WRSN_ERR_ARG, "per-environment node/target count out of range");
        const wrsn_node_spec& ns = node_spec[(size_t)e * (node_spec_stride ? 1 : 0)];
        const wrsn_mc_spec& ms = mc_spec[(size_t)e * (mc_spec_stride ? 1 : 0)];
        if (ns.prob_gp != 1.0) return fail(WRSN_ERR_ARG, "prob_gp != 1 is not supported (Node.py:61 draws Python's MT19937)");
        for (int i = 0; i < n; ++i) { hx[e * NP + i] = node_xy[((size_t)e * d.N + i) * 2]; hy[e * NP + i] = node_xy[((size_t)e * d.N + i) * 2 + 1]; }
        for (int i = 0; i < t; ++i) { tx[e * TP + i] = target_xy[((size_t)e * d.T + i) * 2]; ty[e * TP + i] = target_xy[((size_t)e * d.T + i) * 2 + 1]; }
        WrsnEnvConst& c = ec[e];
        std::memset(&c, 0, sizeof(c));
        c.capacity = ns.capacity; c.threshold = ns.threshold; c.com_range = ns.com_range; c.sen_range = ns.sen_range;
        c.package_size = ns.package_size; c.er = ns.er; c.et = ns.et; c.efs = ns.efs; c.emp = ns.emp; c.max_time = ns.max_time;
        c.mc_capacity = ms.capacity; c.mc_threshold = ms.threshold; c.velocity = ms.velocity; c.pm = ms.pm;
        c.charging_range = ms.charging_range; c.alpha = ms.alpha; c.beta = ms.beta; c.epsilon = ms.epsilon;
        c.bs[0] = bs_xy[e * 2]; c.bs[1] = bs_xy[e * 2 + 1];
        c.warm_up_time = h->cfg.warm_up_time;
        c.n_node = n; c.n_target = t;
    }
    HIPCHK(hipMemcpy(d.node_x + (size_t)env0 * NP, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.node_y + (size_t)env0 * NP, hy.data(), hy.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.target_x + (size_t)env0 * TP, tx.data(), tx.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.target_y + (size_t)env0 * TP, ty.data(), ty.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.ec + env0, ec.data(), ec.size() * sizeof(WrsnEnvConst), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(wrsn_topology_kernel, dim3(nenv), dim3(64), 0, h->stream, h->dev, env0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(ec.data(), d.ec + env0, ec.size() * sizeof(WrsnEnvConst), hipMemcpyDeviceToHost));
    for (int e = 0; e < nenv; ++e) {
        if (ec[e].error == -1) return fail(WRSN_ERR_CAPACITY, "neighbour list capacity exceeded (raise wrsn_cfg.max_degree); env " + std::to_string(env0 + e) + " has " + std::to_string(ec[e].n_edges) + " directed edges");
        if (ec[e].error == -2) return fail(WRSN_ERR_CAPACITY, "coverage list capacity exceeded (raise wrsn_cfg.max_cover); env " + std::to_string(env0 + e));
    }
    {   // size the connected-node lists in LDS to the scenarios (WrsnEnvConst.conn_bound, wrsn_topology_kernel)
        for (int e = 0; e < nenv; ++e) if (ec[e].conn_bound > h->cc_bound) h->cc_bound = ec[e].conn_bound;
        int cc = ((h->cc_bound + 3) / 4) * 4; cc = cc < 4 ? 4 : (cc > WRSN_CONN_CAP ? WRSN_CONN_CAP : cc);
        if (cc != h->dev.CC) { h->dev.CC = cc; int rc2 = configure_launch(h); if (rc2) return rc2; }
    }
    WrsnStepOutDev none; std::memset(&none, 0, sizeof(none));
    int rc = launch_env(h, WRSN_MODE_WARMUP, env0, nenv, nullptr, nullptr, 0, nullptr, none);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->scenario_set = 1;
    return WRSN_OK;
}

int wrsn_reset(wrsn_t* h, const uint8_t* env_mask, const wrsn_step_out* out) {
    if (!h || !out) return fail(WRSN_ERR_ARG, "null argument");
    if (!h->scenario_set) return fail(WRSN_ERR_STATE, "wrsn_set_scenario has not been called");
    WRSN_ON_DEVICE(h);
    // rows of environments the mask leaves out keep every output, agent_id included; what the render pass draws comes
    // from the list the environment kernel writes (WrsnDev.render_agent), not from the caller's agent_id array
    WrsnStepOutDev o; o.agent_id = out->agent_id; o.reward = out->reward; o.terminal = out->terminal;
    o.now = out->now; o.obs = out->obs; o.status = out->status;
    int rc = launch_env(h, WRSN_MODE_RESET, 0, h->dev.B, nullptr, nullptr, 0, env_mask, o);
    if (rc) return rc;
    if (out->obs) return launch_obs(h, h->dev.render_agent, out->obs);
    return WRSN_OK;
}

int wrsn_step(wrsn_t* h, const int32_t* agent_id, const double* action, int32_t auto_reset, const wrsn_step_out* out) {
    if (!h || !out || !agent_id || !action) return fail(WRSN_ERR_ARG, "null argument");
    if (!h->scenario_set) return fail(WRSN_ERR_STATE, "wrsn_set_scenario has not been called");
    WRSN_ON_DEVICE(h);
    // rows with agent_id -2 keep every output (their pending request included)
    WrsnStepOutDev o; o.agent_id = out->agent_id; o.reward = out->reward; o.terminal = out->terminal;
    o.now = out->now; o.obs = out->obs; o.status = out->status;
    return launch_env(h, WRSN_MODE_STEP, 0, h->dev.B, agent_id, action, auto_reset, nullptr, o, out->obs);   // incl. the observations
}

int wrsn_set_obs_reuse(wrsn_t* h, int32_t on) {
    if (!h) return fail(WRSN_ERR_ARG, "null handle");
    h->obs_reuse = on ? 1 : 0;
    return WRSN_OK;
}

int wrsn_set_timing(wrsn_t* h, int32_t on) {
    if (!h) return fail(WRSN_ERR_ARG, "null handle");
    WRSN_ON_DEVICE(h);
    if (on && !h->ev_ok) {
        for (int i = 0; i < 5; ++i) HIPCHK(hipEventCreate(&h->ev[i]));
        h->ev_ok = 1;
    }
    if (!on || !h->timing) h->ev_rec = 0;
    h->timing = on ? 1 : 0;
    return WRSN_OK;
}

int wrsn_kernel_times(wrsn_t* h, float* ms) {
    if (!h || !ms) return fail(WRSN_ERR_ARG, "null argument");
    if (!h->ev_ok || !h->timing) return fail(WRSN_ERR_STATE, "wrsn_set_timing(h, 1) first");
    if (!h->ev_rec) return fail(WRSN_ERR_STATE, "no wrsn_step has run since wrsn_set_timing(h, 1)");
    WRSN_ON_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    ms[0] = ms[1] = ms[2] = ms[3] = 0.f;
    HIPCHK(hipEventElapsedTime(&ms[0], h->ev[0], h->ev[1]));
    HIPCHK(hipEventElapsedTime(&ms[1], h->ev[1], h->ev[2]));
    HIPCHK(hipEventElapsedTime(&ms[2], h->ev[2], h->ev[3]));
    if (h->ev_obs) HIPCHK(hipEventElapsedTime(&ms[3], h->ev[3], h->ev[4]));
    return WRSN_OK;
}

int wrsn_set_step_budget(wrsn_t* h, int32_t work_units) {
    if (!h || work_units < 0) return fail(WRSN_ERR_ARG, "bad step budget");
    WRSN_ON_DEVICE(h);
    h->step_budget = work_units;
    return WRSN_OK;
}

int wrsn_set_step_deadline(wrsn_t* h, int32_t microseconds) {
    if (!h || microseconds < 0 || microseconds > 10000000) return fail(WRSN_ERR_ARG, "bad step deadline");
    h->deadline_ticks = microseconds * 100;                    // wall_clock64 counts at 100 MHz
    return WRSN_OK;
}

int wrsn_density_action(wrsn_t* h, const int32_t* agent_id, const double* dmap, double* action) {
    if (!h || !agent_id || !dmap || !action) return fail(WRSN_ERR_ARG, "null argument");
    if (!h->scenario_set) return fail(WRSN_ERR_STATE, "wrsn_set_scenario has not been called");
    WRSN_ON_DEVICE(h);
    // np.percentile(map, 99.9), method "linear": virtual index (n - 1) q, the two order statistics around it and the weight
    const int n = h->dev.G * h->dev.G;
    const double q = 99.9 / 100.0, vi = (double)(n - 1) * q, lo = std::floor(vi);
    const int n_top = n - (int)lo;                              // elements from the lower order statistic to the maximum
    if (n_top < 1 || n_top > WRSN_DM_KMAX) return fail(WRSN_ERR_ARG, "map size out of range for the percentile selection");
    hipLaunchKernelGGL(wrsn_density_kernel, dim3(h->dev.B), dim3(64), wrsn_density_lds_bytes(), h->stream, h->dev, agent_id, dmap, action,
                       n_top, vi - lo);
    HIPCHK(hipGetLastError());
    return WRSN_OK;
}

int wrsn_rollout_table(wrsn_t* h, double* dst, int32_t zero_after) {
    if (!h || !dst) return fail(WRSN_ERR_ARG, "null argument");
    if (!h->scenario_set) return fail(WRSN_ERR_STATE, "wrsn_set_scenario has not been called");
    WRSN_ON_DEVICE(h);
    hipLaunchKernelGGL(wrsn_rollout_kernel, dim3((h->dev.B + 255) / 256), dim3(256), 0, h->stream, h->dev, dst, (int)zero_after);
    HIPCHK(hipGetLastError());
    return WRSN_OK;
}

static int tr_buffers(const wrsn_transition_buffers* b, WrsnTrBuffers* t) {
    if (!b || b->capacity < 1 || b->action_elems < 1 || !b->pend_state || !b->pend_action || !b->pend_logp || !b->pend_valid || !b->state ||
        !b->action || !b->next_state || !b->reward || !b->logp || !b->now || !b->env || !b->count)
        return fail(WRSN_ERR_ARG, "wrsn_transition_buffers: null pointer or empty geometry");
    t->capacity = b->capacity; t->action_elems = b->action_elems;
    t->pend_state = b->pend_state; t->pend_action = b->pend_action; t->pend_logp = b->pend_logp; t->pend_valid = b->pend_valid;
    t->state = b->state; t->action = b->action; t->next_state = b->next_state; t->reward = b->reward; t->logp = b->logp;
    t->now = b->now; t->env = b->env; t->count = b->count;
    return 0;
}

int wrsn_rollout_record(wrsn_t* h, const wrsn_transition_buffers* buf, const int32_t* agent_id, const float* action, const float* logp,
                        const float* obs) {
    if (!h || !agent_id || !action || !logp || !obs) return fail(WRSN_ERR_ARG, "null argument");
    WrsnTrBuffers t; int rc = tr_buffers(buf, &t); if (rc) return rc;
    WRSN_ON_DEVICE(h);
    hipLaunchKernelGGL(wrsn_tr_record_kernel, dim3(h->dev.B), dim3(256), 0, h->stream, h->dev.B, h->dev.M, h->dev.G, t, agent_id, action, logp, obs);
    HIPCHK(hipGetLastError());
    return WRSN_OK;
}

int wrsn_rollout_collect(wrsn_t* h, const wrsn_transition_buffers* buf, const wrsn_step_out* out) {
    if (!h || !out || !out->agent_id || !out->reward || !out->terminal || !out->now || !out->status || !out->obs)
        return fail(WRSN_ERR_ARG, "wrsn_rollout_collect needs every wrsn_step_out field");
    WrsnTrBuffers t; int rc = tr_buffers(buf, &t); if (rc) return rc;
    WRSN_ON_DEVICE(h);
    hipLaunchKernelGGL(wrsn_tr_collect_kernel, dim3(h->dev.B), dim3(256), 16, h->stream, h->dev.B, h->dev.M, h->dev.G, t, out->agent_id,
                       out->reward, out->now, h->dev.row_state, out->obs);
    HIPCHK(hipGetLastError());
    return WRSN_OK;
}

int wrsn_render(wrsn_t* h, const int32_t* agent_id, float* obs) {
    if (!h || !agent_id || !obs) return fail(WRSN_ERR_ARG, "null argument");
    if (!h->scenario_set) return fail(WRSN_ERR_STATE, "wrsn_set_scenario has not been called");
    WRSN_ON_DEVICE(h);
    return launch_obs(h, agent_id, obs);
}

int wrsn_sync(wrsn_t* h) {
    if (!h) return fail(WRSN_ERR_ARG, "null handle");
    WRSN_ON_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    return WRSN_OK;
}

int wrsn_peek(wrsn_t* h, int32_t what, void* dst) {
    if (!h || !dst) return fail(WRSN_ERR_ARG, "null argument");
    const WrsnDev& d = h->dev;
    const size_t B = d.B, NP = d.NP, N = d.N;
    WRSN_ON_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    auto node_f64 = [&](const double* src) -> int {
        std::vector<double> tmp(B * NP);
        HIPCHK(hipMemcpy(tmp.data(), src, tmp.size() * 8, hipMemcpyDeviceToHost));
        double* o = (double*)dst;
        for (size_t e = 0; e < B; ++e) for (size_t i = 0; i < N; ++i) o[e * N + i] = tmp[e * NP + i];
        return 0;
    };
    auto node_i32 = [&](const int32_t* src, int mode) -> int {
        std::vector<int32_t> tmp(B * NP);
        HIPCHK(hipMemcpy(tmp.data(), src, tmp.size() * 4, hipMemcpyDeviceToHost));
        int32_t* o = (int32_t*)dst;
        for (size_t e = 0; e < B; ++e)
            for (size_t i = 0; i < N; ++i) {
                int32_t v = tmp[e * NP + i];
                o[e * N + i] = mode == 0 ? v : (mode == 1 ? (v & 1) : ((v >> 1) - 1));
            }
        return 0;
    };
    switch (what) {
    case WRSN_PEEK_NODE_ENERGY: return node_f64(d.live.E);
    case WRSN_PEEK_NODE_CS: return node_f64(d.live.CS);
    case WRSN_PEEK_NODE_RR: return node_f64(d.live.RR);
    case WRSN_PEEK_NODE_STATUS: return node_i32(d.live.ls, 1);
    case WRSN_PEEK_NODE_LEVEL: return node_i32(d.live.ls, 2);
    case WRSN_PEEK_NODE_NCOVER: return node_i32(d.ncov, 0);
    case WRSN_PEEK_NODE_DIRECT: return node_i32(d.nflags, 1);
    case WRSN_PEEK_NODE_DEGREE: {
        std::vector<int32_t> tmp(B * (NP + 1));
        HIPCHK(hipMemcpy(tmp.data(), d.nb_off, tmp.size() * 4, hipMemcpyDeviceToHost));
        int32_t* o = (int32_t*)dst;
        for (size_t e = 0; e < B; ++e) for (size_t i = 0; i < N; ++i) o[e * N + i] = tmp[e * (NP + 1) + i + 1] - tmp[e * (NP + 1) + i];
        return 0; }
    case WRSN_PEEK_MC: {
        std::vector<WrsnEnvDyn> dy(B);
        HIPCHK(hipMemcpy(dy.data(), d.live.dyn, B * sizeof(WrsnEnvDyn), hipMemcpyDeviceToHost));
        double* o = (double*)dst;
        for (size_t e = 0; e < B; ++e)
            for (int m = 0; m < d.M; ++m) {
                const WrsnAgent& a = dy[e].ag[m]; double* q = o + (e * d.M + m) * 16;
                q[0] = a.loc[0]; q[1] = a.loc[1]; q[2] = a.energy; q[3] = a.status; q[4] = a.type_charging;
                q[5] = a.cur[0]; q[6] = a.cur[1]; q[7] = a.cur[2]; q[8] = a.n_conn; q[9] = a.excl; q[10] = a.prev_minfit;
                q[11] = a.action[0]; q[12] = a.action[1]; q[13] = a.action[2]; q[14] = 0; q[15] = 0;
            }
        return 0; }
    case WRSN_PEEK_ENV: {
        std::vector<WrsnEnvDyn> dy(B); std::vector<WrsnEnvConst> ec(B);
        HIPCHK(hipMemcpy(dy.data(), d.live.dyn, B * sizeof(WrsnEnvDyn), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ec.data(), d.ec, B * sizeof(WrsnEnvConst), hipMemcpyDeviceToHost));
        double* o = (double*)dst;
        for (size_t e = 0; e < B; ++e) {
            double* q = o + e * 16;
            q[0] = ec[e].frame[0]; q[1] = ec[e].frame[1]; q[2] = ec[e].frame[2]; q[3] = ec[e].frame[3];
            q[4] = ec[e].density; q[5] = ec[e].moving_time_max; q[6] = ec[e].charging_time_max; q[7] = ec[e].avg_nodes_agent;
            q[8] = dy[e].now; q[9] = dy[e].alive; q[10] = (double)dy[e].n_ticks; q[11] = (double)dy[e].n_exact;
            q[12] = (double)dy[e].n_events; q[13] = dy[e].last_minfit; q[14] = ec[e].n_edges; q[15] = ec[e].n_cover;
        }
        return 0; }
    case WRSN_PEEK_TARGETS_ACTIVE: {
        // Network.setLevels marks the targets of every node it reaches (Network.py:45-55); the level words keep what the
        // last setLevels found (a node that died since keeps its level until the next one, exactly like node.level)
        const size_t TP = d.TP, T = d.T;
        std::vector<int32_t> ls(B * NP), off(B * (TP + 1)), idx(B * (size_t)d.CCAP);
        std::vector<WrsnEnvConst> ec(B);
        HIPCHK(hipMemcpy(ls.data(), d.live.ls, ls.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(off.data(), d.tc_off, off.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(idx.data(), d.tc_idx, idx.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ec.data(), d.ec, B * sizeof(WrsnEnvConst), hipMemcpyDeviceToHost));
        int32_t* o = (int32_t*)dst;
        for (size_t e = 0; e < B; ++e)
            for (size_t t = 0; t < T; ++t) {
                int32_t act = 0;
                if ((int)t < ec[e].n_target)
                    for (int32_t p = off[e * (TP + 1) + t]; p < off[e * (TP + 1) + t + 1]; ++p)
                        if ((ls[e * NP + idx[e * (size_t)d.CCAP + p]] >> 1) >= 2) act = 1;     // level >= 1
                o[e * T + t] = act;
            }
        return 0; }
    case 10: {   // diagnostic builds (-DWRSN_PROFILE): int64 [B,25] per-phase cycle totals (+ whole kernel); zeros otherwise
        HIPCHK(hipMemcpy(dst, d.counters, B * 25 * sizeof(int64_t), hipMemcpyDeviceToHost));
        return 0; }
    default: return fail(WRSN_ERR_ARG, "unknown peek selector");
    }
}

int wrsn_counters(wrsn_t* h, int64_t* dst) {
    if (!h || !dst) return fail(WRSN_ERR_ARG, "null argument");
    const size_t B = h->dev.B;
    WRSN_ON_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<WrsnEnvDyn> dy(B);
    HIPCHK(hipMemcpy(dy.data(), h->dev.live.dyn, B * sizeof(WrsnEnvDyn), hipMemcpyDeviceToHost));
    // n_ticks / n_exact / n_events restart at every reset (they live in the snapshot); n_steps is cumulative
    for (int k = 0; k < 8; ++k) dst[k] = 0;
    for (size_t e = 0; e < B; ++e) {
        dst[0] += dy[e].n_ticks; dst[1] += dy[e].n_exact; dst[2] += dy[e].n_events; dst[3] += dy[e].n_steps;
        dst[4] += dy[e].tot_ticks; dst[5] += dy[e].tot_zero_steps;
    }
    return WRSN_OK;
}

int wrsn_synth_network(uint64_t seed, int32_t n_node, int32_t n_target, double side, double com_range,
                       double sen_range, double* node_xy, double* target_xy, double* bs_xy) {
    if (n_node < 2 || n_target < 1 || !node_xy || !target_xy || !bs_xy || !(com_range > 0) || !(sen_range > 0))
        return fail(WRSN_ERR_ARG, "bad generator argument");
    if (side <= 0) side = 1000.0 * std::fmax(1.0, std::sqrt(n_node / 200.0));
    Rng rng(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull);
    const double com = com_range, sen = sen_range;
    const double bx = side / 2, by = side / 2;
    const double hop_lo = 0.62 * com, hop_hi = 0.995 * com, min_sep = 0.56 * com;
    std::vector<int> tips;
    // Nodes next to the base station: 2..4 up to 512 nodes (the shipped scenarios have 2..3; networks of that size are unchanged), one more per
    // 125 nodes above 256 (at most six) for the larger ones -- every packet of the network passes one of them, and with three of them a 1 000-node /
    // 1 000-target network loses its first relay 40 s after the warm-up, which makes a benchmark of resets, not of the dynamics
    int n = 0, n_direct = 2 + rng.below(3);
    if (n_node > 512) { n_direct += (n_node - 256) / 125; if (n_direct > 6) n_direct = 6; }   // (six fit the ring around the base station at the minimum separation)
    long tries = 0;
    while (n < n_node) {
        if (++tries > 4000000L) return fail(WRSN_ERR_ARG, "synthetic generator could not place the nodes (field too small?)");
        double px, py; int par = -1;
        if (n < n_direct) {
            double ang = rng.uni(0, 6.283185307179586), r = rng.uni(0.35 * com, 0.95 * com);
            px = bx + r * std::cos(ang); py = by + r * std::sin(ang);
        } else {
            if (!tips.empty() && rng.uni() < 0.93) par = tips[rng.below((int)tips.size())];
            else par = rng.below(n);
            double ox = node_xy[2 * par] - bx, oy = node_xy[2 * par + 1] - by;
            double ang = std::atan2(oy, ox) + 0.75 * rng.normal(), r = rng.uni(hop_lo, hop_hi);
            px = node_xy[2 * par] + r * std::cos(ang); py = node_xy[2 * par + 1] + r * std::sin(ang);
        }
        if (px < 0 || px > side || py < 0 || py > side) continue;
        bool ok = true;
        for (int k = 0; k < n && ok; ++k) { double dx = node_xy[2 * k] - px, dy = node_xy[2 * k + 1] - py; if (dx * dx + dy * dy < min_sep * min_sep) ok = false; }
        if (!ok) continue;
        node_xy[2 * n] = px; node_xy[2 * n + 1] = py;
        for (size_t k = 0; k < tips.size(); ++k) if (tips[k] == par) { tips.erase(tips.begin() + k); break; }
        tips.push_back(n);
        if (tips.size() > 24) tips.erase(tips.begin());
        ++n;
    }
    // targets: inside 0.93 * sensing range of an owner node, biased towards the outer nodes
    std::vector<double> cum(n_node);
    double dmax = 0; for (int i = 0; i < n_node; ++i) dmax = std::fmax(dmax, std::hypot(node_xy[2 * i] - bx, node_xy[2 * i + 1] - by));
    double acc = 0; for (int i = 0; i < n_node; ++i) { acc += 0.25 + std::hypot(node_xy[2 * i] - bx, node_xy[2 * i + 1] - by) / dmax; cum[i] = acc; }
    for (int t = 0; t < n_target; ++t) {
        double u = rng.uni() * acc; int lo = 0, hi = n_node - 1;
        while (lo < hi) { int mid = (lo + hi) / 2; if (cum[mid] < u) lo = mid + 1; else hi = mid; }
        double ang = rng.uni(0, 6.283185307179586), r = 0.93 * sen * std::sqrt(rng.uni());
        target_xy[2 * t] = node_xy[2 * lo] + r * std::cos(ang); target_xy[2 * t + 1] = node_xy[2 * lo + 1] + r * std::sin(ang);
    }
    bs_xy[0] = bx; bs_xy[1] = by;
    return WRSN_OK;
}

}  // extern "C"
