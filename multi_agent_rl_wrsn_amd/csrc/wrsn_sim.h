// wrsn_sim.h -- gfx950 device code of the WRSN environment step path.
//
// One 64-lane wavefront owns one environment for the whole launch.
//   * Node i lives in lane (i & 63), register slot (i >> 6): energy, consumption rate and the cached
//     per-tick drains are float64 VGPRs (MI355X issues v_fma_f64 at half the f32 rate: exact-looking
//     trajectories for a few cycles per tick).
//   * LDS holds what needs random access: per-node charging rate, level/alive word, cached receiver,
//     scratch for the level BFS / route aggregation / widest-path relaxation / exact packet walk, the
//     chargers, their operate_step processes and their connected-node lists.
//   * Lane 0 is the scalar event processor: it alone reads and writes charger / process / condition state
//     and pops pending items in the reference's discrete-event order (SimPy key (time, priority,
//     insertion id)).  Whenever an item needs O(N) work it posts a request and the whole wave services it
//     between two workgroup barriers: node half/full second, reward priorities, level BFS, the charger
//     energy pre-check, the connected-node scan.  Single writer, barrier-delimited hand-offs.
//
// Reference being restated (read-only at /root/reference):
//   physical_env/network/Node.py, Network.py, BaseStation.py, physical_env/mc/MobileCharger.py,
//   rl_env/WRSN.py.  Line citations are on the individual functions.
#pragma once
#include "wrsn_types.h"

#define WDEV __device__ __forceinline__
// Diagnostic build only (-DWRSN_PROFILE, tools/build_profile.sh): per-phase cycle totals per environment.  Stamps go
// to a buffer of their own (WrsnDev.counters) and no output is computed from them; the product build has none.
#if defined(WRSN_PROFILE) && WRSN_PROFILE == 4
// level 4: breakdown of the scalar event processor and of the service loop (slots listed in tools/diag_scalar.py)
#define WRSN_P4_MARK(var) const long long var = clock64();
#define WRSN_P4_SPAN(slot, a, b) prof_[slot] += (b) - (a);
#define WRSN_P4_CNT(slot, v) prof_[slot] += (v);
#else
#define WRSN_P4_MARK(var)
#define WRSN_P4_SPAN(slot, a, b)
#define WRSN_P4_CNT(slot, v)
#endif
#if defined(WRSN_PROFILE) && WRSN_PROFILE == 4
#define WRSN_PROF_DECL long long prof_[24]; long long prof_t_;
#define WRSN_PROF_ZERO for (int q_ = 0; q_ < 24; ++q_) prof_[q_] = 0;
#define WRSN_PROF_MARK(var)
#define WRSN_PROF_SPAN(slot, a, b)
#define WRSN_PROF_T0
#define WRSN_PROF_ADD(slot)
#define WRSN_PROF_CNT(slot, v)
#define WRSN_PROF_EV(slot, v)
#elif defined(WRSN_PROFILE) && WRSN_PROFILE >= 2
// level 2: event histogram instead of phase timers (slot = process hop fired; 19.. = service kinds)
#define WRSN_PROF_DECL long long prof_[24]; long long prof_t_;
#define WRSN_PROF_ZERO for (int q_ = 0; q_ < 24; ++q_) prof_[q_] = 0;
#define WRSN_PROF_MARK(var)
#define WRSN_PROF_SPAN(slot, a, b)
#define WRSN_PROF_T0
#define WRSN_PROF_ADD(slot)
#define WRSN_PROF_CNT(slot, v)
#define WRSN_PROF_EV(slot, v) prof_[slot] += (v);
#elif defined(WRSN_PROFILE)
#define WRSN_PROF_DECL long long prof_[24]; long long prof_t_;
#define WRSN_PROF_ZERO for (int q_ = 0; q_ < 24; ++q_) prof_[q_] = 0;
#define WRSN_PROF_MARK(var) const long long var = clock64();
#define WRSN_PROF_SPAN(slot, a, b) prof_[slot] += (b) - (a);
#define WRSN_PROF_T0 const long long pt0_ = clock64();
#define WRSN_PROF_ADD(slot) prof_[slot] += clock64() - pt0_;
#define WRSN_PROF_CNT(slot, v) prof_[slot] += (v);
#define WRSN_PROF_EV(slot, v)
#else
#define WRSN_PROF_DECL
#define WRSN_PROF_ZERO
#define WRSN_PROF_T0
#define WRSN_PROF_ADD(slot)
#define WRSN_PROF_CNT(slot, v)
#define WRSN_PROF_MARK(var)
#define WRSN_PROF_SPAN(slot, a, b)
#define WRSN_PROF_EV(slot, v)
#endif
// profile slots: 0 scalar_run  1 grid_run  2 node_half(fast)  3 update_reward  4 exact_walk  5 rebuild_cache  6 set_levels
//                7 min_fitness 8 precheck  9 conn_build  10 load  11 store  12 #services  13 #fused seconds  14 #jumped seconds  15 #generic items
// time-sliced launches (100 MHz ticks): a block that starts less than WRSN_PULL_MARGIN before the deadline does not take its environment (a
// visit costs a load and a store whatever it achieves); a packet-exact second (~70 us) is not begun less than WRSN_EXACT_MARGIN before it,
// unless it is the first item of the visit (every environment a launch takes advances)
#ifndef WRSN_PULL_MARGIN
#ifndef WRSN_NBREG_MAX
#define WRSN_NBREG_MAX 6          // up to this many node slots per lane the packed neighbour ids stay in registers during a sweep
#endif
#define WRSN_PULL_MARGIN 3000
#endif
#ifndef WRSN_TICKS_PER_FUSED_SECOND
#define WRSN_TICKS_PER_FUSED_SECOND 35       // what a second of the time-parallel steady path takes (0.35 us), for the clamp of a fused run to the time left
#endif
#ifndef WRSN_EXACT_MARGIN
#define WRSN_EXACT_MARGIN 6000
#endif
#define WRSN_URGENT 0
#define WRSN_NORMAL 1
#define WRSN_INF (__builtin_inf())

enum {
    PC_NONE = 0, PC_P_INIT, PC_MOVE_INIT, PC_MSTEP_INIT, PC_MSTEP_TIMEOUT, PC_MSTEP_DONE,
    PC_MOVE_DEADWAIT, PC_MOVE_DONE, PC_RECH_INIT, PC_RECH_TIMEOUT, PC_RECH_DONE,
    PC_CHG_INIT, PC_CSTEP_INIT, PC_CSTEP_TIMEOUT, PC_CSTEP_DONE, PC_CHG_DEADWAIT, PC_CHG_DONE,
    PC_P_DONE, PC_FINISHED
};

// requests lane 0 posts to the wave
enum { REQ_STOP = 0, REQ_GRID, REQ_PRECHECK, REQ_CONN };

// ------------------------------------------------------------------ wave-level primitives (64 lanes)
// Reductions run on the DPP crossbar of the VALU (quad_perm / row_half_mirror / row_mirror / row_bcast15 / row_bcast31:
// a handful of cycles per step) instead of ds_bpermute round trips through the LDS; the total lands in lane 63 and
// is broadcast with v_readlane.
#define WRSN_DPP_QP_1032 0xB1
#define WRSN_DPP_QP_2301 0x4E
#define WRSN_DPP_ROW_HALF_MIRROR 0x141
#define WRSN_DPP_ROW_MIRROR 0x140
#define WRSN_DPP_ROW_BCAST15 0x142
#define WRSN_DPP_ROW_BCAST31 0x143

template <int CTRL, int ROW_MASK>
WDEV double dpp_f64(double ident, double v) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
WDEV double lane63_f64(double v) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
WDEV double wv_sum(double v) {
    v += dpp_f64<WRSN_DPP_QP_1032, 0xf>(0.0, v);
    v += dpp_f64<WRSN_DPP_QP_2301, 0xf>(0.0, v);
    v += dpp_f64<WRSN_DPP_ROW_HALF_MIRROR, 0xf>(0.0, v);
    v += dpp_f64<WRSN_DPP_ROW_MIRROR, 0xf>(0.0, v);
    v += dpp_f64<WRSN_DPP_ROW_BCAST15, 0xa>(0.0, v);
    v += dpp_f64<WRSN_DPP_ROW_BCAST31, 0xc>(0.0, v);
    return lane63_f64(v);
}
WDEV void wv_sum2(double& a, double& b) {                   // two independent sums, chains interleaved
    a += dpp_f64<WRSN_DPP_QP_1032, 0xf>(0.0, a); b += dpp_f64<WRSN_DPP_QP_1032, 0xf>(0.0, b);
    a += dpp_f64<WRSN_DPP_QP_2301, 0xf>(0.0, a); b += dpp_f64<WRSN_DPP_QP_2301, 0xf>(0.0, b);
    a += dpp_f64<WRSN_DPP_ROW_HALF_MIRROR, 0xf>(0.0, a); b += dpp_f64<WRSN_DPP_ROW_HALF_MIRROR, 0xf>(0.0, b);
    a += dpp_f64<WRSN_DPP_ROW_MIRROR, 0xf>(0.0, a); b += dpp_f64<WRSN_DPP_ROW_MIRROR, 0xf>(0.0, b);
    a += dpp_f64<WRSN_DPP_ROW_BCAST15, 0xa>(0.0, a); b += dpp_f64<WRSN_DPP_ROW_BCAST15, 0xa>(0.0, b);
    a += dpp_f64<WRSN_DPP_ROW_BCAST31, 0xc>(0.0, a); b += dpp_f64<WRSN_DPP_ROW_BCAST31, 0xc>(0.0, b);
    a = lane63_f64(a); b = lane63_f64(b);
}
template <int CTRL, int ROW_MASK>
WDEV float dpp_f32(float ident, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
WDEV float wv_sumf(float v) {
    v += dpp_f32<WRSN_DPP_QP_1032, 0xf>(0.f, v);
    v += dpp_f32<WRSN_DPP_QP_2301, 0xf>(0.f, v);
    v += dpp_f32<WRSN_DPP_ROW_HALF_MIRROR, 0xf>(0.f, v);
    v += dpp_f32<WRSN_DPP_ROW_MIRROR, 0xf>(0.f, v);
    v += dpp_f32<WRSN_DPP_ROW_BCAST15, 0xa>(0.f, v);
    v += dpp_f32<WRSN_DPP_ROW_BCAST31, 0xc>(0.f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
WDEV void wv_sum2f(float& a, float& b) {
    a += dpp_f32<WRSN_DPP_QP_1032, 0xf>(0.f, a); b += dpp_f32<WRSN_DPP_QP_1032, 0xf>(0.f, b);
    a += dpp_f32<WRSN_DPP_QP_2301, 0xf>(0.f, a); b += dpp_f32<WRSN_DPP_QP_2301, 0xf>(0.f, b);
    a += dpp_f32<WRSN_DPP_ROW_HALF_MIRROR, 0xf>(0.f, a); b += dpp_f32<WRSN_DPP_ROW_HALF_MIRROR, 0xf>(0.f, b);
    a += dpp_f32<WRSN_DPP_ROW_MIRROR, 0xf>(0.f, a); b += dpp_f32<WRSN_DPP_ROW_MIRROR, 0xf>(0.f, b);
    a += dpp_f32<WRSN_DPP_ROW_BCAST15, 0xa>(0.f, a); b += dpp_f32<WRSN_DPP_ROW_BCAST15, 0xa>(0.f, b);
    a += dpp_f32<WRSN_DPP_ROW_BCAST31, 0xc>(0.f, a); b += dpp_f32<WRSN_DPP_ROW_BCAST31, 0xc>(0.f, b);
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63)); b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
}
WDEV double wv_max(double v) {
    const double id = -WRSN_INF;
    v = fmax(v, dpp_f64<WRSN_DPP_QP_1032, 0xf>(id, v));
    v = fmax(v, dpp_f64<WRSN_DPP_QP_2301, 0xf>(id, v));
    v = fmax(v, dpp_f64<WRSN_DPP_ROW_HALF_MIRROR, 0xf>(id, v));
    v = fmax(v, dpp_f64<WRSN_DPP_ROW_MIRROR, 0xf>(id, v));
    v = fmax(v, dpp_f64<WRSN_DPP_ROW_BCAST15, 0xa>(id, v));
    v = fmax(v, dpp_f64<WRSN_DPP_ROW_BCAST31, 0xc>(id, v));
    return lane63_f64(v);
}
WDEV double wv_min(double v) { return -wv_max(-v); }
WDEV int wv_sumi(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, WRSN_DPP_QP_1032, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, WRSN_DPP_QP_2301, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, WRSN_DPP_ROW_HALF_MIRROR, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, WRSN_DPP_ROW_MIRROR, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, WRSN_DPP_ROW_BCAST15, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, WRSN_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return __builtin_amdgcn_readlane(v, 63);
}
WDEV bool wv_any(bool p) { return __ballot(p) != 0ull; }
WDEV int wv_scan_incl(int v, int lane) {
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o); if (lane >= o) v += t; }
    return v;
}

WDEV double dist2(double ax, double ay, double bx, double by) {
    double dx = ax - bx, dy = ay - by;
    return sqrt(dx * dx + dy * dy);
}

// 1/x to ~1e-15 relative: hardware estimate + two Newton steps (the consumers tolerate 1e-5)
WDEV double fast_rcp(double x) {
    double r = (double)(1.0f / (float)x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}

// ------------------------------------------------------------------ the per-environment simulator
// ------------------------------------------------------------------ wave-uniform values
// A value every lane holds identically (clocks, sequence numbers, counters, constants) is moved to scalar registers with
// v_readfirstlane: it then costs SGPRs (which spill into VGPR lanes, 64 per register) instead of a VGPR per value.
WDEV int wu(int v) { return __builtin_amdgcn_readfirstlane(v); }
WDEV double wu(double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); }
WDEV int64_t wu(int64_t v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll)); const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((int64_t)hi << 32) | (int64_t)lo;
}

// ------------------------------------------------------------------ global address space
// A pointer the kernel loads from its descriptor (WrsnDev) is a flat pointer to the compiler: every access through it is a flat_load that
// also counts against the LDS queue.  The arrays of a handle live in HBM: say so, and the access is a global_load with a scalar base.
#ifndef WRSN_GLOBAL_AS
#define WRSN_GLOBAL_AS __attribute__((address_space(1)))
#endif
template <typename T> WDEV const T WRSN_GLOBAL_AS* wrsn_global(const T* p) { return (const T WRSN_GLOBAL_AS*)p; }
template <typename T> WDEV T WRSN_GLOBAL_AS* wrsn_global(T* p) { return (T WRSN_GLOBAL_AS*)p; }

// 16-byte rows of the packed static tables (eight 16-bit node ids): one dwordx4 load from the global address space
struct alignas(16) WrsnU4 { uint32_t x, y, z, w; };
#ifndef WRSN_LD_U4_DEFINED
WDEV WrsnU4 wrsn_ld_u4(const WrsnU4 WRSN_GLOBAL_AS* p) {
    typedef uint32_t v4_ __attribute__((ext_vector_type(4)));
    const v4_ t = *(const v4_ WRSN_GLOBAL_AS*)p;
    WrsnU4 r; r.x = t.x; r.y = t.y; r.z = t.z; r.w = t.w; return r;
}
#endif

// ------------------------------------------------------------------ LDS gathers
// Eight data-dependent LDS words in ONE round trip: the ds_read are issued back to back and waited for once.  Written
// as inline assembly because the register-pressure heuristics of the scheduler otherwise emit read / wait / use eight
// times in a row (the graph sweeps below spent most of their time in exactly that).
#ifndef WRSN_LDS_GATHER_DEFINED
WDEV void wrsn_lds_gather8_b32(const int32_t* base, const int (&idx)[8], int (&out)[8]) {
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (uint32_t)(uintptr_t)(base + idx[k]);      // low half of a flat LDS address = LDS offset
    asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %9\n\tds_read_b32 %2, %10\n\tds_read_b32 %3, %11\n\t"
                 "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]), "=&v"(out[7])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
}
WDEV void wrsn_lds_gather8_b64(const double* base, const int (&idx)[8], double (&out)[8]) {
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (uint32_t)(uintptr_t)(base + idx[k]);
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %9\n\tds_read_b64 %2, %10\n\tds_read_b64 %3, %11\n\t"
                 "ds_read_b64 %4, %12\n\tds_read_b64 %5, %13\n\tds_read_b64 %6, %14\n\tds_read_b64 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]), "=&v"(out[7])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
}
#endif

// Sim<NPL, true> holds everything; Sim<NPL, false> -- the simulator of the step kernel's common path -- has no code for the three rare,
// register-hungry services: the level BFS after a death (set_levels), the routing-cache rebuild (rebuild_cache) and the packet-exact
// second (exact_walk).  When a grid item needs one of them it stops in front of that item exactly like an environment that ran out
// of its launch budget (`need_heavy`), the step kernel stores the environment, calls wrsn_step_env_full -- a NOINLINE device function
// with the full simulator, which loads the environment, runs that one item and suspends (or finishes the step) -- and goes on.
template <int NPL, bool HEAVY = true>
struct Sim {
    // identity / geometry.  Pointers are not kept as members: they are derived on demand from the device descriptor
    // (scalar loads from the constant cache) and from the LDS base, which keeps the hot per-second loop small in
    // registers.
    int env, lane, N, T, M, NP, CC, use_snap;           // CC: capacity of a connected-node list in LDS (WrsnDev.CC)
    const WrsnDev* dp;
    double* smem_;
    // node registers (per lane)
    double E[NPL], CS[NPL], d1[NPL], d2[NPL];
    unsigned am;                       // bit j: node j*64+lane alive
    double cap, thr, max_time, inv_a_b2;   // inv_a_b2 = beta^2 / alpha (1 / the maximal charging rate, WRSN.py:125)
    // wave-uniform registers: identical in every lane.  `now` / `seq` are also advanced by lane 0 while it fires
    // charger events; they are re-broadcast through the mailbox at every hand-off.
    double opmax;
    int64_t n_ticks, n_exact;
    int alive, levels_dirty, cache_dirty, irreg, ring_len, ring_head, safe_ticks, log_pending, err;
    double now; int64_t seq;
    double net_time; int64_t net_seq; double ur_time; int64_t ur_seq; double node_time; int64_t node_seq;   // pending grid items
    int net_phase, net_active, node_phase, frozen, deaths_flag;
    int uns_cnt, uns_node;                                   // walk_range: how many nodes were unsafe, and (one of) them
    double teps;                                             // energy margin of the "may a node run dry" tests
    int dirty;                                               // which state arrays differ from HBM: 1 routing (d1, d2, rcv), 2 level/alive words, 4 CS
    int work, budget;                                        // work units spent in this launch / allowed (0 = unlimited); wave-uniform
    long long t_deadline;                                    // wall clock at which this launch stops taking new grid items (0 = none); wave-uniform
    long long t_exact;                                       // ... at which it stops beginning packet-exact seconds (time-sliced launches; 0 = none)
    int fit_dirty, map1_valid;                               // a grid service ran since last_minfit was evaluated / map 1 of the observation still stands (wave-uniform)
    int need_heavy;                                          // HEAVY = false: the next grid item needs a service this variant has no code for (1); either
                                                             // variant: a packet-exact second was put off because the time-sliced launch is about to end (2)
    int n_items;                                             // grid items completed in this visit (wave-uniform)
    double last_minfit;
    WRSN_PROF_DECL

    using U4 = WrsnU4;
    static constexpr int kChgMax = WRSN_CHG_MAX(64 * NPL);
    // lane-0 bookkeeping of the scalar event processor lives in LDS, not in registers
    struct Scalar { double ev_time, ev2_time; int64_t ev_seq, n_events; int32_t L, pend, pend_idx, ev_valid, ev_kind, ev_idx, ev_prio, ev_uf; };
    static_assert(sizeof(Scalar) == WRSN_LDS_SCALAR_BYTES, "wrsn_lds_bytes must match");

    // ---- static topology / per-environment arrays in HBM
    // per-environment constants are staged in LDS by bind(): a plain global load of them costs a full memory round
    // trip (the backend cannot use scalar loads on mutable global memory)
    WDEV const WrsnEnvConst* EC() const { return (const WrsnEnvConst*)(SS() + 1); }
    WDEV const double WRSN_GLOBAL_AS* NX() const { return wrsn_global(dp->node_x + (size_t)env * NP); }
    WDEV const double WRSN_GLOBAL_AS* NY() const { return wrsn_global(dp->node_y + (size_t)env * NP); }
    WDEV const double WRSN_GLOBAL_AS* DBS() const { return wrsn_global(dp->dist_bs + (size_t)env * NP); }
    WDEV const int32_t WRSN_GLOBAL_AS* NB_OFF() const { return wrsn_global(dp->nb_off + (size_t)env * (NP + 1)); }
    WDEV const int32_t WRSN_GLOBAL_AS* NB_IDX() const { return wrsn_global(dp->nb_idx + (size_t)env * dp->ECAP); }
    WDEV const double WRSN_GLOBAL_AS* NB_DIST() const { return wrsn_global(dp->nb_dist + (size_t)env * dp->ECAP); }
    WDEV const int32_t WRSN_GLOBAL_AS* TC_OFF() const { return wrsn_global(dp->tc_off + (size_t)env * (dp->TP + 1)); }
    WDEV const int32_t WRSN_GLOBAL_AS* TC_IDX() const { return wrsn_global(dp->tc_idx + (size_t)env * dp->CCAP); }
    WDEV const int32_t WRSN_GLOBAL_AS* NCOV() const { return wrsn_global(dp->ncov + (size_t)env * NP); }
    WDEV const int32_t WRSN_GLOBAL_AS* NFLAGS() const { return wrsn_global(dp->nflags + (size_t)env * NP); }   // bit 0 direct node, bit 1 more than 8 neighbours, bits 8.. len(listTargets)
    WDEV const U4 WRSN_GLOBAL_AS* NBP() const { return wrsn_global((const U4*)(dp->nbp + (size_t)env * NP * 4)); }
    WDEV const double WRSN_GLOBAL_AS* NBP_ES() const { return wrsn_global(dp->nbp_es + (size_t)env * NP * 8); }
    WDEV const double WRSN_GLOBAL_AS* ES_BS() const { return wrsn_global(dp->es_bs + (size_t)env * NP); }
    WDEV const U4 WRSN_GLOBAL_AS* ADJM() const { return wrsn_global((const U4*)(dp->adjm + (size_t)env * NP * 8)); }   // [NP][4] uint64 neighbourhood masks (nodes 0..255)
    WDEV const U4 WRSN_GLOBAL_AS* TCP() const { return wrsn_global((const U4*)(dp->tcp + (size_t)env * dp->TP * 4)); }
    WDEV double WRSN_GLOBAL_AS* RING() const { return wrsn_global((use_snap ? dp->snap.ring : dp->live.ring) + (size_t)env * WRSN_RING * NP); }
    WDEV double WRSN_GLOBAL_AS* LOGBUF() const { return wrsn_global((use_snap ? dp->snap.logbuf : dp->live.logbuf) + (size_t)env * NP); }
    // ---- LDS carve-up (must match wrsn_lds_bytes)
    WDEV double* SU() const { return smem_; }
    WDEV int32_t* SLS() const { return (int32_t*)(smem_ + 2 * NP); }
    WDEV int32_t* SRCV() const { return SLS() + NP; }
    // time-parallel steady batch: float CS per node, records / per-second table of the (few) nodes being charged
    WDEV float* SCSF() const { return (float*)(smem_ + 3 * NP); }
    WDEV double* SCHGREC() const { return (double*)SCSF() + NP / 2; }                    // [CHG_MAX][8]: E, d1, d2, rr, node, E_final, CS
    WDEV double* SCHGTAB() const { return SCHGREC() + 8 * kChgMax; }                 // [CHG_MAX][64] energy at the reward instant of second s
    WDEV WrsnAgent* SAG() const { return (WrsnAgent*)(SCHGTAB() + 64 * kChgMax); }
    WDEV WrsnThread* STH() const { return (WrsnThread*)(SAG() + M); }
    WDEV double* SCT() const { return (double*)(STH() + 2 * M); }
    WDEV int64_t* SCS() const { return (int64_t*)(SCT() + (M + 1)); }
    WDEV double* SCONNXY() const { return (double*)(SCS() + (M + 1)); }                 // [M][CC][2] position of every connected node
    WDEV double* SURRATE() const { return SCONNXY() + 2 * M * CC; }
    WDEV double* SURACC() const { return SURRATE() + M * CC; }
    // Node.energyRR (Node.py:31, 137-146) as a SPARSE list of the nodes whose charging rate is not zero: at most a few nodes are inside the
    // charging range of a charger at a time (r02 kept a float64 per node in LDS: 8 KB at 1 024 nodes, the difference between three and four
    // environments per CU there)
    WDEV int RRCAP() const { return M * CC + 32; }
    WDEV double* SRRV() const { return SURACC() + M * CC; }                             // [RRCAP] rates
    WDEV double* SREQD() const { return SRRV() + RRCAP(); }                             // [0] time limit, [1] now, [2] seq (as int64), [3] spare
    WDEV Scalar* SS() const { return (Scalar*)(SREQD() + 4); }
    WDEV int32_t* SREQ() const { return (int32_t*)(EC() + 1); }                           // [0] request, [1] argument, [2] live connections, [3] flags
    WDEV int32_t* SCA() const { return SREQ() + 4; }
    WDEV int32_t* SCTR() const { return SCA() + (M + 1); }
    WDEV int32_t* SCP() const { return SCTR() + (M + 1); }
    WDEV int32_t* SURN() const { return SCP() + (M + 1); }
    WDEV int32_t* SRRN() const { return SURN() + 1; }                                   // entries of the charging-rate list
    WDEV int16_t* SCONN() const { return (int16_t*)(SRRN() + 1); }
    WDEV int16_t* SURIDX() const { return SCONN() + M * CC; }
    WDEV int16_t* SURAGENT() const { return SURIDX() + M * CC; }
    WDEV int16_t* SRRI() const { return SURAGENT() + M * CC; }                          // [RRCAP] nodes of the charging-rate list

    // energyRR of node i += delta (lane 0; the same float64 additions, in the same order, as on a per-node array)
    WDEV void rr_add(int i, double delta) {
        int n = SRRN()[0];
        for (int k = 0; k < n; ++k) {
            if (SRRI()[k] == i) {
                const double v = SRRV()[k] + delta;
                if (v == 0.0) { --n; SRRI()[k] = SRRI()[n]; SRRV()[k] = SRRV()[n]; SRRN()[0] = n; } else SRRV()[k] = v;
                return;
            }
        }
        if (delta == 0.0) return;
        if (n >= RRCAP()) { err = -11; return; }
        SRRI()[n] = (int16_t)i; SRRV()[n] = delta; SRRN()[0] = n + 1;
    }
    // scale * energyRR of the lane's nodes
    WDEV void rr_slots(double (&out)[NPL], double scale) const {
#pragma unroll
        for (int j = 0; j < NPL; ++j) out[j] = 0.0;
        const int n = SRRN()[0];
        for (int k = 0; k < n; ++k) {
            const int i = SRRI()[k]; const double v = SRRV()[k] * scale;
            if ((i & 63) == lane) {
                const int jj = i >> 6;
#pragma unroll
                for (int j = 0; j < NPL; ++j) out[j] = (j == jj) ? v : out[j];
            }
        }
    }

    // -------------------------------------------------------------- setup
    WDEV void bind(const WrsnDev* dp_, int env_, int lane_, double* smem) {
        dp = dp_; env = env_; lane = lane_; M = dp_->M; NP = dp_->NP; CC = dp_->CC;
        smem_ = smem; use_snap = 0;
        {   // stage the constants of this environment in LDS
            const uint64_t* g = (const uint64_t*)(dp_->ec + env_); uint64_t* l = (uint64_t*)EC();
            for (int w = lane; w < (int)(sizeof(WrsnEnvConst) / 8); w += 64) l[w] = g[w];
        }
        __syncthreads();
        N = wu(EC()->n_node); T = wu(EC()->n_target);
        cap = wu(EC()->capacity); thr = wu(EC()->threshold); max_time = wu(EC()->max_time);
        inv_a_b2 = wu((EC()->beta * EC()->beta) / EC()->alpha);
        teps = wu(1e-9 * cap);
        err = 0; deaths_flag = 0; need_heavy = 0; t_deadline = 0; t_exact = 0;
        if (lane == 0) { Scalar* q = SS(); q->pend = 0; q->pend_idx = 0; q->L = 0; q->ev_valid = 0; q->n_events = 0; }
    }

    // -------------------------------------------------------------- static graph data of the lane's nodes
    // Neighbour lists are sorted by (distance, id) when the topology is built, so Node.find_receiver (Node.py:92-100:
    // the first strictly nearer candidate in id order) is "the first eligible neighbour".  The eight nearest ids of a
    // node are packed 16 bit each (0xFFFF = none) next to the send cost of that hop; nodes with more neighbours walk the
    // sorted CSR lists instead.  A routine that sweeps the graph loads these once (independent 16-byte loads) and then only
    // touches LDS; with NPL > 6 the ids would not fit the register file and every sweep re-reads the (L2-resident) table.
    static constexpr bool kNbReg = (NPL <= WRSN_NBREG_MAX);
    struct NbRegs { uint32_t p[kNbReg ? NPL : 1][4]; unsigned ovf, direct; int ncov[NPL]; };
    WDEV void load_neighbors(NbRegs& nb) const {
        nb.ovf = 0; nb.direct = 0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int i = j * 64 + lane;
            const int f = NFLAGS()[i];
            nb.direct |= (unsigned)(f & 1) << j; nb.ncov[j] = f >> 8;
            nb.ovf |= (unsigned)((f >> 1) & 1) << j;
            if (kNbReg) {
                const U4 v = wrsn_ld_u4(NBP() + (i));
                nb.p[kNbReg ? j : 0][0] = v.x; nb.p[kNbReg ? j : 0][1] = v.y; nb.p[kNbReg ? j : 0][2] = v.z; nb.p[kNbReg ? j : 0][3] = v.w;
            }
        }
    }
    // the packed neighbour words of register slot j: from the registers, or (NPL > 6) one coalesced 16-byte load from the static table.
    // A sweep fetches them kNbGrp slots at a time so that the loads of a group are in flight together.
    static constexpr int kNbGrp = kNbReg ? 1 : 4;
    WDEV U4 nb_words(const NbRegs& nb, int j) const {
        if (kNbReg) { U4 v; v.x = nb.p[kNbReg ? j : 0][0]; v.y = nb.p[kNbReg ? j : 0][1]; v.z = nb.p[kNbReg ? j : 0][2]; v.w = nb.p[kNbReg ? j : 0][3]; return v; }
        return wrsn_ld_u4(NBP() + (j * 64 + lane));
    }
    // eight packed ids -> indices (`self` where the slot is empty) and a validity mask
    WDEV static unsigned unpack8(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, int self, int (&idx)[8]) {
        const uint32_t pk[4] = {p0, p1, p2, p3}; unsigned ok = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int lo = (int)(pk[w] & 0xFFFFu), hi = (int)(pk[w] >> 16);
            ok |= (unsigned)(lo < 0xFFFE) << (2 * w); ok |= (unsigned)(hi < 0xFFFE) << (2 * w + 1);
            idx[2 * w] = (lo < 0xFFFE) ? lo : self; idx[2 * w + 1] = (hi < 0xFFFE) ? hi : self;
        }
        return ok;
    }
    // neighbours of node i beyond the packed ones / for NPL > 6: the sorted CSR list
#define WRSN_FOR_NEIGHBORS_CSR(i, nbvar, body)                                                     \
    { const auto off_ = NB_OFF(); const auto idx_ = NB_IDX();                              \
      for (int p_ = off_[i]; p_ < off_[(i) + 1]; ++p_) { const int nbvar = idx_[p_]; body } }

    // nodes covering target t: the first eight ids are packed like the neighbour ids (0xFFFE in the last slot: more
    // than eight, walk the CSR list).  Returns the validity mask, 0xFFFFFFFF for "use the CSR list".
    WDEV unsigned covering8(int t, int (&idx)[8]) const {
        const U4 v = wrsn_ld_u4(TCP() + (t));
        if ((v.w >> 16) == 0xFFFEu) return 0xFFFFFFFFu;
        return unpack8(v.x, v.y, v.z, v.w, 0, idx);
    }

    // Node.find_receiver (Node.py:92-100) of node i from its packed neighbour words: the nearest alive neighbour whose
    // level is lower than lvl, -1 if none; *es = energy of sending one packet to it (Node.py:114-115)
    // *wsel: which packed neighbour (its send cost is NBP_ES()[8 i + wsel]; the caller fetches the costs of all its slots in
    // one memory round trip), or -1 with the cost in *es (CSR path)
    WDEV int find_receiver(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, bool ovf, int i, int lvl, int* wsel_out, double* es) const {
        int r = -1; *wsel_out = -1; *es = 0.0;
        if (!ovf) {
            int wsel = 0; int idx[8], l2[8];
            const unsigned ok = unpack8(p0, p1, p2, p3, i, idx);
            wrsn_lds_gather8_b32(SLS(), idx, l2);
#pragma unroll
            for (int k = 7; k >= 0; --k) {                   // nearest last, so that plain selects keep the nearest
                const int e = (int)((ok >> k) & 1u) & l2[k] & (int)(((l2[k] >> 1) - 1) < lvl);
                r = (e & 1) ? idx[k] : r; wsel = (e & 1) ? k : wsel;
            }
            *wsel_out = (r >= 0) ? wsel : -1;
        } else {
            for (int p = NB_OFF()[i]; p < NB_OFF()[i + 1]; ++p) {
                const int nb = NB_IDX()[p]; const int l2 = SLS()[nb];
                if ((l2 & 1) && ((l2 >> 1) - 1) < lvl) { r = nb; *es = e_send(NB_DIST()[p]); break; }
            }
        }
        return r;
    }

    // -------------------------------------------------------------- state load / store
    // Everything an environment keeps in HBM between two launches comes in with ONE memory round trip: all loads are issued first
    // (global address space, clamped indices instead of predicated loads -- a load behind a branch is a round trip of its own), then the
    // LDS is filled.  r02 took ~30 dependent round trips here (19 K cycles per environment and launch).
    WDEV void load(const WrsnNodeArrays& a) { WRSN_PROF_T0
        const size_t nb = (size_t)env * NP;
        am = 0; dirty = 0;
        const auto gE = wrsn_global(a.E + nb), gCS = wrsn_global(a.CS + nb), gd1 = wrsn_global(a.d1 + nb), gd2 = wrsn_global(a.d2 + nb), gRR = wrsn_global(a.RR + nb);
        const auto gls = wrsn_global(a.ls + nb), grcv = wrsn_global(a.rcv + nb);
        double rr[NPL]; int lsw[NPL], rcw[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int i = j * 64 + lane, ii = i < N ? i : 0;   // the padding up to NP is never read nor written back
            E[j] = gE[ii]; CS[j] = gCS[ii]; d1[j] = gd1[ii]; d2[j] = gd2[ii]; rr[j] = gRR[ii]; lsw[j] = gls[ii];
            rcw[j] = grcv[ii];
        }
        const WrsnEnvDyn* dy = a.dyn + env;
        // charger / process records and connected-node lists: lane w takes 8-byte word w, w + 64, ... (fixed trip counts for the largest
        // configuration, clamped indices)
        constexpr int kAgW = (int)(sizeof(WrsnAgent) / 8), kThW = (int)(sizeof(WrsnThread) / 8);
        constexpr int kAgT = (WRSN_MAX_MC * kAgW + 63) / 64, kThT = (WRSN_MAX_TH * kThW + 63) / 64, kCnT = (WRSN_MAX_MC * WRSN_CONN_CAP + 63) / 64;
        const int nag = M * kAgW, nth = 2 * M * kThW, ncn = M * CC;
        const auto ga = wrsn_global((const uint64_t*)dy->ag), gt = wrsn_global((const uint64_t*)dy->th);
        const auto gc = wrsn_global(a.conn + (size_t)env * WRSN_MAX_MC * WRSN_CONN_CAP);
        const auto gr = wrsn_global(a.conn_xy + (size_t)env * WRSN_MAX_MC * WRSN_CONN_CAP * 2);
        uint64_t va[kAgT], vt[kThT]; int16_t vc[kCnT]; double vx[kCnT], vy[kCnT];
#pragma unroll
        for (int k = 0; k < kAgT; ++k) { const int w = lane + 64 * k; va[k] = ga[w < nag ? w : 0]; }
#pragma unroll
        for (int k = 0; k < kThT; ++k) { const int w = lane + 64 * k; vt[k] = gt[w < nth ? w : 0]; }
#pragma unroll
        for (int k = 0; k < kCnT; ++k) {                      // list m at stride WRSN_CONN_CAP in HBM, CC in LDS
            const int w = lane + 64 * k, wc = w < ncn ? w : 0;
            const int g = (wc / CC) * WRSN_CONN_CAP + (wc % CC);
            vc[k] = gc[g]; vx[k] = gr[2 * g]; vy[k] = gr[2 * g + 1];
        }
        now = wu(dy->now); seq = wu(dy->seq); net_time = wu(dy->net_time); net_seq = wu(dy->net_seq); ur_time = wu(dy->ur_time); ur_seq = wu(dy->ur_seq);
        node_time = wu(dy->node_time); node_seq = wu(dy->node_seq); last_minfit = wu(dy->last_minfit); opmax = wu(dy->opmax);
        n_ticks = wu(dy->n_ticks); n_exact = wu(dy->n_exact);
        net_phase = wu(dy->net_phase); net_active = wu(dy->net_active); node_phase = wu(dy->node_phase); alive = wu(dy->alive);
        levels_dirty = wu(dy->levels_dirty); cache_dirty = wu(dy->cache_dirty); irreg = wu(dy->irreg); ring_len = wu(dy->ring_len);
        ring_head = wu(dy->ring_head); safe_ticks = wu(dy->safe_ticks); frozen = wu(dy->frozen);
        log_pending = wu(dy->log_pending); fit_dirty = wu(dy->fit_dirty); map1_valid = wu(dy->map1_valid);
        const int n_conn0 = dy->n_connected; const int64_t n_ev0 = dy->n_events;
        // ---- everything is on its way: fill the LDS
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int i = j * 64 + lane; const bool real = i < N;
            E[j] = real ? E[j] : 0.0; CS[j] = real ? CS[j] : 0.0; d1[j] = real ? d1[j] : 0.0; d2[j] = real ? d2[j] : 0.0;
            const int ls = real ? lsw[j] : 0; SLS()[i] = ls;
            SRCV()[i] = real ? rcw[j] : -1;
            am |= (unsigned)(ls & 1) << j;
        }
        uint64_t* la = (uint64_t*)SAG(); uint64_t* lt = (uint64_t*)STH();
#pragma unroll
        for (int k = 0; k < kAgT; ++k) { const int w = lane + 64 * k; if (w < nag) la[w] = va[k]; }
#pragma unroll
        for (int k = 0; k < kThT; ++k) { const int w = lane + 64 * k; if (w < nth) lt[w] = vt[k]; }
#pragma unroll
        for (int k = 0; k < kCnT; ++k) { const int w = lane + 64 * k; if (w < ncn) { SCONN()[w] = vc[k]; SCONNXY()[2 * w] = vx[k]; SCONNXY()[2 * w + 1] = vy[k]; } }
        for (int w = lane; w <= M; w += 64) { SCTR()[w] = 0; SCP()[w] = 0; SCA()[w] = 0; SCT()[w] = 0; SCS()[w] = 0; }
        {   // the charging-rate list: the nodes whose energyRR in HBM is not zero
            int base = 0;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int i = j * 64 + lane; const bool nz = i < N && rr[j] != 0.0;
                const unsigned long long mk = __ballot(nz);
                const int pos = base + __popcll(mk & ((1ull << lane) - 1ull));
                if (nz && pos < RRCAP()) { SRRI()[pos] = (int16_t)i; SRRV()[pos] = rr[j]; }
                base += __popcll(mk);
            }
            if (base > RRCAP()) { err = -11; base = RRCAP(); }
            if (lane == 0) SRRN()[0] = base;
        }
        if (lane == 0) { SREQ()[0] = 0; SREQ()[1] = 0; SREQ()[2] = n_conn0; SS()->n_events = n_ev0; SURN()[0] = 0; }
        __syncthreads();
        WRSN_PROF_ADD(10)
    }

    WDEV void store(const WrsnNodeArrays& a, int terminal_pending, int64_t n_steps_add, int susp = 0) { WRSN_PROF_T0
        __syncthreads();
        const size_t nb = (size_t)env * NP;
        const auto gE = wrsn_global(a.E + nb), gCS = wrsn_global(a.CS + nb), gd1 = wrsn_global(a.d1 + nb), gd2 = wrsn_global(a.d2 + nb), gRR = wrsn_global(a.RR + nb);
        const auto gls = wrsn_global(a.ls + nb), grcv = wrsn_global(a.rcv + nb);
        double rrv[NPL]; rr_slots(rrv, 1.0);
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            if (i < N) {                                   // arrays nothing touched since load() are not written back
                gE[i] = E[j]; gRR[i] = rrv[j];
                if (dirty & 4) gCS[i] = CS[j];
                if (dirty & 1) {
                    gd1[i] = d1[j]; gd2[i] = d2[j];
                    grcv[i] = SRCV()[i];
                }
                if (dirty & 2) gls[i] = SLS()[i];
            }
        }
        WrsnEnvDyn WRSN_GLOBAL_AS* dy = wrsn_global(a.dyn + env);
        if (lane == 0) {
            dy->now = now; dy->seq = seq; dy->net_time = net_time; dy->net_seq = net_seq; dy->ur_time = ur_time; dy->ur_seq = ur_seq;
            dy->node_time = node_time; dy->node_seq = node_seq; dy->last_minfit = last_minfit; dy->opmax = opmax;
            dy->n_ticks = n_ticks; dy->n_exact = n_exact; dy->n_events = SS()->n_events; dy->n_steps += n_steps_add;
            dy->net_phase = net_phase; dy->net_active = net_active; dy->node_phase = node_phase; dy->alive = alive;
            dy->levels_dirty = levels_dirty; dy->cache_dirty = cache_dirty; dy->irreg = irreg; dy->ring_len = ring_len;
            dy->ring_head = ring_head; dy->safe_ticks = safe_ticks; dy->frozen = frozen; dy->n_connected = SREQ()[2];
            dy->terminal_pending = terminal_pending; dy->error = err; dy->log_pending = log_pending; dy->susp = susp; dy->fit_dirty = fit_dirty; dy->map1_valid = map1_valid;
        }
        const auto ga = wrsn_global((uint64_t*)(a.dyn + env)->ag); const uint64_t* la = (const uint64_t*)SAG();
        for (int w = lane; w < M * (int)(sizeof(WrsnAgent) / 8); w += 64) ga[w] = la[w];
        const auto gt = wrsn_global((uint64_t*)(a.dyn + env)->th); const uint64_t* lt = (const uint64_t*)STH();
        for (int w = lane; w < 2 * M * (int)(sizeof(WrsnThread) / 8); w += 64) gt[w] = lt[w];
        const auto gc = wrsn_global(a.conn + (size_t)env * WRSN_MAX_MC * WRSN_CONN_CAP);
        const auto gr = wrsn_global(a.conn_xy + (size_t)env * WRSN_MAX_MC * WRSN_CONN_CAP * 2);
        for (int w = lane; w < M * CC; w += 64) {
            const int g = (w / CC) * WRSN_CONN_CAP + (w % CC);
            gc[g] = SCONN()[w]; gr[2 * g] = SCONNXY()[2 * w]; gr[2 * g + 1] = SCONNXY()[2 * w + 1];
        }
        WRSN_PROF_ADD(11)
    }

    WDEV double e_send(double d) const {                    // Node.py:114-115
        double dq = d * d;
        return ((d <= EC()->d0) ? (EC()->et + EC()->efs * dq) : (EC()->et + EC()->emp * (dq * dq))) * EC()->package_size;
    }

    // ============================================================== WAVE SERVICES (all 64 lanes, uniform control flow)

    // -------------------------------------------------------------- Network.setLevels + check_targets (Network.py:37-66, 84-85)
    WDEV void set_levels() { WRSN_PROF_T0
        if constexpr (!HEAVY) { need_heavy = 1; return; } else {
        int oldlv[NPL];
        if (NPL <= 4) {
            // Up to 256 nodes: node sets are NPL 64-bit masks (one ballot per register slot) and the neighbourhood of a
            // node is a mask of the same shape, built with the topology.  A BFS level is then a handful of ANDs per slot:
            // no LDS traffic, no barrier.
            constexpr int W = (NPL <= 4) ? NPL : 1;
            unsigned long long adj[W][W]; int lv[W]; bool al[W];
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const int i = j * 64 + lane;
                const U4 lo = wrsn_ld_u4(ADJM() + (size_t)i * 2), hi = wrsn_ld_u4(ADJM() + (size_t)i * 2 + 1);
                const unsigned long long m[4] = {((unsigned long long)lo.y << 32) | lo.x, ((unsigned long long)lo.w << 32) | lo.z,
                                                 ((unsigned long long)hi.y << 32) | hi.x, ((unsigned long long)hi.w << 32) | hi.z};
#pragma unroll
                for (int w = 0; w < W; ++w) adj[j][w] = m[w];
                const int ls = SLS()[i]; oldlv[j] = (ls >> 1) - 1;
                al[j] = (ls & 1) != 0;
                lv[j] = (al[j] && (NFLAGS()[i] & 1)) ? 1 : -1;
            }
            for (int cur = 1; cur <= N; ++cur) {
                unsigned long long F[W]; unsigned long long anyf = 0ull;
#pragma unroll
                for (int w = 0; w < W; ++w) { F[w] = __ballot(al[w] && lv[w] == cur); anyf |= F[w]; }
                if (anyf == 0ull) break;
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    unsigned long long h = 0ull;
#pragma unroll
                    for (int w = 0; w < W; ++w) h |= adj[j][w] & F[w];
                    if (al[j] && lv[j] == -1 && h != 0ull) lv[j] = cur + 1;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < W; ++j) SLS()[j * 64 + lane] = ((lv[j] + 1) << 1) | (al[j] ? 1 : 0);
            __syncthreads();
        } else {
        NbRegs nbr; load_neighbors(nbr);
        // the packed neighbour words of all nodes go to the scratch array (16 bytes per node, idle during this service): every BFS
        // level then reads them from LDS instead of from the table in global memory
        U4* snb = (U4*)SU();
#pragma unroll
        for (int j = 0; j < NPL; ++j) snb[j * 64 + lane] = wrsn_ld_u4(NBP() + (j * 64 + lane));
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            int ls = SLS()[i]; oldlv[j] = (ls >> 1) - 1;
            int al = ls & 1;
            int lv = (al && ((nbr.direct >> j) & 1u)) ? 1 : -1;
            SLS()[i] = ((lv + 1) << 1) | al;                   // own entry only
        }
        __syncthreads();
        // breadth-first levels, PUSHED from the frontier (r03; r02 let every node look at its eight neighbours in every sweep: 16 slots x
        // 8-word LDS gathers per level, 430 K cycles per search at 1 000 nodes with routes of ~60 hops).  A lane keeps the levels of its own
        // nodes in registers; in sweep `cur` only the slots AT level `cur` read their packed neighbour words and mark every alive, unreached
        // neighbour with level cur + 1 (several writers, one value); then every lane looks its unreached slots up again.  Neighbourhood is
        // symmetric (Node.probe_neighbors, Node.py:80-84), so this reaches exactly the nodes the pull form reaches, at the same levels.
        int lvr[NPL]; unsigned todo = 0;                     // own levels; slots that are alive and unreached
#pragma unroll
        for (int j = 0; j < NPL; ++j) { const int ls = SLS()[j * 64 + lane]; lvr[j] = (ls >> 1) - 1; if ((ls & 1) && lvr[j] == -1) todo |= 1u << j; }
        for (int cur = 1; cur <= N; ++cur) {
            bool front = false;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                if (lvr[j] == cur) {
                    front = true;
                    const int i = j * 64 + lane;
                    const int mark = ((cur + 2) << 1) | 1;
                    if (!((nbr.ovf >> j) & 1u)) {
                        const U4 pk = snb[i];
                        int idx[8], l2[8];
                        const unsigned ok = unpack8(pk.x, pk.y, pk.z, pk.w, i, idx);
                        wrsn_lds_gather8_b32(SLS(), idx, l2);
#pragma unroll
                        for (int k = 0; k < 8; ++k) if (((ok >> k) & 1u) && (l2[k] & 1) && (l2[k] >> 1) == 0) SLS()[idx[k]] = mark;
                    } else WRSN_FOR_NEIGHBORS_CSR(i, nb, { const int l2 = SLS()[nb]; if ((l2 & 1) && (l2 >> 1) == 0) SLS()[nb] = mark; })
                }
            }
            if (!wv_any(front)) break;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                if ((todo >> j) & 1u) { const int l2 = SLS()[j * 64 + lane]; if ((l2 >> 1) != 0) { lvr[j] = (l2 >> 1) - 1; todo &= ~(1u << j); } }
            }
        }
        __syncthreads();
        }
        bool changed = false;
#pragma unroll
        for (int j = 0; j < NPL; ++j) { int i = j * 64 + lane; if (((SLS()[i] >> 1) - 1) != oldlv[j]) changed = true; }
        if (wv_any(changed)) cache_dirty = 1;
        bool bad = false;
        for (int t = lane; t < T; t += 64) {
            int acti = 0; int idx[8];                       // covered by a reached node?
            const unsigned ok = covering8(t, idx);
            if (ok != 0xFFFFFFFFu) {
                int l2[8]; wrsn_lds_gather8_b32(SLS(), idx, l2);
#pragma unroll
                for (int k = 0; k < 8; ++k) acti |= (int)((ok >> k) & 1u) & (int)((l2[k] >> 1) >= 2);
            } else for (int p = TC_OFF()[t]; p < TC_OFF()[t + 1]; ++p) acti |= (int)((SLS()[TC_IDX()[p]] >> 1) >= 2);
            if (!acti) bad = true;
        }
        alive = wv_any(bad) ? 0 : 1;
        levels_dirty = 0; work += 100; dirty |= 2;
        WRSN_PROF_ADD(6)
        }
    }

    // -------------------------------------------------------------- routing cache: receivers + per-tick drains
    // rcv_i = Node.find_receiver (Node.py:92-100) / base station (Node.py:108-111); c1/c2 = packets relayed per tick that
    // arrive before / after the node's own half-charge (sources with lower / higher id; Node.py:57-62 runs in id order).
    WDEV void rebuild_cache() { WRSN_PROF_T0
        if constexpr (!HEAVY) { need_heavy = 1; return; } else {
        int32_t* c1 = (int32_t*)SU(); int32_t* c2 = c1 + NP;
        double es[NPL]; int rc[NPL], wsl[NPL];
        const double er = EC()->e_recv;
        NbRegs nbr; load_neighbors(nbr);
        __syncthreads();
#pragma unroll
        for (int j0 = 0; j0 < NPL; j0 += kNbGrp) {
            U4 pk[kNbGrp];
#pragma unroll
            for (int q = 0; q < kNbGrp; ++q) pk[q] = nb_words(nbr, j0 + q);
#pragma unroll
            for (int q = 0; q < kNbGrp; ++q) {
                const int j = j0 + q;
                int i = j * 64 + lane;
                int ls = SLS()[i]; int lvl = (ls >> 1) - 1;
                int r = -1; double e1 = 0.0; int ws = -1;
                if ((nbr.direct >> j) & 1u) r = -2;
                else r = find_receiver(pk[q].x, pk[q].y, pk[q].z, pk[q].w, (nbr.ovf >> j) & 1u, i, lvl, &ws, &e1);
                if (!(ls & 1)) { r = -1; e1 = 0.0; ws = -1; }
                es[j] = e1; rc[j] = r; wsl[j] = ws;
                SRCV()[i] = r; c1[i] = 0; c2[i] = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < NPL; ++j) {                      // send costs of all slots: independent loads, one round trip
            const int i = j * 64 + lane;
            if (rc[j] == -2) es[j] = ES_BS()[i]; else if (wsl[j] >= 0) es[j] = NBP_ES()[(size_t)i * 8 + wsl[j]]; else if (rc[j] == -1) es[j] = 0.0;
        }
        __syncthreads();
        {   // relays per second of every node: the routes of a lane's sources are walked together (overlapping LDS round trips)
            int v[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) v[j] = (((am >> j) & 1u) && rc[j] >= 0 && nbr.ncov[j] > 0) ? rc[j] : -1;
            for (int guard = 0; guard < N; ++guard) {
                bool any = false;
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    if (v[j] >= 0) { const int i = j * 64 + lane; atomicAdd((i < v[j]) ? &c1[v[j]] : &c2[v[j]], nbr.ncov[j]); v[j] = SRCV()[v[j]]; any = true; }
                }
                if (!any) break;
            }
        }
        __syncthreads();
        double opm = er;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            if ((am >> j) & 1u) {
                d1[j] = (double)c1[i] * (er + es[j]);
                d2[j] = (double)c2[i] * (er + es[j]) + (double)nbr.ncov[j] * es[j];
                opm = fmax(opm, es[j]);
            } else { d1[j] = 0.0; d2[j] = 0.0; }
        }
        __syncthreads();
        opmax = wv_max(opm);
        cache_dirty = 0; irreg = WRSN_RING; safe_ticks = 0; work += 80; dirty |= 1;
        WRSN_PROF_ADD(5)
        }
    }

    // -------------------------------------------------------------- exact k+0.5 instant (a node may run dry this second)
    // Node.py:57-62 + 102-132 process sources in node-id order; a packet is dropped and the node dies as soon as it cannot
    // pay (Node.py:116-117, 126-127), which re-routes everything behind it.  As long as nobody can fail, the order inside a
    // range of sources does not matter, so the second is replayed range by range in closed form (bisection down to one
    // source) and only the single source whose packets may hit a starving node is walked packet by packet on lane 0.
    // LDS records of the packet-by-packet walk: energy and send cost in float64 (SU()[0..NP), SU()[NP..2NP)), receiver in
    // the float-CS array of the steady batch, which is idle here

    // receivers + send cost of every alive node for the current (live status, last levels); returns max op cost
    WDEV double walk_receivers(const NbRegs& nbr, double (&es)[NPL]) {
        double opm = EC()->e_recv;
        __syncthreads();
        int rc[NPL], wsl[NPL];
#pragma unroll
        for (int j0 = 0; j0 < NPL; j0 += kNbGrp) {
            U4 pk[kNbGrp];
#pragma unroll
            for (int q = 0; q < kNbGrp; ++q) pk[q] = nb_words(nbr, j0 + q);
#pragma unroll
            for (int q = 0; q < kNbGrp; ++q) {
                const int j = j0 + q;
                int i = j * 64 + lane; int r = -1; double e1 = 0.0; int ws = -1;
                const int ls = SLS()[i];
                if ((nbr.direct >> j) & 1u) r = -2;
                else r = find_receiver(pk[q].x, pk[q].y, pk[q].z, pk[q].w, (nbr.ovf >> j) & 1u, i, (ls >> 1) - 1, &ws, &e1);
                if (!(ls & 1)) { r = -1; e1 = 0.0; ws = -1; }
                es[j] = e1; rc[j] = r; wsl[j] = ws;
            }
        }
#pragma unroll
        for (int j = 0; j < NPL; ++j) {                      // send costs of all slots: independent loads, one round trip
            const int i = j * 64 + lane;
            if (rc[j] == -2) es[j] = ES_BS()[i]; else if (wsl[j] >= 0) es[j] = NBP_ES()[(size_t)i * 8 + wsl[j]]; else if (rc[j] == -1) es[j] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < NPL; ++j) opm = fmax(opm, es[j]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NPL; ++j) SRCV()[j * 64 + lane] = rc[j];
        __syncthreads();
        return wv_max(opm);
    }

    // sources [a, b) in closed form.  Returns false (nothing changed) when some node might be unable to pay.
    WDEV bool walk_range(int a, int b, const NbRegs& nbr, const double (&es)[NPL], const double (&rrh)[NPL], double (&gain)[NPL], double margin) {
        int32_t* c1 = (int32_t*)SU(); int32_t* c2 = c1 + NP;
        const double er = EC()->e_recv;
#pragma unroll
        for (int j = 0; j < NPL; ++j) { c1[j * 64 + lane] = 0; c2[j * 64 + lane] = 0; }
        __syncthreads();
        {   // every source adds its packets to each relay of its route; the (up to NPL) routes of a lane are walked together so
            // that their LDS round trips overlap
            int v[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) { const int q = j * 64 + lane; v[j] = (q >= a && q < b && ((am >> j) & 1u) && nbr.ncov[j] > 0) ? SRCV()[q] : -1; }
            for (int guard = 0; guard < N; ++guard) {
                bool any = false;
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    if (v[j] >= 0) { const int q = j * 64 + lane; atomicAdd((q < v[j]) ? &c1[v[j]] : &c2[v[j]], nbr.ncov[j]); v[j] = SRCV()[v[j]]; any = true; }
                }
                if (!any) break;
            }
        }
        __syncthreads();
        double en[NPL], gn[NPL]; unsigned unsafe_m = 0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int i = j * 64 + lane;
            en[j] = E[j]; gn[j] = 0.0; bool unsafe = false;
            if ((am >> j) & 1u) {
                const double per = er + es[j];
                double e = E[j] - (double)c1[i] * per;           // sources with a lower id: before the node's own wake
                if (c1[i] > 0 && e - thr < margin) unsafe = true;
                if (i >= a && i < b) {                             // the node wakes inside this range (Node.py:60)
                    double e2 = fmin(e + rrh[j], cap); gn[j] = e2 - e; e = e2;
                    e -= (double)c2[i] * per + (double)nbr.ncov[j] * es[j];
                    if ((c2[i] > 0 || (nbr.ncov[j] > 0 && es[j] > 0.0)) && e - thr < margin) unsafe = true;
                } else if (c2[i] > 0) {
                    e -= (double)c2[i] * per;
                    if (e - thr < margin) unsafe = true;
                }
                en[j] = e;
            }
            if (unsafe) unsafe_m |= 1u << j;
        }
        const bool bad = wv_any(unsafe_m != 0);
        if (bad) {                                           // who: lets exact_walk go straight to the critical source
            uns_cnt = 0; uns_node = -1;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const unsigned long long mk = __ballot((unsafe_m >> j) & 1u);
                uns_cnt += __popcll(mk);
                if (mk && uns_node < 0) uns_node = j * 64 + __popcll((mk & (~mk + 1ull)) - 1ull);
            }
        }
        if (!bad) {
#pragma unroll
            for (int j = 0; j < NPL; ++j) { E[j] = en[j]; gain[j] += gn[j]; }
        }
        __syncthreads();
        return !bad;
    }

    // The range [a, b) failed because exactly one node v would end at / below thr.  Energy of v only falls between its
    // half-charges, so the first source after whose packets v is at / below thr is found by replaying v alone: sources
    // whose route passes v cost it (er + es_v) per packet, its own wake adds the half-charge and costs es_v per own
    // packet.  Returns that source (to be walked packet by packet; everything before it is safe), or -1.
    WDEV int locate_failure(int a, int b, int v, const NbRegs& nbr, const double (&es)[NPL], const double (&rrh)[NPL]) {
        double* pv = SU(); int32_t* rq = (int32_t*)(SU() + 8);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int q = j * 64 + lane;
            if (q == v) { pv[0] = E[j]; pv[1] = EC()->e_recv + es[j]; pv[2] = es[j]; pv[3] = rrh[j]; pv[4] = (double)nbr.ncov[j]; }
            int r = 0;
            if (q >= a && q < b && q != v && ((am >> j) & 1u) && nbr.ncov[j] > 0) {
                int u = SRCV()[q], guard = 0; bool hit = false;
                while (u >= 0 && !hit && guard++ < N) { hit = (u == v); u = SRCV()[u]; }
                r = hit ? nbr.ncov[j] : 0;
            }
            rq[q] = r;
        }
        __syncthreads();
        double e = pv[0]; const double per = pv[1], own = pv[2] * pv[4], rrv = pv[3];
        int found = -1;
        for (int q = a; q < b && found < 0; ++q) {
            const int r = rq[q];
            if (q == v) { e = fmin(e + rrv, cap) - own; if (e - thr <= teps) found = q; }
            else if (r > 0) { e -= per * (double)r; if (e - thr <= teps) found = q; }
        }
        __syncthreads();
        return found;
    }

    // one source, packet by packet (lane 0), exactly as Node.send_package / receive_package; returns #deaths
    // Within one packet the cached receivers stay valid: a death either drops the packet or happens behind it (the sender
    // dies after paying).  Only the NEXT packets see a changed network, so the walk stops after the first packet with a
    // death and reports where to go on (SREQ()[3] = next packet, SREQ()[0] = packets of the source); the caller re-routes
    // with the whole wave and calls again.
    WDEV int walk_single(int q, int p0, double (&es)[NPL], const double (&rrh)[NPL], double (&gain)[NPL]) {
        double* recE = SU(); double* recS = SU() + NP; int32_t* recR = (int32_t*)SCSF();
        const double er = EC()->e_recv;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int i = j * 64 + lane;
            if (p0 == 0 && i == q && ((am >> j) & 1u)) { double e2 = fmin(E[j] + rrh[j], cap); gain[j] += e2 - E[j]; E[j] = e2; }   // the source wakes
            recE[i] = E[j]; recS[i] = es[j]; recR[i] = SRCV()[i];
        }
        __syncthreads();
        if (lane == 0) {
            int deaths = 0, nc = 0, p = p0;
            if (SLS()[q] & 1) {
                nc = NFLAGS()[q] >> 8;
                for (; p < nc && deaths == 0; ++p) {
                    int cur = q;
                    for (int hop = 0; hop <= N; ++hop) {
                        const double wE = recE[cur], esv = recS[cur]; const int r = recR[cur];
                        if (r == -1) { if (wE <= thr) { SLS()[cur] &= ~1; deaths++; } break; }
                        if (wE - thr < esv) { recE[cur] = thr; SLS()[cur] &= ~1; deaths++; break; }
                        double e = wE - esv;
                        recE[cur] = e;
                        if (e <= thr) { SLS()[cur] &= ~1; deaths++; }
                        if (r == -2) break;
                        double e_r = recE[r];
                        if (e_r - thr < er) { recE[r] = thr; SLS()[r] &= ~1; deaths++; break; }
                        recE[r] = e_r - er;
                        cur = r;
                    }
                }
            }
            SREQ()[1] = deaths; SREQ()[3] = p; SREQ()[0] = nc;
        }
        __syncthreads();
        const int deaths = SREQ()[1];
#pragma unroll
        for (int j = 0; j < NPL; ++j) E[j] = recE[j * 64 + lane];
        __syncthreads();
        return deaths;
    }

    WDEV void exact_walk(const double (&rrh)[NPL], bool any_rr) { WRSN_PROF_T0
        if constexpr (!HEAVY) { need_heavy = 1; (void)rrh; (void)any_rr; return; } else {
        double es[NPL], gain[NPL], e_start[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) { e_start[j] = E[j]; gain[j] = 0.0; }
        double margin = 0.0;
        bool any_death = false, need_recv = true;
        NbRegs nbr; load_neighbors(nbr);
        // Everything that is left is tried in one closed form; a range that fails is bisected (its first half is tested,
        // committed if it is safe) until the one source whose packets meet the starving node is isolated: about
        // log2(N) + 2 range evaluations per death.
        // node_half found the whole second unsafe with the routing cache (the same closed form as a range evaluation of
        // [0, N)): that evaluation is not repeated
        int a = 0, hi = N, hint = -1; bool hi_fails = true;  // hi_fails: [a, hi) is known to contain a failure; hint: believed first critical source
        bool first = true;
        for (int guard = 0; a < N && guard < 16 * N + 64; ++guard) {
            if (need_recv) { WRSN_PROF_MARK(x0_) (void)walk_receivers(nbr, es); margin = teps; need_recv = false; WRSN_PROF_MARK(x1_) WRSN_PROF_SPAN(16, x0_, x1_) }   // (re-)route; a range is safe iff nobody ends at / below thr
            if (first) {
                first = false;
                if (uns_cnt == 1 && N > 2) { WRSN_PROF_CNT(21, 1) const int q = locate_failure(0, N, uns_node, nbr, es, rrh); if (q >= 0) { hi = q + 1; hint = q; } }
            }
            if (hi_fails && hi - a <= 1) {
                WRSN_PROF_MARK(x2_)
                for (int p0 = 0;;) {                         // packet by packet; after a packet with a death the wave re-routes
                    const int deaths = walk_single(a, p0, es, rrh, gain);
                    const int next_p = SREQ()[3], nc = SREQ()[0];
                    __syncthreads();
                    if (deaths > 0) {
                        any_death = true;
#pragma unroll
                        for (int j = 0; j < NPL; ++j) {
                            const int i = j * 64 + lane;
                            if (((am >> j) & 1u) && !(SLS()[i] & 1)) { am &= ~(1u << j); CS[j] = 0.0; d1[j] = 0.0; d2[j] = 0.0; }   // Node.check_status
                        }
                        (void)walk_receivers(nbr, es);       // everything behind the dead node is re-routed
                    }
                    if (next_p >= nc) break;
                    p0 = next_p;
                }
                WRSN_PROF_MARK(x3_) WRSN_PROF_SPAN(17, x2_, x3_) WRSN_PROF_CNT(18, 1)
                a += 1; hi = N; hi_fails = false; hint = -1;
                continue;
            }
            const bool use_hint = hi_fails && hint > a && hint < hi;
            const int b = hi_fails ? (use_hint ? hint : a + (hi - a) / 2) : N;
            hint = -1;
            bool okr_ = false;
#if defined(WRSN_PROFILE) && WRSN_PROFILE == 1
            for (int rep_ = 0; rep_ < 2 && !okr_; ++rep_) {      // a failing evaluation has no side effect: the second pass times the same code warm
                WRSN_PROF_MARK(x4_)
                okr_ = walk_range(a, b, nbr, es, rrh, gain, margin);
                WRSN_PROF_MARK(x5_)
                (void)x4_; (void)x5_;
            }
#else
            okr_ = walk_range(a, b, nbr, es, rrh, gain, margin);
#endif
            if (okr_) { a = b; if (a >= hi) { hi = N; hi_fails = false; } }
            else {
                hi = b; hi_fails = true;
                if (uns_cnt == 1 && hi - a > 2) {            // one starving node: go straight to the source that meets it
                    WRSN_PROF_CNT(21, 1)
                    const int q = locate_failure(a, hi, uns_node, nbr, es, rrh);
                    if (q >= a) { hi = q + 1; hint = q; }
                }
            }
        }
        // log_energy of the second: every operation of a surviving node succeeded = start + half-charge gained - end
#pragma unroll
        for (int j = 0; j < NPL; ++j) if ((am >> j) & 1u) LOGBUF()[j * 64 + lane] = e_start[j] + gain[j] - E[j];
        if (any_death) { cache_dirty = 1; levels_dirty = 1; deaths_flag = 1; }
        irreg = WRSN_RING; log_pending = 1; safe_ticks = 0; n_exact++; work += 500; dirty |= 7;
        (void)any_rr;
        __syncthreads();
        WRSN_PROF_ADD(4)
        }
    }

    // May a node run dry in the second that starts now?  (Node state at k+0.5 is the state at the start of the second.)  When not:
    // how many further seconds are safe without looking (safe_ticks); when yes: who (uns_cnt / uns_node).
    WDEV bool second_is_safe(const double (&rrh)[NPL]) {
        unsigned trig = 0; double mn = 1e30;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            if ((am >> j) & 1u) {
                double rr = rrh[j];
                double a = E[j] - d1[j];
                double b = fmin(a + rr, cap) - d2[j];
                // Energy only falls between two half-charges, and an operation fails / kills exactly when the energy
                // after it is <= thr: with the routing cache valid the second is uneventful iff both segment ends
                // stay above thr (teps: far above the rounding of the closed form, far below any operation).  Idle
                // nodes never trigger.
                if ((d1[j] > 0.0 && a - thr <= teps) || (d2[j] > 0.0 && b - thr <= teps)) trig |= 1u << j;
                double ds = d1[j] + d2[j];
                if (ds > 0.0) mn = fmin(mn, (E[j] - thr - opmax) / ds);
            }
        }
        const bool fast = !wv_any(trig != 0);
        if (!fast) {                                         // who: the exact second starts from the critical source when it is one node
            uns_cnt = 0; uns_node = -1;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const unsigned long long mk = __ballot((trig >> j) & 1u);
                uns_cnt += __popcll(mk);
                if (mk && uns_node < 0) uns_node = j * 64 + __popcll((mk & (~mk + 1ull)) - 1ull);
            }
        } else { double m = wv_min(mn); safe_ticks = (m > 4.0) ? (int)fmin(m - 3.0, 1.0e6) : 0; }
        return fast;
    }

    // -------------------------------------------------------------- k+0.5: Node.operate first half for all nodes (Node.py:57-62)
    WDEV void node_half(const double (&rrh)[NPL], const bool any_rr) {
        if (cache_dirty) { rebuild_cache(); if (!HEAVY) return; }
        bool fast = true;
        if (safe_ticks > 0) { safe_ticks--; }
        else {
            fast = second_is_safe(rrh);
            if (!HEAVY && !fast) { need_heavy = 1; return; }  // nothing was touched: the full variant takes this item again
            // time-sliced launch about to end: a packet-exact second is not begun (unless nothing else was done in this visit); the
            // environment stops in front of the item and the next launch, in which it is among the first, takes it again
            if (!fast && t_exact != 0 && n_items > 0 && (long long)wall_clock64() > t_exact) { need_heavy = 2; return; }
        }
        if (fast) {
            if (any_rr) {
#pragma unroll
                for (int j = 0; j < NPL; ++j) if ((am >> j) & 1u) E[j] = fmin(E[j] - d1[j] + rrh[j], cap) - d2[j];
            } else {
#pragma unroll
                for (int j = 0; j < NPL; ++j) E[j] = (E[j] - d1[j]) - d2[j];
            }
            log_pending = 0;
        } else exact_walk(rrh, any_rr);
    }

    // -------------------------------------------------------------- k+1.0: second half + consumption window (Node.py:65-77)
    WDEV void node_full(const double (&rrh)[NPL], const bool any_rr) {
        if (any_rr) {
#pragma unroll
            for (int j = 0; j < NPL; ++j) if ((am >> j) & 1u) E[j] = fmin(E[j] + rrh[j], cap);
        }
        if (irreg > 0) window_update();
        log_pending = 0; n_ticks++;
    }

    // the sliding consumption window of Node.operate's second half (Node.py:71-77) while it is not uniform yet: the ten seconds
    // after a routing change or a packet-exact second
    WDEV void window_update() {
        const int len = ring_len, head = ring_head;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            if ((am >> j) & 1u) {
                int i = j * 64 + lane;
                double lg = log_pending ? LOGBUF()[i] : (d1[j] + d2[j]);
                if (len < WRSN_RING) { RING()[(size_t)len * NP + i] = lg; CS[j] = (CS[j] * len + lg) / (len + 1); }
                else { double old = RING()[(size_t)head * NP + i]; CS[j] = (CS[j] * len - old + lg) / len; RING()[(size_t)head * NP + i] = lg; }
            }
        }
        if (len < WRSN_RING) ring_len = len + 1; else ring_head = (head + 1) % WRSN_RING;
        irreg--; dirty |= 4;
    }

    // alpha / (dist(node, charger) + beta)^2 (Node.py:137, WRSN.py:122) of connected node k of charger m at the charger's
    // CURRENT location (a stale "charging" charger may be on the move); node positions were cached by conn_build
    WDEV double conn_rate_of(int m, int k, int i) const {
        (void)i;
        const double* xy = SCONNXY() + 2 * (m * CC + k);
        double dd = dist2(xy[0], xy[1], SAG()[m].loc[0], SAG()[m].loc[1]) + EC()->beta;
        return EC()->alpha / (dd * dd);
    }

    // -------------------------------------------------------------- WRSN.update_reward (WRSN.py:100-127)
    // priorities: p = CS / (E - thr + 1e-9) (0 for dead nodes), standardised (population std), exp, normalised.
    // Lane 0 keeps a list of (node, charger, rate) for the alive "charging" chargers (ur_build, rebuilt whenever a
    // charger event or a node death changes it); the lane that owns a listed node adds its contribution to the
    // entry's LDS accumulator every second, and the accumulators are folded into agents_exclusive_reward when the
    // grid service ends (ur_flush).  No barrier, no cross-lane fetch per second.
    WDEV void update_reward() { WRSN_PROF_T0
        // float32 pipeline (v_rcp_f32 / v_sqrt_f32 / v_exp_f32, full-rate FMAs, half the DPP traffic): the priorities
        // are ratios of order 1e-4..1e-2 and only enter the reward through 0.2 * excl / avg_nodes_agent; the
        // deviation of excl from the float64 oracle stays below 1e-6 relative (tests/).
        const float epsf = 1e-9f;
        float x[NPL]; float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NPL; ++j) { const float den = (float)(E[j] - thr) + epsf; x[j] = ((am >> j) & 1u) ? (float)CS[j] * __builtin_amdgcn_rcpf(den) : 0.f; s1 += x[j]; s2 = fmaf(x[j], x[j], s2); }
        wv_sum2f(s1, s2);
        const float invn = 1.0f / (float)N;
        const float mean = s1 * invn;
        float var = fmaf(-mean, mean, s2 * invn);
        var = var > 0.f ? var : 0.f;
        float sd = __builtin_amdgcn_sqrtf(var);
        if (sd == 0.f) sd = epsf;
        const float k2 = 1.44269504f * __builtin_amdgcn_rcpf(sd);     // exp(z) = 2^(z * log2 e)
        float ex[NPL]; float es = 0.f;
#pragma unroll
        for (int j = 0; j < NPL; ++j) { const float v = (j * 64 + lane < N) ? __builtin_amdgcn_exp2f((x[j] - mean) * k2) : 0.f; ex[j] = v; es += v; }
        float tot = wv_sumf(es);
        if (tot == 0.f) tot = epsf;
        const double scale = (double)__builtin_amdgcn_rcpf(tot) * inv_a_b2;
        const int n = SURN()[0];
        for (int k = 0; k < n; ++k) {
            const int i = SURIDX()[k];
            if (lane == (i & 63)) {
                const int jj = i >> 6;
                // value selects (not `if (j == jj) x = E[j]`): a pointer select would pin E[] / CS[] in scratch memory
                double Ei = E[0], Ci = CS[0]; float xi = ex[0];
#pragma unroll
                for (int j = 1; j < NPL; ++j) { const double ej = E[j], cj = CS[j]; const float xj = ex[j]; const bool hit = (j == jj); Ei = hit ? ej : Ei; Ci = hit ? cj : Ci; xi = hit ? xj : xi; }
                if ((am >> jj) & 1u) {
                    const double e_no = fmin(Ei - Ci, thr);              // min / max as written (WRSN.py:123-124)
                    const double e_with = fmax(Ei - Ci + SURRATE()[k], cap);
                    SURACC()[k] += (double)xi * (e_with - e_no) * scale;
                }
            }
        }
        WRSN_PROF_ADD(3)
    }

    // -------------------------------------------------------------- time-parallel steady batch
    // nb (<= 64) consecutive whole seconds of the steady path at once: a lane owns one second and loops over (a part of) the nodes,
    // whose state is broadcast from LDS (one 16-byte and one 4-byte read per node, four nodes in flight).  Uncharged
    // nodes follow E0 - (s+1)(d1+d2); the few nodes under charge get their clamped recursion (Node.py:60,68) tabulated
    // per second by one lane each and are added after the main loops (they are staged as "priority 0" nodes, whose
    // term is taken out again).  The reward priorities of every second (update_reward, WRSN.py:100-127) need no
    // cross-lane reduction this way; the contributions of the connected nodes are summed over the seconds with one
    // wave reduction per charger.  Returns false (nothing done) when more than kChgMax nodes are being charged.
    struct alignas(16) D2 { double x, y; };
    struct alignas(8) F2 { float x, y; };
    WDEV bool steady_batch(int nb, const double (&rrh)[NPL], bool any_rr) { WRSN_PROF_T0
        D2* sA = (D2*)SU(); float* sC = SCSF();
        double* rec = SCHGREC(); double* tab = SCHGTAB();
        const float epsf = 1e-9f;
        // -- which nodes are being charged (half-rate != 0, alive)
        int nchg = 0; unsigned cm = 0; int cpos[NPL];
        if (any_rr) {
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const bool c = ((am >> j) & 1u) && rrh[j] != 0.0;
                const unsigned long long mk = __ballot(c);
                cpos[j] = nchg + __popcll(mk & ((1ull << lane) - 1ull));
                if (c) {
                    cm |= 1u << j;
                    if (cpos[j] < kChgMax) { double* r = rec + 8 * cpos[j]; r[0] = E[j]; r[1] = d1[j]; r[2] = d2[j]; r[3] = rrh[j]; r[4] = (double)(j * 64 + lane); r[6] = CS[j]; }
                }
                nchg += __popcll(mk);
            }
            if (nchg > kChgMax) return false;
        }
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int i = j * 64 + lane;
            const bool lin = ((am >> j) & 1u) && !((cm >> j) & 1u);
            D2 v; v.x = lin ? (E[j] - thr) : 1.0; v.y = lin ? (d1[j] + d2[j]) : 0.0;
            sA[i] = v; sC[i] = lin ? (float)CS[j] : 0.f;
        }
        __syncthreads();
        if (lane < nchg) {                                   // clamped recursion of one charged node over the nb seconds
            double* r = rec + 8 * lane;
            double e = r[0]; const double a1 = r[1], a2 = r[2], rr = r[3];
            for (int q = 0; q < nb; ++q) { e = fmin(e - a1 + rr, cap) - a2; tab[lane * 64 + q] = e; e = fmin(e + rr, cap); }
            r[5] = e;
        }
        __syncthreads();
        // lanes = S seconds x P node partitions (S = nb rounded up to a power of two >= 8, P = 64 / S): a short batch
        // costs proportionally less; the partial sums of a second are combined with log2(P) xor-shuffles
        int S = 8; while (S < nb) S <<= 1;
        const int sidx = lane & (S - 1), part = lane / S, P = 64 / S;
        const bool on = sidx < nb && part == 0;              // the lane that owns the second's result
        const int ls = sidx < nb ? sidx : 0;
        const double sp1 = -(double)(ls + 1);
        const int chunk = (((N + P - 1) / P) + 3) & ~3;      // nodes per partition, multiple of 4
        const int i0 = part * chunk;
        const int i1 = (i0 + chunk < N) ? i0 + chunk : N;    // may be <= i0 for the last partitions
        const int i4 = (i1 > i0) ? i0 + ((i1 - i0) & ~3) : i0;
        // -- pass 1: mean / variance of the priorities of "my" second
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        for (int i = i0; i < i4; i += 4) {
            const D2 p0 = sA[i], p1 = sA[i + 1], p2 = sA[i + 2], p3 = sA[i + 3];
            const F2 c01 = *(const F2*)(sC + i), c23 = *(const F2*)(sC + i + 2);
            const double x0 = (double)(c01.x * __builtin_amdgcn_rcpf((float)fma(sp1, p0.y, p0.x) + epsf));
            const double x1 = (double)(c01.y * __builtin_amdgcn_rcpf((float)fma(sp1, p1.y, p1.x) + epsf));
            const double x2 = (double)(c23.x * __builtin_amdgcn_rcpf((float)fma(sp1, p2.y, p2.x) + epsf));
            const double x3 = (double)(c23.y * __builtin_amdgcn_rcpf((float)fma(sp1, p3.y, p3.x) + epsf));
            a0 += x0; a1 += x1; a2 += x2; a3 += x3;
            b0 = fma(x0, x0, b0); b1 = fma(x1, x1, b1); b2 = fma(x2, x2, b2); b3 = fma(x3, x3, b3);
        }
        for (int i = i4; i < i1; ++i) {
            const D2 p0 = sA[i];
            const double x0 = (double)(sC[i] * __builtin_amdgcn_rcpf((float)fma(sp1, p0.y, p0.x) + epsf));
            a0 += x0; b0 = fma(x0, x0, b0);
        }
        if (part == 0) {
            for (int c = 0; c < nchg; ++c) {
                const double x0 = (double)((float)rec[8 * c + 6] * __builtin_amdgcn_rcpf((float)(tab[c * 64 + ls] - thr) + epsf));
                a1 += x0; b1 = fma(x0, x0, b1);
            }
        }
        double s1 = (a0 + a1) + (a2 + a3), s2 = (b0 + b1) + (b2 + b3);
        for (int m = S; m < 64; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
        const double mean = s1 / N;
        double var = s2 / N - mean * mean; var = var > 0.0 ? var : 0.0;
        float sd = __builtin_amdgcn_sqrtf((float)var); if (sd == 0.f) sd = epsf;
        const float k2 = 1.44269504f * __builtin_amdgcn_rcpf(sd);     // exp(z) = 2^(z * log2 e)
        const float mk2 = -(float)mean * k2;
        // -- pass 2: normaliser
        a0 = 0.0; a1 = 0.0; a2 = 0.0; a3 = 0.0;
        for (int i = i0; i < i4; i += 4) {
            const D2 p0 = sA[i], p1 = sA[i + 1], p2 = sA[i + 2], p3 = sA[i + 3];
            const F2 c01 = *(const F2*)(sC + i), c23 = *(const F2*)(sC + i + 2);
            const float x0 = c01.x * __builtin_amdgcn_rcpf((float)fma(sp1, p0.y, p0.x) + epsf);
            const float x1 = c01.y * __builtin_amdgcn_rcpf((float)fma(sp1, p1.y, p1.x) + epsf);
            const float x2 = c23.x * __builtin_amdgcn_rcpf((float)fma(sp1, p2.y, p2.x) + epsf);
            const float x3 = c23.y * __builtin_amdgcn_rcpf((float)fma(sp1, p3.y, p3.x) + epsf);
            a0 += (double)__builtin_amdgcn_exp2f(fmaf(x0, k2, mk2)); a1 += (double)__builtin_amdgcn_exp2f(fmaf(x1, k2, mk2));
            a2 += (double)__builtin_amdgcn_exp2f(fmaf(x2, k2, mk2)); a3 += (double)__builtin_amdgcn_exp2f(fmaf(x3, k2, mk2));
        }
        for (int i = i4; i < i1; ++i) {
            const D2 p0 = sA[i];
            const float x0 = sC[i] * __builtin_amdgcn_rcpf((float)fma(sp1, p0.y, p0.x) + epsf);
            a0 += (double)__builtin_amdgcn_exp2f(fmaf(x0, k2, mk2));
        }
        if (part == 0) {
            const double ez = (double)__builtin_amdgcn_exp2f(fmaf(0.f, k2, mk2));   // what a "priority 0" node added above
            for (int c = 0; c < nchg; ++c) {
                const float x0 = (float)rec[8 * c + 6] * __builtin_amdgcn_rcpf((float)(tab[c * 64 + ls] - thr) + epsf);
                a1 += (double)__builtin_amdgcn_exp2f(fmaf(x0, k2, mk2)) - ez;
            }
        }
        double es = (a0 + a1) + (a2 + a3);
        for (int m = S; m < 64; m <<= 1) es += __shfl_xor(es, m);
        float tot = (float)es; if (tot == 0.f) tot = epsf;
        const double scale = (double)__builtin_amdgcn_rcpf(tot) * inv_a_b2;
        // -- connected entries: contribution of every second, summed over the seconds (entries are grouped by charger)
        const int n = SURN()[0];
        double acc = 0.0;
        for (int k = 0; k < n; ++k) {
            const int i = SURIDX()[k];
            int c = -1;
            for (int cc = 0; cc < nchg; ++cc) if ((int)rec[8 * cc + 4] == i) c = cc;
            double e, cs;
            if (c >= 0) { e = tab[c * 64 + ls]; cs = rec[8 * c + 6]; }
            else { const D2 p0 = sA[i]; e = fma(sp1, p0.y, p0.x) + thr; cs = (double)sC[i]; }
            if (on && (SLS()[i] & 1)) {
                const float x = (float)cs * __builtin_amdgcn_rcpf((float)(e - thr) + epsf);
                const double ex = (double)__builtin_amdgcn_exp2f(fmaf(x, k2, mk2));
                const double e_no = fmin(e - cs, thr);                   // min / max as written (WRSN.py:123-124)
                const double e_with = fmax(e - cs + SURRATE()[k], cap);
                acc += ex * (e_with - e_no) * scale;
            }
            if (k == n - 1 || SURAGENT()[k + 1] != SURAGENT()[k]) {
                const double r = wv_sum(acc);
                if (lane == 0) SURACC()[k] += r;
                acc = 0.0;
            }
        }
        // -- advance the node registers by nb seconds
        const double dnb = (double)nb;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const double lin = E[j] - dnb * (d1[j] + d2[j]);
            E[j] = ((cm >> j) & 1u) ? rec[8 * cpos[j] + 5] : lin;
        }
        __syncthreads();
        WRSN_PROF_ADD(2)
        return true;
    }

    // lane 0: the list update_reward iterates (alive chargers whose action type is "charging", their connected nodes)
    WDEV void ur_build() {
        int n = 0;
        for (int m = 0; m < M; ++m) {
            if (SAG()[m].status == 0 || !SAG()[m].type_charging) continue;
            const int nc = SAG()[m].n_conn;
            for (int k = 0; k < nc; ++k) {
                const int i = SCONN()[m * CC + k];
                if (!(SLS()[i] & 1)) continue;
                SURIDX()[n] = (int16_t)i; SURAGENT()[n] = (int16_t)m; SURRATE()[n] = conn_rate_of(m, k, i); SURACC()[n] = 0.0; ++n;
            }
        }
        SURN()[0] = n;
    }

    // fold the per-entry sums into agents_exclusive_reward (end of a grid service; all lanes call, lane 0 works)
    WDEV void ur_flush() {
        __syncthreads();
        if (lane == 0) {
            const int n = SURN()[0];
            for (int k = 0; k < n; ++k) { SAG()[SURAGENT()[k]].excl += SURACC()[k]; SURACC()[k] = 0.0; }
        }
        __syncthreads();
    }

    // -------------------------------------------------------------- WRSN.get_network_fitness -> np.min (WRSN.py:188-220)
    // label-correcting widest path; the fixed point does not depend on visiting order, so relax in parallel.
    WDEV double min_fitness() { WRSN_PROF_T0
        NbRegs nbr; load_neighbors(nbr);
        double* t = SU(); double lt[NPL], tc[NPL];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            bool al = (am >> j) & 1u;
            lt[j] = al ? ((CS[j] == 0.0) ? WRSN_INF : (E[j] - thr) / CS[j]) : 0.0;
            tc[j] = (al && ((nbr.direct >> j) & 1u)) ? lt[j] : -1.0;
            t[i] = tc[j];
        }
        __syncthreads();
        // labels only grow towards the unique fixed point, so the relaxation may read labels other lanes are updating
        // in the same sweep: fewer sweeps than Jacobi, one barrier per sweep
        unsigned nd = 0;                                     // non-direct alive slots of this lane
#pragma unroll
        for (int j = 0; j < NPL; ++j) if (((am >> j) & 1u) && !((nbr.direct >> j) & 1u)) nd |= 1u << j;
        for (int it = 0; it <= N; ++it) {
            bool ch = false;
#pragma unroll
            for (int j0 = 0; j0 < NPL; j0 += kNbGrp) {
                U4 pk[kNbGrp];
#pragma unroll
                for (int q = 0; q < kNbGrp; ++q) pk[q] = nb_words(nbr, j0 + q);
#pragma unroll
                for (int q = 0; q < kNbGrp; ++q) {
                    const int j = j0 + q;
                    const int i = j * 64 + lane;
                    double best = -1.0;                     // dead / unreached neighbours hold -1
                    if (!((nbr.ovf >> j) & 1u)) {
                        int idx[8]; double tn[8];
                        const unsigned ok = unpack8(pk[q].x, pk[q].y, pk[q].z, pk[q].w, i, idx);
                        wrsn_lds_gather8_b64(t, idx, tn);
#pragma unroll
                        for (int k = 0; k < 8; ++k) best = fmax(best, ((ok >> k) & 1u) ? tn[k] : -1.0);
                    } else WRSN_FOR_NEIGHBORS_CSR(i, nb, { best = fmax(best, t[nb]); })
                    const double cand = fmin(lt[j], best);
                    if (((nd >> j) & 1u) && cand > tc[j]) { tc[j] = cand; t[i] = cand; ch = true; }
                }
            }
            __syncthreads();
            if (!wv_any(ch)) break;
        }
        double mn = WRSN_INF;
        for (int q = lane; q < T; q += 64) {
            double v = 0.0; int idx[8];
            const unsigned ok = covering8(q, idx);
            if (ok != 0xFFFFFFFFu) {
                double tn[8]; wrsn_lds_gather8_b64(t, idx, tn);
#pragma unroll
                for (int k = 0; k < 8; ++k) v = fmax(v, ((ok >> k) & 1u) ? tn[k] : 0.0);
            } else for (int p = TC_OFF()[q]; p < TC_OFF()[q + 1]; ++p) v = fmax(v, t[TC_IDX()[p]]);
            mn = fmin(mn, v);
        }
        mn = wv_min(mn);
        __syncthreads();
        WRSN_PROF_ADD(7)
        return mn;
    }

    // -------------------------------------------------------------- charger energy pre-check sum (MobileCharger.py:111-115)
    WDEV double precheck(int ti) {
        const double dx = STH()[ti].phy[0], dy = STH()[ti].phy[1];
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            if ((am >> j) & 1u) {
                int i = j * 64 + lane;
                double dis = dist2(dx, dy, NX()[i], NY()[i]);
                if (dis <= EC()->charging_range) part += EC()->alpha / ((dis + EC()->beta) * (dis + EC()->beta));
            }
        }
        return wv_sum(part);
    }

    // -------------------------------------------------------------- connected_nodes of a charger (MobileCharger.py:55-58)
    // every node (alive or not) within charging range of the charger, id order; caches the node positions
    WDEV void conn_build(int a) {
        const double lx = SAG()[a].loc[0], ly = SAG()[a].loc[1];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            const double px = (i < N) ? NX()[i] : 0.0, py = (i < N) ? NY()[i] : 0.0;
            double dis = (i < N) ? dist2(px, py, lx, ly) : 0.0;
            bool in = (i < N) && (dis <= EC()->charging_range);
            unsigned long long mk = __ballot(in);
            if (in) {
                int pos = cnt + __popcll(mk & ((1ull << lane) - 1ull));
                if (pos < CC) {
                    SCONN()[a * CC + pos] = (int16_t)i;
                    SCONNXY()[2 * (a * CC + pos)] = px; SCONNXY()[2 * (a * CC + pos) + 1] = py;
                }
            }
            cnt += __popcll(mk);
        }
        if (cnt > CC) { err = -9; cnt = CC; }                 // cannot happen: CC bounds the nodes any disc of the charging range holds (wrsn_topology_kernel)
        if (lane == 0) { SAG()[a].n_conn = cnt; SAG()[a].conn_loc[0] = lx; SAG()[a].conn_loc[1] = ly; }
    }

    // the launch's deadline (wrsn_set_step_deadline) has passed: like an exhausted work budget
    WDEV bool past_deadline() const { return t_deadline != 0 && (long long)wall_clock64() > t_deadline; }

    // ============================================================== GRID LOOP (all lanes, wave-uniform registers)
    // The periodic items of the reference -- Network.operate (k+0.1 setLevels/check_targets, k+1.0 alive check),
    // update_reward (k+1.0) and the Node.operate block (k+0.5, k+1.0) -- are popped here in (time, seq) order until the
    // next item would not be strictly earlier than `t_limit` (the next charger / condition event).  When no node is
    // being charged, no reward priority is consumed, the consumption window is uniform and no node can run dry, whole
    // seconds are skipped in closed form (E -= j * (d1 + d2)).
    // `tie_seq` >= 0: the limit is a NORMAL-priority event with that sequence number, and the grid items AT t_limit that
    // were scheduled before it (smaller sequence number) go first too -- one service instead of one per item.
    WDEV void grid_run(double t_limit, bool one, bool ur_flag, int64_t tie_seq) {
        deaths_flag = 0;
        // Node.energyRR only changes when lane 0 connects / disconnects a charger, i.e. between two grid services
        const bool any_rr = SREQ()[2] > 0;
        double rrh[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) rrh[j] = 0.0;
        if (any_rr) rr_slots(rrh, 0.5);
        for (long guard = 0; guard < 4000000L; ++guard) {
            WRSN_PROF_MARK(lh0_)
            if (frozen) break;
            if (budget > 0 && guard > 0 && (work >= budget || past_deadline())) break;   // out of budget / time: the next launch asks for this service again
            int k = 1; double bt = ur_time; int64_t bs = ur_seq;
            if (node_time < bt || (node_time == bt && node_seq < bs)) { k = 2; bt = node_time; bs = node_seq; }
            if (net_active && (net_time < bt || (net_time == bt && net_seq < bs))) { k = 0; bt = net_time; bs = net_seq; }
            if (!one && !(bt < t_limit || (tie_seq >= 0 && bt == t_limit && bs < tie_seq))) break;
            bool do_ur = false, fused = false; int nrep = 0;
            const double kk = floor(bt);
            // ---- canonical start of a second (setLevels@k+0.1 is a no-op, nodes@k+0.5, reward/alive-check/nodes@k+1.0) on
            //      the steady path: nobody can run dry, the consumption window is uniform, the routing cache is valid
            // (while the consumption window is still filling after a routing change -- irreg > 0 -- the same holds for ONE second at a
            //  time: its five items run as one pass with the window update of Node.py:71-77 at the end, instead of five loop turns)
            // (a second in which the safe horizon has run out -- safe_ticks == 0, a node is within a few seconds of its threshold -- is
            //  looked at first: when nobody runs dry in it, it takes the same single pass)
            const bool irr = irreg > 0;
            bool guarded = false;
            if (!one && !cache_dirty && !log_pending && !levels_dirty && node_phase == 0 &&
                (net_active ? (k == 0 && net_phase == 0) : (k == 2)) && ur_time == kk + 1.0) {
                // whole seconds that fit before the next charger event / max_time and stay inside the safe horizon
                double jf = floor(fmin(t_limit, kk + 1.0e6) - kk);
                if (kk + jf >= t_limit) jf -= 1.0;
                if (net_active) { double jm = floor(fmin(max_time, kk + 1.0e6) - kk); if (kk + jm >= max_time) jm -= 1.0; jf = fmin(jf, jm); }
                int j = (int)fmin(jf, (double)safe_ticks);
                if (safe_ticks == 0 && jf >= 1.0) { work += 4; if (second_is_safe(rrh)) { guarded = true; j = 1; } }
                if (budget > 0 && j > 1 && j > budget - work) j = (budget - work > 1) ? budget - work : 1;
                if ((irr || guarded) && j > 1) j = 1;
                if (t_deadline != 0 && j > 8 && (any_rr || ur_flag)) {
                    // what is left of the launch, in seconds of the time-parallel steady path (about 0.35 us each, 35 ticks of the 100 MHz clock)
                    const long long left = (t_deadline - (long long)wall_clock64()) / WRSN_TICKS_PER_FUSED_SECOND;
                    if ((long long)j > left) j = left > 8 ? (int)left : 8;
                }
                if (j >= 1) {
                    if (!any_rr && !ur_flag && !irr && !guarded) {
                        // nothing but the constant per-second drain: closed form
                        const double dj = (double)j;
#pragma unroll
                        for (int q = 0; q < NPL; ++q) E[q] -= dj * (d1[q] + d2[q]);
                        WRSN_PROF_CNT(14, j) work += 1;
                    } else { fused = true; nrep = j; do_ur = ur_flag; WRSN_PROF_CNT(13, j) }
                    const double ke = kk + (double)j;
                    ur_time = ke + 1.0; node_time = ke + 1.0 * 0.5;
                    if (net_active) { net_time = ke + 1.0 / 10.0; seq += 5 * (int64_t)j; ur_seq = seq - 3; net_seq = seq - 2; node_seq = seq - 1; }
                    else { seq += 3 * (int64_t)j; ur_seq = seq - 2; node_seq = seq - 1; }
                    now = ke; n_ticks += j; n_items += j; if (!guarded) safe_ticks -= j;
                    if (!fused) continue;
                }
            }
            WRSN_PROF_MARK(lh1_) WRSN_PROF_SPAN(22, lh0_, lh1_) WRSN_PROF_CNT(23, 1)
            if (!fused) {
                // ---- one item (every O(N) routine has exactly one call site: the kernel has to fit the instruction cache)
                WRSN_PROF_CNT(15, 1) work += 8;
                WRSN_PROF_MARK(gi0_)
                const double now_before = now;
                now = bt;
                if (k == 0) {
                    if (net_phase == 0) {                    // Network.py:75-78
                        if (levels_dirty) set_levels();
                        if (!HEAVY && need_heavy) { now = now_before; break; }
                        if (alive == 0) frozen = 1;          // terminal at the next return; node state is no longer observable
                        net_phase = 1; net_time = now + 9.0 * 1.0 / 10.0; net_seq = seq++;
                    } else {                                 // Network.py:78-80
                        if (alive == 0 || now >= max_time) net_active = 0;
                        else { net_phase = 0; net_time = now + 1.0 / 10.0; net_seq = seq++; }
                    }
                } else if (k == 1) {
                    do_ur = ur_flag; nrep = do_ur ? 1 : 0;
                    ur_time = now + 1.0; ur_seq = seq++;
                } else {
                    if (node_phase == 0) {
                        node_half(rrh, any_rr);
                        if (need_heavy) { now = now_before; break; }
                        node_phase = 1;
                    } else { node_full(rrh, any_rr); node_phase = 0; }
                    node_time = now + 1.0 * 0.5; node_seq = seq++;
                }
                n_items++;
                WRSN_PROF_MARK(gi1_) WRSN_PROF_SPAN(8, gi0_, gi1_)
            }
            // ---- the steady loop: k+0.5 drain and half-charge (Node.py:60), reward instant, k+1.0 half-charge (Node.py:68),
            //      repeated for every whole second taken above; a lone reward item of the generic path runs it once
            int q0 = 0;
            if (fused && do_ur && SURN()[0] > 0) {
                while (nrep - q0 >= 8) {                     // batches of up to 64 seconds, one lane per second
                    const int nb = (nrep - q0 < 64) ? (nrep - q0) : 64;
                    if (!steady_batch(nb, rrh, any_rr)) break;
                    q0 += nb; work += nb;
                }
            }
            work += 4 * (nrep - q0);
            WRSN_PROF_MARK(ps0_) WRSN_PROF_CNT(20, nrep - q0)
            for (int q = q0; q < nrep; ++q) {
                if (fused) {
                    if (any_rr) {
#pragma unroll
                        for (int j = 0; j < NPL; ++j) { const double e = fmin(E[j] - d1[j] + rrh[j], cap) - d2[j]; E[j] = ((am >> j) & 1u) ? e : E[j]; }
                    } else {
#pragma unroll
                        for (int j = 0; j < NPL; ++j) E[j] = (E[j] - d1[j]) - d2[j];
                    }
                }
                if (do_ur) update_reward();                  // the one call site
                if (fused && any_rr) {
#pragma unroll
                    for (int j = 0; j < NPL; ++j) { const double e = fmin(E[j] + rrh[j], cap); E[j] = ((am >> j) & 1u) ? e : E[j]; }
                }
            }
            WRSN_PROF_MARK(ps1_) WRSN_PROF_SPAN(19, ps0_, ps1_)
            if (fused && irr) { window_update(); work += 8; }
            if (fused) continue;
            if (one || deaths_flag) break;
        }
        { WRSN_PROF_MARK(uf0_) if (ur_flag) ur_flush(); WRSN_PROF_MARK(uf1_) WRSN_PROF_SPAN(9, uf0_, uf1_) }
    }

    // ============================================================== SCALAR EVENT PROCESSOR (lane 0 only)

    WDEV void th_sched(int ti, int pc, int prio, double time) { STH()[ti].pc = pc; STH()[ti].prio = prio; STH()[ti].time = time; STH()[ti].seq = seq++; }

    WDEV void mc_check_status(int a) {                       // MobileCharger.py:134-140
        if (SAG()[a].energy <= EC()->mc_threshold) { SAG()[a].status = 0; SAG()[a].energy = EC()->mc_threshold; }
    }

    WDEV bool agent_single(int a, int ti) const {            // no other live operate_step process acts on this charger
        for (int i = 0; i < 2 * M; ++i) {
            if (i == ti) continue;
            int pc = STH()[i].pc;
            if (pc != PC_NONE && pc != PC_FINISHED && STH()[i].agent == a) return false;
        }
        return true;
    }

    // ---- fast-forward: unit sub-steps of move() / charge() whose only effect is on the charger itself are kept
    // virtual and applied lazily (identical arithmetic, applied in order); the process resumes at the timeout of the
    // last unit sub-step.  Only taken in generic position; ties with the node grid use the per-sub-step path.
    // t + 1.0 + 1.0 ... (n times) with the same roundings as n successive additions: adding 1.0 is exact inside a
    // binade, so only the (at most ~12) power-of-two crossings are done one by one
    WDEV static double add_ones(double t, int n) {
        while (n > 0) {
            int ex; (void)frexp(t, &ex);                      // t in [2^(ex-1), 2^ex)
            const double top = ldexp(1.0, ex);
            double room = floor(top - t); if (top - t == room) room -= 1.0;   // additions that stay below 2^ex
            if (t < 1.0 || room < 1.0) { t = t + 1.0; --n; continue; }
            const int m = (double)n < room ? n : (int)room;
            t = t + (double)m; n -= m;
        }
        return t;
    }
    // number of unit sub-steps (ends at ft+1, ft+2, ...) that have ended by time t, at most n
    WDEV static int due_units(double ft, double t, int n) {
        if (!(t >= ft + 1.0)) return 0;
        double c = floor(t - ft);
        int m = c < (double)n ? (int)c : n;
        while (m > 0 && add_ones(ft, m) > t) --m;            // guard the rounding of the estimate
        while (m < n && add_ones(ft, m + 1) <= t) ++m;
        return m;
    }

    WDEV void ff_apply(int ti, double t, bool all) {
        const int a = STH()[ti].agent;
        int n = STH()[ti].ff_n; double ft = STH()[ti].ff_t;
        if (STH()[ti].ff == 1) {
            const double ux = STH()[ti].mvec[0] / STH()[ti].total_time * 1.0, uy = STH()[ti].mvec[1] / STH()[ti].total_time * 1.0;
            const double de = EC()->pm * 1.0 * EC()->velocity;
            // closed form of m identical unit sub-steps: positions / energy differ from the step-by-step sums by a few
            // ulps (1e-13 relative), the time is accumulated exactly like successive `now + 1.0`
            const int m = all ? n : due_units(ft, t, n);
            const double dm = (double)m;
            SAG()[a].loc[0] = SAG()[a].loc[0] + dm * ux; SAG()[a].loc[1] = SAG()[a].loc[1] + dm * uy; SAG()[a].energy = SAG()[a].energy - dm * de;
            ft = add_ones(ft, m); n -= m;
        } else if (STH()[ti].ff == 2) {
            const double cr = SAG()[a].charging_rate;
            const int m = all ? n : due_units(ft, t, n);
            const double dm = (double)m;
            double c2 = SAG()[a].cur[2] - dm;                 // x - 1.0 is exact for x >= 1, so m unit steps == x - m
            SAG()[a].energy = SAG()[a].energy - dm * (cr * 1.0); SAG()[a].cur[2] = c2 > 0.0 ? c2 : 0.0; STH()[ti].tmp = STH()[ti].tmp - dm;
            ft = add_ones(ft, m); n -= m;
        }
        STH()[ti].ff_n = n; STH()[ti].ff_t = ft;
        if (n == 0) STH()[ti].ff = 0;
    }

    WDEV void ff_sync_all(double t) {
        for (int i = 0; i < 2 * M; ++i) if (STH()[i].ff != 0 && STH()[i].pc != PC_NONE && STH()[i].pc != PC_FINISHED) ff_apply(i, t, false);
    }

    // a connected node died (or a second process now shares the charger): return to the per-sub-step path at the
    // sub-step in flight
    WDEV void ff_fallback(int ti) {
        if (STH()[ti].ff == 0) return;
        ff_apply(ti, now, false);
        if (STH()[ti].ff == 0) return;                         // everything virtual was already due: the resume event stands
        const int kind = STH()[ti].ff;
        STH()[ti].ff = 0; STH()[ti].ff_n = 0;
        th_sched(ti, kind == 1 ? PC_MSTEP_TIMEOUT : PC_CSTEP_TIMEOUT, WRSN_NORMAL, STH()[ti].ff_t + 1.0);
    }

    WDEV void mc_move_loop(int ti) {                         // MobileCharger.py:85-94 from the top of `while True`
        int a = STH()[ti].agent;
        if (STH()[ti].moving_time <= 0.0) { th_sched(ti, PC_MOVE_DONE, WRSN_NORMAL, now); return; }
        if (SAG()[a].status == 0) { th_sched(ti, PC_MOVE_DEADWAIT, WRSN_NORMAL, now + STH()[ti].moving_time); return; }
        double mt = dist2(STH()[ti].m_dest[0], STH()[ti].m_dest[1], SAG()[a].loc[0], SAG()[a].loc[1]) / EC()->velocity;
        STH()[ti].moving_time = mt;
        double s = mt < 1.0 ? mt : 1.0;
        const double pmv = EC()->pm * EC()->velocity;
        double lim = (SAG()[a].energy - EC()->mc_threshold) / pmv;
        STH()[ti].span = s < lim ? s : lim;
        // fast-forward: a moving charger is observed by nobody until the run returns -- except through update_reward
        // when it is (stale) "charging" with connected nodes -- so whole-second sub-steps can stay virtual
        if (mt > 3.0 && lim > 3.0 && !(SAG()[a].type_charging && SAG()[a].n_conn > 0) && agent_single(a, ti)) {
            double nf = floor(fmin(mt, lim)) - 1.0;          // the remainder (> 1 s) and the energy margin go through the exact path
            if (nf > 100000.0) nf = 100000.0;
            int n = (int)nf;
            if (n >= 2) {
                const double t = add_ones(now, n);           // same float accumulation as n successive timeouts
                STH()[ti].span = 1.0; STH()[ti].ff = 1; STH()[ti].ff_n = n - 1; STH()[ti].ff_t = now;
                th_sched(ti, PC_MSTEP_TIMEOUT, WRSN_NORMAL, t);
                return;
            }
        }
        WRSN_PROF_EV(0, (mt > 3.0 && SAG()[a].type_charging && SAG()[a].n_conn > 0) ? 1 : 0) WRSN_PROF_EV(18, (mt > 3.0 && !agent_single(a, ti)) ? 1 : 0)
        th_sched(ti, PC_MSTEP_INIT, WRSN_URGENT, now);
    }

    // Node.charger_connection / charger_disconnection over connected_nodes (Node.py:134-146); sign = +1 / -1
    WDEV int mc_connect(int a, double sign) {
        const int nc = SAG()[a].n_conn;
        double cr = SAG()[a].charging_rate; int cnt = 0;
        for (int k = 0; k < nc; ++k) {
            int i = SCONN()[a * CC + k];
            if (!(SLS()[i] & 1)) continue;
            double r = conn_rate_of(a, k, i);
            rr_add(i, sign * r); cr += sign * r; cnt++;
        }
        SAG()[a].charging_rate = cr;
        if (sign > 0) { SAG()[a].n_live = cnt; SREQ()[2] += cnt; }
        else { SREQ()[2] -= SAG()[a].n_live; SAG()[a].n_live = 0; if (SREQ()[2] < 0) SREQ()[2] = 0; }
        return cnt;
    }

    WDEV void mc_charge_loop(int ti) {                       // MobileCharger.py:59-69 from the top of `while True`
        int a = STH()[ti].agent;
        if (STH()[ti].tmp == 0.0) { th_sched(ti, PC_CHG_DONE, WRSN_NORMAL, now); return; }
        if (SAG()[a].status == 0) { SAG()[a].cur[2] = 0.0; th_sched(ti, PC_CHG_DEADWAIT, WRSN_NORMAL, now + STH()[ti].tmp); return; }
        double span = STH()[ti].tmp < 1.0 ? STH()[ti].tmp : 1.0;
        if (SAG()[a].charging_rate != 0.0) { double lim = (SAG()[a].energy - EC()->mc_threshold) / SAG()[a].charging_rate; if (lim < span) span = lim; }
        STH()[ti].cspan = span;
        // fast-forward: in generic position (no sub-step boundary on a node sampling instant k+0.5 / k+1.0) the
        // disconnect/reconnect pair of every boundary cancels, so the connection is made once and the whole-second
        // sub-steps stay virtual.  A node death or the charger running dry return to the exact path.
        const double tmp = STH()[ti].tmp;
        const double fr = now - floor(now);
        if (tmp > 3.0 && fr != 0.0 && fr != 0.5 && SAG()[a].charging_rate == 0.0 && agent_single(a, ti)) {
            double nf = floor(tmp); if (nf == tmp) nf -= 1.0;          // unit sub-steps that are not the last one
            // rate this sub-step would connect with
            double cr = 0.0; const int nc = SAG()[a].n_conn;
            for (int k = 0; k < nc; ++k) { int i = SCONN()[a * CC + k]; if (SLS()[i] & 1) cr += conn_rate_of(a, k, i); }
            if (cr > 0.0) { double ne = floor((SAG()[a].energy - EC()->mc_threshold) / cr) - 1.0; if (ne < nf) nf = ne; }
            if (nf > 100000.0) nf = 100000.0;
            int n = (int)nf;
            // every boundary now + q keeps (up to the last bits) the fractional part of `now`; stay clear of the grid
            bool ok = n >= 2 && fr > 1e-6 && fr < 1.0 - 1e-6 && (fr < 0.5 - 1e-6 || fr > 0.5 + 1e-6);
            const double t = add_ones(now, ok ? n : 0);
            if (ok) {
                mc_connect(a, 1.0);                          // charge_step #1 connects (MobileCharger.py:40-41)
                STH()[ti].cspan = 1.0; STH()[ti].ff = 2; STH()[ti].ff_n = n - 1; STH()[ti].ff_t = now;
                th_sched(ti, PC_CSTEP_TIMEOUT, WRSN_NORMAL, t);
                return;
            }
        }
        th_sched(ti, PC_CSTEP_INIT, WRSN_URGENT, now);
    }

    WDEV void cond_trigger(int j) {                          // Condition.succeed(): NORMAL at now
        if (SCTR()[j]) return;
        SCTR()[j] = 1; SCP()[j] = 1; SCT()[j] = now; SCS()[j] = seq++;
    }

    WDEV void p_init_tail(int ti, double tmp) {              // MobileCharger.py:110, 116-121 / 128-130
        const int a = STH()[ti].agent;
        const double dx = STH()[ti].phy[0], dy = STH()[ti].phy[1], ct = STH()[ti].phy[2];
        double used = dist2(dx, dy, SAG()[a].loc[0], SAG()[a].loc[1]) * EC()->pm;
        used += tmp * ct;
        used += dist2(dx, dy, EC()->bs[0], EC()->bs[1]) * EC()->pm;
        SAG()[a].cur[0] = dx; SAG()[a].cur[1] = dy; SAG()[a].cur[2] = ct;
        SAG()[a].type_charging = 0;
        if (used > SAG()[a].energy - EC()->mc_threshold - EC()->mc_capacity / 200.0) { STH()[ti].stage = 0; STH()[ti].m_dest[0] = EC()->bs[0]; STH()[ti].m_dest[1] = EC()->bs[1]; }
        else { STH()[ti].stage = 2; STH()[ti].m_dest[0] = dx; STH()[ti].m_dest[1] = dy; }
        if (SAG()[a].cur_thread == ti) {                     // launch ordering: when the process WRSN.step waits for will finish
            const double way = (STH()[ti].stage == 0)
                ? dist2(EC()->bs[0], EC()->bs[1], SAG()[a].loc[0], SAG()[a].loc[1]) + dist2(dx, dy, EC()->bs[0], EC()->bs[1])
                : dist2(dx, dy, SAG()[a].loc[0], SAG()[a].loc[1]);
            SAG()[a].t_done = now + way / EC()->velocity + ct;
        }
        th_sched(ti, PC_MOVE_INIT, WRSN_URGENT, now);
    }

    // returns a wave request (REQ_PRECHECK / REQ_CONN) or 0
    WDEV int thread_fire(int ti) {
        const int a = STH()[ti].agent;
        if (STH()[ti].ff != 0) ff_apply(ti, now, true);        // the resume event: every virtual sub-step precedes it
        WRSN_PROF_EV(STH()[ti].pc, 1)
        switch (STH()[ti].pc) {
        case PC_P_INIT:                                      // MobileCharger.py:105-115: needs the O(N) sum
            for (int i = 0; i < 2 * M; ++i) if (i != ti && STH()[i].agent == a) ff_fallback(i);   // a second process on this charger
            SS()->pend = REQ_PRECHECK; SS()->pend_idx = ti; return REQ_PRECHECK;
        case PC_MOVE_INIT: {                                 // MobileCharger.py:82-84
            double mt = dist2(STH()[ti].m_dest[0], STH()[ti].m_dest[1], SAG()[a].loc[0], SAG()[a].loc[1]) / EC()->velocity;
            STH()[ti].moving_time = mt; STH()[ti].total_time = mt;
            STH()[ti].mvec[0] = STH()[ti].m_dest[0] - SAG()[a].loc[0]; STH()[ti].mvec[1] = STH()[ti].m_dest[1] - SAG()[a].loc[1];
            mc_move_loop(ti);
            break; }
        case PC_MSTEP_INIT:                                  // MobileCharger.py:76
            th_sched(ti, PC_MSTEP_TIMEOUT, WRSN_NORMAL, now + STH()[ti].span);
            break;
        case PC_MSTEP_TIMEOUT:                               // MobileCharger.py:77-78
            SAG()[a].loc[0] = SAG()[a].loc[0] + STH()[ti].mvec[0] / STH()[ti].total_time * STH()[ti].span;
            SAG()[a].loc[1] = SAG()[a].loc[1] + STH()[ti].mvec[1] / STH()[ti].total_time * STH()[ti].span;
            SAG()[a].energy -= EC()->pm * STH()[ti].span * EC()->velocity;
            th_sched(ti, PC_MSTEP_DONE, WRSN_NORMAL, now);
            break;
        case PC_MSTEP_DONE:                                  // MobileCharger.py:95-96
            STH()[ti].moving_time -= STH()[ti].span;
            mc_check_status(a);
            mc_move_loop(ti);
            break;
        case PC_MOVE_DEADWAIT:
            th_sched(ti, PC_MOVE_DONE, WRSN_NORMAL, now);
            break;
        case PC_MOVE_DONE:
            if (STH()[ti].stage == 0) th_sched(ti, PC_RECH_INIT, WRSN_URGENT, now);                    // :123
            else { SAG()[a].type_charging = 1; th_sched(ti, PC_CHG_INIT, WRSN_URGENT, now); }         // :125-126 / :131-132
            break;
        case PC_RECH_INIT:                                   // MobileCharger.py:99-103
            if (dist2(SAG()[a].loc[0], SAG()[a].loc[1], EC()->bs[0], EC()->bs[1]) <= EC()->epsilon) {
                SAG()[a].loc[0] = EC()->bs[0]; SAG()[a].loc[1] = EC()->bs[1]; SAG()[a].energy = EC()->mc_capacity;
            }
            th_sched(ti, PC_RECH_TIMEOUT, WRSN_NORMAL, now + 0.0);
            break;
        case PC_RECH_TIMEOUT:
            th_sched(ti, PC_RECH_DONE, WRSN_NORMAL, now);
            break;
        case PC_RECH_DONE:                                   // :124
            STH()[ti].stage = 2; STH()[ti].m_dest[0] = STH()[ti].phy[0]; STH()[ti].m_dest[1] = STH()[ti].phy[1];
            th_sched(ti, PC_MOVE_INIT, WRSN_URGENT, now);
            break;
        case PC_CHG_INIT:                                    // MobileCharger.py:52-58: needs the O(N) range scan
            STH()[ti].tmp = STH()[ti].phy[2];
            SS()->pend = REQ_CONN; SS()->pend_idx = ti; return REQ_CONN;
        case PC_CSTEP_INIT:                                  // MobileCharger.py:40-44
            mc_connect(a, 1.0);
            th_sched(ti, PC_CSTEP_TIMEOUT, WRSN_NORMAL, now + STH()[ti].cspan);
            break;
        case PC_CSTEP_TIMEOUT: {                             // MobileCharger.py:45-50
            SAG()[a].energy = SAG()[a].energy - SAG()[a].charging_rate * STH()[ti].cspan;
            double rem = SAG()[a].cur[2] - STH()[ti].cspan;
            SAG()[a].cur[2] = rem > 0.0 ? rem : 0.0;
            mc_connect(a, -1.0);
            SAG()[a].charging_rate = 0.0;
            th_sched(ti, PC_CSTEP_DONE, WRSN_NORMAL, now);
            break; }
        case PC_CSTEP_DONE:                                  // MobileCharger.py:70-72
            STH()[ti].tmp -= STH()[ti].cspan;
            mc_check_status(a);
            mc_charge_loop(ti);
            break;
        case PC_CHG_DEADWAIT:
            th_sched(ti, PC_CHG_DONE, WRSN_NORMAL, now);
            break;
        case PC_CHG_DONE:
            th_sched(ti, PC_P_DONE, WRSN_NORMAL, now);
            break;
        case PC_P_DONE:                                      // the process event is processed: conditions of WRSN.step see it
            STH()[ti].pc = PC_FINISHED;
            for (int j = 1; j <= SS()->L; ++j) if (SAG()[SCA()[j - 1]].cur_thread == ti) cond_trigger(j);
            break;
        default: break;
        }
        return 0;
    }

    WDEV int new_thread(int agent, double p0, double p1, double p2) {
        for (int i = 0; i < 2 * M; ++i) {
            int pc = STH()[i].pc;
            if (pc == PC_NONE || (pc == PC_FINISHED && SAG()[STH()[i].agent].cur_thread != i)) {
                STH()[i].agent = agent; STH()[i].phy[0] = p0; STH()[i].phy[1] = p1; STH()[i].phy[2] = p2;
                STH()[i].stage = 0; STH()[i].moving_time = 0; STH()[i].total_time = 0; STH()[i].span = 0; STH()[i].tmp = 0; STH()[i].cspan = 0;
                STH()[i].ff = 0; STH()[i].ff_n = 0; STH()[i].ff_t = 0;
                th_sched(i, PC_P_INIT, WRSN_URGENT, now);
                return i;
            }
        }
        return -1;
    }

    WDEV bool agent_at_rest(int m) const {                   // WRSN.py:66 / :322
        return dist2(SAG()[m].loc[0], SAG()[m].loc[1], SAG()[m].cur[0], SAG()[m].cur[1]) < 1e-9 && SAG()[m].cur[2] == 0.0;
    }

    // update_reward consumes the priorities only for alive chargers whose action type is "charging" and that have
    // connected nodes (WRSN.py:116-126); bit 1: such a charger is being moved by a (stale) process, so its location
    // has to be brought up to date before every reward instant
    WDEV int ur_flags() const {
        int f = 0;
        for (int m = 0; m < M; ++m) {
            if (SAG()[m].status != 0 && SAG()[m].type_charging && SAG()[m].n_conn > 0) {
                f |= 1;
                for (int i = 0; i < 2 * M; ++i) if (STH()[i].agent == m && STH()[i].ff == 1 && STH()[i].pc != PC_NONE && STH()[i].pc != PC_FINISHED) f |= 2;
            }
        }
        return f;
    }

    WDEV static bool key_less(double t1, int p1, int64_t s1, double t2, int p2, int64_t s2) {
        if (t1 != t2) return t1 < t2;
        if (p1 != p2) return p1 < p2;
        return s1 < s2;
    }

    // Fire charger / condition events in order until the wave has to do something: run the grid up to the next
    // event (REQ_GRID), an O(N) service (REQ_PRECHECK / REQ_CONN), or the run stops (REQ_STOP).
    // The cached "next event" lives in registers while lane 0 runs and in the LDS Scalar block between two services.
    // Every lane calls scalar_run: lane 0 decides and fires (scalar_iter), the whole wave looks for the next event
    // (event_scan) whenever lane 0's cached one is spent.  The return value is wave-uniform; *arg, *t_lim_out,
    // *flags_out and the advanced `now` / `seq` are lane 0's.
    struct EvCache { int valid, kind, idx, prio, uf; double time, t2; int64_t seq; int64_t fired; };
    WDEV static int lane0(int v) { return __builtin_amdgcn_readlane(v, 0); }
    WDEV static double lane_f64(double v, int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    }

    // The pending charger-process / condition event with the smallest SimPy key (time, priority, insertion id), and the
    // earliest time among the others (t2): one lane per candidate -- lanes 0 .. 2M-1 the operate_step processes, the next L
    // lanes the `|` conditions of the step -- and three wave minima instead of a lane-0 loop over the candidates.
    WDEV void event_scan(const int L, int& kind, int& idx, double& bt, int& bp, int64_t& bs, double& t2) {
        const int nth = 2 * M;
        bool valid = false; double t = WRSN_INF; int pr = 0; int64_t sq = 0; int kd = -1, ix = 0;
        if (lane < nth) {
            const int pc = STH()[lane].pc;
            valid = pc != PC_NONE && pc != PC_FINISHED; t = STH()[lane].time; pr = STH()[lane].prio; sq = STH()[lane].seq; kd = 3; ix = lane;
        } else if (lane - nth < L) {
            const int j = lane - nth + 1;
            valid = SCP()[j] != 0; t = SCT()[j]; pr = WRSN_NORMAL; sq = SCS()[j]; kd = 4; ix = j;
        }
        const unsigned long long anyv = __ballot(valid);
        if (anyv == 0ull) { kind = -1; idx = 0; bt = 0.0; bp = 0; bs = 0; t2 = WRSN_INF; return; }
        const double tv = valid ? t : WRSN_INF;
        const double tmin = wv_min(tv);
        // insertion ids are unique and far below 2^52: (priority, id) orders as one exact double
        const double k2 = (valid && tv == tmin) ? (double)sq + (double)pr * 4503599627370496.0 : WRSN_INF;
        const double kmin = wv_min(k2);
        const unsigned long long win = __ballot(valid && tv == tmin && k2 == kmin);
        const int w = __popcll((win & (~win + 1ull)) - 1ull);  // the winning lane
        t2 = wv_min(lane == w ? WRSN_INF : tv);
        kind = __builtin_amdgcn_readlane(kd, w); idx = __builtin_amdgcn_readlane(ix, w); bp = __builtin_amdgcn_readlane(pr, w);
        bt = lane_f64(t, w);
        const unsigned slo = (unsigned)__builtin_amdgcn_readlane((int)(sq & 0xffffffffll), w); const int shi = __builtin_amdgcn_readlane((int)(sq >> 32), w);
        bs = ((int64_t)shi << 32) | (int64_t)slo;
    }

    WDEV int scalar_run(double svc, bool use_limit, double limit, int* arg, double* t_lim_out, int* flags_out) {
        Scalar* ss = SS();
        WRSN_PROF_MARK(sr0_)
        WRSN_P4_MARK(q0_)
        EvCache ev; ev.valid = ss->ev_valid; ev.kind = ss->ev_kind; ev.idx = ss->ev_idx; ev.prio = ss->ev_prio; ev.uf = ss->ev_uf;
        ev.time = ss->ev_time; ev.t2 = ss->ev2_time; ev.seq = ss->ev_seq; ev.fired = 0;
        const int L = ss->L; const int pend0 = ss->pend, pend_idx0 = ss->pend_idx;
        int pend_out = 0, r = REQ_STOP;
        if (lane == 0) {
            switch (pend0) {                                  // finish the item that asked for the service
            case REQ_PRECHECK: p_init_tail(pend_idx0, svc); ev.valid = 0; break;
            case REQ_CONN: mc_charge_loop(pend_idx0); ev.valid = 0; break;
            case REQ_GRID:
                if (deaths_flag) {                           // a node died: chargers connected to it re-plan on the exact path
                    for (int i = 0; i < 2 * M; ++i) if (STH()[i].ff == 2 && STH()[i].pc != PC_NONE && STH()[i].pc != PC_FINISHED && SAG()[STH()[i].agent].n_live > 0) ff_fallback(i);
                    ev.valid = 0;
                }
                break;
            default: break;
            }
        }
        WRSN_P4_MARK(q1_) WRSN_P4_SPAN(8, q0_, q1_)
        long guard = 0;
        for (;; ++guard) {                                   // a step spans at most a few thousand seconds
            if (guard >= 4000000L) { err = -6; r = REQ_STOP; break; }
            WRSN_P4_CNT(1, 1) WRSN_P4_MARK(q2_)
            if (lane0(ev.valid) == 0) { WRSN_P4_CNT(3, 1)     // charger / condition state only changes when one of them fires
                __syncthreads();                             // what lane 0 wrote is what the other lanes read
                event_scan(L, ev.kind, ev.idx, ev.time, ev.prio, ev.seq, ev.t2);
                ev.valid = 1; ev.uf = -1;
                WRSN_P4_MARK(q3_) WRSN_P4_SPAN(2, q2_, q3_)
            }
            int rr = -1;
            if (lane == 0) rr = scalar_iter(ev, L, &pend_out, use_limit, limit, arg, t_lim_out, flags_out);
            rr = lane0(rr);
            if (rr >= 0) { r = rr; break; }
        }
        if (lane == 0) {
            ss->ev_valid = ev.valid; ss->ev_kind = ev.kind; ss->ev_idx = ev.idx; ss->ev_prio = ev.prio; ss->ev_uf = ev.uf;
            ss->ev_time = ev.time; ss->ev2_time = ev.t2; ss->ev_seq = ev.seq; ss->n_events += ev.fired; ss->pend = pend_out;
        }
        return r;
    }

    // lane 0: one decision of the event machine on the cached next event.  Returns -1 (go on: the event fired, or the cache is
    // spent) or the request for the wave.
    WDEV int scalar_iter(EvCache& ev, const int L, int* pend_out, bool use_limit, double limit, int* arg, double* t_lim_out, int* flags_out) {
            WRSN_P4_MARK(q4_)
            const int kind = ev.kind, idx = ev.idx, bp = ev.prio; const double bt = ev.time; const int64_t bs = ev.seq;
            const bool have_ev = kind >= 0;
            // next grid item (wave-uniform registers; lane 0 holds the same copy)
            bool have_grid = !frozen;
            double gt = ur_time; int64_t gs = ur_seq;
            if (have_grid) {
                if (node_time < gt || (node_time == gt && node_seq < gs)) { gt = node_time; gs = node_seq; }
                if (net_active && (net_time < gt || (net_time == gt && net_seq < gs))) { gt = net_time; gs = net_seq; }
            }
            if (!have_ev && !have_grid) { err = -7; return REQ_STOP; }      // cannot happen while a charger process runs
            double t_lim = have_ev ? bt : WRSN_INF;
            if (use_limit && limit < t_lim) t_lim = limit;
            if (have_grid && (gt < t_lim || (have_ev && gt == bt)) && ev.uf < 0) {
                // the reward entry list is only needed by a grid service: (re)build it lazily
                WRSN_P4_MARK(q5_) WRSN_P4_CNT(5, 1)
                ev.uf = ur_flags();
                if (ev.uf & 1) ur_build(); else SURN()[0] = 0;
                WRSN_P4_MARK(q6_) WRSN_P4_SPAN(4, q5_, q6_)
                WRSN_PROF_EV(21, 1) WRSN_PROF_EV(23, SURN()[0])
            }
            // items AT the event's instant that were scheduled before it precede it as well (same time, same NORMAL priority,
            // smaller sequence number): the service may take them along
            const bool tie_ok = have_ev && t_lim == bt && bp == WRSN_NORMAL;
            if (have_grid && (gt < t_lim || (tie_ok && gt == bt && gs < bs))) {
                const int uf = ev.uf;
                if (uf & 2) { WRSN_P4_MARK(q7_) ff_sync_all(gt); ur_build(); *arg = 1; WRSN_P4_MARK(q8_) WRSN_P4_SPAN(9, q7_, q8_) }   // stale "charging" mover: one item at a time, location kept current
                else if (tie_ok) { *arg = 2; ((int64_t*)SREQD())[3] = bs; }
                else *arg = 0;
                *t_lim_out = t_lim; *flags_out = uf & 1;
                WRSN_P4_MARK(q9_) WRSN_P4_SPAN(10, q4_, q9_)
                WRSN_PROF_EV(19, 1) WRSN_PROF_EV(20, (uf & 2) ? 1 : 0)
                *pend_out = REQ_GRID; return REQ_GRID;
            }
            if (use_limit && !(have_ev && bt < limit)) { now = limit; return REQ_STOP; }
            if (have_grid && have_ev && gt == bt && key_less(gt, WRSN_NORMAL, gs, bt, bp, bs)) {
                const int uf = ev.uf;
                if (uf & 2) { ff_sync_all(gt); ur_build(); }
                *arg = 1; *t_lim_out = t_lim; *flags_out = uf & 1;
                WRSN_PROF_EV(22, 1)
                *pend_out = REQ_GRID; return REQ_GRID;            // tie at one instant: exactly one grid item goes first
            }
            now = bt; ev.fired++; ev.valid = 0;
            WRSN_P4_MARK(q10_) WRSN_P4_SPAN(10, q4_, q10_)
            if (kind == 3) {
                WRSN_P4_MARK(q11_) WRSN_P4_CNT(7, 1)
                int r = thread_fire(idx);
                WRSN_P4_MARK(q12_) WRSN_P4_SPAN(6, q11_, q12_)
                if (r) { *arg = (r == REQ_CONN) ? STH()[idx].agent : idx; *pend_out = r; return r; }
                // the same process usually owns the next event too (its hops at one instant, or its next timeout is
                // the earliest): no rescan when nothing else can come first
                const int pc2 = STH()[idx].pc;
                if (pc2 != PC_NONE && pc2 != PC_FINISHED) {
                    const double t2 = STH()[idx].time;
                    if (t2 < ev.t2 && (!have_grid || t2 < gt)) {
                        ev.kind = 3; ev.idx = idx; ev.time = t2; ev.prio = STH()[idx].prio; ev.seq = STH()[idx].seq;
                        ev.valid = 1; ev.uf = -1;
                    }
                }
            } else {
                SCP()[idx] = 0;
                if (idx == L) return REQ_STOP;               // StopSimulation
                cond_trigger(idx + 1);
            }
            return -1;
    }

    // drive the environment until the run stops: lane 0 fires charger events, the wave runs the grid and the O(N) services
    // `budget` > 0 bounds the work of one launch in units of roughly 400 cycles: 1 per second of the time-parallel steady
    // batch, 4 per second of the per-second steady path, 8 per generic grid item, 16 per service, 80 / 100 / 500 per
    // routing-cache rebuild / level BFS / packet-exact second.  The grid service stops at an item boundary when the
    // budget is used up, the run is suspended in front of the next grid service (returns true) and the following
    // launch goes on from the stored state: lane 0 simply takes the same decision again (none of its side effects is
    // not idempotent).  Deterministic: the launch a request appears in does not depend on timing.
    WDEV bool run(bool use_limit, double limit, int budget_ = 0) {
        double svc = 0.0;
        work = 0; budget = budget_; n_items = 0;
        bool suspended = false, stopped = false;
        for (long guard = 0; guard < 8000000L; ++guard) {
            WRSN_P4_MARK(r0_)
            { WRSN_PROF_T0
            {
                int arg = 0, fl = 0; double tl = 0.0;
                WRSN_P4_MARK(r1_)
                const int req = scalar_run(svc, use_limit, limit, &arg, &tl, &fl);     // every lane; lane 0 holds the details
                WRSN_P4_MARK(r2_) WRSN_P4_SPAN(0, r1_, r2_)
                if (lane == 0) { SREQ()[0] = req; SREQ()[1] = arg; SREQ()[3] = fl; SREQD()[0] = tl; SREQD()[1] = now; ((int64_t*)SREQD())[2] = seq; }
            }
            __syncthreads();
            WRSN_PROF_ADD(0) WRSN_PROF_CNT(12, 1) }
            const int req = SREQ()[0], arg = SREQ()[1];
            now = wu(SREQD()[1]); seq = wu(((const int64_t*)SREQD())[2]);      // lane 0 advanced them while firing events
            WRSN_P4_MARK(r3_) WRSN_P4_SPAN(11, r0_, r3_) WRSN_P4_CNT(20, 1)
            if (req == REQ_STOP) { stopped = true; break; }
            work += 16;
            // (guard > 0: whatever the budget -- a user budget of a few units, or the tapered share of a late block -- the first request of a launch
            //  is served, so that every environment advances by at least one grid item per launch)
            if (budget > 0 && req == REQ_GRID && guard > 0 && (work >= budget || past_deadline())) { suspended = true; break; }
            switch (req) {
            case REQ_GRID: { WRSN_PROF_T0 WRSN_P4_CNT(21, 1) fit_dirty = 1; map1_valid = 0; grid_run(SREQD()[0], (arg & 1) != 0, SREQ()[3] != 0, (arg & 2) ? ((const int64_t*)SREQD())[3] : (int64_t)-1); WRSN_PROF_ADD(1) } break;
            case REQ_PRECHECK: { svc = precheck(arg); } break;
            case REQ_CONN: { conn_build(arg); } break;
            default: break;
            }
            __syncthreads();
            WRSN_P4_MARK(r4_) if (req == REQ_GRID) { WRSN_P4_SPAN(14, r3_, r4_) } else if (req == REQ_PRECHECK) { WRSN_P4_SPAN(12, r3_, r4_) } else { WRSN_P4_SPAN(13, r3_, r4_) }
            if (need_heavy) { suspended = true; break; }     // a packet-exact second put off to the next (time-sliced) launch
        }
        if (!stopped && !suspended) err = -10;               // the service loop ran out: the environment is stuck, report it (status < 0)
        __syncthreads();
        if (lane == 0 && !suspended) ff_sync_all(now);       // bring virtual charger sub-steps up to the return instant
        __syncthreads();
        return suspended;
    }
};

// ------------------------------------------------------------------ the environment kernels
// wrsn_warmup_kernel: t = 0 .. warm_up_time with no charger activity, snapshot into d.snap      (WRSN.py:41-64)
// wrsn_step_kernel  : WRSN.step (WRSN.py:289-330); with `reset_call` (or auto-reset of a terminal environment) it
//                     restores the snapshot into d.live and emits the reset request instead        (WRSN.py:66-75)
// Two kernels so that the event machine and every O(N) routine are instantiated once per code object.
// two waves per SIMD (256 registers) for up to 256 nodes: the event machine is latency-bound, a second wave hides it
// two waves per SIMD (256 registers) for up to 256 nodes, one above.  (r03 measured three waves per SIMD at 168 registers, with and
// without the rare services in the code object: slower, the common path spills there -- DESIGN.md 4.1)
#ifndef WRSN_WAVES_PER_SIMD
#define WRSN_WAVES_PER_SIMD(NPL_) ((NPL_) <= 4 ? 2 : 1)
#endif
// WRSN_KERNEL_INLINE = true (the product): the step kernel is ONE code object with the rare services inlined.  false builds the r03
// experiment -- the common-path simulator Sim<NPL, false> in the kernel, the full one behind a noinline call for the one grid item that
// needs it (wrsn_step_env_full): measured 2 x slower (1.03 ms against 0.51 ms per launch, profiles/r03_ab_load_eventmachine.log), as was
// the first form of it (the three services as noinline functions taking the node registers through memory: 1.24 ms -- the register
// allocator puts the spills of a call at the head of the enclosing hot region, not at the rare call site).  tools/ab_build.sh.
#ifndef WRSN_KERNEL_INLINE
#define WRSN_KERNEL_INLINE true
#endif
template <int NPL>
__global__ void __launch_bounds__(64, WRSN_WAVES_PER_SIMD(NPL)) wrsn_warmup_kernel(const WrsnDev* __restrict__ dp, int env0) {
    extern __shared__ double smem[];
    const int env = env0 + blockIdx.x;
    const int lane = threadIdx.x;
    if (env >= dp->B) return;
    Sim<NPL, true> s;                                        // (one-off: the full simulator, everything inline)
    s.bind(dp, env, lane, smem);
    const WrsnEnvConst* ec = s.EC();
    // NetworkIO.makeNetwork + Node.__init__ (Node.py:12-43) + t = 0 process start-up
    s.am = 0;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        int i = j * 64 + lane;
        bool real = i < s.N;
        s.E[j] = real ? ec->capacity : 0.0; s.CS[j] = 0.0; s.d1[j] = 0.0; s.d2[j] = 0.0;
        int al = (real && ec->capacity > ec->threshold) ? 1 : 0;
        s.SLS()[i] = al; s.SRCV()[i] = -1;
        s.am |= (unsigned)al << j;
    }
    s.now = 0.0; s.seq = 0; s.last_minfit = 0.0; s.opmax = 0.0;
    s.n_ticks = s.n_exact = 0;
    s.alive = 1; s.levels_dirty = 1; s.cache_dirty = 1; s.irreg = WRSN_RING; s.ring_len = 0; s.ring_head = 0;
    s.safe_ticks = 0; s.frozen = 0; s.log_pending = 0;
    uint64_t* la = (uint64_t*)s.SAG();
    for (int w = lane; w < s.M * (int)(sizeof(WrsnAgent) / 8); w += 64) la[w] = 0;
    uint64_t* lt = (uint64_t*)s.STH();
    for (int w = lane; w < 2 * s.M * (int)(sizeof(WrsnThread) / 8); w += 64) lt[w] = 0;
    for (int w = lane; w < s.M * s.CC; w += 64) { s.SCONN()[w] = 0; s.SCONNXY()[2 * w] = 0.0; s.SCONNXY()[2 * w + 1] = 0.0; }
    for (int w = lane; w <= s.M; w += 64) { s.SCTR()[w] = 0; s.SCP()[w] = 0; s.SCA()[w] = 0; s.SCT()[w] = 0; s.SCS()[w] = 0; }
    __syncthreads();
    if (lane == 0) {
        s.SREQ()[0] = 0; s.SREQ()[1] = 0; s.SREQ()[2] = 0; s.SURN()[0] = 0; s.SRRN()[0] = 0;
        for (int m = 0; m < s.M; ++m) {                      // MobileCharger.__init__ + WRSN.py:44-49
            s.SAG()[m].loc[0] = ec->bs[0]; s.SAG()[m].loc[1] = ec->bs[1]; s.SAG()[m].energy = ec->mc_capacity; s.SAG()[m].charging_rate = 0.0;
            s.SAG()[m].status = 1; s.mc_check_status(m);
            s.SAG()[m].type_charging = 0; s.SAG()[m].n_conn = 0; s.SAG()[m].cur_thread = -1; s.SAG()[m].n_live = 0;
            s.SAG()[m].cur[0] = ec->bs[0]; s.SAG()[m].cur[1] = ec->bs[1]; s.SAG()[m].cur[2] = 0.0;
            s.SAG()[m].excl = 0.0; s.SAG()[m].prev_minfit = 0.0;
            s.SAG()[m].conn_loc[0] = ec->bs[0]; s.SAG()[m].conn_loc[1] = ec->bs[1];
        }
    }
    // Network.operate -> timeout(0.1); update_reward body at t = 0 (no charger is charging) -> timeout(1); nodes -> timeout(0.5)
    s.net_active = 1; s.net_phase = 0; s.net_time = s.now + 1.0 / 10.0; s.net_seq = s.seq++;
    s.ur_time = s.now + 1.0; s.ur_seq = s.seq++;
    s.node_phase = 0; s.node_time = s.now + 1.0 * 0.5; s.node_seq = s.seq++;
    s.use_snap = 1;
    __syncthreads();
    s.run(true, ec->warm_up_time);                           // env.run(until=warm_up_time): stops before that instant's NORMAL events
    double fit = s.min_fitness();
    s.last_minfit = fit; s.fit_dirty = 0; s.map1_valid = 0;
    if (lane == 0) {
        for (int m = 0; m < s.M; ++m) {                      // WRSN.py:59-64
            s.SAG()[m].action[0] = (ec->bs[0] - ec->frame[0]) / (ec->frame[1] - ec->frame[0]);
            s.SAG()[m].action[1] = (ec->bs[1] - ec->frame[2]) / (ec->frame[3] - ec->frame[2]);
            s.SAG()[m].action[2] = 0.0;
            s.SAG()[m].cur_thread = s.new_thread(m, s.SAG()[m].cur[0], s.SAG()[m].cur[1], s.SAG()[m].cur[2]);
            s.SAG()[m].prev_minfit = fit; s.SAG()[m].excl = 0.0;
        }
    }
    s.dirty = 7;
    s.store(dp->snap, 0, 0);
}

// `handoff`: 0 = one block per environment in the launch order (`block0`: the launch covers blocks block0 .. of it); 3 = time-sliced launch.
//            4 = (wrsn_step_env_full only) go on with the step in flight of this environment, whatever the caller's rows say.
// Returns (work units spent << 1) | 1 when the HEAVY = false simulator stopped in front of a grid item it has no code for.
template <int NPL, bool HEAVY>
__device__ __forceinline__ int wrsn_step_env(const WrsnDev* __restrict__ dp, int env, int reset_call, const int32_t* __restrict__ agent_id,
                                             const double* __restrict__ action, int auto_reset, int budget, long long epoch,
                                             const uint8_t* __restrict__ env_mask, const WrsnStepOutDev& out, int handoff, int deadline, double* smem, long long t_end);

// The full simulator, out of line: one grid item of the step in flight of `env` (budget 1: the item in front of which the common-path
// simulator stopped -- level BFS, routing rebuild, packet-exact second -- and whatever charger events follow it up to the next grid
// service), then the environment is stored again, suspended or finished.  Its own register allocation and its own spills; the call
// sits where the caller holds next to nothing in registers.
template <int NPL>
__device__ __noinline__ void wrsn_step_env_full(const WrsnDev* dp, int env, long long epoch, WrsnStepOutDev out) {
    extern __shared__ double smem[];
    (void)wrsn_step_env<NPL, true>(dp, env, 0, nullptr, nullptr, 0, 1, epoch, nullptr, out, 4, 0, smem, 0);
}

template <int NPL, bool HEAVY>
__global__ void __launch_bounds__(64, WRSN_WAVES_PER_SIMD(NPL))
wrsn_step_kernel(const WrsnDev* __restrict__ dp, int reset_call, const int32_t* __restrict__ agent_id, const double* __restrict__ action, int auto_reset,
                 int budget, long long epoch, int slots, const uint8_t* __restrict__ env_mask, WrsnStepOutDev out, int handoff, int deadline, int block0) {
    extern __shared__ double smem[];
    // Block b of a step launch takes environment order[b]: the environments sorted by the work their WRSN.step still needs,
    // longest first (wrsn_estimate_kernel / wrsn_sort_kernel run in front of every step launch).  The duration of a WRSN.step is
    // heavy-tailed and a launch ends with its last wave, so the long jobs have to start first; every environment appears exactly once
    // in the order, so a launch owns an environment through one block only.
    // The heavy launch has one block per environment of the batch too; block b takes entry b of the list and the blocks behind the end of the
    // list leave at once (the host does not know its length).
    int env = blockIdx.x; long long t_slice_end = 0;
    if (handoff == 3) {
        // Time-sliced launch (wrsn_set_step_deadline): the blocks of the launch share ONE deadline (`deadline` ticks after the first of them
        // started).  Block b takes environment (start + b) mod B of this launch's cyclic order; the hardware hands the blocks to the wave slots
        // in index order as slots become free, so every slot is busy from the first microsecond on.  A wave runs its WRSN.step until it
        // returns or the deadline passes (then it stops at the next item boundary, like a wave out of budget).  A block that only starts
        // when the launch is about to end leaves at once and does NOT touch its environment: the action waits in the latch
        // (wrsn_latch_kernel), the row says "in flight", and the next launch starts its cyclic order there.
        const int B = dp->B;
        if ((int)blockIdx.x >= B) return;
        // (the launch's start was stamped by wrsn_latch_kernel, a few microseconds ago: thousands of blocks racing for one stamp with an
        //  atomic -- and for one "last block taken" word -- took 70 us to get going; the word is spread over 64 addresses)
        const long long tn = (long long)wall_clock64();       // wave-uniform (s_memrealtime)
        const long long t_end = wu((int64_t)*dp->launch_t0) + deadline;
        // too late to do anything for this environment in this launch?  (margin: at most a quarter of the time slice)
        if (tn > t_end - (deadline / 4 < WRSN_PULL_MARGIN ? deadline / 4 : WRSN_PULL_MARGIN) && blockIdx.x != 0) return;
        if (threadIdx.x == 0) atomicMax(&dp->queue[8 + (blockIdx.x & 63)], (int)blockIdx.x + 1);    // the next launch starts behind the last environment taken
        env = dp->queue[1] + (int)blockIdx.x; env = env >= B ? env - B : env;
        if (dp->qskip[env]) return;
        t_slice_end = t_end;
    } else if (!reset_call) {
        // (`block0`: a step call may come as two launches over the two halves of the launch order -- see wrsn_step in wrsn_api.hip)
        const int bidx = (int)blockIdx.x + block0;
        env = dp->order[bidx];
        if (env < 0 || env >= dp->B) return;
        if (budget > 0) {
            // blocks are dispatched in index order: a block far behind the first `slots` ones starts late, and what it is
            // allowed to spend shrinks accordingly so that the launch does not wait for late long jobs
            // (`slots` packs three launch parameters: wave slots of the device, block at which the taper starts, blocks over which the
            //  budget falls to zero -- the floor of a quarter applies before that)
            const int n_slots = slots & 0xFFFF, t_start = (slots >> 16) & 0xFF, t_len = (slots >> 24) & 0xFF;
            const int k = bidx - t_start * (n_slots / 8);
#ifndef WRSN_BUDGET_FLOOR
#define WRSN_BUDGET_FLOOR 4
#endif
            if (k > 0) { const int cut = (int)((long long)budget * k / (t_len * (n_slots / 8))); budget = (budget - cut > budget / WRSN_BUDGET_FLOOR) ? budget - cut : budget / WRSN_BUDGET_FLOOR; }
        }
    }
    if (env < 0 || env >= dp->B) return;
    for (int pass = 0; pass < 256; ++pass) {
        const int r = wrsn_step_env<NPL, HEAVY>(dp, env, reset_call, agent_id, action, auto_reset, budget, epoch, env_mask, out, handoff, deadline, smem, t_slice_end);   // the one call site
        if (HEAVY || !(r & 1)) break;
        // the common-path simulator stopped in front of a rare service: the full simulator takes that one item, then this one goes on
        __syncthreads();
        wrsn_step_env_full<NPL>(dp, env, epoch, out);
        __syncthreads();
        if (wu(dp->live.dyn[env].susp) == 0) break;        // the step ended in there: its request is written
        if (budget > 0) { budget -= (r >> 1) + 64; if (budget < 16) budget = 16; }
    }
}

template <int NPL, bool HEAVY>
__device__ __forceinline__ int wrsn_step_env(const WrsnDev* __restrict__ dp, int env, int reset_call, const int32_t* __restrict__ agent_id,
                                             const double* __restrict__ action, int auto_reset, int budget, long long epoch,
                                             const uint8_t* __restrict__ env_mask, const WrsnStepOutDev& out, int handoff, int deadline, double* smem, long long t_end) {
    const int lane = threadIdx.x;
    bool do_reset = reset_call != 0;
    // a row nobody handles in this launch is not rendered and none of its outputs is touched
    if (reset_call && env_mask && env_mask[env] == 0) { if (lane == 0) { dp->render_agent[env] = -1; dp->row_state[env] = 0; } return 0; }
    int aid = -1, resume = 0;
    const double* act_src = action + (size_t)env * 3;
    if (handoff == 4) resume = 1;                          // the full simulator called for one item of the step in flight
    else if (handoff == 3) {                               // time-sliced launch: the action waits in the latch, the caller's row is not looked at
        const WrsnEnvDyn* dy = dp->live.dyn + env;
        resume = dy->susp;
        if (!resume) { if (!dy->lat_valid) return 0; aid = dy->lat_agent; act_src = dy->lat_action; }
        if (auto_reset && dy->terminal_pending) do_reset = true;
    } else if (!reset_call) {
        aid = agent_id[env];
        if (aid == -2) { if (lane == 0) { dp->render_agent[env] = -1; dp->row_state[env] = 0; } return 0; }
        resume = dp->live.dyn[env].susp;                   // a step in flight goes on; agent_id / action are not looked at
        if (auto_reset && dp->live.dyn[env].terminal_pending) do_reset = true;
    }
    if (reset_call && lane == 0) dp->live.dyn[env].lat_valid = 0;   // a reset environment holds no latched action
    Sim<NPL, HEAVY> s;
    s.bind(dp, env, lane, smem);
    if (handoff == 3) { s.t_deadline = t_end; s.t_exact = t_end - (deadline / 2 < WRSN_EXACT_MARGIN ? deadline / 2 : WRSN_EXACT_MARGIN); }   // the time-sliced launch stamped its start itself
    else if (deadline > 0 && budget > 0 && !reset_call) {    // common deadline of the launch: `deadline` ticks after its first wave started
        long long t0 = 0;
        const long long tn = (long long)wall_clock64();      // wave-uniform (s_memrealtime)
        if (lane == 0) {
            const unsigned long long old = atomicCAS((unsigned long long*)dp->launch_t0, 0ull, (unsigned long long)tn);
            t0 = old ? (long long)old : tn;
        }
        {   // lane 0's stamp for the whole wave (v_readlane: the other lanes hold nothing)
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(t0 & 0xffffffffll), 0); const int hi = __builtin_amdgcn_readlane((int)(t0 >> 32), 0);
            s.t_deadline = (((long long)hi << 32) | (long long)lo) + deadline;
        }
    }
#ifdef WRSN_PROFILE
    for (int q_ = 0; q_ < 24; ++q_) s.prof_[q_] = 0;
    const long long kt0_ = clock64();
    const long long wt0_ = wall_clock64();
#endif
    const WrsnEnvConst* ec = s.EC();
    { WRSN_P4_MARK(k0_) s.load(do_reset ? dp->snap : dp->live); WRSN_P4_MARK(k1_)
#if defined(WRSN_PROFILE) && WRSN_PROFILE == 4
      s.prof_[16] += k1_ - k0_; s.prof_[18] += k0_ - kt0_;
#endif
    }
    if (do_reset) { s.dirty = 7; s.map1_valid = 0; }       // the snapshot goes to the live arrays in full; a rendered map 1 no longer stands
    const int64_t ticks0 = s.n_ticks;
    int terminal = 0, susp = 0;
    if (do_reset) {
        const double* rs = dp->snap.ring + (size_t)env * WRSN_RING * s.NP; double* rl = dp->live.ring + (size_t)env * WRSN_RING * s.NP;
        for (int w = lane; w < WRSN_RING * s.NP; w += 64) rl[w] = rs[w];
        for (int w = lane; w < s.NP; w += 64) dp->live.logbuf[(size_t)env * s.NP + w] = dp->snap.logbuf[(size_t)env * s.NP + w];
        if (lane == 0) {
            int agent = -1;
            for (int m = s.M - 1; m >= 0; --m) if (s.agent_at_rest(m)) agent = m;
            if (out.agent_id) out.agent_id[env] = agent;
            if (out.reward) out.reward[env] = 0.0;
            if (out.terminal) out.terminal[env] = (s.alive == 1) ? 0 : 1;
            if (out.now) out.now[env] = s.now;
            if (out.status) out.status[env] = reset_call ? 0 : 3;
            dp->render_agent[env] = agent; dp->row_state[env] = 2;
        }
    } else {
        // ------------------------------------------------------ WRSN.step
        if (lane == 0 && resume) {
            const WrsnEnvDyn* dy = dp->live.dyn + env;
            s.SS()->L = dy->cond_L;
            for (int j = 0; j <= s.M; ++j) { s.SCA()[j] = dy->cond_agent[j]; s.SCTR()[j] = dy->cond_trig[j]; s.SCP()[j] = dy->cond_pend[j]; s.SCT()[j] = dy->cond_time[j]; s.SCS()[j] = dy->cond_seq[j]; }
            s.SREQ()[3] = 0;
        }
        if (lane == 0 && !resume) {
            int st0 = 0;
            dp->live.dyn[env].step_t0 = s.now;
            if (aid >= 0 && aid < s.M) {
                double act[3];
                for (int k = 0; k < 3; ++k) { double v = act_src[k]; act[k] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }   // np.clip (WRSN.py:299)
                s.SAG()[aid].action[0] = act[0]; s.SAG()[aid].action[1] = act[1]; s.SAG()[aid].action[2] = act[2];
                double p0 = act[0] * (ec->frame[1] - ec->frame[0]) + ec->frame[0];  // translate (WRSN.py:95-98)
                double p1 = act[1] * (ec->frame[3] - ec->frame[2]) + ec->frame[2];
                double p2 = ec->charging_time_max * act[2];
                int ti = s.new_thread(aid, p0, p1, p2);
                if (ti < 0) s.err = -8; else s.SAG()[aid].cur_thread = ti;
                s.SAG()[aid].prev_minfit = s.last_minfit;      // WRSN.py:304 (node state is unchanged since the last return)
                s.SAG()[aid].excl = 0.0;                       // WRSN.py:305
            }
            // general_process = net_process | p_a0 | p_a1 ... over chargers alive now (WRSN.py:307-310)
            s.SS()->L = 0;
            for (int m = 0; m < s.M; ++m) if (s.SAG()[m].status != 0) { s.SCA()[s.SS()->L] = m; s.SS()->L++; }
            if (s.SS()->L == 0) st0 = 2;                           // reference: run() never returns; deliberate deviation
            else {
                for (int j = 1; j <= s.SS()->L; ++j) {             // Condition.__init__ checks processed operands at once
                    int ti = s.SAG()[s.SCA()[j - 1]].cur_thread;
                    if (ti >= 0 && s.STH()[ti].pc == PC_FINISHED) s.cond_trigger(j);
                }
            }
            s.SREQ()[3] = st0;
        }
        __syncthreads();
        const int st0 = s.SREQ()[3];
        double fit = 0.0;
        if (st0 == 2) terminal = 1;
        else {
            susp = s.run(false, 0.0, budget) ? 1 : 0;        // env.run(until=general_process)
            if (susp) { }
            else if (s.alive == 0) terminal = 1;             // WRSN.py:312-320
            else if (!s.fit_dirty) fit = s.last_minfit;      // no grid item ran since the last evaluation (a return at the instant of the
                                                             // call: the bookkeeping steps at t = warm_up_time): node state, hence fitness, unchanged
            else { WRSN_P4_MARK(k2_) fit = s.min_fitness(); s.last_minfit = fit; s.fit_dirty = 0; WRSN_P4_MARK(k3_)
#if defined(WRSN_PROFILE) && WRSN_PROFILE == 4
                s.prof_[15] += k3_ - k2_;
#endif
            }
        }
        if (lane == 0 && susp) {                             // no request yet: status 4, the next launch goes on
            WrsnEnvDyn* dy = dp->live.dyn + env;
            dy->cond_L = s.SS()->L;
            for (int j = 0; j <= s.M; ++j) { dy->cond_agent[j] = s.SCA()[j]; dy->cond_trig[j] = s.SCTR()[j]; dy->cond_pend[j] = s.SCP()[j]; dy->cond_time[j] = s.SCT()[j]; dy->cond_seq[j] = s.SCS()[j]; }
            if (out.agent_id) out.agent_id[env] = -1;
            if (out.reward) out.reward[env] = 0.0;
            if (out.terminal) out.terminal[env] = 0;
            if (out.now) out.now[env] = s.now;
            if (out.status) out.status[env] = (s.err != 0) ? -4 : 4;
            dp->render_agent[env] = -1; dp->row_state[env] = 3;
        }
        if (lane == 0 && !susp) {
            int agent = -1, status = st0; double reward = 0.0;
            if (!terminal) {
                for (int m = s.M - 1; m >= 0; --m) if (s.agent_at_rest(m)) agent = m;   // lowest id (WRSN.py:321-322)
                if (agent >= 0) {                            // get_reward (WRSN.py:222-227)
                    double term_all = fit - s.SAG()[agent].prev_minfit;
                    double term_excl = s.SAG()[agent].excl / ec->avg_nodes_agent;
                    reward = (term_all * 0.8 + 0.2 * term_excl) / (ec->charging_time_max + ec->moving_time_max);
                } else status = 1;                           // reference falls off the end and returns None
            }
            if (s.err != 0) status = -4;
            {   // rollout table (wrsn_rollout_table): accumulated where the request is produced, no extra launch
                WrsnEnvDyn* dy = dp->live.dyn + env;
                if (agent >= 0) dy->roll[agent] += reward;
                if (terminal) { dy->roll[WRSN_MAX_MC] += 1.0; dy->roll[WRSN_MAX_MC + 1] += s.now; }
                dy->roll[WRSN_MAX_MC + 2] += 1.0;
                if (s.now == dy->step_t0) dy->tot_zero_steps += 1;
            }
            if (out.agent_id) out.agent_id[env] = agent;
            if (out.reward) out.reward[env] = reward;
            if (out.terminal) out.terminal[env] = (uint8_t)terminal;
            if (out.now) out.now[env] = s.now;
            if (out.status) out.status[env] = status;
            dp->render_agent[env] = agent; dp->row_state[env] = terminal ? 4 : 1;
        }
    }
    { WRSN_P4_MARK(k4_) s.store(dp->live, terminal, (do_reset || susp) ? 0 : 1, susp); WRSN_P4_MARK(k5_)
#if defined(WRSN_PROFILE) && WRSN_PROFILE == 4
      s.prof_[17] += k5_ - k4_;
#endif
    }
    if (lane == 0 && !do_reset) dp->live.dyn[env].tot_ticks += s.n_ticks - ticks0;
    if (handoff == 3 && lane == 0 && (!resume || do_reset)) dp->live.dyn[env].lat_valid = 0;   // the latched action has been taken up
#ifdef WRSN_PROFILE
    if (lane == 0) { for (int q_ = 0; q_ < 24; ++q_) dp->counters[(size_t)env * 24 + q_] += s.prof_[q_]; dp->counters[(size_t)dp->B * 24 + env] += clock64() - kt0_; }
#if WRSN_PROFILE >= 3
    if (lane == 0) { dp->counters[(size_t)env * 24 + 22] = wt0_; dp->counters[(size_t)env * 24 + 23] = wall_clock64(); }
#endif
#endif
    return (!HEAVY && susp && s.need_heavy == 1) ? ((s.work << 1) | 1) : 0;
}

// ------------------------------------------------------------------ work-queue launches: latch the actions, preset the rows
// wrsn_latch_kernel runs in front of a work-queue step launch (wrsn_set_step_deadline), one thread per environment:
//   * a row the caller marks -2 is nobody's in this call (qskip);
//   * a fresh action (the environment has no step in flight and nothing latched) goes into the environment's latch: the wave that takes
//     the environment up -- in this launch or a later one -- starts the WRSN.step from there;
//   * every row is preset to "in flight" (agent -1, status 4, not rendered); the waves overwrite the rows of the environments they finish;
//   * thread 0 moves the start of the cyclic order behind the environments the previous launch handed out and rewinds the queue.
__global__ void __launch_bounds__(256) wrsn_latch_kernel(WrsnDev d, const int32_t* __restrict__ agent_id, const double* __restrict__ action, WrsnStepOutDev out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const long long tn = (long long)wall_clock64();
    if (e == 0) {
        int handed = 0;
        for (int k = 0; k < 64; ++k) { handed = d.queue[8 + k] > handed ? d.queue[8 + k] : handed; d.queue[8 + k] = 0; }
        handed = handed < d.B ? handed : d.B;
        int st = d.queue[1] + handed; st = st >= d.B ? st - d.B : st;
        d.queue[1] = st; *d.launch_t0 = tn;                   // the time slice of the step launch behind this kernel starts now
    }
    if (e >= d.B) return;
    const int aid = agent_id[e];
    if (aid == -2) { d.qskip[e] = 1; d.render_agent[e] = -1; d.row_state[e] = 0; return; }
    d.qskip[e] = 0;
    WrsnEnvDyn* dy = d.live.dyn + e;
    if (!dy->susp && !dy->lat_valid) {
        dy->lat_agent = aid; dy->lat_action[0] = action[(size_t)e * 3]; dy->lat_action[1] = action[(size_t)e * 3 + 1]; dy->lat_action[2] = action[(size_t)e * 3 + 2];
        dy->lat_valid = 1;
    }
    if (out.agent_id) out.agent_id[e] = -1;
    if (out.reward) out.reward[e] = 0.0;
    if (out.terminal) out.terminal[e] = 0;
    if (out.now) out.now[e] = dy->now;
    if (out.status) out.status[e] = 4;
    d.render_agent[e] = -1; d.row_state[e] = 3;
}

// ------------------------------------------------------------------ launch order of a step call: longest job first
// wrsn_estimate_kernel: one thread per environment.  A WRSN.step runs until the first alive charger finishes its action
// (WRSN.py:307-311), and when that happens is known in advance: a charger's action takes dist / velocity + charge time whatever its
// energy (MobileCharger.py:81-97: a charger that runs dry waits the remaining time out), recorded as WrsnAgent.t_done when the action
// started; for the charger that is handed an action in THIS call it follows from the action itself.  Work of the launch ~ simulated
// seconds to go, plus a surcharge when a node may run dry on the way (packet-exact second + re-routing).
// wrsn_sort_kernel: stable counting sort of the (work, environment) keys in LDS, one workgroup; ties by environment index: the order
// is a pure function of the environment states, not of timing.
#define WRSN_EST_THREADS 64      // one wave per workgroup: the ~45 scattered cache lines an environment costs are spread over as many CUs as possible
__global__ void __launch_bounds__(WRSN_EST_THREADS) wrsn_estimate_kernel(WrsnDev d, const int32_t* __restrict__ agent_id, const double* __restrict__ action,
                                                            int auto_reset, int BP2) {
    const int e = blockIdx.x * WRSN_EST_THREADS + threadIdx.x;
    if (e >= BP2) return;
    if (e >= d.B) { d.order_key[e] = 0xFFFFFFFFu; return; }  // padding sorts to the end
    const WrsnEnvDyn* dy = d.live.dyn + e; const WrsnEnvConst* ec = d.ec + e;
    // everything the estimate may need is loaded up front, unconditionally (independent loads: one memory round trip instead of a
    // chain behind the tests below -- 64 waves cannot hide a chain)
    const int aid = agent_id[e];
    const int term_p = dy->terminal_pending, susp = dy->susp, frozen = dy->frozen, safe_t = dy->safe_ticks;
    const double now = dy->now;
    const double ax0 = action[(size_t)e * 3], ay0 = action[(size_t)e * 3 + 1], az0 = action[(size_t)e * 3 + 2];
    const double f0 = ec->frame[0], f1 = ec->frame[1], f2 = ec->frame[2], f3 = ec->frame[3], vel = ec->velocity, ctm = ec->charging_time_max;
    int st[WRSN_MAX_MC]; double td[WRSN_MAX_MC], lx[WRSN_MAX_MC], ly[WRSN_MAX_MC];
#pragma unroll
    for (int m = 0; m < WRSN_MAX_MC; ++m) {
        const WrsnAgent* a = dy->ag + (m < d.M ? m : 0);
        st[m] = a->status; td[m] = a->t_done; lx[m] = a->loc[0]; ly[m] = a->loc[1];
    }
    double w = 0.0;
    if (aid != -2 && !(auto_reset && term_p)) {
        double t_first = 1.0e30;
#pragma unroll
        for (int m = 0; m < WRSN_MAX_MC; ++m) {
            if (m >= d.M || st[m] == 0) continue;
            double t = td[m];
            if (!susp && m == aid) {                         // the action of this call: translate (WRSN.py:95-98) + move + charge
                double ax = ax0, ay = ay0, az = az0;
                ax = ax < 0.0 ? 0.0 : (ax > 1.0 ? 1.0 : ax); ay = ay < 0.0 ? 0.0 : (ay > 1.0 ? 1.0 : ay); az = az < 0.0 ? 0.0 : (az > 1.0 ? 1.0 : az);
                const double px = ax * (f1 - f0) + f0, py = ay * (f3 - f2) + f2;
                t = now + dist2(px, py, lx[m], ly[m]) / vel + ctm * az;
            }
            t_first = t < t_first ? t : t_first;
        }
        w = t_first - now;
        w = w > 0.0 ? (w < 1.0e4 ? w : 1.0e4) : 0.0;
        if (frozen) w = 0.0;                                 // network declared dead: the rest of the step is a jump
        else if ((double)safe_t < w) w += 250.0;             // a node may run dry before the step ends
        w += 8.0;                                            // load / events / fitness / store of a step that has anything to do
    }
    unsigned q = (unsigned)(w * 4.0); q = q > 0xFFFFu ? 0xFFFFu : q;
    d.order_key[e] = ((0xFFFFu - q) << 13) | (unsigned)e;    // ascending sort = longest first, ties by environment index
}

#ifndef WRSN_SORT_THREADS
#define WRSN_SORT_THREADS 1024                               // the CPU emulator of tests/emu runs at most 256 fibers per block and says so
#endif
// Launch order = a stable counting sort of the (work, environment) keys in one workgroup.  The 16-bit work figure is taken in 2 048
// buckets (8 simulated seconds each: the order is a scheduling hint, not a result), padding keys in one more at the end:
//   1. histogram with LDS atomics; 2. exclusive scan of the bucket counts (wave scans + wave totals); 3. scatter: a wave owns 64 K
//   consecutive environments and the waves take their turn one after the other, so equal buckets keep the order of the
//   environment indices (ties by index, as before) and the result is a pure function of the keys.
// About a third of the time of the bitonic network it replaces (78 compare-exchange stages at 4 096 keys).
#define WRSN_ORDER_BUCKETS 2048
static inline int wrsn_sort_lds_bytes() { return (WRSN_ORDER_BUCKETS + 1 + WRSN_SORT_THREADS / 64 + 1) * 4; }
template <int K>
__global__ void __launch_bounds__(WRSN_SORT_THREADS) wrsn_sort_kernel(WrsnDev d, int BP2) {
    extern __shared__ double smem[];
    constexpr int NBK = WRSN_ORDER_BUCKETS, T = WRSN_SORT_THREADS, NW = T / 64, PER = (NBK + 1 + T - 1) / T;
    int* hist = (int*)smem; int* wsum = hist + NBK + 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i <= NBK; i += T) hist[i] = 0;
    if (tid == 0) *d.launch_t0 = 0;
    __syncthreads();
    unsigned key[K]; int bk[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int e = wave * 64 * K + j * 64 + lane;         // a wave owns 64 K consecutive environments, a lane every 64th of them
        key[j] = (e < BP2) ? d.order_key[e] : 0xFFFFFFFFu;
        bk[j] = (key[j] == 0xFFFFFFFFu) ? NBK : (int)(key[j] >> 18);     // key >> 13 = 0xFFFF - work; 32 of those per bucket
        if (e < BP2) atomicAdd(&hist[bk[j]], 1);
    }
    __syncthreads();
    {   // exclusive scan: a thread owns PER consecutive buckets
        int c[PER], own = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) { const int i = tid * PER + q; c[q] = (i <= NBK) ? hist[i] : 0; own += c[q]; }
        const int incl = wv_scan_incl(own, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = incl - own;
        for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
        for (int q = 0; q < PER; ++q) { const int i = tid * PER + q; if (i <= NBK) hist[i] = base; base += c[q]; }
    }
    __syncthreads();
    for (int w = 0; w < NW; ++w) {                           // one wave at a time: environments of a bucket stay in index order
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int e = wave * 64 * K + j * 64 + lane;
                if (e < BP2) { const int pos = atomicAdd(&hist[bk[j]], 1); d.order[pos] = (int)(key[j] & 0x1FFFu); }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ topology kernel (one wave per environment)
// Network.__init__ frame/density (Network.py:16-27), Node.probe_neighbors / probe_targets (Node.py:80-90),
// BaseStation.probe_neighbors (BaseStation.py:20-23), WRSN.reset constants (WRSN.py:50-52).
__global__ void __launch_bounds__(64) wrsn_topology_kernel(WrsnDev d, int env0) {
    const int env = env0 + blockIdx.x, lane = threadIdx.x;
    if (env >= d.B) return;
    WrsnEnvConst* ec = d.ec + env;
    const int N = ec->n_node, T = ec->n_target, NP = d.NP;
    size_t nb = (size_t)env * NP;
    const double *nx = d.node_x + nb, *ny = d.node_y + nb;
    const double *tx = d.target_x + (size_t)env * d.TP, *ty = d.target_y + (size_t)env * d.TP;
    double* dbs = d.dist_bs + nb;
    int32_t *nb_off = d.nb_off + (size_t)env * (NP + 1), *nb_idx = d.nb_idx + (size_t)env * d.ECAP;
    double* nb_dist = d.nb_dist + (size_t)env * d.ECAP;
    int32_t *tc_off = d.tc_off + (size_t)env * (d.TP + 1), *tc_idx = d.tc_idx + (size_t)env * d.CCAP;
    int32_t *ncov = d.ncov + nb, *nflags = d.nflags + nb;
    uint32_t* nbp = d.nbp + nb * 4; double* nbp_es = d.nbp_es + nb * 8; double* es_bs = d.es_bs + nb;
    uint32_t* tcp = d.tcp + (size_t)env * d.TP * 4;
    uint32_t* adjm = d.adjm + nb * 8;
    // energy of sending one packet over distance dd (Node.py:114-115), the formula of Sim::e_send
    const double d0_ = sqrt(ec->efs / ec->emp);
    auto e_send_t = [&](double dd) { const double dq = dd * dd; return ((dd <= d0_) ? (ec->et + ec->efs * dq) : (ec->et + ec->emp * (dq * dq))) * ec->package_size; };
    const double bx = ec->bs[0], by = ec->bs[1], com = ec->com_range, sen = ec->sen_range;
    int error = 0;
    // frame over nodes and the base station
    double x0 = bx, x1 = bx, y0 = by, y1 = by;
    for (int i = lane; i < NP; i += 64) {
        if (i < N) {
            double x = nx[i], y = ny[i];
            x0 = fmin(x0, x); x1 = fmax(x1, x); y0 = fmin(y0, y); y1 = fmax(y1, y);
            double db = dist2(bx, by, x, y);
            dbs[i] = db; nflags[i] = (db <= com) ? 1 : 0; es_bs[i] = e_send_t(db);
        } else { dbs[i] = 0.0; nflags[i] = 0; es_bs[i] = 0.0; }
    }
    x0 = wv_min(x0); x1 = wv_max(x1); y0 = wv_min(y0); y1 = wv_max(y1);
    {   // Morton order of the nodes on a 16 x 16 grid of the frame (ties by id): the observation kernel walks the nodes in
        // this order so that the eight nodes of a chunk are neighbours in the plane
        auto morton = [&](int i) {
            int qx = (int)((nx[i] - x0) / (x1 - x0) * 16.0), qy = (int)((ny[i] - y0) / (y1 - y0) * 16.0);
            qx = qx < 0 ? 0 : (qx > 15 ? 15 : qx); qy = qy < 0 ? 0 : (qy > 15 ? 15 : qy);
            int k = 0;
            for (int b = 0; b < 4; ++b) k |= (((qx >> b) & 1) << (2 * b + 1)) | (((qy >> b) & 1) << (2 * b));
            return k * 2048 + i;
        };
        int32_t* xo = d.xorder + nb;
        for (int i = lane; i < NP; i += 64) {
            if (i < N) { const int key = morton(i); int pos = 0; for (int k = 0; k < N; ++k) pos += (morton(k) < key) ? 1 : 0; xo[pos] = i; }
            else xo[i] = i;
        }
    }
    // neighbour lists sorted by (distance, id): Node.find_receiver (Node.py:92-100) keeps the first strictly nearer
    // candidate in id order, which is the minimum of (distance, id) over the candidates
    int base = 0;
    for (int i0 = 0; i0 < NP; i0 += 64) {
        int i = i0 + lane; int cnt = 0;
        if (i < N) for (int k = 0; k < N; ++k) if (k != i && dist2(nx[k], ny[k], nx[i], ny[i]) <= com) cnt++;
        int incl = wv_scan_incl(cnt, lane);
        int off = base + incl - cnt;
        nb_off[i] = off;
        if (i < N && off + cnt <= d.ECAP) {
            int p = off;
            for (int k = 0; k < N; ++k) {
                double dd = dist2(nx[k], ny[k], nx[i], ny[i]);
                if (k != i && dd <= com) {
                    int q = p;                               // insertion keeps equal distances in id order
                    while (q > off && nb_dist[q - 1] > dd) { nb_idx[q] = nb_idx[q - 1]; nb_dist[q] = nb_dist[q - 1]; --q; }
                    nb_idx[q] = k; nb_dist[q] = dd; ++p;
                }
            }
        }
        if (i < NP) {
            const bool fits = i < N && off + cnt <= d.ECAP;
            for (int w = 0; w < 4; ++w) {
                uint32_t lo = (fits && 2 * w < cnt) ? (uint32_t)nb_idx[off + 2 * w] : 0xFFFFu;
                uint32_t hi = (fits && 2 * w + 1 < cnt) ? (uint32_t)nb_idx[off + 2 * w + 1] : 0xFFFFu;
                nbp[i * 4 + w] = lo | (hi << 16);
            }
            for (int w = 0; w < 8; ++w) nbp_es[i * 8 + w] = (fits && w < cnt) ? e_send_t(nb_dist[off + w]) : 0.0;
            if (cnt > 8) nflags[i] |= 2;
            unsigned long long am4[4] = {0ull, 0ull, 0ull, 0ull};                 // neighbourhood as a node-set mask (nodes 0..255)
            if (fits) for (int p = off; p < off + cnt; ++p) { const int k = nb_idx[p]; if (k < 256) am4[k >> 6] |= 1ull << (k & 63); }
            for (int w = 0; w < 4; ++w) { adjm[(size_t)i * 8 + 2 * w] = (uint32_t)am4[w]; adjm[(size_t)i * 8 + 2 * w + 1] = (uint32_t)(am4[w] >> 32); }
        }
        base += __shfl(incl, 63);
    }
    if (lane == 0) nb_off[NP] = base;
    if (base > d.ECAP) error = -1;
    // covered targets per node; target -> covering nodes, node-id order
    for (int i = lane; i < NP; i += 64) {
        int c = 0;
        if (i < N) for (int t = 0; t < T; ++t) if (dist2(nx[i], ny[i], tx[t], ty[t]) <= sen) c++;
        ncov[i] = c; nflags[i] |= c << 8;
    }
    int tbase = 0;
    for (int t0 = 0; t0 < d.TP; t0 += 64) {
        int t = t0 + lane; int cnt = 0;
        if (t < T) for (int k = 0; k < N; ++k) if (dist2(nx[k], ny[k], tx[t], ty[t]) <= sen) cnt++;
        int incl = wv_scan_incl(cnt, lane);
        int off = tbase + incl - cnt;
        tc_off[t] = off;
        if (t < T && off + cnt <= d.CCAP) { int p = off; for (int k = 0; k < N; ++k) if (dist2(nx[k], ny[k], tx[t], ty[t]) <= sen) tc_idx[p++] = k; }
        if (t < d.TP) {
            const bool fits = t < T && off + cnt <= d.CCAP;
            for (int w = 0; w < 4; ++w) {
                uint32_t lo = (fits && 2 * w < cnt) ? (uint32_t)tc_idx[off + 2 * w] : 0xFFFFu;
                uint32_t hi = (fits && 2 * w + 1 < cnt) ? (uint32_t)tc_idx[off + 2 * w + 1] : 0xFFFFu;
                if (w == 3 && cnt > 8) hi = 0xFFFEu;
                tcp[t * 4 + w] = lo | (hi << 16);
            }
        }
        tbase += __shfl(incl, 63);
    }
    if (lane == 0) tc_off[d.TP] = tbase;
    if (tbase > d.CCAP) error = -2;
    // connected_nodes of a charger (MobileCharger.py:55-58) are the nodes inside a disc of the charging range: any two of them are at most
    // two ranges apart, so the list never holds more nodes than the fullest disc of twice the range around a node does
    int cbound = 1;
    {
        const double r2 = 2.0 * ec->charging_range;
        for (int i = lane; i < N; i += 64) { int c = 0; for (int k = 0; k < N; ++k) if (dist2(nx[k], ny[k], nx[i], ny[i]) <= r2) c++; cbound = c > cbound ? c : cbound; }
        cbound = (int)wv_max((double)cbound);
    }
    if (lane == 0) {
        ec->conn_bound = cbound;
        ec->frame[0] = x0; ec->frame[1] = x1; ec->frame[2] = y0; ec->frame[3] = y1;
        ec->density = (double)N / ((x1 - x0) * (y1 - y0));
        ec->moving_time_max = dist2(x0, y0, x1, y1) / ec->velocity;
        ec->charging_time_max = (ec->capacity - ec->threshold) / (ec->alpha / (ec->beta * ec->beta));
        ec->avg_nodes_agent = ec->density * 3.141592653589793 * (ec->charging_range * ec->charging_range);
        ec->e_recv = ec->er * ec->package_size;
        ec->d0 = sqrt(ec->efs / ec->emp);
        ec->n_edges = base; ec->n_cover = tbase; ec->error = error;
    }
}

// ------------------------------------------------------------------ density map -> action (WRSN.py:229-297), one wave per environment
// The policy of runner/IPPO.py emits a G x G density map; WRSN.step turns it into [x, y, charging-time fraction]:
//   :293-296  a map that is not a probability map goes through exp / (sum + eps)
//   :234      arg-max cell (np.argmax: first maximum)
//   :236-238  box of +- charging range around the cell centre
//   :239-249  charging spot in the box that maximises sum_{alive, within range} CS/(E - thr) * alpha/(d + beta)^2.
//             The reference runs SciPy L-BFGS-B from the box centre; here a deterministic bounded search (33 x 33 grid +
//             the node positions, then a shrinking 7 x 7 pattern) -- its objective value is >= the optimiser's on every
//             case tested, the spot itself is "parity unpinned" (DESIGN.md 2).
//   :276-287  third component = map[argmax] / sum(map >= 99.9th percentile), np.percentile's linear interpolation
//             between the two order statistics (host passes how many elements lie above the lower one and the weight).
#define WRSN_DM_KMAX 24
#define WRSN_DM_NODES 128
static inline int wrsn_density_lds_bytes() { return (64 * WRSN_DM_KMAX + 3 * WRSN_DM_NODES) * 8; }
__global__ void __launch_bounds__(64) wrsn_density_kernel(WrsnDev d, const int32_t* __restrict__ agent_id, const double* __restrict__ dmap,
                                                           double* __restrict__ act, int n_top, double gamma) {
    extern __shared__ double smem[];
    const int env = blockIdx.x, lane = threadIdx.x;
    if (env >= d.B) return;
    const int aid = agent_id[env];
    if (aid < 0 || aid >= d.M) return;
    const WrsnEnvConst* ec = d.ec + env;
    const int G = d.G, n = G * G, N = ec->n_node;
    const double* a = dmap + (size_t)env * n;
    double* my = smem + lane * WRSN_DM_KMAX;                // descending local top-n_top of this lane
    double* lx = smem + 64 * WRSN_DM_KMAX; double* ly = lx + WRSN_DM_NODES; double* lw = ly + WRSN_DM_NODES;
    // ---- probability map or logits (WRSN.py:293-296)
    double mn = WRSN_INF, mx = -WRSN_INF, sm = 0.0, se = 0.0;
    for (int i = lane; i < n; i += 64) { const double v = a[i]; mn = fmin(mn, v); mx = fmax(mx, v); sm += v; se += exp(v); }
    mn = wv_min(mn); mx = wv_max(mx); sm = wv_sum(sm); se = wv_sum(se);
    const bool is_prob = mn >= 0.0 && mx <= 1.0 && fabs(sm - 1.0) <= 1e-8 + 1e-5;      // np.isclose(sum, 1)
    const double den = se + 1e-9;
    // ---- arg-max (first maximum) and the largest n_top values
    for (int k = 0; k < n_top; ++k) my[k] = -WRSN_INF;
    double bv = -WRSN_INF; int bi = n;
    for (int i = lane; i < n; i += 64) {
        const double v = is_prob ? a[i] : exp(a[i]) / den;
        if (v > bv) { bv = v; bi = i; }
        if (v > my[n_top - 1]) { int k = n_top - 1; while (k > 0 && my[k - 1] < v) { my[k] = my[k - 1]; --k; } my[k] = v; }
    }
    for (int m = 1; m < 64; m <<= 1) {
        const double ov = __shfl_xor(bv, m); const int oi = __shfl_xor(bi, m);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    double t_hi = 0.0, t_lo = 0.0; int head = 0;
    for (int r = 0; r < n_top; ++r) {                        // merge the lane lists: r-th largest of the whole map
        const double cand = (head < n_top) ? my[head] : -WRSN_INF;
        const double m = wv_max(cand);
        const unsigned long long hit = __ballot(cand == m);
        if ((hit & (~hit + 1ull)) == (1ull << lane)) ++head; // the lowest lane holding it pops
        if (r == n_top - 2) t_hi = m;
        if (r == n_top - 1) t_lo = m;
    }
    if (n_top < 2) t_hi = t_lo;
    const double diff = t_hi - t_lo;                          // np.percentile (linear): _lerp(a, b, t)
    double thr = t_lo + diff * gamma;
    if (gamma >= 0.5) thr = t_hi - diff * (1.0 - gamma);
    double ks = 0.0;
    for (int i = lane; i < n; i += 64) { const double v = is_prob ? a[i] : exp(a[i]) / den; if (v >= thr) ks += v; }
    ks = wv_sum(ks);
    // ---- box around the arg-max cell and the nodes that can be in range of a point of it
    const double x0 = ec->frame[0], y0 = ec->frame[2], W = ec->frame[1] - x0, H = ec->frame[3] - y0, unit = 1.0 / G, rng = ec->charging_range;
    const int ci = bi / G, cj = bi - ci * G;
    const double lbx = ((ci + 0.5) * unit - rng / W) * W + x0, ubx = ((ci + 0.5) * unit + rng / W) * W + x0;
    const double lby = ((cj + 0.5) * unit - rng / H) * H + y0, uby = ((cj + 0.5) * unit + rng / H) * H + y0;
    const double cx = (lbx + ubx) / 2, cy = (lby + uby) / 2, hx = (ubx - lbx) / 2, hy = (uby - lby) / 2;
    const size_t nb = (size_t)env * d.NP;
    int cnt = 0;
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        bool in = false; double px = 0.0, py = 0.0, w = 0.0;
        if (i < N && (d.live.ls[nb + i] & 1)) {
            px = d.node_x[nb + i]; py = d.node_y[nb + i];
            in = fabs(px - cx) <= hx + rng && fabs(py - cy) <= hy + rng;
            w = d.live.CS[nb + i] / (d.live.E[nb + i] - ec->threshold);
        }
        const unsigned long long mk = __ballot(in);
        if (in) { const int pos = cnt + __popcll(mk & ((1ull << lane) - 1ull)); if (pos < WRSN_DM_NODES) { lx[pos] = px; ly[pos] = py; lw[pos] = w; } }
        cnt += __popcll(mk);
    }
    if (cnt > WRSN_DM_NODES) cnt = WRSN_DM_NODES;
    __syncthreads();
    const double alpha = ec->alpha, beta = ec->beta;
    auto objective = [&](double px, double py) {
        double r = 0.0;
        for (int k = 0; k < cnt; ++k) { const double dd = dist2(px, py, lx[k], ly[k]); if (dd <= rng) r += lw[k] * alpha / ((dd + beta) * (dd + beta)); }
        return r;
    };
    auto clampd = [](double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); };
    // best of: 33 x 33 grid over the box (its centre, the optimiser's start, is point (16, 16)) and the nodes themselves
    double best = -1.0, bx = cx, by = cy; int bidx = 1 << 30;
    const int ncand = 33 * 33 + cnt;
    for (int c = lane; c < ncand; c += 64) {
        double px, py;
        if (c < 33 * 33) { const int u = c / 33, v = c - u * 33; px = lbx + (ubx - lbx) * (u / 32.0); py = lby + (uby - lby) * (v / 32.0); }
        else { px = clampd(lx[c - 33 * 33], lbx, ubx); py = clampd(ly[c - 33 * 33], lby, uby); }
        const double f = objective(px, py);
        if (f > best) { best = f; bx = px; by = py; bidx = c; }
    }
    for (int m = 1; m < 64; m <<= 1) {
        const double of = __shfl_xor(best, m), ox = __shfl_xor(bx, m), oy = __shfl_xor(by, m); const int oi = __shfl_xor(bidx, m);
        if (of > best || (of == best && oi < bidx)) { best = of; bx = ox; by = oy; bidx = oi; }
    }
    // shrinking 7 x 7 pattern around the incumbent
    double sx = (ubx - lbx) / 32.0, sy = (uby - lby) / 32.0;
    for (int it = 0; it < 40 && (sx > 1e-9 * (ubx - lbx) || sy > 1e-9 * (uby - lby)); ++it) {
        double f = -1.0, px = bx, py = by; int pi = 1 << 30;
        if (lane < 49) {
            const int u = lane / 7, v = lane - u * 7;
            px = clampd(bx + (u - 3) * (sx / 3.0), lbx, ubx); py = clampd(by + (v - 3) * (sy / 3.0), lby, uby);
            f = objective(px, py); pi = (lane == 24) ? -1 : lane;   // the incumbent wins ties
        }
        for (int m = 1; m < 64; m <<= 1) {
            const double of = __shfl_xor(f, m), ox = __shfl_xor(px, m), oy = __shfl_xor(py, m); const int oi = __shfl_xor(pi, m);
            if (of > f || (of == f && oi < pi)) { f = of; px = ox; py = oy; pi = oi; }
        }
        if (pi == -1) { sx /= 3.0; sy /= 3.0; } else { bx = px; by = py; best = f; }
    }
    if (lane == 0) {
        act[(size_t)env * 3 + 0] = (bx - x0) / W; act[(size_t)env * 3 + 1] = (by - y0) / H;   // down_mapping (WRSN.py:86-88)
        act[(size_t)env * 3 + 2] = bv / ks;
    }
}

// rollout table [B][M + 3] (returns per charger, finished episodes, sum of lifetimes, completed steps) from the per-environment accumulators
__global__ void __launch_bounds__(256) wrsn_rollout_kernel(WrsnDev d, double* __restrict__ dst, int zero_after) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= d.B) return;
    WrsnEnvDyn* dy = d.live.dyn + e;
    const int M = d.M;
    for (int k = 0; k < M + 3; ++k) {
        const int q = (k < M) ? k : WRSN_MAX_MC + (k - M);
        dst[(size_t)e * (M + 3) + k] = dy->roll[q];
        if (zero_after) dy->roll[q] = 0.0;
    }
}

struct WrsnTrue { static constexpr bool value = true; };
struct WrsnFalse { static constexpr bool value = false; };
#define WRSN_OBS_MAXROWS 16
// ------------------------------------------------------------------ observation kernel: WRSN.get_state (WRSN.py:130-186)
// map_1[i][j] = sum over alive nodes of w_n g(x_i - x_n; hX) g(y_j - y_n; hY) is a rank-N sum of separable Gaussians,
// i.e. the GEMM (w .* Gx)^T Gy with M = N = G (100) and K = #nodes: it runs on the matrix cores with the exact-f32
// v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain over the nodes).  One 256-thread workgroup per environment: wave w
// owns the 32-row band w of the (padded 128 x 128) map = 4 accumulator tiles; node chunks are expanded in registers into
// A[k][i] = w_k g(x_i - x_k), B[k][j] = g(y_j - y_k).  Maps 2-4 are at most M rank-1 terms and stay on the VALU.
#define WRSN_OBS_CH 8
#define WRSN_OBS_LD 128
#define WRSN_OBS_AG_FLOATS ((WRSN_MAX_MC * (int)sizeof(WrsnAgent) + 15) / 16 * 4)   // LDS floats of the staged chargers
#ifndef WRSN_V16F_DEFINED
typedef float wrsn_v16f __attribute__((ext_vector_type(16)));
#endif
#ifndef WRSN_WAVE_FIRST_DEFINED
// value of the first active lane for the whole wave (the CPU emulator of tests/emu supplies its own rendezvous)
WDEV int wrsn_wave_first(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif
// `reuse` (wrsn_set_obs_reuse): map 1 depends on node state only.  When no grid item ran since it was last rendered into this very row
// (a WRSN.step that returns at the instant it was called -- 40 % of the steps of short episodes -- or the same charger asked twice),
// the row still holds it: only maps 2..4, which depend on the asking charger, are written.
// `row_map` (optional): block b renders environment row_map[map0 + b] -- a step call renders the half of the batch whose steps are short
// (by the launch order) while the long half is still being stepped, and the rest afterwards.
__global__ void __launch_bounds__(256, 4) wrsn_obs_kernel(WrsnDev d, const int32_t* __restrict__ agent_id, float* __restrict__ obs, int reuse,
                                                           const int32_t* __restrict__ row_map, int map0) {
    extern __shared__ double smem[];
    const int env = row_map ? row_map[map0 + (int)blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;
    if (env < 0 || env >= d.B) return;
    const int aid = agent_id[env];
    if (aid < 0 || aid >= d.M) return;
    const WrsnEnvConst* ec = d.ec + env;
    const int N = ec->n_node, NP = d.NP, G = d.G, M = d.M;
    const size_t nb = (size_t)env * NP;
    const WrsnEnvDyn* dy = d.live.dyn + env;
    // node coordinates (normalised to the frame) as float32 OFFSETS from the centre of each 32-row band (x) and of each 32-column
    // tile (y): the differences the Gaussians need are then float32 subtractions of small numbers (|.| < 0.7, error ~3e-8) instead
    // of a float64 subtraction and a conversion per element
    const int NS = NP + 2 * WRSN_OBS_CH;
    float* pxb = (float*)smem;                             // [3][NS]  x - (32 w + 16) / G
    float* pyt = pxb + 3 * NS;                             // [4][NS]  y - (32 t + 16) / G
    float* wf = pyt + 4 * NS;                              // [NS] weight as float32 (0: dead / padding)
    float* bbox = wf + NS;                                 // [NP / CH + 2][4] x / y range of the weighted nodes of a chunk
    float* A = bbox + 4 * (NP / WRSN_OBS_CH + 2);          // the chargers of the environment (WRSN_OBS_AG_FLOATS), then the term rows
    const double fx0 = ec->frame[0], fy0 = ec->frame[2];
    const double W = ec->frame[1] - fx0, H = ec->frame[3] - fy0;
    const double unit = 1.0 / G;
    const double hX = ec->charging_range / W, hY = ec->charging_range / H;
    const float inv2hx = (float)(-1.0 / (2.0 * hX * hX)), inv2hy = (float)(-1.0 / (2.0 * hY * hY));
    float* out = obs + (size_t)env * 4 * G * G;
    WrsnEnvDyn* dyw = d.live.dyn + env;
    const bool keep1 = reuse && dyw->map1_valid && dyw->map1_ptr == (uint64_t)(uintptr_t)out;   // block-uniform
    // Role of a wave: 0..2 = the 32-row band of map 1 it computes on the matrix cores, 3 = the store wave (rows 96.. of map 1 on the
    // VALU, then maps 2..4).  The roles rotate with the block index so that the (matrix-core-free) store waves of the blocks resident
    // on a CU do not all sit on the same SIMD.
    const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = (hw_wave + (int)((blockIdx.x ^ (blockIdx.x >> 3) ^ (blockIdx.x >> 8)) & 3u)) & 3;   // wave-uniform
    const int l = tid & 63, half = l >> 5, l31 = l & 31;
    const int row0 = 32 * wave;                            // this wave's band of map rows
    const bool band = row0 < G;
#ifdef WRSN_OBS_PROF
    long long ot_[6]; ot_[0] = clock64();
#define WRSN_OBS_STAMP(k) ot_[k] = clock64();
#else
#define WRSN_OBS_STAMP(k)
#endif
    // the chargers of the environment go to LDS in one coalesced round trip (the term rows below read a dozen of their fields)
    WrsnAgent* sag = (WrsnAgent*)A;
    {
        const uint64_t* ga = (const uint64_t*)dy->ag; uint64_t* la = (uint64_t*)sag;
        for (int w = tid; w < M * (int)(sizeof(WrsnAgent) / 8); w += 256) la[w] = ga[w];
    }
    if (!keep1) {   // node parameters once: w_n = (CS / (alpha/beta^2)) / ((E - thr) / (cap - thr))   (WRSN.py:146)
        // one float64 division per node: w = CS span / (a_b2 (E - thr)); the frame scaling multiplies by the reciprocals of W and H
        // (the results are rounded to float32 right below)
        const double a_b2 = ec->alpha / (ec->beta * ec->beta), thr = ec->threshold, span = ec->capacity - ec->threshold;
        const double invW = 1.0 / W, invH = 1.0 / H;
        for (int n = tid; n < NP; n += 256) {
            double w = 0.0, cx = 0.0, cy = 0.0;
            const int src = (n < N) ? d.xorder[nb + n] : 0;  // position n of the Morton order (static, built with the topology)
            // all five loads are issued before the first use: one memory round trip instead of a chain behind the alive test
            const int lsw = d.live.ls[nb + src];
            const double px = d.node_x[nb + src], py = d.node_y[nb + src], en = d.live.E[nb + src], csv = d.live.CS[nb + src];
            if (n < N && (lsw & 1)) {
                cx = (px - fx0) * invW; cy = (py - fy0) * invH;
                w = (csv * span) / (a_b2 * (en - thr));
            }
#pragma unroll
            for (int w = 0; w < 3; ++w) pxb[w * NS + n] = (float)(cx - (32 * w + 16) * unit);
#pragma unroll
            for (int t = 0; t < 4; ++t) pyt[t * NS + n] = (float)(cy - (32 * t + 16) * unit);
            wf[n] = (float)w;
        }
        for (int n = NP + tid; n < NS; n += 256) {
            wf[n] = 0.f;
            for (int w = 0; w < 3; ++w) pxb[w * NS + n] = 0.f;
            for (int t = 0; t < 4; ++t) pyt[t * NS + n] = 0.f;
        }
    }
    wrsn_v16f acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    __syncthreads();                                       // pc / wf ready
    if (!keep1) for (int c = tid; c < NP / WRSN_OBS_CH + 2; c += 256) {
        float xlo = 3.0e38f, xhi = -3.0e38f, ylo = 3.0e38f, yhi = -3.0e38f;
        for (int k = 0; k < WRSN_OBS_CH; ++k) {
            const int n = c * WRSN_OBS_CH + k;
            if (wf[n] != 0.f) {                                // a zero weight (dead node, padding) reaches nothing
                const float x = pxb[n] + (float)(16 * unit), y = pyt[n] + (float)(16 * unit);
                xlo = fminf(xlo, x); xhi = fmaxf(xhi, x); ylo = fminf(ylo, y); yhi = fmaxf(yhi, y);
            }
        }
        bbox[4 * c + 0] = xlo; bbox[4 * c + 1] = xhi; bbox[4 * c + 2] = ylo; bbox[4 * c + 3] = yhi;
    }
    // maps 2..4: own charger (map 2), others charging (map 3), others moving (map 4): at most M rank-1 terms.  Their term rows
    // gx[t][G] (already scaled) and gy[t][G] go to LDS rows of their own; wave 3 (the "store wave", below) turns them into the
    // three maps while waves 0..2 are busy with the matrix cores.
    float* tx = A + WRSN_OBS_AG_FLOATS; float* ty = tx + WRSN_MAX_MC * WRSN_OBS_LD; int* tmap = (int*)(ty + WRSN_MAX_MC * WRSN_OBS_LD);
    int* row_ctr = tmap + WRSN_MAX_MC;                     // next pair of rows of maps 2..4 to be written (claimed with an LDS atomic)
    {
    const WrsnAgent* ag = sag;
    for (int o = 0; o < M; ++o) {
        int mp; double cxo, cyo, hx, hy, val;
        if (o == aid) {
            mp = 1;
            cxo = (ag[o].loc[0] - fx0) / W; cyo = (ag[o].loc[1] - fy0) / H;
            const double tmp = H < W ? H : W;
            hx = 0.5 * tmp / W; hy = 0.5 * tmp / H;
            val = ag[o].energy / ec->mc_capacity;
        } else {
            cxo = (ag[o].cur[0] - fx0) / W; cyo = (ag[o].cur[1] - fy0) / H; hx = hX; hy = hY;
            if (ag[o].type_charging) { mp = 2; val = ag[o].cur[2] / ec->charging_time_max; }          // map_3: others not "moving"
            else { mp = 3; val = (dist2(ag[o].loc[0], ag[o].loc[1], ag[o].cur[0], ag[aid].cur[1]) / ec->velocity) / ec->moving_time_max; }   // map_4, mixed index as in WRSN.py:184
        }
        const double ihx = -0.5 / (hx * hx), ihy = -0.5 / (hy * hy);      // block-uniform
        for (int idx = tid; idx < 2 * G; idx += 256) {
            const bool isx = idx < G; const int c = isx ? idx : idx - G;
            const double cen = unit / 2 + c * unit;
            const double df = cen - (isx ? cxo : cyo);
            const float g = __expf((float)(df * df * (isx ? ihx : ihy)));
            if (isx) tx[o * WRSN_OBS_LD + c] = g * (float)val; else ty[o * WRSN_OBS_LD + c] = g;
        }
        if (tid == 0) tmap[o] = mp;
    }
    if (tid == 0) *row_ctr = 0;
    }
    __syncthreads();
    WRSN_OBS_STAMP(1)
    WRSN_OBS_STAMP(2)
    // Map 1: every wave works on its own 32-row band and walks only the chunks (eight nodes each, Morton order: neighbours
    // in the plane) with a weighted node that reaches a row of the band -- a Gaussian further than 6.5 bandwidths away
    // contributes less than 7e-10 of its peak and is skipped.  Per chunk a lane expands the Gaussians of its row and of its
    // column in every tile in reach for four of the eight nodes (20 v_exp_f32, float32 offsets from the band / tile centre)
    // and the wave issues up to 16 MFMAs.  No barrier of any kind: a band's time follows its own work, and the waves of
    // the other resident blocks fill the matrix-core and VALU gaps.
    const float kx = inv2hx * 1.44269504f, ky = inv2hy * 1.44269504f;       // exp(t) = 2^(t log2 e)
    const float off = (float)((l31 - 15.5) * unit);        // this lane's row (A) / column (B) relative to the band / tile centre
    const float rx = 6.5f * (float)hX, ry = 6.5f * (float)hY;
    const float band_lo = (float)((row0 + 0.5) * unit), band_hi = (float)((row0 + 31.5) * unit);
    const int nchunk = (N + WRSN_OBS_CH - 1) / WRSN_OBS_CH;
    const bool mfma_wave = wave < 3 && band && !keep1;     // rows 0..95 on the matrix cores; wave 3 is the store wave
    if (mfma_wave) {
        const float* pxw = pxb + wave * NS;
        for (int c = 0; c < nchunk; ++c) {
            const bool need = __builtin_amdgcn_readfirstlane((int)(bbox[4 * c + 0] - rx <= band_hi && bbox[4 * c + 1] + rx >= band_lo)) != 0;
            if (!need) continue;
            // the same cut-off along y: which of the four 32-column tiles the chunk's nodes reach at all (wave-uniform)
            const float cy_lo = bbox[4 * c + 2] - ry, cy_hi = bbox[4 * c + 3] + ry;
            int tmask = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) tmask |= (int)(cy_lo <= (float)((32 * t + 31.5) * unit) && cy_hi >= (float)((32 * t + 0.5) * unit)) << t;
            tmask = __builtin_amdgcn_readfirstlane(tmask);
            const float* wn = wf + c * WRSN_OBS_CH;
            // Operand layout of v_mfma_f32_32x32x2_f32: lane l feeds element [l & 31][k = l >> 5] of both operands.  With k = node
            // (l >> 5) + 2 m of the chunk, a lane therefore consumes exactly the Gaussians of ITS row / column l & 31: they are expanded
            // straight into registers (20 v_exp_f32 per lane and chunk) -- no LDS staging, no barrier between expansion and MFMA.
            float av[WRSN_OBS_CH / 2], bv[4][WRSN_OBS_CH / 2];
#pragma unroll
            for (int m = 0; m < WRSN_OBS_CH / 2; ++m) {
                const int n = half + 2 * m;
                const float df = off - pxw[c * WRSN_OBS_CH + n];
                av[m] = __builtin_amdgcn_exp2f(df * df * kx) * wn[n];          // weight * g(x_row - x_n)
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if ((tmask >> t) & 1) {
                    const float* pyw = pyt + t * NS + c * WRSN_OBS_CH;
#pragma unroll
                    for (int m = 0; m < WRSN_OBS_CH / 2; ++m) {
                        const float df = off - pyw[half + 2 * m];
                        bv[t][m] = __builtin_amdgcn_exp2f(df * df * ky);        // g(y_col - y_n)
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < WRSN_OBS_CH / 2; ++m) bv[t][m] = 0.f;
                }
            }
            // D[m][n] = sum_k Ay[m][k] Bx[k][n] with m = map column inside tile t, n = band row: the accumulator registers of a
            // lane then run along the map's COLUMNS (row = lane & 31), which lets the store below write 16 bytes at a time
#pragma unroll
            for (int m = 0; m < WRSN_OBS_CH / 2; ++m) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if ((tmask >> t) & 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t][m], av[m], acc[t], 0, 0, 0);
            }
        }
    }
    WRSN_OBS_STAMP(3)
    // map 1 store.  C/D layout: col (n) = lane & 31 = band row, row (m) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) = column in the
    // tile: registers 4 g .. 4 g + 3 of a lane are four consecutive map columns of one map row -> one 16-byte store
    if (mfma_wave) {
        const int i = row0 + l31;
        if (i < G) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int j = 32 * t + 8 * g4 + 4 * half;
                    if (j + 3 < G) {
                        float4 v; v.x = acc[t][4 * g4]; v.y = acc[t][4 * g4 + 1]; v.z = acc[t][4 * g4 + 2]; v.w = acc[t][4 * g4 + 3];
                        *(float4*)(out + (size_t)i * G + j) = v;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (j + q < G) out[(size_t)i * G + j + q] = acc[t][4 * g4 + q];
                    }
                }
            }
        }
    }
    WRSN_OBS_STAMP(4)
    if (tid == 0 && !keep1) { dyw->map1_valid = 1; dyw->map1_ptr = (uint64_t)(uintptr_t)out; }   // (only consulted with `reuse`)
    if (wave == 3 && !keep1) {
        // ---- the store wave, part 1.  Rows 96 .. G-1 of map 1 (4 rows at G = 100: a fourth matrix-core band would be 7/8 idle) as
        // rank-1 updates on the VALU: a lane owns columns l and l + 64, four rows at a time in registers; the same Morton order,
        // chunk cut-off (6.5 bandwidths) and float32 fma chain as the matrix-core bands.
        for (int rb = 96; rb < G; rb += 4) {
            float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
            const float lo = (float)((rb + 0.5) * unit), hi = (float)((rb + 3.5) * unit);
            float offr[4];                                     // rows rb .. rb + 3 relative to the centre of band 2 (row 80)
#pragma unroll
            for (int q = 0; q < 4; ++q) offr[q] = (float)((rb + q - 80 + 0.5) * unit);
            const float* px2 = pxb + 2 * NS; const float* py0 = pyt + half * NS; const float* py1 = pyt + (2 + half) * NS;   // columns l, l + 64
            for (int c = 0; c < nchunk; ++c) {
                if (!(bbox[4 * c + 0] - rx <= hi && bbox[4 * c + 1] + rx >= lo)) continue;      // wave-uniform (LDS broadcast values)
                float wn[WRSN_OBS_CH], xs[WRSN_OBS_CH], y0s[WRSN_OBS_CH], y1s[WRSN_OBS_CH];
#pragma unroll
                for (int k = 0; k < WRSN_OBS_CH; ++k) { const int n = c * WRSN_OBS_CH + k; wn[k] = wf[n]; xs[k] = px2[n]; y0s[k] = py0[n]; y1s[k] = py1[n]; }
#pragma unroll
                for (int k = 0; k < WRSN_OBS_CH; ++k) {
                    const float d0 = off - y0s[k], d1 = off - y1s[k];
                    const float g0 = __builtin_amdgcn_exp2f(d0 * d0 * ky), g1 = __builtin_amdgcn_exp2f(d1 * d1 * ky);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float df = offr[q] - xs[k];
                        const float gx = __builtin_amdgcn_exp2f(df * df * kx) * wn[k];       // weight 0 (dead node / padding): adds +0
                        a0[q] = fmaf(g0, gx, a0[q]); a1[q] = fmaf(g1, gx, a1[q]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (rb + q < G) { if (l < G) out[(size_t)(rb + q) * G + l] = a0[q]; if (l + 64 < G) out[(size_t)(rb + q) * G + l + 64] = a1[q]; }
            }
        }
    }
    {   // ---- maps 2..4, shared by all four waves: the store wave starts at once, the matrix-core waves join when their band is
        // written.  A wave claims the next pair of map rows from an LDS counter (lanes 0..31 take row 2 p, lanes 32..63 row 2 p + 1);
        // a lane owns FOUR consecutive columns (their y factors sit in registers) and writes 16 bytes per map and row: one
        // broadcast LDS read and four FMAs per charger.
        const int nq = (G + 3) >> 2, jq = l31, j = 4 * jq;
        float gq[WRSN_MAX_MC][4]; int mps[WRSN_MAX_MC];
#pragma unroll
        for (int o = 0; o < WRSN_MAX_MC; ++o) {
            mps[o] = (o < M) ? __builtin_amdgcn_readfirstlane(tmap[o]) : 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) gq[o][c] = (o < M && jq < nq && j + c < G) ? ty[o * WRSN_OBS_LD + j + c] : 0.f;
        }
        float* o2 = out + (size_t)G * G + j; float* o3 = o2 + (size_t)G * G; float* o4 = o3 + (size_t)G * G;
        const bool vec = ((G & 3) == 0);
        const int npair = (G + 1) >> 1;
        for (;;) {
            int rp = 0;
            if (l == 0) rp = atomicAdd(row_ctr, 1);
            rp = wrsn_wave_first(rp);
            if (rp >= npair) break;
            const int i = 2 * rp + half;
            if (i < G && jq < nq) {
                float v1[4] = {0.f, 0.f, 0.f, 0.f}, v2[4] = {0.f, 0.f, 0.f, 0.f}, v3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < WRSN_MAX_MC; ++o) {
                    if (o < M) {
                        const float t = tx[o * WRSN_OBS_LD + i];
                        if (mps[o] == 1) { for (int c = 0; c < 4; ++c) v1[c] = fmaf(t, gq[o][c], v1[c]); }
                        else if (mps[o] == 2) { for (int c = 0; c < 4; ++c) v2[c] = fmaf(t, gq[o][c], v2[c]); }
                        else { for (int c = 0; c < 4; ++c) v3[c] = fmaf(t, gq[o][c], v3[c]); }
                    }
                }
                if (vec) {
                    float4 a; a.x = v1[0]; a.y = v1[1]; a.z = v1[2]; a.w = v1[3]; *(float4*)(o2 + (size_t)i * G) = a;
                    float4 b; b.x = v2[0]; b.y = v2[1]; b.z = v2[2]; b.w = v2[3]; *(float4*)(o3 + (size_t)i * G) = b;
                    float4 d4; d4.x = v3[0]; d4.y = v3[1]; d4.z = v3[2]; d4.w = v3[3]; *(float4*)(o4 + (size_t)i * G) = d4;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (j + c < G) { o2[(size_t)i * G + c] = v1[c]; o3[(size_t)i * G + c] = v2[c]; o4[(size_t)i * G + c] = v3[c]; }
                }
            }
        }
    }
#ifdef WRSN_OBS_PROF
    ot_[5] = clock64();
    if (tid == 0) for (int q = 0; q < 5; ++q) d.counters[(size_t)env * 24 + q] = ot_[q + 1] - ot_[q];
    if (tid == 0) d.counters[(size_t)env * 24 + 5] = ot_[0];
#endif
}

static inline int wrsn_obs_lds_bytes(int G, int NP) { (void)G; return (NP + 2 * WRSN_OBS_CH) * 8 * 4 + 16 * (NP / WRSN_OBS_CH + 2) + WRSN_OBS_AG_FLOATS * 4 + 2 * WRSN_MAX_MC * WRSN_OBS_LD * 4 + 64; }
