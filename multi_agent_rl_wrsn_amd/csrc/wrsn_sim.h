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

// ------------------------------------------------------------------ the per-environment simulator
template <int NPL>
struct Sim {
    // identity / geometry
    int env, lane, N, T, M, NP;
    const WrsnEnvConst* ec;
    const double *nx, *ny, *dbs, *nb_dist;
    const int32_t *nb_off, *nb_idx, *tc_off, *tc_idx, *ncov, *nflags;
    double *ring, *logbuf;
    // LDS
    double *sRR, *sU;
    int32_t *sLS, *sRcv;
    WrsnAgent* sAg;
    WrsnThread* sTh;
    double* sCT; int64_t* sCS; int32_t *sCA, *sCTr, *sCP;
    double* sConnRate; int16_t* sConn;
    int32_t* sReq;                     // mailbox: [0] request, [1] argument, [2] live charger-node connections, [3] flags
    double* sReqD;                     //          [0] time limit, [1] now, [2] seq (as int64)
    // node registers (per lane)
    double E[NPL], CS[NPL], d1[NPL], d2[NPL];
    unsigned am;                       // bit j: node j*64+lane alive
    double cap, thr;
    // wave-uniform registers: identical in every lane.  `now` / `seq` are also advanced by lane 0 while it fires
    // charger events; they are re-broadcast through the mailbox at every hand-off.
    double opmax;
    int64_t n_ticks, n_exact, n_events;
    int alive, levels_dirty, cache_dirty, irreg, ring_len, ring_head, safe_ticks, log_pending, err;
    double now; int64_t seq;
    double net_time; int64_t net_seq; double ur_time; int64_t ur_seq; double node_time; int64_t node_seq;   // pending grid items
    int net_phase, net_active, node_phase, frozen, deaths_flag;
    double last_minfit;
    // scalar registers: meaningful in lane 0 only
    int L, pend, pend_idx;

    // -------------------------------------------------------------- setup of pointers
    WDEV void bind(const WrsnDev& d, int env_, int lane_, double* smem) {
        env = env_; lane = lane_; N = d.ec[env_].n_node; T = d.ec[env_].n_target; M = d.M; NP = d.NP;
        ec = d.ec + env;
        size_t nb = (size_t)env * NP;
        nx = d.node_x + nb; ny = d.node_y + nb; dbs = d.dist_bs + nb;
        nb_off = d.nb_off + (size_t)env * (NP + 1); nb_idx = d.nb_idx + (size_t)env * d.ECAP; nb_dist = d.nb_dist + (size_t)env * d.ECAP;
        tc_off = d.tc_off + (size_t)env * (d.TP + 1); tc_idx = d.tc_idx + (size_t)env * d.CCAP;
        ncov = d.ncov + nb; nflags = d.nflags + nb;
        ring = d.live.ring + (size_t)env * WRSN_RING * NP; logbuf = d.live.logbuf + nb;
        sRR = smem; sU = sRR + NP;
        sLS = (int32_t*)(sU + 2 * NP); sRcv = sLS + NP;
        sAg = (WrsnAgent*)(sRcv + NP);
        sTh = (WrsnThread*)(sAg + M);
        sCT = (double*)(sTh + 2 * M);
        sCS = (int64_t*)(sCT + (M + 1));
        sCA = (int32_t*)(sCS + (M + 1)); sCTr = sCA + (M + 1); sCP = sCTr + (M + 1);
        uintptr_t p = (uintptr_t)(sCP + (M + 1)); p = (p + 7) & ~(uintptr_t)7;
        sConnRate = (double*)p;
        sConn = (int16_t*)(sConnRate + M * WRSN_CONN_CAP);
        p = (uintptr_t)(sConn + M * WRSN_CONN_CAP); p = (p + 7) & ~(uintptr_t)7;
        sReq = (int32_t*)p; sReqD = (double*)(sReq + 4);
        cap = ec->capacity; thr = ec->threshold;
        err = 0; pend = 0; pend_idx = 0; L = 0; deaths_flag = 0;
    }

    // -------------------------------------------------------------- state load / store
    WDEV void load(const WrsnNodeArrays& a) {
        size_t nb = (size_t)env * NP;
        am = 0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            E[j] = a.E[nb + i]; CS[j] = a.CS[nb + i]; d1[j] = a.d1[nb + i]; d2[j] = a.d2[nb + i];
            sRR[i] = a.RR[nb + i];
            int ls = a.ls[nb + i]; sLS[i] = ls; sRcv[i] = a.rcv[nb + i];
            am |= (unsigned)(ls & 1) << j;
        }
        const WrsnEnvDyn* dy = a.dyn + env;
        now = dy->now; seq = dy->seq; net_time = dy->net_time; net_seq = dy->net_seq; ur_time = dy->ur_time; ur_seq = dy->ur_seq;
        node_time = dy->node_time; node_seq = dy->node_seq; last_minfit = dy->last_minfit; opmax = dy->opmax;
        n_ticks = dy->n_ticks; n_exact = dy->n_exact; n_events = dy->n_events;
        net_phase = dy->net_phase; net_active = dy->net_active; node_phase = dy->node_phase; alive = dy->alive;
        levels_dirty = dy->levels_dirty; cache_dirty = dy->cache_dirty; irreg = dy->irreg; ring_len = dy->ring_len;
        ring_head = dy->ring_head; safe_ticks = dy->safe_ticks; frozen = dy->frozen;
        log_pending = dy->log_pending;
        const uint64_t* ga = (const uint64_t*)dy->ag; uint64_t* la = (uint64_t*)sAg;
        for (int w = lane; w < M * (int)(sizeof(WrsnAgent) / 8); w += 64) la[w] = ga[w];
        const uint64_t* gt = (const uint64_t*)dy->th; uint64_t* lt = (uint64_t*)sTh;
        for (int w = lane; w < 2 * M * (int)(sizeof(WrsnThread) / 8); w += 64) lt[w] = gt[w];
        const int16_t* gc = a.conn + (size_t)env * WRSN_MAX_MC * WRSN_CONN_CAP;
        const double* gr = a.conn_rate + (size_t)env * WRSN_MAX_MC * WRSN_CONN_CAP;
        for (int w = lane; w < M * WRSN_CONN_CAP; w += 64) { sConn[w] = gc[w]; sConnRate[w] = gr[w]; }
        for (int w = lane; w <= M; w += 64) { sCTr[w] = 0; sCP[w] = 0; sCA[w] = 0; sCT[w] = 0; sCS[w] = 0; }
        if (lane == 0) { sReq[0] = 0; sReq[1] = 0; sReq[2] = dy->n_connected; }
        __syncthreads();
    }

    WDEV void store(const WrsnNodeArrays& a, int terminal_pending, int64_t n_steps_add) {
        __syncthreads();
        size_t nb = (size_t)env * NP;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            a.E[nb + i] = E[j]; a.CS[nb + i] = CS[j]; a.d1[nb + i] = d1[j]; a.d2[nb + i] = d2[j];
            a.RR[nb + i] = sRR[i]; a.ls[nb + i] = sLS[i]; a.rcv[nb + i] = sRcv[i];
        }
        WrsnEnvDyn* dy = a.dyn + env;
        if (lane == 0) {
            dy->now = now; dy->seq = seq; dy->net_time = net_time; dy->net_seq = net_seq; dy->ur_time = ur_time; dy->ur_seq = ur_seq;
            dy->node_time = node_time; dy->node_seq = node_seq; dy->last_minfit = last_minfit; dy->opmax = opmax;
            dy->n_ticks = n_ticks; dy->n_exact = n_exact; dy->n_events = n_events; dy->n_steps += n_steps_add;
            dy->net_phase = net_phase; dy->net_active = net_active; dy->node_phase = node_phase; dy->alive = alive;
            dy->levels_dirty = levels_dirty; dy->cache_dirty = cache_dirty; dy->irreg = irreg; dy->ring_len = ring_len;
            dy->ring_head = ring_head; dy->safe_ticks = safe_ticks; dy->frozen = frozen; dy->n_connected = sReq[2];
            dy->terminal_pending = terminal_pending; dy->error = err; dy->log_pending = log_pending;
        }
        uint64_t* ga = (uint64_t*)dy->ag; const uint64_t* la = (const uint64_t*)sAg;
        for (int w = lane; w < M * (int)(sizeof(WrsnAgent) / 8); w += 64) ga[w] = la[w];
        uint64_t* gt = (uint64_t*)dy->th; const uint64_t* lt = (const uint64_t*)sTh;
        for (int w = lane; w < 2 * M * (int)(sizeof(WrsnThread) / 8); w += 64) gt[w] = lt[w];
        int16_t* gc = a.conn + (size_t)env * WRSN_MAX_MC * WRSN_CONN_CAP;
        double* gr = a.conn_rate + (size_t)env * WRSN_MAX_MC * WRSN_CONN_CAP;
        for (int w = lane; w < M * WRSN_CONN_CAP; w += 64) { gc[w] = sConn[w]; gr[w] = sConnRate[w]; }
    }

    WDEV double e_send(double d) const {                    // Node.py:114-115
        double dq = d * d;
        return ((d <= ec->d0) ? (ec->et + ec->efs * dq) : (ec->et + ec->emp * (dq * dq))) * ec->package_size;
    }

    // ============================================================== WAVE SERVICES (all 64 lanes, uniform control flow)

    // -------------------------------------------------------------- Network.setLevels + check_targets (Network.py:37-66, 84-85)
    WDEV void set_levels() {
        int oldlv[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            int ls = sLS[i]; oldlv[j] = (ls >> 1) - 1;
            int al = ls & 1;
            int lv = (al && (nflags[i] & 1)) ? 1 : -1;
            sLS[i] = ((lv + 1) << 1) | al;                   // own entry only
        }
        __syncthreads();
        for (int cur = 1; cur <= N; ++cur) {
            bool ch = false; int newls[NPL];
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                int i = j * 64 + lane;
                int ls = sLS[i]; newls[j] = ls;
                if ((ls & 1) && (ls >> 1) == 0) {           // alive, level == -1
                    bool hit = false;
                    for (int p = nb_off[i]; p < nb_off[i + 1]; ++p) {
                        int l2 = sLS[nb_idx[p]];
                        if ((l2 & 1) && ((l2 >> 1) - 1) == cur) { hit = true; break; }
                    }
                    if (hit) { newls[j] = ((cur + 2) << 1) | 1; ch = true; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NPL; ++j) sLS[j * 64 + lane] = newls[j];
            __syncthreads();
            if (!wv_any(ch)) break;
        }
        bool changed = false;
#pragma unroll
        for (int j = 0; j < NPL; ++j) { int i = j * 64 + lane; if (((sLS[i] >> 1) - 1) != oldlv[j]) changed = true; }
        if (wv_any(changed)) cache_dirty = 1;
        bool bad = false;
        for (int t = lane; t < T; t += 64) {
            bool act = false;
            for (int p = tc_off[t]; p < tc_off[t + 1]; ++p) if ((sLS[tc_idx[p]] >> 1) >= 2) act = true;   // covered by a reached node
            if (!act) bad = true;
        }
        alive = wv_any(bad) ? 0 : 1;
        levels_dirty = 0;
    }

    // -------------------------------------------------------------- routing cache: receivers + per-tick drains
    // rcv_i = Node.find_receiver (Node.py:92-100) / base station (Node.py:108-111); c1/c2 = packets relayed per tick that
    // arrive before / after the node's own half-charge (sources with lower / higher id; Node.py:57-62 runs in id order).
    WDEV void rebuild_cache() {
        int32_t* c1 = (int32_t*)sU; int32_t* c2 = c1 + NP;
        double es[NPL]; int rc[NPL];
        const double er = ec->e_recv, com = ec->com_range;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            int ls = sLS[i]; int lvl = (ls >> 1) - 1;
            int r = -1; double dd = 0.0;
            if (ls & 1) {
                if (dbs[i] > com) {
                    double bd = 0.0;
                    for (int p = nb_off[i]; p < nb_off[i + 1]; ++p) {
                        int nb = nb_idx[p]; int l2 = sLS[nb];
                        if ((l2 & 1) && ((l2 >> 1) - 1) < lvl) {
                            double dist = nb_dist[p];
                            if (r < 0 || dist < bd) { r = nb; bd = dist; }
                        }
                    }
                    dd = bd;
                } else { r = -2; dd = dbs[i]; }
            }
            es[j] = (r != -1) ? e_send(dd) : 0.0; rc[j] = r;
            sRcv[i] = r; c1[i] = 0; c2[i] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            if (((am >> j) & 1u) && rc[j] >= 0) {
                int n = ncov[i];
                if (n > 0) {
                    int a = rc[j], guard = 0;
                    while (a >= 0 && guard++ < N) { atomicAdd((i < a) ? &c1[a] : &c2[a], n); a = sRcv[a]; }
                }
            }
        }
        __syncthreads();
        double opm = er;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            if ((am >> j) & 1u) {
                d1[j] = (double)c1[i] * (er + es[j]);
                d2[j] = (double)c2[i] * (er + es[j]) + (double)ncov[i] * es[j];
                opm = fmax(opm, es[j]);
            } else { d1[j] = 0.0; d2[j] = 0.0; }
        }
        __syncthreads();
        opmax = wv_max(opm);
        cache_dirty = 0; irreg = WRSN_RING; safe_ticks = 0;
    }

    // -------------------------------------------------------------- exact in-order packet walk of one k+0.5 instant
    // (Node.py:57-62 + 102-132 literally, node-id order); only taken when a node may run out of energy this tick.
    WDEV int live_receiver(int i, double* dd) {               // Node.find_receiver with live status
        int lvl = (sLS[i] >> 1) - 1; int r = -1; double bd = 0.0;
        for (int p = nb_off[i]; p < nb_off[i + 1]; ++p) {
            int nb = nb_idx[p]; int l2 = sLS[nb];
            if ((l2 & 1) && ((l2 >> 1) - 1) < lvl) { double dist = nb_dist[p]; if (r < 0 || dist < bd) { r = nb; bd = dist; } }
        }
        *dd = bd; return r;
    }

    struct WalkRec { double E; int32_t rcv; float es; };      // one LDS read per visited node

    WDEV void exact_walk(bool any_rr) {
        WalkRec* rec = (WalkRec*)sU;                            // 16 B per node = the whole scratch
        const double er = ec->e_recv;
        double e_start[NPL];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            e_start[j] = E[j];
            double dd = 0.0; int r = sRcv[i];                    // send cost towards the cached receiver
            if (r == -2) dd = dbs[i];
            else if (r >= 0) { for (int p = nb_off[i]; p < nb_off[i + 1]; ++p) if (nb_idx[p] == r) { dd = nb_dist[p]; break; } }
            WalkRec w; w.E = E[j]; w.rcv = r; w.es = (r != -1) ? (float)e_send(dd) : 0.0f;
            rec[i] = w;
            if (any_rr) logbuf[i] = 0.0;                        // becomes the half-charge actually gained at the node's wake
        }
        __syncthreads();
        if (lane == 0) {
            int deaths = 0;
            for (int i = 0; i < N; ++i) {
                if (!(sLS[i] & 1)) continue;
                if (any_rr) {
                    double e0 = rec[i].E, e1 = e0 + sRR[i] * 0.5;
                    e1 = e1 < cap ? e1 : cap;
                    rec[i].E = e1; logbuf[i] = e1 - e0;
                }
                const int nc = ncov[i];
                for (int p = 0; p < nc; ++p) {
                    int cur = i;
                    for (int hop = 0; hop <= N; ++hop) {
                        WalkRec w = rec[cur];
                        int r; double es;
                        if (deaths == 0) { r = w.rcv; es = (double)w.es; }
                        else if (nflags[cur] & 1) { r = -2; es = e_send(dbs[cur]); }
                        else { double dd; r = live_receiver(cur, &dd); es = (r >= 0) ? e_send(dd) : 0.0; }
                        if (r == -1) { if (w.E <= thr) { sLS[cur] &= ~1; deaths++; } break; }
                        if (w.E - thr < es) { rec[cur].E = thr; sLS[cur] &= ~1; deaths++; break; }
                        double e = w.E - es;
                        rec[cur].E = e;
                        if (e <= thr) { sLS[cur] &= ~1; deaths++; }
                        if (r == -2) break;
                        double e_r = rec[r].E;
                        if (e_r - thr < er) { rec[r].E = thr; sLS[r] &= ~1; deaths++; break; }
                        rec[r].E = e_r - er;
                        cur = r;
                    }
                }
            }
        }
        __syncthreads();
        bool died = false;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            double e_end = rec[i].E;
            if ((am >> j) & 1u) {
                // every operation of a surviving node succeeded: log_energy = start + half-charge gained - end
                double gain = any_rr ? logbuf[i] : 0.0;
                logbuf[i] = e_start[j] + gain - e_end;
                if (!(sLS[i] & 1)) { am &= ~(1u << j); CS[j] = 0.0; d1[j] = 0.0; d2[j] = 0.0; died = true; }   // Node.check_status
            }
            E[j] = e_end;
        }
        if (wv_any(died)) { cache_dirty = 1; levels_dirty = 1; deaths_flag = 1; }
        irreg = WRSN_RING; log_pending = 1; safe_ticks = 0; n_exact++;
        __syncthreads();
    }

    // -------------------------------------------------------------- k+0.5: Node.operate first half for all nodes (Node.py:57-62)
    WDEV void node_half(const double (&rrh)[NPL], const bool any_rr) {
        if (cache_dirty) rebuild_cache();
        bool fast = true;
        if (safe_ticks > 0) { safe_ticks--; }
        else {
            bool trig = false; double mn = 1e30;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                if ((am >> j) & 1u) {
                    double rr = rrh[j];
                    double a = E[j] - d1[j];
                    double b = fmin(a + rr, cap) - d2[j];
                    // a node can only run dry while it pays for an operation: idle nodes never trigger
                    if ((d1[j] > 0.0 && a - thr < opmax) || (d2[j] > 0.0 && b - thr < opmax)) trig = true;
                    double ds = d1[j] + d2[j];
                    if (ds > 0.0) mn = fmin(mn, (E[j] - thr - opmax) / ds);
                }
            }
            fast = !wv_any(trig);
            if (fast) { double m = wv_min(mn); safe_ticks = (m > 4.0) ? (int)fmin(m - 3.0, 1.0e6) : 0; }
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
        } else exact_walk(any_rr);
    }

    // -------------------------------------------------------------- k+1.0: second half + consumption window (Node.py:65-77)
    WDEV void node_full(const double (&rrh)[NPL], const bool any_rr) {
        if (any_rr) {
#pragma unroll
            for (int j = 0; j < NPL; ++j) if ((am >> j) & 1u) E[j] = fmin(E[j] + rrh[j], cap);
        }
        if (irreg > 0) {
            const int len = ring_len, head = ring_head;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                if ((am >> j) & 1u) {
                    int i = j * 64 + lane;
                    double lg = log_pending ? logbuf[i] : (d1[j] + d2[j]);
                    if (len < WRSN_RING) { ring[(size_t)len * NP + i] = lg; CS[j] = (CS[j] * len + lg) / (len + 1); }
                    else { double old = ring[(size_t)head * NP + i]; CS[j] = (CS[j] * len - old + lg) / len; ring[(size_t)head * NP + i] = lg; }
                }
            }
            if (len < WRSN_RING) ring_len = len + 1; else ring_head = (head + 1) % WRSN_RING;
            irreg--;
        }
        log_pending = 0; n_ticks++;
    }

    WDEV double conn_rate_of(int m, int k, int i) const {    // alpha / (dist(node, charger) + beta)^2 (Node.py:137, WRSN.py:122)
        if (sAg[m].loc[0] == sAg[m].conn_loc[0] && sAg[m].loc[1] == sAg[m].conn_loc[1]) return sConnRate[m * WRSN_CONN_CAP + k];
        double dd = dist2(nx[i], ny[i], sAg[m].loc[0], sAg[m].loc[1]) + ec->beta;
        return ec->alpha / (dd * dd);
    }

    // -------------------------------------------------------------- WRSN.update_reward (WRSN.py:100-127)
    // priorities: p = CS / (E - thr + 1e-9) (0 for dead nodes), standardised (population std), exp, normalised.
    // mean and variance come from one fused pair of sums; the node that owns a connected entry computes its own
    // contribution and posts it to LDS, lane 0 adds them up in list order.
    WDEV void update_reward() {
        const double eps = 1e-9;
        double x[NPL]; double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) { x[j] = ((am >> j) & 1u) ? (CS[j] / (E[j] - thr + eps)) : 0.0; s1 += x[j]; s2 += x[j] * x[j]; }
        wv_sum2(s1, s2);
        const double mean = s1 / N;
        double var = s2 / N - mean * mean;
        if (!(var > 0.0)) var = 0.0;
        double sd = sqrt(var);
        if (sd == 0.0) sd = eps;
        double ex[NPL]; double es = 0.0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) { ex[j] = (j * 64 + lane < N) ? (double)expf((float)((x[j] - mean) / sd)) : 0.0; es += ex[j]; }
        double tot = wv_sum(es);
        if (tot == 0.0) tot = eps;
        const double a_b2 = ec->alpha / (ec->beta * ec->beta);
        double* post = sU;                                   // [M][CONN_CAP] contributions
        for (int m = 0; m < M; ++m) {
            if (sAg[m].status == 0 || !sAg[m].type_charging) continue;
            const int nc = sAg[m].n_conn;
            for (int k = 0; k < nc; ++k) {
                const int i = sConn[m * WRSN_CONN_CAP + k];
                if (lane == (i & 63)) {
                    const int jj = i >> 6;
                    double Ei = E[0], Ci = CS[0], xi = ex[0];
#pragma unroll
                    for (int j = 1; j < NPL; ++j) if (j == jj) { Ei = E[j]; Ci = CS[j]; xi = ex[j]; }
                    double c = 0.0;
                    if (sLS[i] & 1) {
                        const double rate = conn_rate_of(m, k, i);
                        const double e_no = fmin(Ei - Ci, thr);          // min / max as written (WRSN.py:123-124)
                        const double e_with = fmax(Ei - Ci + rate, cap);
                        c = (xi / tot) * (e_with - e_no) / a_b2;
                    }
                    post[m * WRSN_CONN_CAP + k] = c;
                }
            }
        }
        __syncthreads();
        if (lane == 0) {
            for (int m = 0; m < M; ++m) {
                if (sAg[m].status == 0 || !sAg[m].type_charging) continue;
                double incentive = 0.0; const int nc = sAg[m].n_conn;
                for (int k = 0; k < nc; ++k) incentive += post[m * WRSN_CONN_CAP + k];
                sAg[m].excl += incentive;
            }
        }
        __syncthreads();
    }

    // -------------------------------------------------------------- WRSN.get_network_fitness -> np.min (WRSN.py:188-220)
    // label-correcting widest path; the fixed point does not depend on visiting order, so relax in parallel.
    WDEV double min_fitness() {
        double* t = sU; double lt[NPL], tc[NPL];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            bool al = (am >> j) & 1u;
            lt[j] = al ? ((CS[j] == 0.0) ? WRSN_INF : (E[j] - thr) / CS[j]) : 0.0;
            tc[j] = (al && (nflags[i] & 1)) ? lt[j] : -1.0;
            t[i] = tc[j];
        }
        __syncthreads();
        for (int it = 0; it <= N; ++it) {
            bool ch = false;
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                int i = j * 64 + lane;
                if (((am >> j) & 1u) && !(nflags[i] & 1)) {
                    double best = -1.0;
                    for (int p = nb_off[i]; p < nb_off[i + 1]; ++p) { int nb = nb_idx[p]; if (sLS[nb] & 1) best = fmax(best, t[nb]); }
                    double cand = fmin(lt[j], best);
                    if (cand > tc[j]) { tc[j] = cand; ch = true; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NPL; ++j) t[j * 64 + lane] = tc[j];
            __syncthreads();
            if (!wv_any(ch)) break;
        }
        double mn = WRSN_INF;
        for (int q = lane; q < T; q += 64) {
            double v = 0.0;
            for (int p = tc_off[q]; p < tc_off[q + 1]; ++p) v = fmax(v, t[tc_idx[p]]);
            mn = fmin(mn, v);
        }
        mn = wv_min(mn);
        __syncthreads();
        return mn;
    }

    // -------------------------------------------------------------- charger energy pre-check sum (MobileCharger.py:111-115)
    WDEV double precheck(int ti) {
        const double dx = sTh[ti].phy[0], dy = sTh[ti].phy[1];
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            if ((am >> j) & 1u) {
                int i = j * 64 + lane;
                double dis = dist2(dx, dy, nx[i], ny[i]);
                if (dis <= ec->charging_range) part += ec->alpha / ((dis + ec->beta) * (dis + ec->beta));
            }
        }
        return wv_sum(part);
    }

    // -------------------------------------------------------------- connected_nodes of a charger (MobileCharger.py:55-58)
    // every node (alive or not) within charging range of the charger, id order; caches the connection rate
    WDEV void conn_build(int a) {
        const double lx = sAg[a].loc[0], ly = sAg[a].loc[1];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            double dis = (i < N) ? dist2(nx[i], ny[i], lx, ly) : 0.0;
            bool in = (i < N) && (dis <= ec->charging_range);
            unsigned long long mk = __ballot(in);
            if (in) {
                int pos = cnt + __popcll(mk & ((1ull << lane) - 1ull));
                if (pos < WRSN_CONN_CAP) {
                    sConn[a * WRSN_CONN_CAP + pos] = (int16_t)i;
                    double dd = dis + ec->beta;
                    sConnRate[a * WRSN_CONN_CAP + pos] = ec->alpha / (dd * dd);
                }
            }
            cnt += __popcll(mk);
        }
        if (cnt > WRSN_CONN_CAP) { err = -9; cnt = WRSN_CONN_CAP; }
        if (lane == 0) { sAg[a].n_conn = cnt; sAg[a].conn_loc[0] = lx; sAg[a].conn_loc[1] = ly; }
    }

    // ============================================================== GRID LOOP (all lanes, wave-uniform registers)
    // The periodic items of the reference -- Network.operate (k+0.1 setLevels/check_targets, k+1.0 alive check),
    // update_reward (k+1.0) and the Node.operate block (k+0.5, k+1.0) -- are popped here in (time, seq) order until the
    // next item would not be strictly earlier than `t_limit` (the next charger / condition event).  When no node is
    // being charged, no reward priority is consumed, the consumption window is uniform and no node can run dry, whole
    // seconds are skipped in closed form (E -= j * (d1 + d2)).
    WDEV void grid_run(double t_limit, bool one, bool ur_flag) {
        deaths_flag = 0;
        // Node.energyRR only changes when lane 0 connects / disconnects a charger, i.e. between two grid services
        const bool any_rr = sReq[2] > 0;
        double rrh[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) rrh[j] = any_rr ? sRR[j * 64 + lane] * 0.5 : 0.0;
        for (long guard = 0; guard < 4000000L; ++guard) {
            if (frozen) break;
            int k = 1; double bt = ur_time; int64_t bs = ur_seq;
            if (node_time < bt || (node_time == bt && node_seq < bs)) { k = 2; bt = node_time; bs = node_seq; }
            if (net_active && (net_time < bt || (net_time == bt && net_seq < bs))) { k = 0; bt = net_time; bs = net_seq; }
            if (!one && !(bt < t_limit)) break;
            // ---- canonical start of a second: setLevels @k+0.1 (no-op), nodes @k+0.5, reward/alive-check/nodes @k+1.0
            const double kk = floor(bt);
            if (!one && node_phase == 0 && (net_active ? (k == 0 && net_phase == 0) : (k == 2)) && !levels_dirty && ur_time == kk + 1.0) {
                if (!any_rr && !ur_flag && irreg == 0 && !cache_dirty && !log_pending && safe_ticks > 0) {
                    // nothing but the constant per-second drain happens: skip j whole seconds in closed form
                    double jf = floor(fmin(t_limit, kk + 1.0e6) - kk);
                    if (kk + jf >= t_limit) jf -= 1.0;
                    if (net_active) { double jm = floor(fmin(ec->max_time, kk + 1.0e6) - kk); if (kk + jm >= ec->max_time) jm -= 1.0; jf = fmin(jf, jm); }
                    const int j = (int)fmin(jf, (double)safe_ticks);
                    if (j >= 1) {
                        const double dj = (double)j;
#pragma unroll
                        for (int q = 0; q < NPL; ++q) E[q] -= dj * (d1[q] + d2[q]);
                        const double ke = kk + dj;
                        ur_time = ke + 1.0; node_time = ke + 1.0 * 0.5;
                        if (net_active) { net_time = ke + 1.0 / 10.0; seq += 5 * (int64_t)j; ur_seq = seq - 3; net_seq = seq - 2; node_seq = seq - 1; n_events += 5 * (int64_t)j; }
                        else { seq += 3 * (int64_t)j; ur_seq = seq - 2; node_seq = seq - 1; n_events += 3 * (int64_t)j; }
                        now = ke; n_ticks += j; safe_ticks -= j;
                        continue;
                    }
                }
                if (kk + 1.0 < t_limit && (!net_active || kk + 1.0 < ec->max_time)) {
                    // one whole second, straight-line: the five items keep their order, nothing can interleave
                    now = kk + 1.0 * 0.5;
                    node_half(rrh, any_rr);
                    if (deaths_flag) {                       // lane 0 must re-plan chargers before time moves on
                        if (net_active) { const double t1 = kk + 1.0 / 10.0; net_phase = 1; net_time = t1 + 9.0 * 1.0 / 10.0; net_seq = seq++; n_events++; }
                        node_phase = 1; node_time = now + 1.0 * 0.5; node_seq = seq++; n_events++;
                        break;
                    }
                    now = kk + 1.0;
                    if (ur_flag) update_reward();
                    node_full(rrh, any_rr);
                    ur_time = now + 1.0; node_time = now + 1.0 * 0.5;
                    if (net_active) { net_time = now + 1.0 / 10.0; seq += 5; ur_seq = seq - 3; net_seq = seq - 2; node_seq = seq - 1; n_events += 5; }
                    else { seq += 3; ur_seq = seq - 2; node_seq = seq - 1; n_events += 3; }
                    continue;
                }
            }
            // ---- generic path: one item
            now = bt; n_events++;
            if (k == 0) {
                if (net_phase == 0) {                        // Network.py:75-78
                    if (levels_dirty) set_levels();
                    if (alive == 0) frozen = 1;              // terminal at the next return; node state is no longer observable
                    net_phase = 1; net_time = now + 9.0 * 1.0 / 10.0; net_seq = seq++;
                } else {                                     // Network.py:78-80
                    if (alive == 0 || now >= ec->max_time) net_active = 0;
                    else { net_phase = 0; net_time = now + 1.0 / 10.0; net_seq = seq++; }
                }
            } else if (k == 1) {
                if (ur_flag) update_reward();
                ur_time = now + 1.0; ur_seq = seq++;
            } else {
                if (node_phase == 0) { node_half(rrh, any_rr); node_phase = 1; } else { node_full(rrh, any_rr); node_phase = 0; }
                node_time = now + 1.0 * 0.5; node_seq = seq++;
            }
            if (one || deaths_flag) break;
        }
    }

    // ============================================================== SCALAR EVENT PROCESSOR (lane 0 only)

    WDEV void th_sched(int ti, int pc, int prio, double time) { sTh[ti].pc = pc; sTh[ti].prio = prio; sTh[ti].time = time; sTh[ti].seq = seq++; }

    WDEV void mc_check_status(int a) {                       // MobileCharger.py:134-140
        if (sAg[a].energy <= ec->mc_threshold) { sAg[a].status = 0; sAg[a].energy = ec->mc_threshold; }
    }

    WDEV bool agent_single(int a, int ti) const {            // no other live operate_step process acts on this charger
        for (int i = 0; i < 2 * M; ++i) {
            if (i == ti) continue;
            int pc = sTh[i].pc;
            if (pc != PC_NONE && pc != PC_FINISHED && sTh[i].agent == a) return false;
        }
        return true;
    }

    // ---- fast-forward: unit sub-steps of move() / charge() whose only effect is on the charger itself are kept
    // virtual and applied lazily (identical arithmetic, applied in order); the process resumes at the timeout of the
    // last unit sub-step.  Only taken in generic position; ties with the node grid use the per-sub-step path.
    WDEV void ff_apply(int ti, double t, bool all) {
        const int a = sTh[ti].agent;
        int n = sTh[ti].ff_n; double ft = sTh[ti].ff_t;
        if (sTh[ti].ff == 1) {
            const double ux = sTh[ti].mvec[0] / sTh[ti].total_time * 1.0, uy = sTh[ti].mvec[1] / sTh[ti].total_time * 1.0;
            const double de = ec->pm * 1.0 * ec->velocity;
            double lx = sAg[a].loc[0], ly = sAg[a].loc[1], e = sAg[a].energy;
            while (n > 0 && (all || ft + 1.0 <= t)) { lx = lx + ux; ly = ly + uy; e -= de; ft = ft + 1.0; --n; }
            sAg[a].loc[0] = lx; sAg[a].loc[1] = ly; sAg[a].energy = e;
        } else if (sTh[ti].ff == 2) {
            const double cr = sAg[a].charging_rate;
            double e = sAg[a].energy, c2 = sAg[a].cur[2], tmp = sTh[ti].tmp;
            while (n > 0 && (all || ft + 1.0 <= t)) {
                e = e - cr * 1.0; c2 = (c2 - 1.0) > 0.0 ? (c2 - 1.0) : 0.0; tmp -= 1.0; ft = ft + 1.0; --n;
            }
            sAg[a].energy = e; sAg[a].cur[2] = c2; sTh[ti].tmp = tmp;
        }
        sTh[ti].ff_n = n; sTh[ti].ff_t = ft;
        if (n == 0) sTh[ti].ff = 0;
    }

    WDEV void ff_sync_all(double t) {
        for (int i = 0; i < 2 * M; ++i) if (sTh[i].ff != 0 && sTh[i].pc != PC_NONE && sTh[i].pc != PC_FINISHED) ff_apply(i, t, false);
    }

    // a connected node died (or a second process now shares the charger): return to the per-sub-step path at the
    // sub-step in flight
    WDEV void ff_fallback(int ti) {
        if (sTh[ti].ff == 0) return;
        ff_apply(ti, now, false);
        if (sTh[ti].ff == 0) return;                         // everything virtual was already due: the resume event stands
        const int kind = sTh[ti].ff;
        sTh[ti].ff = 0; sTh[ti].ff_n = 0;
        th_sched(ti, kind == 1 ? PC_MSTEP_TIMEOUT : PC_CSTEP_TIMEOUT, WRSN_NORMAL, sTh[ti].ff_t + 1.0);
    }

    WDEV void mc_move_loop(int ti) {                         // MobileCharger.py:85-94 from the top of `while True`
        int a = sTh[ti].agent;
        if (sTh[ti].moving_time <= 0.0) { th_sched(ti, PC_MOVE_DONE, WRSN_NORMAL, now); return; }
        if (sAg[a].status == 0) { th_sched(ti, PC_MOVE_DEADWAIT, WRSN_NORMAL, now + sTh[ti].moving_time); return; }
        double mt = dist2(sTh[ti].m_dest[0], sTh[ti].m_dest[1], sAg[a].loc[0], sAg[a].loc[1]) / ec->velocity;
        sTh[ti].moving_time = mt;
        double s = mt < 1.0 ? mt : 1.0;
        const double pmv = ec->pm * ec->velocity;
        double lim = (sAg[a].energy - ec->mc_threshold) / pmv;
        sTh[ti].span = s < lim ? s : lim;
        // fast-forward: a moving charger is observed by nobody until the run returns -- except through update_reward
        // when it is (stale) "charging" with connected nodes -- so whole-second sub-steps can stay virtual
        if (mt > 3.0 && lim > 3.0 && !(sAg[a].type_charging && sAg[a].n_conn > 0) && agent_single(a, ti)) {
            double nf = floor(fmin(mt, lim)) - 1.0;          // the remainder (> 1 s) and the energy margin go through the exact path
            if (nf > 100000.0) nf = 100000.0;
            int n = (int)nf;
            if (n >= 2) {
                double t = now;
                for (int q = 0; q < n; ++q) t = t + 1.0;     // same float accumulation as n successive timeouts
                sTh[ti].span = 1.0; sTh[ti].ff = 1; sTh[ti].ff_n = n - 1; sTh[ti].ff_t = now;
                th_sched(ti, PC_MSTEP_TIMEOUT, WRSN_NORMAL, t);
                return;
            }
        }
        th_sched(ti, PC_MSTEP_INIT, WRSN_URGENT, now);
    }

    // Node.charger_connection / charger_disconnection over connected_nodes (Node.py:134-146); sign = +1 / -1
    WDEV int mc_connect(int a, double sign) {
        const int nc = sAg[a].n_conn;
        double cr = sAg[a].charging_rate; int cnt = 0;
        for (int k = 0; k < nc; ++k) {
            int i = sConn[a * WRSN_CONN_CAP + k];
            if (!(sLS[i] & 1)) continue;
            double r = conn_rate_of(a, k, i);
            sRR[i] += sign * r; cr += sign * r; cnt++;
        }
        sAg[a].charging_rate = cr;
        if (sign > 0) { sAg[a].n_live = cnt; sReq[2] += cnt; }
        else { sReq[2] -= sAg[a].n_live; sAg[a].n_live = 0; if (sReq[2] < 0) sReq[2] = 0; }
        return cnt;
    }

    WDEV void mc_charge_loop(int ti) {                       // MobileCharger.py:59-69 from the top of `while True`
        int a = sTh[ti].agent;
        if (sTh[ti].tmp == 0.0) { th_sched(ti, PC_CHG_DONE, WRSN_NORMAL, now); return; }
        if (sAg[a].status == 0) { sAg[a].cur[2] = 0.0; th_sched(ti, PC_CHG_DEADWAIT, WRSN_NORMAL, now + sTh[ti].tmp); return; }
        double span = sTh[ti].tmp < 1.0 ? sTh[ti].tmp : 1.0;
        if (sAg[a].charging_rate != 0.0) { double lim = (sAg[a].energy - ec->mc_threshold) / sAg[a].charging_rate; if (lim < span) span = lim; }
        sTh[ti].cspan = span;
        // fast-forward: in generic position (no sub-step boundary on a node sampling instant k+0.5 / k+1.0) the
        // disconnect/reconnect pair of every boundary cancels, so the connection is made once and the whole-second
        // sub-steps stay virtual.  A node death or the charger running dry return to the exact path.
        const double tmp = sTh[ti].tmp;
        const double fr = now - floor(now);
        if (tmp > 3.0 && fr != 0.0 && fr != 0.5 && sAg[a].charging_rate == 0.0 && agent_single(a, ti)) {
            double nf = floor(tmp); if (nf == tmp) nf -= 1.0;          // unit sub-steps that are not the last one
            // rate this sub-step would connect with
            double cr = 0.0; const int nc = sAg[a].n_conn;
            for (int k = 0; k < nc; ++k) { int i = sConn[a * WRSN_CONN_CAP + k]; if (sLS[i] & 1) cr += conn_rate_of(a, k, i); }
            if (cr > 0.0) { double ne = floor((sAg[a].energy - ec->mc_threshold) / cr) - 1.0; if (ne < nf) nf = ne; }
            if (nf > 100000.0) nf = 100000.0;
            int n = (int)nf;
            bool ok = n >= 2;
            double t = now;
            for (int q = 0; q < n && ok; ++q) { t = t + 1.0; double f2 = t - floor(t); if (f2 == 0.0 || f2 == 0.5) ok = false; }
            if (ok) {
                mc_connect(a, 1.0);                          // charge_step #1 connects (MobileCharger.py:40-41)
                sTh[ti].cspan = 1.0; sTh[ti].ff = 2; sTh[ti].ff_n = n - 1; sTh[ti].ff_t = now;
                th_sched(ti, PC_CSTEP_TIMEOUT, WRSN_NORMAL, t);
                return;
            }
        }
        th_sched(ti, PC_CSTEP_INIT, WRSN_URGENT, now);
    }

    WDEV void cond_trigger(int j) {                          // Condition.succeed(): NORMAL at now
        if (sCTr[j]) return;
        sCTr[j] = 1; sCP[j] = 1; sCT[j] = now; sCS[j] = seq++;
    }

    WDEV void p_init_tail(int ti, double tmp) {              // MobileCharger.py:110, 116-121 / 128-130
        const int a = sTh[ti].agent;
        const double dx = sTh[ti].phy[0], dy = sTh[ti].phy[1], ct = sTh[ti].phy[2];
        double used = dist2(dx, dy, sAg[a].loc[0], sAg[a].loc[1]) * ec->pm;
        used += tmp * ct;
        used += dist2(dx, dy, ec->bs[0], ec->bs[1]) * ec->pm;
        sAg[a].cur[0] = dx; sAg[a].cur[1] = dy; sAg[a].cur[2] = ct;
        sAg[a].type_charging = 0;
        if (used > sAg[a].energy - ec->mc_threshold - ec->mc_capacity / 200.0) { sTh[ti].stage = 0; sTh[ti].m_dest[0] = ec->bs[0]; sTh[ti].m_dest[1] = ec->bs[1]; }
        else { sTh[ti].stage = 2; sTh[ti].m_dest[0] = dx; sTh[ti].m_dest[1] = dy; }
        th_sched(ti, PC_MOVE_INIT, WRSN_URGENT, now);
    }

    // returns a wave request (REQ_PRECHECK / REQ_CONN) or 0
    WDEV int thread_fire(int ti) {
        const int a = sTh[ti].agent;
        if (sTh[ti].ff != 0) ff_apply(ti, now, true);        // the resume event: every virtual sub-step precedes it
        switch (sTh[ti].pc) {
        case PC_P_INIT:                                      // MobileCharger.py:105-115: needs the O(N) sum
            for (int i = 0; i < 2 * M; ++i) if (i != ti && sTh[i].agent == a) ff_fallback(i);   // a second process on this charger
            pend = REQ_PRECHECK; pend_idx = ti; return REQ_PRECHECK;
        case PC_MOVE_INIT: {                                 // MobileCharger.py:82-84
            double mt = dist2(sTh[ti].m_dest[0], sTh[ti].m_dest[1], sAg[a].loc[0], sAg[a].loc[1]) / ec->velocity;
            sTh[ti].moving_time = mt; sTh[ti].total_time = mt;
            sTh[ti].mvec[0] = sTh[ti].m_dest[0] - sAg[a].loc[0]; sTh[ti].mvec[1] = sTh[ti].m_dest[1] - sAg[a].loc[1];
            mc_move_loop(ti);
            break; }
        case PC_MSTEP_INIT:                                  // MobileCharger.py:76
            th_sched(ti, PC_MSTEP_TIMEOUT, WRSN_NORMAL, now + sTh[ti].span);
            break;
        case PC_MSTEP_TIMEOUT:                               // MobileCharger.py:77-78
            sAg[a].loc[0] = sAg[a].loc[0] + sTh[ti].mvec[0] / sTh[ti].total_time * sTh[ti].span;
            sAg[a].loc[1] = sAg[a].loc[1] + sTh[ti].mvec[1] / sTh[ti].total_time * sTh[ti].span;
            sAg[a].energy -= ec->pm * sTh[ti].span * ec->velocity;
            th_sched(ti, PC_MSTEP_DONE, WRSN_NORMAL, now);
            break;
        case PC_MSTEP_DONE:                                  // MobileCharger.py:95-96
            sTh[ti].moving_time -= sTh[ti].span;
            mc_check_status(a);
            mc_move_loop(ti);
            break;
        case PC_MOVE_DEADWAIT:
            th_sched(ti, PC_MOVE_DONE, WRSN_NORMAL, now);
            break;
        case PC_MOVE_DONE:
            if (sTh[ti].stage == 0) th_sched(ti, PC_RECH_INIT, WRSN_URGENT, now);                    // :123
            else { sAg[a].type_charging = 1; th_sched(ti, PC_CHG_INIT, WRSN_URGENT, now); }         // :125-126 / :131-132
            break;
        case PC_RECH_INIT:                                   // MobileCharger.py:99-103
            if (dist2(sAg[a].loc[0], sAg[a].loc[1], ec->bs[0], ec->bs[1]) <= ec->epsilon) {
                sAg[a].loc[0] = ec->bs[0]; sAg[a].loc[1] = ec->bs[1]; sAg[a].energy = ec->mc_capacity;
            }
            th_sched(ti, PC_RECH_TIMEOUT, WRSN_NORMAL, now + 0.0);
            break;
        case PC_RECH_TIMEOUT:
            th_sched(ti, PC_RECH_DONE, WRSN_NORMAL, now);
            break;
        case PC_RECH_DONE:                                   // :124
            sTh[ti].stage = 2; sTh[ti].m_dest[0] = sTh[ti].phy[0]; sTh[ti].m_dest[1] = sTh[ti].phy[1];
            th_sched(ti, PC_MOVE_INIT, WRSN_URGENT, now);
            break;
        case PC_CHG_INIT:                                    // MobileCharger.py:52-58: needs the O(N) range scan
            sTh[ti].tmp = sTh[ti].phy[2];
            pend = REQ_CONN; pend_idx = ti; return REQ_CONN;
        case PC_CSTEP_INIT:                                  // MobileCharger.py:40-44
            mc_connect(a, 1.0);
            th_sched(ti, PC_CSTEP_TIMEOUT, WRSN_NORMAL, now + sTh[ti].cspan);
            break;
        case PC_CSTEP_TIMEOUT: {                             // MobileCharger.py:45-50
            sAg[a].energy = sAg[a].energy - sAg[a].charging_rate * sTh[ti].cspan;
            double rem = sAg[a].cur[2] - sTh[ti].cspan;
            sAg[a].cur[2] = rem > 0.0 ? rem : 0.0;
            mc_connect(a, -1.0);
            sAg[a].charging_rate = 0.0;
            th_sched(ti, PC_CSTEP_DONE, WRSN_NORMAL, now);
            break; }
        case PC_CSTEP_DONE:                                  // MobileCharger.py:70-72
            sTh[ti].tmp -= sTh[ti].cspan;
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
            sTh[ti].pc = PC_FINISHED;
            for (int j = 1; j <= L; ++j) if (sAg[sCA[j - 1]].cur_thread == ti) cond_trigger(j);
            break;
        default: break;
        }
        return 0;
    }

    WDEV int new_thread(int agent, double p0, double p1, double p2) {
        for (int i = 0; i < 2 * M; ++i) {
            int pc = sTh[i].pc;
            if (pc == PC_NONE || (pc == PC_FINISHED && sAg[sTh[i].agent].cur_thread != i)) {
                sTh[i].agent = agent; sTh[i].phy[0] = p0; sTh[i].phy[1] = p1; sTh[i].phy[2] = p2;
                sTh[i].stage = 0; sTh[i].moving_time = 0; sTh[i].total_time = 0; sTh[i].span = 0; sTh[i].tmp = 0; sTh[i].cspan = 0;
                sTh[i].ff = 0; sTh[i].ff_n = 0; sTh[i].ff_t = 0;
                th_sched(i, PC_P_INIT, WRSN_URGENT, now);
                return i;
            }
        }
        return -1;
    }

    WDEV bool agent_at_rest(int m) const {                   // WRSN.py:66 / :322
        return dist2(sAg[m].loc[0], sAg[m].loc[1], sAg[m].cur[0], sAg[m].cur[1]) < 1e-9 && sAg[m].cur[2] == 0.0;
    }

    // update_reward consumes the priorities only for alive chargers whose action type is "charging" and that have
    // connected nodes (WRSN.py:116-126); bit 1: such a charger is being moved by a (stale) process, so its location
    // has to be brought up to date before every reward instant
    WDEV int ur_flags() const {
        int f = 0;
        for (int m = 0; m < M; ++m) {
            if (sAg[m].status != 0 && sAg[m].type_charging && sAg[m].n_conn > 0) {
                f |= 1;
                for (int i = 0; i < 2 * M; ++i) if (sTh[i].agent == m && sTh[i].ff == 1 && sTh[i].pc != PC_NONE && sTh[i].pc != PC_FINISHED) f |= 2;
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
    WDEV int scalar_run(double svc, bool use_limit, double limit, int* arg, double* t_lim_out, int* flags_out) {
        switch (pend) {                                      // finish the item that asked for the service
        case REQ_PRECHECK: p_init_tail(pend_idx, svc); break;
        case REQ_CONN: mc_charge_loop(pend_idx); break;
        case REQ_GRID:
            if (deaths_flag) {                               // a node died: chargers connected to it re-plan on the exact path
                for (int i = 0; i < 2 * M; ++i) if (sTh[i].ff == 2 && sTh[i].pc != PC_NONE && sTh[i].pc != PC_FINISHED && sAg[sTh[i].agent].n_live > 0) ff_fallback(i);
            }
            break;
        default: break;
        }
        pend = 0;
        for (long guard = 0; guard < 4000000L; ++guard) {   // a step spans at most a few thousand seconds
            int kind = -1, idx = 0; double bt = 0.0; int bp = 0; int64_t bs = 0;
#define WRSN_CONSIDER(K, I, T_, P_, S_) if (kind < 0 || key_less((T_), (P_), (S_), bt, bp, bs)) { kind = (K); idx = (I); bt = (T_); bp = (P_); bs = (S_); }
            for (int i = 0; i < 2 * M; ++i) {
                int pc = sTh[i].pc;
                if (pc != PC_NONE && pc != PC_FINISHED) { WRSN_CONSIDER(3, i, sTh[i].time, sTh[i].prio, sTh[i].seq) }
            }
            for (int j = 1; j <= L; ++j) if (sCP[j]) { WRSN_CONSIDER(4, j, sCT[j], WRSN_NORMAL, sCS[j]) }
#undef WRSN_CONSIDER
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
            if (have_grid && gt < t_lim) {
                const int uf = ur_flags();
                if (uf & 2) { ff_sync_all(gt); *arg = 1; }   // stale "charging" mover: one item at a time, location kept current
                else *arg = 0;
                *t_lim_out = t_lim; *flags_out = uf & 1;
                pend = REQ_GRID; return REQ_GRID;
            }
            if (use_limit && !(have_ev && bt < limit)) { now = limit; return REQ_STOP; }
            if (have_grid && have_ev && gt == bt && key_less(gt, WRSN_NORMAL, gs, bt, bp, bs)) {
                const int uf = ur_flags();
                if (uf & 2) ff_sync_all(gt);
                *arg = 1; *t_lim_out = t_lim; *flags_out = uf & 1;
                pend = REQ_GRID; return REQ_GRID;            // tie at one instant: exactly one grid item goes first
            }
            now = bt; n_events++;
            if (kind == 3) {
                int r = thread_fire(idx);
                if (r) { *arg = (r == REQ_CONN) ? sTh[idx].agent : idx; return r; }
            } else {
                sCP[idx] = 0;
                if (idx == L) return REQ_STOP;               // StopSimulation
                cond_trigger(idx + 1);
            }
        }
        err = -6;
        return REQ_STOP;
    }

    // drive the environment until the run stops: lane 0 fires charger events, the wave runs the grid and the O(N) services
    WDEV void run(bool use_limit, double limit) {
        double svc = 0.0;
        for (long guard = 0; guard < 8000000L; ++guard) {
            if (lane == 0) {
                int arg = 0, fl = 0; double tl = 0.0;
                int req = scalar_run(svc, use_limit, limit, &arg, &tl, &fl);
                sReq[0] = req; sReq[1] = arg; sReq[3] = fl; sReqD[0] = tl; sReqD[1] = now; ((int64_t*)sReqD)[2] = seq;
            }
            __syncthreads();
            const int req = sReq[0], arg = sReq[1];
            now = sReqD[1]; seq = ((const int64_t*)sReqD)[2];      // lane 0 advanced them while firing events
            if (req == REQ_STOP) break;
            switch (req) {
            case REQ_GRID: grid_run(sReqD[0], arg != 0, sReq[3] != 0); break;
            case REQ_PRECHECK: svc = precheck(arg); break;
            case REQ_CONN: conn_build(arg); break;
            default: break;
            }
            __syncthreads();
        }
        __syncthreads();
        if (lane == 0) ff_sync_all(now);                     // bring virtual charger sub-steps up to the return instant
        __syncthreads();
    }
};

// ------------------------------------------------------------------ the environment kernel
// mode WARMUP: t = 0 .. warm_up_time with no charger activity, snapshot into d.snap  (WRSN.py:41-64)
// mode RESET : restore the snapshot into d.live and emit the reset request           (WRSN.py:66-75)
// mode STEP  : WRSN.step                                                            (WRSN.py:289-330)
template <int NPL>
__global__ void __launch_bounds__(64) wrsn_env_kernel(WrsnDev d, int mode, int env0, const int32_t* __restrict__ agent_id,
                                                      const double* __restrict__ action, int auto_reset,
                                                      const uint8_t* __restrict__ env_mask, WrsnStepOutDev out) {
    extern __shared__ double smem[];
    const int env = env0 + blockIdx.x;
    const int lane = threadIdx.x;
    if (env >= d.B) return;
    Sim<NPL> s;
    s.bind(d, env, lane, smem);
    const WrsnEnvConst* ec = s.ec;

    if (mode == WRSN_MODE_WARMUP) {
        // NetworkIO.makeNetwork + Node.__init__ (Node.py:12-43) + t = 0 process start-up
        s.am = 0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            int i = j * 64 + lane;
            bool real = i < s.N;
            s.E[j] = real ? ec->capacity : 0.0; s.CS[j] = 0.0; s.d1[j] = 0.0; s.d2[j] = 0.0;
            int al = (real && ec->capacity > ec->threshold) ? 1 : 0;
            s.sRR[i] = 0.0; s.sLS[i] = al; s.sRcv[i] = -1;
            s.am |= (unsigned)al << j;
        }
        s.now = 0.0; s.seq = 0; s.last_minfit = 0.0; s.opmax = 0.0;
        s.n_ticks = s.n_exact = s.n_events = 0;
        s.alive = 1; s.levels_dirty = 1; s.cache_dirty = 1; s.irreg = WRSN_RING; s.ring_len = 0; s.ring_head = 0;
        s.safe_ticks = 0; s.frozen = 0; s.log_pending = 0; s.L = 0;
        uint64_t* la = (uint64_t*)s.sAg;
        for (int w = lane; w < s.M * (int)(sizeof(WrsnAgent) / 8); w += 64) la[w] = 0;
        uint64_t* lt = (uint64_t*)s.sTh;
        for (int w = lane; w < 2 * s.M * (int)(sizeof(WrsnThread) / 8); w += 64) lt[w] = 0;
        for (int w = lane; w < s.M * WRSN_CONN_CAP; w += 64) { s.sConn[w] = 0; s.sConnRate[w] = 0.0; }
        for (int w = lane; w <= s.M; w += 64) { s.sCTr[w] = 0; s.sCP[w] = 0; s.sCA[w] = 0; s.sCT[w] = 0; s.sCS[w] = 0; }
        __syncthreads();
        if (lane == 0) {
            s.sReq[0] = 0; s.sReq[1] = 0; s.sReq[2] = 0;
            for (int m = 0; m < s.M; ++m) {                  // MobileCharger.__init__ + WRSN.py:44-49
                s.sAg[m].loc[0] = ec->bs[0]; s.sAg[m].loc[1] = ec->bs[1]; s.sAg[m].energy = ec->mc_capacity; s.sAg[m].charging_rate = 0.0;
                s.sAg[m].status = 1; s.mc_check_status(m);
                s.sAg[m].type_charging = 0; s.sAg[m].n_conn = 0; s.sAg[m].cur_thread = -1; s.sAg[m].n_live = 0;
                s.sAg[m].cur[0] = ec->bs[0]; s.sAg[m].cur[1] = ec->bs[1]; s.sAg[m].cur[2] = 0.0;
                s.sAg[m].excl = 0.0; s.sAg[m].prev_minfit = 0.0;
                s.sAg[m].conn_loc[0] = ec->bs[0]; s.sAg[m].conn_loc[1] = ec->bs[1];
            }
        }
        // Network.operate -> timeout(0.1); update_reward body at t = 0 (no charger is charging) -> timeout(1); nodes -> timeout(0.5)
        s.net_active = 1; s.net_phase = 0; s.net_time = s.now + 1.0 / 10.0; s.net_seq = s.seq++;
        s.ur_time = s.now + 1.0; s.ur_seq = s.seq++;
        s.node_phase = 0; s.node_time = s.now + 1.0 * 0.5; s.node_seq = s.seq++;
        s.ring = d.snap.ring + (size_t)env * WRSN_RING * s.NP; s.logbuf = d.snap.logbuf + (size_t)env * s.NP;
        __syncthreads();
        s.run(true, ec->warm_up_time);                       // env.run(until=warm_up_time): stops before that instant's NORMAL events
        double fit = s.min_fitness();
        s.last_minfit = fit;
        if (lane == 0) {
            for (int m = 0; m < s.M; ++m) {                  // WRSN.py:59-64
                s.sAg[m].action[0] = (ec->bs[0] - ec->frame[0]) / (ec->frame[1] - ec->frame[0]);
                s.sAg[m].action[1] = (ec->bs[1] - ec->frame[2]) / (ec->frame[3] - ec->frame[2]);
                s.sAg[m].action[2] = 0.0;
                s.sAg[m].cur_thread = s.new_thread(m, s.sAg[m].cur[0], s.sAg[m].cur[1], s.sAg[m].cur[2]);
                s.sAg[m].prev_minfit = fit; s.sAg[m].excl = 0.0;
            }
        }
        s.store(d.snap, 0, 0);
        return;
    }

    bool do_reset = (mode == WRSN_MODE_RESET);
    if (mode == WRSN_MODE_RESET && env_mask && env_mask[env] == 0) return;
    int aid = -1;
    if (mode == WRSN_MODE_STEP) {
        aid = agent_id[env];
        if (aid == -2) return;
        if (auto_reset && d.live.dyn[env].terminal_pending) do_reset = true;
    }

    if (do_reset) {
        s.load(d.snap);
        const double* rs = d.snap.ring + (size_t)env * WRSN_RING * s.NP; double* rl = d.live.ring + (size_t)env * WRSN_RING * s.NP;
        for (int w = lane; w < WRSN_RING * s.NP; w += 64) rl[w] = rs[w];
        for (int w = lane; w < s.NP; w += 64) d.live.logbuf[(size_t)env * s.NP + w] = d.snap.logbuf[(size_t)env * s.NP + w];
        if (lane == 0) {
            int agent = -1;
            for (int m = s.M - 1; m >= 0; --m) if (s.agent_at_rest(m)) agent = m;
            if (out.agent_id) out.agent_id[env] = agent;
            if (out.reward) out.reward[env] = 0.0;
            if (out.terminal) out.terminal[env] = (s.alive == 1) ? 0 : 1;
            if (out.now) out.now[env] = s.now;
            if (out.status) out.status[env] = (mode == WRSN_MODE_STEP) ? 3 : 0;
        }
        s.store(d.live, 0, 0);
        return;
    }

    // ---------------------------------------------------------- WRSN.step
    s.load(d.live);
    if (lane == 0) {
        int st0 = 0;
        if (aid >= 0 && aid < s.M) {
            double act[3];
            for (int k = 0; k < 3; ++k) { double v = action[(size_t)env * 3 + k]; act[k] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }   // np.clip (WRSN.py:299)
            s.sAg[aid].action[0] = act[0]; s.sAg[aid].action[1] = act[1]; s.sAg[aid].action[2] = act[2];
            double p0 = act[0] * (ec->frame[1] - ec->frame[0]) + ec->frame[0];      // translate (WRSN.py:95-98)
            double p1 = act[1] * (ec->frame[3] - ec->frame[2]) + ec->frame[2];
            double p2 = ec->charging_time_max * act[2];
            int ti = s.new_thread(aid, p0, p1, p2);
            if (ti < 0) s.err = -8; else s.sAg[aid].cur_thread = ti;
            s.sAg[aid].prev_minfit = s.last_minfit;          // WRSN.py:304 (node state is unchanged since the last return)
            s.sAg[aid].excl = 0.0;                           // WRSN.py:305
        }
        // general_process = net_process | p_a0 | p_a1 ... over chargers alive now (WRSN.py:307-310)
        s.L = 0;
        for (int m = 0; m < s.M; ++m) if (s.sAg[m].status != 0) { s.sCA[s.L] = m; s.L++; }
        if (s.L == 0) st0 = 2;                               // reference: run() never returns; deliberate deviation
        else {
            for (int j = 1; j <= s.L; ++j) {                 // Condition.__init__ checks processed operands at once
                int ti = s.sAg[s.sCA[j - 1]].cur_thread;
                if (ti >= 0 && s.sTh[ti].pc == PC_FINISHED) s.cond_trigger(j);
            }
        }
        s.sReq[3] = st0;
    }
    __syncthreads();
    const int st0 = s.sReq[3];
    int terminal = 0;
    double fit = 0.0;
    if (st0 == 2) terminal = 1;
    else {
        s.run(false, 0.0);                                   // env.run(until=general_process)
        if (s.alive == 0) terminal = 1;                      // WRSN.py:312-320
        else { fit = s.min_fitness(); s.last_minfit = fit; }
    }
    if (lane == 0) {
        int agent = -1, status = st0; double reward = 0.0;
        if (!terminal) {
            for (int m = s.M - 1; m >= 0; --m) if (s.agent_at_rest(m)) agent = m;   // lowest id (WRSN.py:321-322)
            if (agent >= 0) {                                // get_reward (WRSN.py:222-227)
                double term_all = fit - s.sAg[agent].prev_minfit;
                double term_excl = s.sAg[agent].excl / ec->avg_nodes_agent;
                reward = (term_all * 0.8 + 0.2 * term_excl) / (ec->charging_time_max + ec->moving_time_max);
            } else status = 1;                               // reference falls off the end and returns None
        }
        if (s.err != 0) status = -4;
        if (out.agent_id) out.agent_id[env] = agent;
        if (out.reward) out.reward[env] = reward;
        if (out.terminal) out.terminal[env] = (uint8_t)terminal;
        if (out.now) out.now[env] = s.now;
        if (out.status) out.status[env] = status;
    }
    s.store(d.live, terminal, 1);
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
    const double bx = ec->bs[0], by = ec->bs[1], com = ec->com_range, sen = ec->sen_range;
    int error = 0;
    // frame over nodes and the base station
    double x0 = bx, x1 = bx, y0 = by, y1 = by;
    for (int i = lane; i < NP; i += 64) {
        if (i < N) {
            double x = nx[i], y = ny[i];
            x0 = fmin(x0, x); x1 = fmax(x1, x); y0 = fmin(y0, y); y1 = fmax(y1, y);
            double db = dist2(bx, by, x, y);
            dbs[i] = db; nflags[i] = (db <= com) ? 1 : 0;
        } else { dbs[i] = 0.0; nflags[i] = 0; }
    }
    x0 = wv_min(x0); x1 = wv_max(x1); y0 = wv_min(y0); y1 = wv_max(y1);
    // neighbour lists, id order
    int base = 0;
    for (int i0 = 0; i0 < NP; i0 += 64) {
        int i = i0 + lane; int cnt = 0;
        if (i < N) for (int k = 0; k < N; ++k) if (k != i && dist2(nx[k], ny[k], nx[i], ny[i]) <= com) cnt++;
        int incl = wv_scan_incl(cnt, lane);
        int off = base + incl - cnt;
        nb_off[i] = off;
        if (i < N && off + cnt <= d.ECAP) {
            int p = off;
            for (int k = 0; k < N; ++k) { double dd = dist2(nx[k], ny[k], nx[i], ny[i]); if (k != i && dd <= com) { nb_idx[p] = k; nb_dist[p] = dd; ++p; } }
        }
        base += __shfl(incl, 63);
    }
    if (lane == 0) nb_off[NP] = base;
    if (base > d.ECAP) error = -1;
    // covered targets per node; target -> covering nodes, node-id order
    for (int i = lane; i < NP; i += 64) {
        int c = 0;
        if (i < N) for (int t = 0; t < T; ++t) if (dist2(nx[i], ny[i], tx[t], ty[t]) <= sen) c++;
        ncov[i] = c;
    }
    int tbase = 0;
    for (int t0 = 0; t0 < d.TP; t0 += 64) {
        int t = t0 + lane; int cnt = 0;
        if (t < T) for (int k = 0; k < N; ++k) if (dist2(nx[k], ny[k], tx[t], ty[t]) <= sen) cnt++;
        int incl = wv_scan_incl(cnt, lane);
        int off = tbase + incl - cnt;
        tc_off[t] = off;
        if (t < T && off + cnt <= d.CCAP) { int p = off; for (int k = 0; k < N; ++k) if (dist2(nx[k], ny[k], tx[t], ty[t]) <= sen) tc_idx[p++] = k; }
        tbase += __shfl(incl, 63);
    }
    if (lane == 0) tc_off[d.TP] = tbase;
    if (tbase > d.CCAP) error = -2;
    if (lane == 0) {
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

// ------------------------------------------------------------------ observation kernel: WRSN.get_state (WRSN.py:130-186)
// 256 threads per environment.  map_1 is a rank-N sum of separable Gaussians: node chunks are expanded into
// w*g(x) and g(y) rows in LDS and each thread accumulates a ROWS x 4 register tile; maps 2-4 are rank-1 terms.
#define WRSN_OBS_CH 32
#define WRSN_OBS_MAXROWS 16
__global__ void __launch_bounds__(256) wrsn_obs_kernel(WrsnDev d, const int32_t* __restrict__ agent_id, float* __restrict__ obs) {
    extern __shared__ double smem[];
    const int env = blockIdx.x, tid = threadIdx.x;
    const int aid = agent_id[env];
    if (aid < 0 || aid >= d.M) return;
    const WrsnEnvConst* ec = d.ec + env;
    const int N = ec->n_node, NP = d.NP, G = d.G, M = d.M;
    const size_t nb = (size_t)env * NP;
    const WrsnEnvDyn* dy = d.live.dyn + env;
    float* A = (float*)smem;                        // [CH][G]  weight * g(x - x_n)
    float* Bm = A + WRSN_OBS_CH * G;                // [CH][G]  g(y - y_n)
    double* pc = (double*)(Bm + WRSN_OBS_CH * G);   // [CH][3]  cx, cy, weight
    const double W = ec->frame[1] - ec->frame[0], H = ec->frame[3] - ec->frame[2];
    const double unit = 1.0 / G;
    const int CG = (G + 3) / 4;                     // column groups of 4
    const int RG = 256 / CG;                        // row groups
    const int RPG = (G + RG - 1) / RG;              // rows per group (<= WRSN_OBS_MAXROWS)
    const int cg = tid % CG, rg = tid / CG;
    const bool worker = rg < RG;
    const int i0 = rg * RPG, j0 = cg * 4;
    float acc[WRSN_OBS_MAXROWS][4];
#pragma unroll
    for (int r = 0; r < WRSN_OBS_MAXROWS; ++r) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f; }
    const double hX = ec->charging_range / W, hY = ec->charging_range / H;
    const float inv2hx = (float)(-1.0 / (2.0 * hX * hX)), inv2hy = (float)(-1.0 / (2.0 * hY * hY));
    float* out = obs + (size_t)env * 4 * G * G;

    for (int c0 = 0; c0 < N; c0 += WRSN_OBS_CH) {
        __syncthreads();
        if (tid < WRSN_OBS_CH) {
            int n = c0 + tid; double w = 0.0, cx = 0.0, cy = 0.0;
            if (n < N && (d.live.ls[nb + n] & 1)) {
                cx = (d.node_x[nb + n] - ec->frame[0]) / W; cy = (d.node_y[nb + n] - ec->frame[2]) / H;
                double e = d.live.E[nb + n], cs = d.live.CS[nb + n];
                w = (cs / (ec->alpha / (ec->beta * ec->beta))) / ((e - ec->threshold) / (ec->capacity - ec->threshold));
            }
            pc[tid * 3 + 0] = cx; pc[tid * 3 + 1] = cy; pc[tid * 3 + 2] = w;
        }
        __syncthreads();
        for (int idx = tid; idx < WRSN_OBS_CH * 2 * G; idx += 256) {
            int n = idx / (2 * G), r = idx - n * 2 * G;
            bool isx = r < G; int c = isx ? r : r - G;
            double cen = unit / 2 + c * unit;
            float df = (float)(cen - pc[n * 3 + (isx ? 0 : 1)]);
            float g = __expf(df * df * (isx ? inv2hx : inv2hy));
            if (isx) A[n * G + c] = g * (float)pc[n * 3 + 2]; else Bm[n * G + c] = g;
        }
        __syncthreads();
        if (worker) {
            for (int n = 0; n < WRSN_OBS_CH; ++n) {
                float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
                if (j0 < G) b0 = Bm[n * G + j0];
                if (j0 + 1 < G) b1 = Bm[n * G + j0 + 1];
                if (j0 + 2 < G) b2 = Bm[n * G + j0 + 2];
                if (j0 + 3 < G) b3 = Bm[n * G + j0 + 3];
#pragma unroll
                for (int r = 0; r < WRSN_OBS_MAXROWS; ++r) {
                    if (r < RPG && i0 + r < G) {
                        float a = A[n * G + i0 + r];
                        acc[r][0] += a * b0; acc[r][1] += a * b1; acc[r][2] += a * b2; acc[r][3] += a * b3;
                    }
                }
            }
        }
    }
    __syncthreads();
    // rank-1 terms of maps 2..4: rows gx/gy per term in LDS
    // term list: own charger (map 2), others charging (map 3), others moving (map 4)
    float* gx = A; float* gy = A + G;
    const WrsnAgent* ag = dy->ag;
    // map 1 store
    if (worker) {
#pragma unroll
        for (int r = 0; r < WRSN_OBS_MAXROWS; ++r) {
            if (r < RPG && i0 + r < G) {
                float* o = out + (size_t)(i0 + r) * G + j0;
                for (int c = 0; c < 4; ++c) if (j0 + c < G) o[c] = acc[r][c];
            }
        }
    }
    // maps 2..4
    for (int mp = 1; mp < 4; ++mp) {
#pragma unroll
        for (int r = 0; r < WRSN_OBS_MAXROWS; ++r) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f; }
        for (int o = 0; o < M; ++o) {
            double cxo, cyo, hx, hy, val;
            if (mp == 1) {
                if (o != aid) continue;
                cxo = (ag[o].loc[0] - ec->frame[0]) / W; cyo = (ag[o].loc[1] - ec->frame[2]) / H;
                double tmp = H < W ? H : W;
                hx = 0.5 * tmp / W; hy = 0.5 * tmp / H;
                val = ag[o].energy / ec->mc_capacity;
            } else {
                if (o == aid) continue;
                if (mp == 2 && !ag[o].type_charging) continue;      // map_3: others not "moving"
                if (mp == 3 && ag[o].type_charging) continue;       // map_4: others not "charging"
                cxo = (ag[o].cur[0] - ec->frame[0]) / W; cyo = (ag[o].cur[1] - ec->frame[2]) / H;
                hx = hX; hy = hY;
                if (mp == 2) val = ag[o].cur[2] / ec->charging_time_max;
                else val = (dist2(ag[o].loc[0], ag[o].loc[1], ag[o].cur[0], ag[aid].cur[1]) / ec->velocity) / ec->moving_time_max;   // mixed index as in WRSN.py:184
            }
            __syncthreads();
            for (int idx = tid; idx < 2 * G; idx += 256) {
                bool isx = idx < G; int c = isx ? idx : idx - G;
                double cen = unit / 2 + c * unit;
                double df = cen - (isx ? cxo : cyo); double h = isx ? hx : hy;
                float g = __expf((float)(df * df / (-2.0 * h * h)));
                if (isx) gx[c] = g * (float)val; else gy[c] = g;
            }
            __syncthreads();
            if (worker) {
#pragma unroll
                for (int r = 0; r < WRSN_OBS_MAXROWS; ++r) {
                    if (r < RPG && i0 + r < G) {
                        float a = gx[i0 + r];
                        for (int c = 0; c < 4; ++c) if (j0 + c < G) acc[r][c] += a * gy[j0 + c];
                    }
                }
            }
        }
        if (worker) {
#pragma unroll
            for (int r = 0; r < WRSN_OBS_MAXROWS; ++r) {
                if (r < RPG && i0 + r < G) {
                    float* o = out + (size_t)mp * G * G + (size_t)(i0 + r) * G + j0;
                    for (int c = 0; c < 4; ++c) if (j0 + c < G) o[c] = acc[r][c];
                }
            }
        }
    }
}

static inline int wrsn_obs_lds_bytes(int G) { return WRSN_OBS_CH * G * 4 * 2 + WRSN_OBS_CH * 3 * 8 + 64; }
