/*
 * wrsn_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, float64) of the
 * reference's WRSN step path: physical_env/network/{Node,Network,BaseStation}.py,
 * physical_env/mc/MobileCharger.py and rl_env/WRSN.py of
 * nguyenngocbaocmt02/multi_agent_rl_wrsn (read-only at /root/reference).
 *
 * It is the checker for the HIP product path (tests/, __graft_entry__.smoke(), and
 * bench.py's cpu_baseline leg) and is never imported, linked or executed by the
 * product package.  Parity pinning: checked against golden vectors produced by
 * running the reference's own classes in the build container (oracle/refharness/,
 * tests/golden/); the reference's discrete-event kernel SimPy 4.0.1 is absent from
 * the image, so those vectors are pinned to "reference classes + a stand-in that
 * follows SimPy 4.0.1's (time, priority, insertion) rule" -- see DESIGN.md.
 *
 * Structure: no generator/heap machinery.  Everything the reference schedules is
 * one of a handful of pending items, each carrying the SimPy queue key
 * (time, priority, seq):
 *   NET   Network.operate   (Network.py:69-81)   alternating setLevels / alive check
 *   UR    WRSN.update_reward (WRSN.py:100-127)   every integer second
 *   NODE  the N Node.operate processes (Node.py:45-78) -- they are inserted
 *         consecutively and therefore stay one contiguous block (k+0.5 / k+1.0)
 *   THREAD one per MobileCharger.operate_step process (MobileCharger.py:105-132);
 *         every SimPy hop (Initialize, Timeout, sub-process completion) of the nested
 *         move/move_step/charge/charge_step/recharge processes is one event
 *   COND  the nested `|` conditions WRSN.step builds (WRSN.py:307-311)
 * The BaseStation timeout (BaseStation.py:28-31) is a no-op and is not modelled.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define WO_URGENT 0
#define WO_NORMAL 1
#define WO_MAX_MC 16
#define WO_MAX_THREADS (4 * WO_MAX_MC)

typedef struct {
    int32_t n_node, n_target, n_mc, map_size;
    double warm_up_time;
    /* node_phy_spe (hanoi1000n50.yaml:1-11) */
    double capacity, threshold, com_range, sen_range, prob_gp, package_size, er, et, efs, emp;
    double max_time;
    /* mc_types/default.yaml:2-9 */
    double mc_capacity, mc_threshold, velocity, pm, charging_range, alpha, beta, epsilon;
} wo_cfg;

typedef struct {
    int32_t agent_id;   /* -1: none */
    int32_t terminal;
    int32_t status;     /* 0 ok, 1 fell off the end (reference returns None), 2 all MCs dead */
    int32_t pad;
    double reward;
    double now;
} wo_out;

enum { /* thread program counters: the pending SimPy event of an operate_step process tree */
    PC_NONE = 0, PC_P_INIT, PC_MOVE_INIT, PC_MSTEP_INIT, PC_MSTEP_TIMEOUT, PC_MSTEP_DONE,
    PC_MOVE_DEADWAIT, PC_MOVE_DONE, PC_RECH_INIT, PC_RECH_TIMEOUT, PC_RECH_DONE,
    PC_CHG_INIT, PC_CSTEP_INIT, PC_CSTEP_TIMEOUT, PC_CSTEP_DONE, PC_CHG_DEADWAIT, PC_CHG_DONE,
    PC_P_DONE, PC_FINISHED
};

typedef struct {
    int pc, agent, prio, stage;       /* stage: 0 = moving to BS (detour), 2 = moving to dest */
    double time; int64_t seq;
    double phy[3];                    /* phy_action of this process */
    double m_dest[2], moving_time, mvec[2], total_time, span;   /* move() locals */
    double tmp, cspan;                /* charge() locals */
} wo_thread;

typedef struct {
    double loc[2], energy, charging_rate, cur[3];
    int status, type_charging;        /* cur_action_type == "charging" */
    int n_conn; int *conn;            /* connected_nodes (ids, id order) */
    int cur_thread;                   /* agents_process[id] */
    double prev_minfit, excl;         /* min(agents_prev_fitness[id]), agents_exclusive_reward[id] */
    double action[3];                 /* agents_action[id] (normalised, clipped) */
} wo_agent;

typedef struct wo_t {
    wo_cfg c;
    int N, T, M, G;
    double *nx, *ny, *tx, *ty, bs[2];
    /* static topology, built at t=0 (Node.probe_neighbors / probe_targets, BaseStation.probe_neighbors) */
    int *nb_off, *nb_idx; double *nb_dist;
    int *cv_off, *cv_idx;             /* node -> covered targets (target id order) */
    int *direct; int n_direct; uint8_t *near_bs;
    double *dist_bs;
    double frame[4], nodes_density;
    double moving_time_max, charging_time_max, avg_nodes_agent;
    /* node state */
    double *E, *RR, *CS, *loge, *ring; int *ring_len, *status, *level;
    /* receiver cache: pure function of (level, status) */
    int *rcv; int64_t *rcv_epoch; int64_t epoch;
    int *targets_active; int alive;
    /* pending items */
    double now; int64_t seq;
    int net_phase, net_active; double net_time; int64_t net_seq;   /* phase 0: SL pending, 1: NC pending */
    double ur_time; int64_t ur_seq;
    int node_phase; double node_time; int64_t node_seq;             /* phase 0: NA pending, 1: NB pending */
    wo_thread th[WO_MAX_THREADS];
    wo_agent ag[WO_MAX_MC];
    /* conditions of the current step() */
    int L, cond_agent[WO_MAX_MC]; int cond_trig[WO_MAX_MC + 1], cond_pending[WO_MAX_MC + 1];
    double cond_time[WO_MAX_MC + 1]; int64_t cond_seq[WO_MAX_MC + 1];
    int stop_flag;
    /* statistics */
    int64_t n_ticks, n_hops, n_events;
    double *tmp_nt, *tmp_tt;          /* node_t / target_t scratch for fitness */
    int *bfs_a, *bfs_b, *lvl_old; int bfs_cap;
} wo_t;

static double dist2d(double ax, double ay, double bx, double by) {
    double dx = ax - bx, dy = ay - by;
    return sqrt(dx * dx + dy * dy);   /* scipy euclidean = sqrt(dot(u-v,u-v)) */
}

/* ------------------------------------------------------------------ creation */

wo_t *wo_create(const wo_cfg *cfg, const double *node_xy, const double *target_xy, const double *bs_xy) {
    wo_t *w = (wo_t *)calloc(1, sizeof(wo_t));
    w->c = *cfg;
    int N = w->N = cfg->n_node, T = w->T = cfg->n_target;
    w->M = cfg->n_mc; w->G = cfg->map_size;
    if (w->M > WO_MAX_MC || w->M < 1) { free(w); return NULL; }
    w->nx = malloc(sizeof(double) * N); w->ny = malloc(sizeof(double) * N);
    w->tx = malloc(sizeof(double) * (T + 1)); w->ty = malloc(sizeof(double) * (T + 1));
    for (int i = 0; i < N; i++) { w->nx[i] = node_xy[2 * i]; w->ny[i] = node_xy[2 * i + 1]; }
    for (int t = 0; t < T; t++) { w->tx[t] = target_xy[2 * t]; w->ty[t] = target_xy[2 * t + 1]; }
    w->bs[0] = bs_xy[0]; w->bs[1] = bs_xy[1];
    /* Node.probe_neighbors (Node.py:80-84): other nodes within com_range (<=), id order */
    w->nb_off = calloc(N + 1, sizeof(int));
    for (int i = 0; i < N; i++) {
        int c = 0;
        for (int j = 0; j < N; j++)
            if (j != i && dist2d(w->nx[j], w->ny[j], w->nx[i], w->ny[i]) <= cfg->com_range) c++;
        w->nb_off[i + 1] = w->nb_off[i] + c;
    }
    w->nb_idx = malloc(sizeof(int) * (w->nb_off[N] + 1));
    w->nb_dist = malloc(sizeof(double) * (w->nb_off[N] + 1));
    for (int i = 0; i < N; i++) {
        int p = w->nb_off[i];
        for (int j = 0; j < N; j++) {
            double d = dist2d(w->nx[j], w->ny[j], w->nx[i], w->ny[i]);
            if (j != i && d <= cfg->com_range) { w->nb_idx[p] = j; w->nb_dist[p] = d; p++; }
        }
    }
    /* Node.probe_targets (Node.py:86-90): targets within sen_range (<=), id order */
    w->cv_off = calloc(N + 1, sizeof(int));
    for (int i = 0; i < N; i++) {
        int c = 0;
        for (int t = 0; t < T; t++)
            if (dist2d(w->nx[i], w->ny[i], w->tx[t], w->ty[t]) <= cfg->sen_range) c++;
        w->cv_off[i + 1] = w->cv_off[i] + c;
    }
    w->cv_idx = malloc(sizeof(int) * (w->cv_off[N] + 1));
    for (int i = 0; i < N; i++) {
        int p = w->cv_off[i];
        for (int t = 0; t < T; t++)
            if (dist2d(w->nx[i], w->ny[i], w->tx[t], w->ty[t]) <= cfg->sen_range) w->cv_idx[p++] = t;
    }
    /* BaseStation.probe_neighbors (BaseStation.py:20-23): within the NODE's com_range */
    w->direct = malloc(sizeof(int) * (N + 1)); w->near_bs = calloc(N + 1, 1);
    w->dist_bs = malloc(sizeof(double) * (N + 1));
    w->n_direct = 0;
    for (int i = 0; i < N; i++) {
        w->dist_bs[i] = dist2d(w->bs[0], w->bs[1], w->nx[i], w->ny[i]);
        if (w->dist_bs[i] <= cfg->com_range) { w->direct[w->n_direct++] = i; w->near_bs[i] = 1; }
    }
    /* Network.__init__ frame (Network.py:16-27): nodes and base station, not targets */
    w->frame[0] = w->frame[1] = w->bs[0]; w->frame[2] = w->frame[3] = w->bs[1];
    for (int i = 0; i < N; i++) {
        if (w->nx[i] < w->frame[0]) w->frame[0] = w->nx[i];
        if (w->nx[i] > w->frame[1]) w->frame[1] = w->nx[i];
        if (w->ny[i] < w->frame[2]) w->frame[2] = w->ny[i];
        if (w->ny[i] > w->frame[3]) w->frame[3] = w->ny[i];
    }
    w->nodes_density = (double)N / ((w->frame[1] - w->frame[0]) * (w->frame[3] - w->frame[2]));
    /* WRSN.reset derived constants (WRSN.py:50-52) */
    w->moving_time_max = dist2d(w->frame[0], w->frame[2], w->frame[1], w->frame[3]) / cfg->velocity;
    w->charging_time_max = (cfg->capacity - cfg->threshold) / (cfg->alpha / (cfg->beta * cfg->beta));
    w->avg_nodes_agent = w->nodes_density * M_PI * (cfg->charging_range * cfg->charging_range);
    w->E = malloc(sizeof(double) * N); w->RR = malloc(sizeof(double) * N); w->CS = malloc(sizeof(double) * N);
    w->loge = malloc(sizeof(double) * N); w->ring = malloc(sizeof(double) * N * 10);
    w->ring_len = malloc(sizeof(int) * N); w->status = malloc(sizeof(int) * N); w->level = malloc(sizeof(int) * N);
    w->rcv = malloc(sizeof(int) * N); w->rcv_epoch = malloc(sizeof(int64_t) * N);
    w->targets_active = malloc(sizeof(int) * (T + 1));
    w->tmp_nt = malloc(sizeof(double) * N);
    w->tmp_tt = malloc(sizeof(double) * (T + 1));
    w->bfs_cap = w->nb_off[N] + N + 8;                      /* a frontier never exceeds (#directed edges + N) entries */
    w->bfs_a = malloc(sizeof(int) * w->bfs_cap); w->bfs_b = malloc(sizeof(int) * w->bfs_cap);
    w->lvl_old = malloc(sizeof(int) * (N + 1));
    for (int m = 0; m < w->M; m++) w->ag[m].conn = malloc(sizeof(int) * (N + 1));
    return w;
}

void wo_destroy(wo_t *w) {
    if (!w) return;
    free(w->nx); free(w->ny); free(w->tx); free(w->ty); free(w->nb_off); free(w->nb_idx); free(w->nb_dist);
    free(w->cv_off); free(w->cv_idx); free(w->direct); free(w->near_bs); free(w->dist_bs);
    free(w->E); free(w->RR); free(w->CS); free(w->loge); free(w->ring); free(w->ring_len); free(w->status);
    free(w->level); free(w->rcv); free(w->rcv_epoch); free(w->targets_active); free(w->tmp_nt);
    free(w->bfs_a); free(w->bfs_b); free(w->lvl_old); free(w->tmp_tt);
    for (int m = 0; m < w->M; m++) free(w->ag[m].conn);
    free(w);
}

/* ------------------------------------------------------------------ node model */

static void node_check_status(wo_t *w, int i) {            /* Node.py:148-151 */
    if (w->E[i] <= w->c.threshold) {
        if (w->status[i] != 0) w->epoch++;
        w->status[i] = 0; w->CS[i] = 0.0;
    }
}

static int node_find_receiver(wo_t *w, int i) {            /* Node.py:92-100; -1 = None */
    if (w->rcv_epoch[i] == w->epoch) return w->rcv[i];
    int best = -1; double bd = 0.0;
    for (int p = w->nb_off[i]; p < w->nb_off[i + 1]; p++) {
        int j = w->nb_idx[p];
        if (w->level[j] < w->level[i] && w->status[j] == 1) {
            if (best < 0 || w->nb_dist[p] < bd) { best = j; bd = w->nb_dist[p]; }   /* np.argmin: first minimum */
        }
    }
    w->rcv[i] = best; w->rcv_epoch[i] = w->epoch;
    return best;
}

static void node_receive_package(wo_t *w, int i);

static void node_send_package(wo_t *w, int i) {            /* Node.py:106-122 */
    const wo_cfg *c = &w->c;
    double d0 = sqrt(c->efs / c->emp);
    int receiver; double d;
    w->n_hops++;
    if (w->dist_bs[i] > c->com_range) {
        receiver = node_find_receiver(w, i);
        d = 0.0;
        if (receiver >= 0) d = dist2d(w->nx[i], w->ny[i], w->nx[receiver], w->ny[receiver]);
    } else {
        receiver = -2;                                       /* the base station */
        d = w->dist_bs[i];
    }
    if (receiver != -1) {
        double e_send = ((d <= d0) ? (c->et + c->efs * (d * d)) : (c->et + c->emp * ((d * d) * (d * d)))) * c->package_size;
        if (w->E[i] - c->threshold < e_send) {
            w->E[i] = c->threshold;
        } else {
            w->E[i] -= e_send;
            if (receiver >= 0) node_receive_package(w, receiver);   /* BaseStation.receive_package: no-op */
            w->loge[i] += e_send;
        }
    }
    node_check_status(w, i);
}

static void node_receive_package(wo_t *w, int i) {         /* Node.py:124-132 */
    const wo_cfg *c = &w->c;
    double e_receive = c->er * c->package_size;
    if (w->E[i] - c->threshold < e_receive) {
        w->E[i] = c->threshold;
    } else {
        w->E[i] -= e_receive;
        node_send_package(w, i);
        w->loge[i] += e_receive;
    }
    node_check_status(w, i);
}

static void node_block_half(wo_t *w) {                     /* Node.py:57-62, all nodes in id order */
    const wo_cfg *c = &w->c;
    for (int i = 0; i < w->N; i++) {
        if (w->status[i] == 0) continue;                   /* process ends */
        double e = w->E[i] + w->RR[i] * 1.0 * 0.5;
        w->E[i] = e < c->capacity ? e : c->capacity;
        /* prob_gp == 1: random.random() < 1 always holds */
        for (int p = w->cv_off[i]; p < w->cv_off[i + 1]; p++) node_send_package(w, i);   /* generate_packages */
    }
}

static void node_block_full(wo_t *w) {                     /* Node.py:65-77 then :54 */
    const wo_cfg *c = &w->c;
    for (int i = 0; i < w->N; i++) {
        if (w->status[i] == 0) continue;
        double e = w->E[i] + w->RR[i] * 1.0 * 0.5;
        w->E[i] = e < c->capacity ? e : c->capacity;
        int len_log = w->ring_len[i];
        double *log = w->ring + (size_t)i * 10;
        if (len_log < 10) {
            log[len_log] = w->loge[i];
            w->CS[i] = (w->CS[i] * len_log + w->loge[i]) / (len_log + 1);
            w->ring_len[i] = len_log + 1;
        } else {
            w->CS[i] = (w->CS[i] * len_log - log[0] + w->loge[i]) / len_log;
            memmove(log, log + 1, 9 * sizeof(double));
            log[9] = w->loge[i];
        }
        w->loge[i] = 0.0;
    }
}

/* ------------------------------------------------------------------ network */

static void net_set_levels(wo_t *w) {                      /* Network.py:37-66 */
    int N = w->N;
    int changed = 0;
    int *tmp1 = w->bfs_a, *tmp2 = w->bfs_b; int n1 = 0, n2 = 0;
    int *old = w->lvl_old;                                  /* previous levels, to detect changes */
    for (int i = 0; i < N; i++) { old[i] = w->level[i]; w->level[i] = -1; }
    for (int k = 0; k < w->n_direct; k++) {
        int i = w->direct[k];
        if (w->status[i] == 1) { w->level[i] = 1; tmp1[n1++] = i; }
    }
    for (int t = 0; t < w->T; t++) w->targets_active[t] = 0;
    while (n1 > 0) {
        n2 = 0;
        for (int a = 0; a < n1; a++) {
            int i = tmp1[a];
            for (int p = w->cv_off[i]; p < w->cv_off[i + 1]; p++) w->targets_active[w->cv_idx[p]] = 1;
            for (int p = w->nb_off[i]; p < w->nb_off[i + 1]; p++) {
                int j = w->nb_idx[p];
                if (w->status[j] == 1 && w->level[j] == -1) { tmp2[n2++] = j; w->level[j] = w->level[i] + 1; }
            }
        }
        int *sw = tmp1; tmp1 = tmp2; tmp2 = sw; n1 = n2;
    }
    for (int i = 0; i < N; i++) if (old[i] != w->level[i]) changed = 1;
    if (changed) w->epoch++;
}

static int net_check_targets(wo_t *w) {                    /* Network.py:84-85 */
    int m = 1;
    for (int t = 0; t < w->T; t++) if (w->targets_active[t] < m) m = w->targets_active[t];
    return m;                                               /* min([]) would raise; T >= 1 assumed */
}

/* ------------------------------------------------------------------ WRSN reward / fitness / state */

static void wrsn_update_reward(wo_t *w) {                  /* WRSN.py:100-127 */
    const wo_cfg *c = &w->c; int N = w->N;
    const double eps = 1e-9;
    int need = 0;
    for (int m = 0; m < w->M; m++) if (w->ag[m].status != 0 && w->ag[m].type_charging && w->ag[m].n_conn > 0) need = 1;
    if (!need) return;                                      /* priorities are unused otherwise */
    double *pr = w->tmp_nt;
    double sum = 0.0;
    for (int i = 0; i < N; i++) {
        pr[i] = (w->status[i] != 0) ? (w->CS[i] / (w->E[i] - c->threshold + eps)) : 0.0;
        sum += pr[i];
    }
    double mean = sum / N, var = 0.0;
    for (int i = 0; i < N; i++) var += (pr[i] - mean) * (pr[i] - mean);
    double std = sqrt(var / N);
    if (std == 0) std = eps;
    double tsum = 0.0;
    for (int i = 0; i < N; i++) { pr[i] = exp((pr[i] - mean) / std); tsum += pr[i]; }
    if (tsum == 0) tsum = eps;
    for (int i = 0; i < N; i++) pr[i] = pr[i] / tsum;
    for (int m = 0; m < w->M; m++) {
        wo_agent *a = &w->ag[m];
        if (a->status == 0) continue;
        if (!a->type_charging) continue;
        double incentive = 0.0;
        for (int k = 0; k < a->n_conn; k++) {
            int i = a->conn[k];
            if (w->status[i] == 1) {
                double dd = dist2d(w->nx[i], w->ny[i], a->loc[0], a->loc[1]) + c->beta;
                double chargingRate = c->alpha / (dd * dd);
                double e_no = fmin(w->E[i] - w->CS[i], c->threshold);              /* min/max as written */
                double e_with = fmax(w->E[i] - w->CS[i] + chargingRate, c->capacity);
                incentive += pr[i] * (e_with - e_no) / (c->alpha / (c->beta * c->beta));
            }
        }
        a->excl += incentive;
    }
}

static double wrsn_min_fitness(wo_t *w) {                  /* WRSN.py:188-220, returns np.min(target_t) */
    const wo_cfg *c = &w->c; int N = w->N;
    double *node_t = w->tmp_nt;
    int *tmp1 = w->bfs_a, *tmp2 = w->bfs_b; int n1 = 0, n2 = 0;
    for (int i = 0; i < N; i++) node_t[i] = -1.0;
    for (int k = 0; k < w->n_direct; k++) {
        int i = w->direct[k];
        if (w->status[i] == 1) {
            tmp1[n1++] = i;
            node_t[i] = (w->CS[i] == 0) ? INFINITY : (w->E[i] - c->threshold) / w->CS[i];
        }
    }
    while (n1 > 0) {
        n2 = 0;
        for (int a = 0; a < n1; a++) {
            int i = tmp1[a];
            for (int p = w->nb_off[i]; p < w->nb_off[i + 1]; p++) {
                int j = w->nb_idx[p];
                if (w->status[j] != 1) continue;
                double lt = (w->CS[j] == 0) ? INFINITY : (w->E[j] - c->threshold) / w->CS[j];
                if (node_t[j] == -1.0 || (node_t[i] > node_t[j] && lt > node_t[j])) {
                    tmp2[n2++] = j;                          /* at most one push per directed edge per round */
                    node_t[j] = lt < node_t[i] ? lt : node_t[i];
                }
            }
        }
        int *sw = tmp1; tmp1 = tmp2; tmp2 = sw; n1 = n2;
    }
    /* target_t[t] = max over covering nodes (any status) of node_t, starting from 0 (WRSN.py:216-219) */
    double *target_t = w->tmp_tt;
    for (int t = 0; t < w->T; t++) target_t[t] = 0.0;
    for (int i = 0; i < N; i++)
        for (int p = w->cv_off[i]; p < w->cv_off[i + 1]; p++)
            if (node_t[i] > target_t[w->cv_idx[p]]) target_t[w->cv_idx[p]] = node_t[i];
    double mn = INFINITY;
    for (int t = 0; t < w->T; t++) if (target_t[t] < mn) mn = target_t[t];
    return mn;
}

static void down_mapping(const wo_t *w, double x, double y, double *ox, double *oy) {   /* WRSN.py:86-88 */
    *ox = (x - w->frame[0]) / (w->frame[1] - w->frame[0]);
    *oy = (y - w->frame[2]) / (w->frame[3] - w->frame[2]);
}

static double gfunc(double x, double h) { return exp(x * x / (-2 * (h * h))); }        /* WRSN.py:16-18 */

void wo_get_state(wo_t *w, int agent_id, double *out) {   /* WRSN.py:130-186; out[4][G][G], axis0 <-> x */
    const wo_cfg *c = &w->c; int G = w->G, N = w->N;
    double unit = 1.0 / G;
    double *gx = malloc(sizeof(double) * G * 2), *gy = gx + G;
    double W = w->frame[1] - w->frame[0], H = w->frame[3] - w->frame[2];
    wo_agent *a = &w->ag[agent_id];
    memset(out, 0, sizeof(double) * 4 * G * G);
    double *m1 = out, *m2 = out + G * G, *m3 = out + 2 * G * G, *m4 = out + 3 * G * G;
    for (int n = 0; n < N; n++) {
        if (w->status[n] == 0) continue;
        double cx, cy; down_mapping(w, w->nx[n], w->ny[n], &cx, &cy);
        double hX = c->charging_range / W, hY = c->charging_range / H;
        double wgt = ((w->CS[n] / (c->alpha / (c->beta * c->beta))) / ((w->E[n] - c->threshold) / (c->capacity - c->threshold)));
        for (int i = 0; i < G; i++) { gx[i] = gfunc((unit / 2 + i * unit) - cx, hX); gy[i] = gfunc((unit / 2 + i * unit) - cy, hY); }
        for (int i = 0; i < G; i++) for (int j = 0; j < G; j++) m1[i * G + j] += wgt * gx[i] * gy[j];
    }
    {
        double cx, cy; down_mapping(w, a->loc[0], a->loc[1], &cx, &cy);
        double tmp = H < W ? H : W;
        double hX = 0.5 * tmp / W, hY = 0.5 * tmp / H;
        for (int i = 0; i < G; i++) { gx[i] = gfunc((unit / 2 + i * unit) - cx, hX); gy[i] = gfunc((unit / 2 + i * unit) - cy, hY); }
        for (int i = 0; i < G; i++) for (int j = 0; j < G; j++) m2[i * G + j] += (a->energy / c->mc_capacity) * gx[i] * gy[j];
    }
    for (int o = 0; o < w->M; o++) {
        wo_agent *b = &w->ag[o];
        if (o == agent_id) continue;
        double cx, cy; down_mapping(w, b->cur[0], b->cur[1], &cx, &cy);
        double hX = c->charging_range / W, hY = c->charging_range / H;
        for (int i = 0; i < G; i++) { gx[i] = gfunc((unit / 2 + i * unit) - cx, hX); gy[i] = gfunc((unit / 2 + i * unit) - cy, hY); }
        if (b->type_charging) {            /* map_3: others that are not "moving" */
            double v = b->cur[2] / w->charging_time_max;
            for (int i = 0; i < G; i++) for (int j = 0; j < G; j++) m3[i * G + j] += v * gx[i] * gy[j];
        } else {                           /* map_4: others that are not "charging"; mixed index as in WRSN.py:184 */
            double v = (dist2d(b->loc[0], b->loc[1], b->cur[0], a->cur[1]) / c->velocity) / w->moving_time_max;
            for (int i = 0; i < G; i++) for (int j = 0; j < G; j++) m4[i * G + j] += gx[i] * gy[j] * v;
        }
    }
    free(gx);
}

/* ------------------------------------------------------------------ mobile charger process tree */

static void th_sched(wo_t *w, wo_thread *t, int pc, int prio, double time) {
    t->pc = pc; t->prio = prio; t->time = time; t->seq = w->seq++;
}

static void mc_check_status(wo_t *w, wo_agent *a) {        /* MobileCharger.py:134-140 */
    if (a->energy <= w->c.mc_threshold) { a->status = 0; a->energy = w->c.mc_threshold; }
}

static void mc_move_loop(wo_t *w, wo_thread *t) {          /* MobileCharger.py:85-97, from the top of `while True` */
    const wo_cfg *c = &w->c; wo_agent *a = &w->ag[t->agent];
    if (t->moving_time <= 0) { th_sched(w, t, PC_MOVE_DONE, WO_NORMAL, w->now); return; }
    if (a->status == 0) { th_sched(w, t, PC_MOVE_DEADWAIT, WO_NORMAL, w->now + t->moving_time); return; }
    t->moving_time = dist2d(t->m_dest[0], t->m_dest[1], a->loc[0], a->loc[1]) / c->velocity;
    double s = t->moving_time < 1.0 ? t->moving_time : 1.0;
    double lim = (a->energy - c->mc_threshold) / (c->pm * c->velocity);
    t->span = s < lim ? s : lim;
    th_sched(w, t, PC_MSTEP_INIT, WO_URGENT, w->now);      /* env.process(self.move_step(...)) */
}

static void mc_start_move(wo_t *w, wo_thread *t, double dx, double dy) {
    t->m_dest[0] = dx; t->m_dest[1] = dy;
    th_sched(w, t, PC_MOVE_INIT, WO_URGENT, w->now);       /* env.process(self.move(destination)) */
}

static void mc_charge_loop(wo_t *w, wo_thread *t) {        /* MobileCharger.py:59-72, from the top of `while True` */
    const wo_cfg *c = &w->c; wo_agent *a = &w->ag[t->agent];
    if (t->tmp == 0) { th_sched(w, t, PC_CHG_DONE, WO_NORMAL, w->now); return; }
    if (a->status == 0) { a->cur[2] = 0; th_sched(w, t, PC_CHG_DEADWAIT, WO_NORMAL, w->now + t->tmp); return; }
    double span = t->tmp < 1.0 ? t->tmp : 1.0;
    if (a->charging_rate != 0) {
        double lim = (a->energy - c->mc_threshold) / a->charging_rate;
        if (lim < span) span = lim;
    }
    t->cspan = span;
    th_sched(w, t, PC_CSTEP_INIT, WO_URGENT, w->now);      /* env.process(self.charge_step(t=span)) */
}

static void cond_on_process_done(wo_t *w, int thread_idx);

static void thread_fire(wo_t *w, int ti) {
    const wo_cfg *c = &w->c; wo_thread *t = &w->th[ti]; wo_agent *a = &w->ag[t->agent];
    switch (t->pc) {
    case PC_P_INIT: {                                       /* MobileCharger.py:105-121/128-130 */
        double dx = t->phy[0], dy = t->phy[1], chargingTime = t->phy[2];
        double used = dist2d(dx, dy, a->loc[0], a->loc[1]) * c->pm;
        double tmp = 0;
        for (int i = 0; i < w->N; i++) {
            double dis = dist2d(dx, dy, w->nx[i], w->ny[i]);
            if (dis <= c->charging_range && w->status[i] == 1) tmp += c->alpha / ((dis + c->beta) * (dis + c->beta));
        }
        used += tmp * chargingTime;
        used += dist2d(dx, dy, w->bs[0], w->bs[1]) * c->pm;
        a->cur[0] = t->phy[0]; a->cur[1] = t->phy[1]; a->cur[2] = t->phy[2];
        a->type_charging = 0;
        if (used > a->energy - c->mc_threshold - c->mc_capacity / 200.0) { t->stage = 0; mc_start_move(w, t, w->bs[0], w->bs[1]); }
        else { t->stage = 2; mc_start_move(w, t, dx, dy); }
        break; }
    case PC_MOVE_INIT:                                      /* MobileCharger.py:82-84 */
        t->moving_time = dist2d(t->m_dest[0], t->m_dest[1], a->loc[0], a->loc[1]) / c->velocity;
        t->mvec[0] = t->m_dest[0] - a->loc[0]; t->mvec[1] = t->m_dest[1] - a->loc[1];
        t->total_time = t->moving_time;
        mc_move_loop(w, t);
        break;
    case PC_MSTEP_INIT:                                     /* MobileCharger.py:76 */
        th_sched(w, t, PC_MSTEP_TIMEOUT, WO_NORMAL, w->now + t->span);
        break;
    case PC_MSTEP_TIMEOUT:                                  /* MobileCharger.py:77-78 */
        a->loc[0] = a->loc[0] + t->mvec[0] / t->total_time * t->span;
        a->loc[1] = a->loc[1] + t->mvec[1] / t->total_time * t->span;
        a->energy -= c->pm * t->span * c->velocity;
        th_sched(w, t, PC_MSTEP_DONE, WO_NORMAL, w->now);
        break;
    case PC_MSTEP_DONE:                                     /* MobileCharger.py:95-96 */
        t->moving_time -= t->span;
        mc_check_status(w, a);
        mc_move_loop(w, t);
        break;
    case PC_MOVE_DEADWAIT:
        th_sched(w, t, PC_MOVE_DONE, WO_NORMAL, w->now);
        break;
    case PC_MOVE_DONE:
        if (t->stage == 0) th_sched(w, t, PC_RECH_INIT, WO_URGENT, w->now);            /* :123 */
        else { a->type_charging = 1; th_sched(w, t, PC_CHG_INIT, WO_URGENT, w->now); } /* :125-126 / :131-132 */
        break;
    case PC_RECH_INIT:                                      /* MobileCharger.py:99-103 */
        if (dist2d(a->loc[0], a->loc[1], w->bs[0], w->bs[1]) <= c->epsilon) {
            a->loc[0] = w->bs[0]; a->loc[1] = w->bs[1]; a->energy = c->mc_capacity;
        }
        th_sched(w, t, PC_RECH_TIMEOUT, WO_NORMAL, w->now + 0);
        break;
    case PC_RECH_TIMEOUT:
        th_sched(w, t, PC_RECH_DONE, WO_NORMAL, w->now);
        break;
    case PC_RECH_DONE:                                      /* :124 */
        t->stage = 2; mc_start_move(w, t, t->phy[0], t->phy[1]);
        break;
    case PC_CHG_INIT:                                       /* MobileCharger.py:52-58 */
        t->tmp = t->phy[2];
        a->n_conn = 0;
        for (int i = 0; i < w->N; i++)
            if (dist2d(w->nx[i], w->ny[i], a->loc[0], a->loc[1]) <= c->charging_range) a->conn[a->n_conn++] = i;
        mc_charge_loop(w, t);
        break;
    case PC_CSTEP_INIT:                                     /* MobileCharger.py:40-44 + Node.py:134-139 */
        for (int k = 0; k < a->n_conn; k++) {
            int i = a->conn[k];
            if (w->status[i] == 0) continue;
            double dd = dist2d(w->nx[i], w->ny[i], a->loc[0], a->loc[1]) + c->beta;
            double r = c->alpha / (dd * dd);
            w->RR[i] += r; a->charging_rate += r;
        }
        th_sched(w, t, PC_CSTEP_TIMEOUT, WO_NORMAL, w->now + t->cspan);
        break;
    case PC_CSTEP_TIMEOUT:                                  /* MobileCharger.py:45-50 + Node.py:141-146 */
        a->energy = a->energy - a->charging_rate * t->cspan;
        a->cur[2] = (a->cur[2] - t->cspan) > 0 ? (a->cur[2] - t->cspan) : 0;
        for (int k = 0; k < a->n_conn; k++) {
            int i = a->conn[k];
            if (w->status[i] == 0) continue;
            double dd = dist2d(w->nx[i], w->ny[i], a->loc[0], a->loc[1]) + c->beta;
            double r = c->alpha / (dd * dd);
            w->RR[i] -= r; a->charging_rate -= r;
        }
        a->charging_rate = 0;
        th_sched(w, t, PC_CSTEP_DONE, WO_NORMAL, w->now);
        break;
    case PC_CSTEP_DONE:                                     /* MobileCharger.py:70-72 */
        t->tmp -= t->cspan;
        mc_check_status(w, a);
        mc_charge_loop(w, t);
        break;
    case PC_CHG_DEADWAIT:
        th_sched(w, t, PC_CHG_DONE, WO_NORMAL, w->now);
        break;
    case PC_CHG_DONE:                                       /* operate_step returns */
        th_sched(w, t, PC_P_DONE, WO_NORMAL, w->now);
        break;
    case PC_P_DONE:
        t->pc = PC_FINISHED;
        cond_on_process_done(w, ti);
        break;
    default: break;
    }
}

/* ------------------------------------------------------------------ conditions of WRSN.step (WRSN.py:307-311) */

static void cond_trigger(wo_t *w, int j) {                  /* C_j.succeed(): NORMAL at now */
    if (w->cond_trig[j]) return;
    w->cond_trig[j] = 1; w->cond_pending[j] = 1; w->cond_time[j] = w->now; w->cond_seq[j] = w->seq++;
}

static void cond_on_process_done(wo_t *w, int ti) {
    for (int j = 1; j <= w->L; j++)
        if (w->ag[w->cond_agent[j - 1]].cur_thread == ti) cond_trigger(w, j);
}

static void cond_fire(wo_t *w, int j) {
    w->cond_pending[j] = 0;
    if (j == w->L) w->stop_flag = 1;                        /* StopSimulation */
    else cond_trigger(w, j + 1);
}

/* ------------------------------------------------------------------ event loop */

static int key_less(double t1, int p1, int64_t s1, double t2, int p2, int64_t s2) {
    if (t1 != t2) return t1 < t2;
    if (p1 != p2) return p1 < p2;
    return s1 < s2;
}

/* process the next pending item; returns 0 when `limit` (exclusive, URGENT stop of run(until=number)) is reached */
static int wo_step_event(wo_t *w, int use_limit, double limit) {
    int kind = -1, idx = 0; double bt = 0; int bp = 0; int64_t bs = 0;
#define CONSIDER(K, I, T_, P_, S_) do { if (kind < 0 || key_less((T_), (P_), (S_), bt, bp, bs)) { kind = (K); idx = (I); bt = (T_); bp = (P_); bs = (S_); } } while (0)
    if (w->net_active) CONSIDER(0, 0, w->net_time, WO_NORMAL, w->net_seq);
    CONSIDER(1, 0, w->ur_time, WO_NORMAL, w->ur_seq);
    CONSIDER(2, 0, w->node_time, WO_NORMAL, w->node_seq);
    for (int i = 0; i < WO_MAX_THREADS; i++)
        if (w->th[i].pc != PC_NONE && w->th[i].pc != PC_FINISHED) CONSIDER(3, i, w->th[i].time, w->th[i].prio, w->th[i].seq);
    for (int j = 1; j <= w->L; j++) if (w->cond_pending[j]) CONSIDER(4, j, w->cond_time[j], WO_NORMAL, w->cond_seq[j]);
#undef CONSIDER
    if (use_limit && !(bt < limit)) { w->now = limit; return 0; }
    w->now = bt; w->n_events++;
    switch (kind) {
    case 0:
        if (w->net_phase == 0) {                            /* Network.py:75-78 */
            net_set_levels(w); w->alive = net_check_targets(w);
            w->net_phase = 1; w->net_time = w->now + 9.0 * 1 / 10.0; w->net_seq = w->seq++;
        } else {                                            /* Network.py:78-80 */
            if (w->alive == 0 || w->now >= w->c.max_time) w->net_active = 0;
            else { w->net_phase = 0; w->net_time = w->now + 1 / 10.0; w->net_seq = w->seq++; }
        }
        break;
    case 1:
        wrsn_update_reward(w);
        w->ur_time = w->now + 1.0; w->ur_seq = w->seq++;
        break;
    case 2:
        if (w->node_phase == 0) { node_block_half(w); w->node_phase = 1; }
        else { node_block_full(w); w->node_phase = 0; w->n_ticks++; }
        w->node_time = w->now + 1 * 0.5; w->node_seq = w->seq++;
        break;
    case 3: thread_fire(w, idx); break;
    case 4: cond_fire(w, idx); break;
    }
    return 1;
}

static int new_thread(wo_t *w, int agent, const double phy[3]) {
    for (int i = 0; i < WO_MAX_THREADS; i++) {
        if (w->th[i].pc == PC_NONE || (w->th[i].pc == PC_FINISHED && w->ag[w->th[i].agent].cur_thread != i)) {
            memset(&w->th[i], 0, sizeof(wo_thread));
            w->th[i].agent = agent; w->th[i].phy[0] = phy[0]; w->th[i].phy[1] = phy[1]; w->th[i].phy[2] = phy[2];
            th_sched(w, &w->th[i], PC_P_INIT, WO_URGENT, w->now);
            return i;
        }
    }
    return -1;
}

static int agent_at_rest(const wo_t *w, int m) {           /* WRSN.py:66 / :322 */
    const wo_agent *a = &w->ag[m];
    return dist2d(a->loc[0], a->loc[1], a->cur[0], a->cur[1]) < 1e-9 && a->cur[2] == 0;
}

int wo_reset(wo_t *w, wo_out *out) {                       /* NetworkIO.makeNetwork + WRSN.reset (WRSN.py:41-83) */
    const wo_cfg *c = &w->c; int N = w->N;
    for (int i = 0; i < N; i++) {
        w->E[i] = c->capacity; w->RR[i] = 0; w->CS[i] = 0; w->loge[i] = 0; w->ring_len[i] = 0;
        w->status[i] = 1; w->level[i] = 0; w->rcv_epoch[i] = -1;
        if (w->E[i] <= c->threshold) { w->status[i] = 0; }  /* Node.__init__ check_status */
    }
    /* `level` is None until the first setLevels; find_receiver is never called before it */
    w->epoch = 1; w->alive = 1;
    for (int t = 0; t < w->T; t++) w->targets_active[t] = 1;
    w->now = 0; w->seq = 0; w->stop_flag = 0; w->L = 0;
    w->n_ticks = w->n_hops = w->n_events = 0;
    memset(w->th, 0, sizeof(w->th));
    for (int j = 0; j <= WO_MAX_MC; j++) w->cond_trig[j] = w->cond_pending[j] = 0;
    for (int m = 0; m < w->M; m++) {                        /* MobileCharger.__init__ + WRSN.py:44-49 */
        wo_agent *a = &w->ag[m];
        a->loc[0] = w->bs[0]; a->loc[1] = w->bs[1]; a->energy = c->mc_capacity; a->charging_rate = 0;
        a->status = 1; mc_check_status(w, a);
        a->type_charging = 0; a->n_conn = 0; a->cur_thread = -1;
        a->cur[0] = w->bs[0]; a->cur[1] = w->bs[1]; a->cur[2] = 0;
        a->excl = 0; a->prev_minfit = 0;
    }
    /* t = 0: Network.operate starts (timeout 0.1), update_reward body runs once, nodes start (timeout 0.5) */
    w->net_active = 1; w->net_phase = 0; w->net_time = w->now + 1 / 10.0; w->net_seq = w->seq++;
    wrsn_update_reward(w); w->ur_time = w->now + 1.0; w->ur_seq = w->seq++;
    w->node_phase = 0; w->node_time = w->now + 1 * 0.5; w->node_seq = w->seq++;
    /* env.run(until=warm_up_time): URGENT stop, i.e. before the NORMAL events of that instant */
    while (wo_step_event(w, 1, c->warm_up_time)) {}
    int terminal = (w->alive == 1) ? 0 : 1;
    double fit = wrsn_min_fitness(w);
    for (int m = 0; m < w->M; m++) {                        /* WRSN.py:59-64 */
        wo_agent *a = &w->ag[m];
        double ax, ay; down_mapping(w, w->bs[0], w->bs[1], &ax, &ay);
        a->action[0] = ax; a->action[1] = ay; a->action[2] = 0;
        double phy[3] = { a->cur[0], a->cur[1], a->cur[2] };
        a->cur_thread = new_thread(w, m, phy);
        a->prev_minfit = fit; a->excl = 0.0;
    }
    out->agent_id = -1; out->reward = 0.0; out->terminal = terminal; out->now = w->now; out->status = 0;
    for (int m = 0; m < w->M; m++) if (agent_at_rest(w, m)) { out->agent_id = m; break; }
    return 0;
}

/* WRSN.step (WRSN.py:289-330). action: normalised 3-vector (density_map=False path). agent_id < 0: "just run". */
int wo_step(wo_t *w, int agent_id, const double *action, wo_out *out) {
    const wo_cfg *c = &w->c;
    out->status = 0; out->reward = 0.0; out->terminal = 0; out->agent_id = -1;
    if (agent_id >= 0) {
        wo_agent *a = &w->ag[agent_id];
        double act[3];
        for (int k = 0; k < 3; k++) { double v = action[k]; act[k] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }   /* np.clip */
        a->action[0] = act[0]; a->action[1] = act[1]; a->action[2] = act[2];
        double phy[3];                                       /* translate, WRSN.py:95-98 */
        phy[0] = act[0] * (w->frame[1] - w->frame[0]) + w->frame[0];
        phy[1] = act[1] * (w->frame[3] - w->frame[2]) + w->frame[2];
        phy[2] = w->charging_time_max * act[2];
        int ti = new_thread(w, agent_id, phy);
        if (ti < 0) return -1;
        a->cur_thread = ti;
        a->prev_minfit = wrsn_min_fitness(w);                /* WRSN.py:304 */
        a->excl = 0;                                         /* WRSN.py:305 */
    }
    /* general_process = net_process | p_a0 | p_a1 ... over agents alive NOW (WRSN.py:307-310) */
    for (int j = 0; j <= WO_MAX_MC; j++) w->cond_trig[j] = w->cond_pending[j] = 0;
    w->L = 0;
    for (int m = 0; m < w->M; m++) if (w->ag[m].status != 0) w->cond_agent[w->L++] = m;
    if (w->L == 0) {          /* reference: run(until=net_process) never returns; deliberate deviation */
        out->terminal = 1; out->status = 2; out->now = w->now; return 0;
    }
    for (int j = 1; j <= w->L; j++) {                        /* Condition.__init__: processed operands are checked at once */
        int ti = w->ag[w->cond_agent[j - 1]].cur_thread;
        if (ti >= 0 && w->th[ti].pc == PC_FINISHED) cond_trigger(w, j);
    }
    w->stop_flag = 0;
    while (!w->stop_flag) wo_step_event(w, 0, 0.0);          /* env.run(until=general_process) */
    out->now = w->now;
    if (w->alive == 0) { out->terminal = 1; return 0; }      /* WRSN.py:312-320 */
    for (int m = 0; m < w->M; m++) {
        if (agent_at_rest(w, m)) {                           /* WRSN.py:321-330 */
            wo_agent *a = &w->ag[m];
            double fitness = wrsn_min_fitness(w);
            double term_all = fitness - a->prev_minfit;
            double term_excl = a->excl / w->avg_nodes_agent;
            out->reward = ((term_all * 0.8 + 0.2 * term_excl) / (w->charging_time_max + w->moving_time_max));
            out->agent_id = m;
            return 0;
        }
    }
    out->status = 1;                                         /* reference falls off the end: returns None */
    (void)c;
    return 0;
}

/* ------------------------------------------------------------------ inspection (parity tests) */

void wo_peek_nodes(const wo_t *w, double *E, double *CS, double *RR, int32_t *status, int32_t *level) {
    for (int i = 0; i < w->N; i++) {
        if (E) E[i] = w->E[i];
        if (CS) CS[i] = w->CS[i];
        if (RR) RR[i] = w->RR[i];
        if (status) status[i] = w->status[i];
        if (level) level[i] = w->level[i];
    }
}

/* per MC: loc x, loc y, energy, status, type_charging, cur0, cur1, cur2, n_conn, excl, prev_minfit, action0..2 -> 14 doubles */
void wo_peek_mcs(const wo_t *w, double *out) {
    for (int m = 0; m < w->M; m++) {
        const wo_agent *a = &w->ag[m]; double *o = out + 14 * m;
        o[0] = a->loc[0]; o[1] = a->loc[1]; o[2] = a->energy; o[3] = a->status; o[4] = a->type_charging;
        o[5] = a->cur[0]; o[6] = a->cur[1]; o[7] = a->cur[2]; o[8] = a->n_conn; o[9] = a->excl; o[10] = a->prev_minfit;
        o[11] = a->action[0]; o[12] = a->action[1]; o[13] = a->action[2];
    }
}

/* frame[4], nodes_density, moving_time_max, charging_time_max, avg_nodes_agent, now, alive, n_ticks, n_hops, n_events, min fitness */
void wo_peek_env(wo_t *w, double *out) {
    out[0] = w->frame[0]; out[1] = w->frame[1]; out[2] = w->frame[2]; out[3] = w->frame[3];
    out[4] = w->nodes_density; out[5] = w->moving_time_max; out[6] = w->charging_time_max; out[7] = w->avg_nodes_agent;
    out[8] = w->now; out[9] = w->alive; out[10] = (double)w->n_ticks; out[11] = (double)w->n_hops; out[12] = (double)w->n_events;
    out[13] = wrsn_min_fitness(w);
}

int wo_peek_topology(const wo_t *w, int32_t *degree, int32_t *n_cover, int32_t *direct) {
    for (int i = 0; i < w->N; i++) {
        if (degree) degree[i] = w->nb_off[i + 1] - w->nb_off[i];
        if (n_cover) n_cover[i] = w->cv_off[i + 1] - w->cv_off[i];
        if (direct) direct[i] = w->near_bs[i];
    }
    return w->nb_off[w->N];
}

void wo_peek_targets(const wo_t *w, int32_t *active) { for (int t = 0; t < w->T; t++) active[t] = w->targets_active[t]; }
