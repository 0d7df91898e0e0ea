// wrsn_types.h -- plain structs shared by the host C-ABI layer and the gfx950 kernels.
// Names follow the reference's domain (nodes, targets, mobile chargers, base station).
#pragma once
#include <stdint.h>

#define WRSN_WAVE 64
#define WRSN_MAX_MC 8                    // mobile chargers per environment
#define WRSN_MAX_TH (2 * WRSN_MAX_MC)    // operate_step processes alive at once (current + superseded)
#define WRSN_CONN_CAP 32                 // nodes inside one charger's charging range: stride of the connected-node lists in HBM (the lists in LDS are
                                         // sized to the scenario: WrsnDev.CC <= WRSN_CONN_CAP)
#define WRSN_RING 10                     // Node.log window (Node.py:71-77)

// run modes of the environment kernel
#define WRSN_MODE_STEP 0
#define WRSN_MODE_WARMUP 1
#define WRSN_MODE_RESET 2

// per-environment constants: scenario + charger parameters and what Network.__init__ /
// WRSN.reset derive from them (Network.py:16-27, WRSN.py:50-52)
struct WrsnEnvConst {
    double capacity, threshold, com_range, sen_range, package_size, er, et, efs, emp, max_time;
    double mc_capacity, mc_threshold, velocity, pm, charging_range, alpha, beta, epsilon;
    double bs[2], frame[4], density, moving_time_max, charging_time_max, avg_nodes_agent;
    double e_recv, d0, warm_up_time;
    int32_t n_node, n_target, n_edges, n_cover, error;
    int32_t conn_bound;               // upper bound of len(MobileCharger.connected_nodes): the most nodes within twice the charging range of one node
};

// MobileCharger object state (MobileCharger.py:6-32) + the per-agent lists WRSN keeps (WRSN.py:33-38)
struct WrsnAgent {
    double loc[2], energy, charging_rate, cur[3];   // location, energy, chargingRate, cur_phy_action
    double prev_minfit, excl, action[3];            // min(agents_prev_fitness), agents_exclusive_reward, agents_action
    double conn_loc[2];                             // charger location the cached connection rates were computed for
    double t_done;                                  // env.now at which the action in progress ends (move + charge; exact unless the charger
                                                    // runs dry): launch ordering only, nothing in the simulation reads it
    int32_t status, type_charging, n_conn, cur_thread;
    int32_t n_live, pad;                            // connections made by the running charge sub-step
};

// one operate_step process tree (MobileCharger.py:105-132): its single pending event + generator locals
struct WrsnThread {
    double time; int64_t seq;
    double phy[3];
    double m_dest[2], moving_time, mvec[2], total_time, span;   // move() locals
    double tmp, cspan;                                          // charge() locals
    double ff_t;                                                // fast-forward: end time of the last applied unit sub-step
    int32_t pc, agent, prio, stage;
    int32_t ff, ff_n;                                           // fast-forward kind (0 none, 1 move, 2 charge) and unit sub-steps still virtual
};

// everything else that persists between two step() calls of one environment
struct WrsnEnvDyn {
    double now; int64_t seq;
    double net_time; int64_t net_seq;        // Network.operate pending timeout
    double ur_time; int64_t ur_seq;          // WRSN.update_reward pending timeout
    double node_time; int64_t node_seq;      // the block of Node.operate timeouts
    double last_minfit, opmax;
    int64_t n_ticks, n_exact, n_events, n_steps;
    // cumulative since create, never reset: simulated seconds executed by WRSN.step calls (warm-up excluded), completed
    // WRSN.step calls that returned at the instant they started (the bookkeeping returns at t = warm_up_time)
    int64_t tot_ticks, tot_zero_steps;
    double step_t0;                          // env.now when the WRSN.step in progress (or last completed) was called
    int32_t net_phase, net_active, node_phase, alive;
    int32_t levels_dirty, cache_dirty, irreg, ring_len;
    int32_t ring_head, safe_ticks, frozen, terminal_pending;
    int32_t fit_dirty, map1_valid;           // node state changed (a grid service ran) since last_minfit was evaluated / since map 1 of the
    uint64_t map1_ptr;                       // observation was rendered into the row at this address (wrsn_set_obs_reuse; written by the observation kernel)
    int32_t n_connected, error, log_pending, susp;   // susp: WRSN.step in flight (work budget of a launch used up)
    // error (0 = none; the row reports status WRSN_ERR_CAPACITY and the code stays readable through wrsn_peek): -6 event loop guard, -7 no
    // event and no grid item, -8 no free process record, -9 connection list longer than its bound, -10 service loop ran out, -11 charging-rate
    // list full (more than M * CC + 32 nodes with a non-zero energyRR)
    // work-queue launches (wrsn_set_step_deadline): the action of a WRSN.step call that no wave has taken up yet (wrsn_latch_kernel); the
    // environment may wait for its turn over several launches, during which the caller's row says "in flight" and is ignored
    int32_t lat_valid, lat_agent; double lat_action[3];
    // the `|` conditions of the step in flight (WRSN.py:307-311); only meaningful while susp != 0
    double cond_time[WRSN_MAX_MC + 1]; int64_t cond_seq[WRSN_MAX_MC + 1];
    int32_t cond_agent[WRSN_MAX_MC + 1], cond_trig[WRSN_MAX_MC + 1], cond_pend[WRSN_MAX_MC + 1], cond_L;
    // rollout accumulators since create (never reset): sum of rewards per charger, finished episodes, sum of lifetimes
    // (env.now at terminal), completed WRSN.step calls
    double roll[WRSN_MAX_MC + 3];
    WrsnAgent ag[WRSN_MAX_MC];
    WrsnThread th[WRSN_MAX_TH];
};

// device memory of one handle (all arrays dense over the B environments)
struct WrsnNodeArrays {
    double *E, *CS, *RR, *d1, *d2;    // [B][NP]   energy, energyCS, energyRR, cached per-tick drain before / after the own half-charge
    double *ring;                     // [B][10][NP] Node.log window
    double *logbuf;                   // [B][NP]   Node.log_energy of a second whose k+0.5 half ran the exact walk
    int32_t *ls;                      // [B][NP]   ((level + 1) << 1) | alive
    int32_t *rcv;                     // [B][NP]   cached receiver: node id, -2 base station, -1 none
    int16_t *conn;                    // [B][MAX_MC][CONN_CAP] connected_nodes of each charger
    double *conn_xy;                  // [B][MAX_MC][CONN_CAP][2] position of every connected node
    WrsnEnvDyn *dyn;                  // [B]
};

struct WrsnDev {
    int32_t B, N, T, M, G, NP, TP, ECAP, CCAP;
    int32_t CC;                       // capacity of a charger's connected-node list in LDS: the largest number of nodes any disc of the charging
                                      // range can hold in this handle's scenarios (a multiple of 4, <= WRSN_CONN_CAP; wrsn_set_scenario)
    WrsnEnvConst *ec;                 // [B]
    double *node_x, *node_y, *dist_bs;   // [B][NP]
    double *target_x, *target_y;      // [B][TP]
    int32_t *nb_off, *nb_idx;         // [B][NP+1], [B][ECAP]  Node.neighbors sorted by (distance, id)
    double *nb_dist;                  // [B][ECAP]
    int32_t *tc_off, *tc_idx;         // [B][TP+1], [B][CCAP]  target -> covering nodes (id order)
    int32_t *ncov, *nflags;           // [B][NP]  len(listTargets); bit 0: in BaseStation.direct_nodes, bit 1: > 8 neighbours, bits 8..: ncov
    uint32_t *nbp;                    // [B][NP][4]  the eight nearest neighbour ids, 16 bit each (0xFFFF none), sorted by (distance, id)
    double *nbp_es;                   // [B][NP][8]  energy of sending one packet to that neighbour
    double *es_bs;                    // [B][NP]     energy of sending one packet to the base station
    uint32_t *adjm;                   // [B][NP][8]  neighbourhood of a node as a set mask over nodes 0..255 (4 x 64 bit; level BFS of N <= 256)
    int32_t *xorder;                  // [B][NP]     node ids in Morton order of their position (observation kernel)
    uint32_t *tcp;                    // [B][TP][4]  the first eight covering node ids of a target, packed alike
    WrsnNodeArrays live, snap;        // current state / post-warm-up snapshot
    int64_t *counters;                // [4]
    uint32_t *order_key;              // [BP2]  launch order of a step call, longest job first: (0xFFFF - estimated work) << 13 | environment,
    int32_t *order;                   // [BP2]  sorted ascending; order[b] = environment of block b (BP2 = B rounded up to a power of two)
    int32_t *row_state;               // [B]    what the last environment launch did with the row: 0 left untouched, 1 WRSN.step completed
                                      //        (fresh request), 2 reset / auto-reset request, 3 step still in flight, 4 terminal return
    long long *launch_t0;             // [1]    wall clock (100 MHz) at which the first wave of the current step launch started; zeroed by the
                                      //        sort kernel in front of it (wrsn_set_step_deadline)
    // time-sliced launches: queue[1] = environment the cyclic order of this launch starts at, queue[8 + (b & 63)] = 1 + the last block b that
    // took its environment (spread over 64 words: one word is a serial point for thousands of starting blocks)
    int32_t *queue;                   // [8 + 64]
    uint8_t *qskip;                   // [B]    the caller marked the row -2 in this call: nobody takes the environment
    int32_t *render_agent;            // [B]    charger whose observation the launch's render pass draws (-1: none); written by the
                                      //        environment kernel for every row, including the rows it leaves untouched
};

struct WrsnStepOutDev {
    int32_t *agent_id; double *reward; uint8_t *terminal; double *now; float *obs; int32_t *status;
};

#define WRSN_LDS_SCALAR_BYTES 64
// nodes under charge handled by the time-parallel steady batch (more fall back to the per-second path): four up to 256 nodes -- three
// chargers with one or two nodes in range each --, eight up to 512, six above (three
// environments of 1 024 nodes x 8 chargers then fit the 160 KB of a CU instead of two)
#define WRSN_CHG_MAX(NP_) ((NP_) > 512 ? 6 : ((NP_) > 256 ? 8 : 4))
// LDS of one environment wave; must match the carve-up of Sim (wrsn_sim.h)
static inline int wrsn_lds_bytes(int NP, int M, int CC) {
    int b = 0;
    b += 2 * NP * 8 + 2 * NP * 4;                     // 2 scratch arrays, level/alive words, cached receivers
    b += (M * CC + 32) * (8 + 2) + 4;                 // the charging rates (Node.energyRR) as a sparse list: rates, nodes, count
    b += M * (int)sizeof(WrsnAgent) + 2 * M * (int)sizeof(WrsnThread);
    b += (M + 1) * (8 + 8);                           // condition times / seqs
    b += 4 * M * CC * 8;                              // connected-node positions (x, y), reward-entry rates and accumulators
    b += 4 * 8 + WRSN_LDS_SCALAR_BYTES + 4 * 4;       // mailbox doubles, scalar bookkeeping, mailbox ints
    b += 3 * (M + 1) * 4 + 4;                         // condition agent / triggered / pending, reward-entry count
    b += 3 * M * CC * 2;                              // connected-node ids, reward-entry node / charger
    const int chg = WRSN_CHG_MAX(NP);
    b += NP * 4 + chg * 8 * 8 + chg * 64 * 8;         // steady batch: float CS, charged-node records and table
    b += (int)sizeof(WrsnEnvConst);                   // constants of the environment
    return (b + 31) & ~15;
}
