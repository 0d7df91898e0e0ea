/*
 * wrsn_hip.h -- C-ABI of libwrsn_hip.so, the MI355X (gfx950) implementation of the
 * WRSN environment step path of nguyenngocbaocmt02/multi_agent_rl_wrsn.
 *
 * The reference exposes this path as a Python class, not an FFI:
 *     rl_env/WRSN.py:21   class WRSN(gym.Env)
 *     rl_env/WRSN.py:22   WRSN(scenario_path, agent_type_path, num_agent, map_size, warm_up_time, density_map)
 *     rl_env/WRSN.py:41   reset()
 *     rl_env/WRSN.py:289  step(agent_id, input_action)
 * The entry points below are what a ctypes binding for a *batched* version of that class
 * binds (INTEGRATION.md shows the stub); the Python classes `VecWRSN` / `WRSN` of
 * multi_agent_rl_wrsn_amd are built on exactly these calls.
 *
 * Conventions
 *   - every function returns 0 on success or a negative wrsn_status; nothing throws
 *     across the boundary; wrsn_last_error() returns a thread-local message;
 *   - a handle is bound to one HIP device and one stream; calls on one handle must be
 *     serialised by the caller, different handles are independent (one per GPU);
 *   - pointers documented "device" are caller-owned HIP device pointers (for instance
 *     torch.Tensor.data_ptr()); pointers documented "host" are ordinary host memory;
 *   - the library owns only its internal environment state;
 *   - step/reset are asynchronous on the handle's stream (no host synchronisation).
 */
#ifndef WRSN_HIP_H
#define WRSN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wrsn_handle wrsn_t;

enum wrsn_status {
    WRSN_OK = 0,
    WRSN_ERR_ARG = -1,        /* bad argument */
    WRSN_ERR_HIP = -2,        /* HIP runtime error, see wrsn_last_error() */
    WRSN_ERR_NO_DEVICE = -3,  /* no usable gfx950 device */
    WRSN_ERR_CAPACITY = -4,   /* neighbour / coverage / connection list capacity exceeded */
    WRSN_ERR_STATE = -5       /* call sequence error (e.g. step before scenario set) */
};

/* Batch geometry.  Mirrors the constructor arguments of WRSN (rl_env/WRSN.py:22-30). */
typedef struct wrsn_cfg {
    int32_t n_env;            /* B: number of independent environments on this handle            */
    int32_t n_node;           /* N: sensor nodes per environment (max over the batch)            */
    int32_t n_target;         /* T: targets per environment (max over the batch)                 */
    int32_t n_mc;             /* M: mobile chargers, `num_agent` (WRSN.py:26); 1..8               */
    int32_t map_size;         /* G: observation is 4 x G x G (WRSN.py:27,31); 4..128              */
    int32_t device;           /* HIP device ordinal                                               */
    int32_t max_degree;       /* average neighbour-list capacity per node (0 = default 24)        */
    int32_t max_cover;        /* average covered-target capacity per node (0 = default 8)         */
    double  warm_up_time;     /* `warm_up_time` (WRSN.py:29,53), simulated seconds                */
} wrsn_cfg;

/* node_phy_spe of a scenario file (network_scenarios/hanoi1000n50.yaml:1-11), in this order. */
typedef struct wrsn_node_spec {
    double capacity, threshold, com_range, sen_range, prob_gp, package_size, er, et, efs, emp;
    double max_time;          /* scenario key `max_time` (NetworkIO.py:34, Network.py:78)         */
} wrsn_node_spec;

/* mc_types/default.yaml:2-9 */
typedef struct wrsn_mc_spec {
    double capacity, threshold, velocity, pm, charging_range, alpha, beta, epsilon;
} wrsn_mc_spec;

/* Per-call outputs of reset/step: the numeric content of the request dict WRSN.step returns
 * (WRSN.py:323-330; terminal form :313-320).  All pointers are DEVICE pointers, caller-owned;
 * obs may be NULL (no observation is rendered). */
typedef struct wrsn_step_out {
    int32_t *agent_id;        /* [B]  id of the charger that needs an action; -1 = none (terminal / fell off) */
    double  *reward;          /* [B]  get_reward (WRSN.py:222-227); 0 when agent_id < 0            */
    uint8_t *terminal;        /* [B]  1 when net.alive == 0 at return (WRSN.py:312)                */
    double  *now;             /* [B]  env.now at return                                            */
    float   *obs;             /* [B,4,G,G] get_state(agent_id) (WRSN.py:130-186); untouched rows for agent_id < 0 */
    int32_t *status;          /* [B]  0 ok; 1 step fell off the end (reference returns None);
                                      2 every charger dead (reference would hang); 3 auto-reset performed;
                                      4 the step is still in flight (wrsn_set_step_budget), call wrsn_step again;
                                      negative: per-env error (capacity)                           */
} wrsn_step_out;

/* what wrsn_peek copies; dst is a HOST pointer, dense [B, ...] in the listed dtype */
enum wrsn_peek_what {
    WRSN_PEEK_NODE_ENERGY = 0,   /* double [B,N]   Node.energy                                   */
    WRSN_PEEK_NODE_CS = 1,       /* double [B,N]   Node.energyCS                                  */
    WRSN_PEEK_NODE_RR = 2,       /* double [B,N]   Node.energyRR                                  */
    WRSN_PEEK_NODE_STATUS = 3,   /* int32  [B,N]   Node.status                                    */
    WRSN_PEEK_NODE_LEVEL = 4,    /* int32  [B,N]   Node.level                                     */
    WRSN_PEEK_MC = 5,            /* double [B,M,16] loc_x, loc_y, energy, status, charging, cur_x, cur_y,
                                                    cur_t, n_conn, excl, prev_minfit, act0, act1, act2, 0, 0 */
    WRSN_PEEK_ENV = 6,           /* double [B,16]  xmin,xmax,ymin,ymax,density,moving_time_max,charging_time_max,
                                                    avg_nodes_agent,now,alive,ticks,exact_ticks,events,min_fitness,
                                                    n_edges,n_cover                                 */
    WRSN_PEEK_NODE_DEGREE = 7,   /* int32  [B,N]   len(Node.neighbors)                            */
    WRSN_PEEK_NODE_NCOVER = 8,   /* int32  [B,N]   len(Node.listTargets)                          */
    WRSN_PEEK_NODE_DIRECT = 9,   /* int32  [B,N]   node in BaseStation.direct_nodes               */
    WRSN_PEEK_TARGETS_ACTIVE = 11 /* int32 [B,T]   Network.targets_active (Network.py:9,45-55): target covered by a node the last
                                                    setLevels reached; 0 beyond an environment's own target count.  (10 is
                                                    taken by the per-phase cycle counters of diagnostic builds.)       */
};

/* Create a handle for B environments on cfg->device.  Fails with WRSN_ERR_NO_DEVICE when no HIP
 * device is usable: there is no CPU fallback. */
int wrsn_create(const wrsn_cfg *cfg, wrsn_t **out);
void wrsn_destroy(wrsn_t *h);

/* Bind the handle to a HIP stream (hipStream_t passed as void*; NULL = default stream). */
int wrsn_set_stream(wrsn_t *h, void *hip_stream);

/* Load scenarios for environments [env0, env0+nenv): replaces NetworkIO.makeNetwork
 * (NetworkIO.py:19-34) + Network.__init__ (Network.py:4-33) + the t=0 probes
 * (Node.py:80-90, BaseStation.py:20-23).  HOST pointers:
 *   node_xy   [nenv, n_node, 2]   target_xy [nenv, n_target, 2]   bs_xy [nenv, 2]
 *   n_node_env / n_target_env [nenv] actual sizes (NULL = cfg sizes for all)
 *   node_spec [nenv] (or one spec broadcast when spec_stride == 0), mc_spec likewise.
 * Builds topology on the device, runs the warm-up (WRSN.py:53) on the device and caches the
 * post-warm-up snapshot reset() restores (the state after run(until=warm_up_time) is a pure
 * function of the scenario).  Synchronous. */
int wrsn_set_scenario(wrsn_t *h, int32_t env0, int32_t nenv,
                      const double *node_xy, const double *target_xy, const double *bs_xy,
                      const int32_t *n_node_env, const int32_t *n_target_env,
                      const wrsn_node_spec *node_spec, int32_t node_spec_stride,
                      const wrsn_mc_spec *mc_spec, int32_t mc_spec_stride);

/* WRSN.reset (WRSN.py:41-83) for the environments whose env_mask byte is non-zero
 * (DEVICE pointer [B]; NULL = all).  Outputs follow the reset request (agent 0, reward 0).  Rows of environments
 * the mask leaves out are not written at all: their pending request (agent_id included) stays valid. */
int wrsn_reset(wrsn_t *h, const uint8_t *env_mask, const wrsn_step_out *out);

/* WRSN.step (WRSN.py:289-330) for every environment, density_map=False path.
 *   agent_id [B] DEVICE int32: charger receiving `action` (>= 0), -1 = "just run" (WRSN.py:290),
 *                              -2 = leave this environment untouched (none of its output rows is written);
 *   action   [B,3] DEVICE double: normalised action, clipped to [0,1] inside (WRSN.py:299).
 * auto_reset != 0: an environment whose previous return was terminal is reset instead of stepped
 * and reports status 3 with the reset request (agent 0, reward 0). */
int wrsn_step(wrsn_t *h, const int32_t *agent_id, const double *action, int32_t auto_reset,
              const wrsn_step_out *out);

/* Bound the work one wrsn_step launch spends on an environment (0, the default: every environment runs its
 * WRSN.step to the end, like the blocking call of the reference).  The duration of a WRSN.step is heavy-tailed (it
 * runs until the next charger finishes: 1 .. several thousand simulated seconds), so in a batch a launch waits for
 * its slowest environment.  With a budget (in work units of roughly 400 shader cycles, counted per simulated second, grid item,
 * service, routing rebuild and packet-exact second; deterministic, not wall-clock) an environment whose step is not finished reports status 4 / agent_id -1
 * and the next wrsn_step goes on with it, ignoring agent_id/action of that row.  Every WRSN.step is still executed
 * in full and returns the same request (agent, time, terminal flag identical; rewards to ~1e-10, because a
 * suspension may split a closed-form jump / a batch of the float32 priority pipeline in two); only the launch it is reported in changes. */
int wrsn_set_step_budget(wrsn_t *h, int32_t work_units);

/* TIME-SLICED launches: a common deadline for the waves of a wrsn_step launch, in microseconds after it started (0, the default:
 * none).  The launch walks the environments in a cyclic order that starts where the previous launch stopped; a wave runs its
 * environment's WRSN.step until it returns or the deadline passes -- then it stops at the next grid-item boundary exactly as if a work
 * budget were used up (status 4 / agent_id -1; the next wrsn_step goes on with it) -- and a wave that would only start when the slice is
 * (nearly) over does not touch its environment at all: the action given for that row is kept in the environment's latch, the row
 * reports status 4 like a step in flight, and the next launches take it up (the caller passes -1 / anything for a status-4 row, as with
 * a step budget; -2 still leaves a row untouched for one call, a reset drops a latched action).  Every environment a launch takes
 * advances by at least one item, a packet-exact second is not begun in the last ~60 us of a slice.  The wave slots stay busy for the
 * whole slice -- with one work cap per environment a quarter of the slot time of a launch is idle while the capped waves finish -- and
 * no launch-order kernels are needed.  The requests are the same as ever (tests: agent, time, terminal identical, rewards to ~1e-9);
 * WHICH launch reports a request depends on timing (the work budget alone is deterministic).  A step budget, if set, still caps what one
 * visit may spend. */
int wrsn_set_step_deadline(wrsn_t *h, int32_t microseconds);

/* WRSN.density_map_to_action (WRSN.py:229-287) with the normalisation of WRSN.step (WRSN.py:293-296), for the
 * `density_map=True` policies of runner/IPPO.py: dmap DEVICE double [B, G, G] (probability map or logits), agent_id
 * DEVICE int32 [B] (< 0: row skipped), action DEVICE double [B, 3] = [x, y, map[argmax] / sum(map >= 99.9th
 * percentile)], ready for wrsn_step.  Arg-max cell, box and third component follow the reference exactly; the
 * charging spot inside the box comes from a deterministic bounded search instead of SciPy's L-BFGS-B (objective value
 * >= the optimiser's; parity of the spot itself is unpinned).  Asynchronous on the handle's stream. */
int wrsn_density_action(wrsn_t *h, const int32_t *agent_id, const double *dmap, double *action);

/* Rollout table accumulated by the step kernel since create (or since the last call with zero_after != 0):
 * dst DEVICE double [B, M + 3] = sum of rewards per charger (the returns IPPO consumes, IPPO.py:80-81), finished
 * episodes, sum of env.now at terminal, completed WRSN.step calls.  Asynchronous on the handle's stream; this is the
 * buffer a data-parallel trainer all-gathers (one collective per rollout, no per-step reduction kernels). */
int wrsn_rollout_table(wrsn_t *h, double *dst, int32_t zero_after);

/* Roll-out bookkeeping of the asynchronous-agent batch: what controller/ippo/IPPO.py:119-156 (and controller/ppo/PPO.py:
 * 115-152) keep in Python lists per environment, for B environments on the device.  All pointers are caller-owned DEVICE
 * memory (torch tensors); the library keeps no roll-out state of its own.
 *   pend_*   : per (environment, charger) the observation the charger last acted on, its raw policy output
 *              (`input_action`: action_elems = 3, or G*G for density-map policies) and the log-probability;
 *   the rest : per charger a list of `capacity` transitions (prev_state, input_action, log-prob, reward, state) plus the
 *              environment index and env.now of every transition; count[m] = transitions appended for charger m so far
 *              (it keeps counting past `capacity`: the excess is dropped, not stored). */
typedef struct wrsn_transition_buffers {
    int32_t capacity, action_elems;
    float   *pend_state;      /* [B, M, 4, G, G] */
    float   *pend_action;     /* [B, M, action_elems] */
    float   *pend_logp;       /* [B, M] */
    uint8_t *pend_valid;      /* [B, M]  1: the charger has acted in the running episode */
    float   *state;           /* [M, capacity, 4, G, G]  request["prev_state"]   (IPPO.py:150) */
    float   *action;          /* [M, capacity, action_elems]  request["input_action"] (IPPO.py:151) */
    float   *next_state;      /* [M, capacity, 4, G, G]  request["state"]        (IPPO.py:152) */
    float   *reward;          /* [M, capacity]           request["reward"]       (IPPO.py:153) */
    float   *logp;            /* [M, capacity]           log_probs_pre[agent]    (IPPO.py:154) */
    double  *now;             /* [M, capacity]           env.now at the return */
    int32_t *env;             /* [M, capacity]           environment of the transition */
    int32_t *count;           /* [M] */
} wrsn_transition_buffers;

/* The chargers named by agent_id (DEVICE int32 [B], < 0: row skipped) are about to be given `action` (DEVICE float
 * [B, action_elems], the policy's raw output) chosen with log-probability logp (DEVICE float [B]) on observation obs (DEVICE
 * float [B,4,G,G]): remember them as pending (IPPO.py:141-142).  Call before wrsn_step with the same agent_id. */
int wrsn_rollout_record(wrsn_t *h, const wrsn_transition_buffers *buf, const int32_t *agent_id, const float *action,
                        const float *logp, const float *obs);

/* After wrsn_step / wrsn_reset wrote `out` (every field non-NULL): rows whose WRSN.step completed in that launch and returned a
 * charger with a pending action append one transition to that charger's list (IPPO.py:146-155: a charger that has not acted yet
 * in the episode is skipped); terminal rows and (auto-)reset rows discard what was pending (IPPO.py:144-145); rows the launch left
 * untouched (agent -2 / masked out) or whose step is still in flight (status 4) are skipped -- the library remembers per row what
 * the last environment launch did, so a request is consumed once however often this is called. */
int wrsn_rollout_collect(wrsn_t *h, const wrsn_transition_buffers *buf, const wrsn_step_out *out);

/* Render get_state(agent) for arbitrary agents (DEVICE int32 [B], < 0 = skip) into obs (DEVICE). */
int wrsn_render(wrsn_t *h, const int32_t *agent_id, float *obs);

/* Copy internal state to HOST memory (parity tests, `net` / `agents` views).  Synchronises. */
int wrsn_peek(wrsn_t *h, int32_t what, void *dst);

/* Observation reuse.  Map 1 of get_state (WRSN.py:137-147) depends on node state only, maps 2..4 on the asking charger.  With
 * on != 0 the caller promises that an `obs` row the library wrote keeps its content until the library writes it again (same buffer
 * passed call after call, never modified, never re-allocated at the same address with other content); the render pass then leaves map 1
 * of a row alone when no simulated second has passed since it rendered that row at that address (a WRSN.step that returns at the
 * instant it was called) and writes only maps 2..4.  Results are bit-identical; default off. */
int wrsn_set_obs_reuse(wrsn_t *h, int32_t on);

/* Per-kernel timing of the step path with HIP events recorded on the handle's stream (the stream the kernels are launched on).
 * wrsn_set_timing(h, 1) makes every following wrsn_step record four events; wrsn_kernel_times waits for the last call and
 * returns, in milliseconds: ms[0] launch-order kernels (work estimate + sort), ms[1] step kernel, ms[2] continuation launch of the
 * two-launch variant (0 otherwise), ms[3] observation kernel (0 when no observation was requested).  Measurement only. */
int wrsn_set_timing(wrsn_t *h, int32_t on);
int wrsn_kernel_times(wrsn_t *h, float *ms);

/* Wait for the handle's stream. */
int wrsn_sync(wrsn_t *h);

/* Device counters, summed over the environments.  HOST pointer to 8 x int64.  Synchronises.
 *   [0] simulated seconds, [1] packet-exact seconds, [2] charger events of the episodes in progress (they restart at
 *       every reset: the warm-up is part of them);
 *   [3] completed WRSN.step calls since create;
 *   [4] simulated seconds executed inside WRSN.step calls since create (warm-up excluded, never reset);
 *   [5] completed WRSN.step calls since create that returned at the instant they were called (the bookkeeping returns
 *       at t = warm_up_time, SURVEY.md A.4, and same-instant completions);  [6], [7] reserved (0). */
int wrsn_counters(wrsn_t *h, int64_t *dst);

/* Seeded synthetic network generator (host code): fills HOST arrays
 * node_xy [n_node,2], target_xy [n_target,2], bs_xy [2].  side <= 0 selects
 * 1000 * max(1, sqrt(n_node / 200)) metres. */
int wrsn_synth_network(uint64_t seed, int32_t n_node, int32_t n_target, double side,
                       double com_range, double sen_range,
                       double *node_xy, double *target_xy, double *bs_xy);

const char *wrsn_last_error(void);
const char *wrsn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* WRSN_HIP_H */
