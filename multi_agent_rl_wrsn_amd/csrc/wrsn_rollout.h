// wrsn_rollout.h -- device-side roll-out bookkeeping of the asynchronous-agent batch (gfx950).
//
// Restates, for B environments at once, what the reference's trainer does per environment with Python lists
// (controller/ippo/IPPO.py:119-156, the same in controller/ppo/PPO.py:115-152):
//   * when charger a of an environment is given an action, the trainer remembers the observation it acted on, the raw
//     policy output (`input_action`: a 3-vector, or the G x G density map) and its log-probability (IPPO.py:141-142;
//     WRSN.step keeps prev_state / input_action per agent, WRSN.py:292,303);
//   * when a later WRSN.step returns that charger (not terminal), one transition (prev_state, input_action, log-prob,
//     reward, state) is appended to the charger's lists (IPPO.py:150-155); a charger returned before it ever acted in
//     the episode is skipped (IPPO.py:146-147: the first transition of every agent is dropped);
//   * a terminal return ends the episode: what the chargers had pending is discarded (IPPO.py:144-145).
// Stored terminals are therefore all False and cal_rt_adv's bootstrap term vanishes (IPPO.py:80-81): returns == rewards.
//
// Two kernels, both pure data movement (HBM-bound; 16-byte accesses):
//   wrsn_tr_record_kernel   pending[env][agent] <- (observation row, action row, log-prob)       one block per environment
//   wrsn_tr_collect_kernel  transition[agent][slot] <- (pending state/action/log-prob, reward, observation row)
#pragma once
#include <stdint.h>

struct WrsnTrBuffers {                    // mirrors wrsn_transition_buffers of include/wrsn_hip.h (device pointers)
    int32_t capacity, action_elems;
    float* pend_state; float* pend_action; float* pend_logp; uint8_t* pend_valid;
    float* state; float* action; float* next_state; float* reward; float* logp; double* now; int32_t* env; int32_t* count;
};

__device__ __forceinline__ void wrsn_tr_copy(float* __restrict__ dst, const float* __restrict__ src, int n, int tid, int nthreads) {
    // rows are 16-byte aligned whenever n is a multiple of 4 (G*G and 4*G*G with even G; 3-vectors take the tail loop)
    const int n4 = ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) ? (n >> 2) : 0;
    const float4* s4 = (const float4*)src; float4* d4 = (float4*)dst;
    for (int i = tid; i < n4; i += nthreads) d4[i] = s4[i];
    for (int i = 4 * n4 + tid; i < n; i += nthreads) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) wrsn_tr_record_kernel(int B, int M, int G, WrsnTrBuffers t, const int32_t* __restrict__ agent_id,
                                                             const float* __restrict__ action, const float* __restrict__ logp,
                                                             const float* __restrict__ obs) {
    const int e = blockIdx.x;
    if (e >= B) return;
    const int a = agent_id[e];
    if (a < 0 || a >= M) return;
    const int S = 4 * G * G, A = t.action_elems;
    const size_t slot = (size_t)e * M + a;
    wrsn_tr_copy(t.pend_state + slot * S, obs + (size_t)e * S, S, threadIdx.x, 256);
    wrsn_tr_copy(t.pend_action + slot * A, action + (size_t)e * A, A, threadIdx.x, 256);
    if (threadIdx.x == 0) { t.pend_logp[slot] = logp[e]; t.pend_valid[slot] = 1; }
}

__global__ void __launch_bounds__(256) wrsn_tr_collect_kernel(int B, int M, int G, WrsnTrBuffers t, const int32_t* __restrict__ agent_id,
                                                              const double* __restrict__ reward, const double* __restrict__ now,
                                                              int32_t* __restrict__ row_state, const float* __restrict__ obs) {
    extern __shared__ double smem[];                          // one int: the slot the block's transition goes to
    int* s_slot = (int*)smem;
    const int e = blockIdx.x;
    if (e >= B) return;
    // what the last environment launch did with this row (WrsnDev.row_state): only a row whose WRSN.step completed in that launch
    // carries a fresh request; rows left untouched (agent -2 / masked out) or still in flight (step budget) are skipped
    const int st = row_state[e];
    if (st == 0 || st == 3) return;
    __syncthreads();
    if (threadIdx.x == 0) row_state[e] = 0;                   // consumed: a second call after the same launch finds nothing
    if (st == 2 || st == 4) {                                 // episode over (terminal return) or restarted ((auto-)reset): pending actions are discarded
        if ((int)threadIdx.x < M) t.pend_valid[(size_t)e * M + threadIdx.x] = 0;
        return;
    }
    const int a = agent_id[e];
    if (a < 0 || a >= M) return;
    const size_t pslot = (size_t)e * M + a;
    if (!t.pend_valid[pslot]) return;                         // this charger has not acted yet in this episode (IPPO.py:146-147)
    if (threadIdx.x == 0) *s_slot = atomicAdd(&t.count[a], 1);
    __syncthreads();
    const int slot = *s_slot;
    if (slot >= t.capacity) return;                           // buffer full: counted, not stored
    const int S = 4 * G * G, A = t.action_elems;
    const size_t q = (size_t)a * t.capacity + slot;
    wrsn_tr_copy(t.state + q * S, t.pend_state + pslot * S, S, threadIdx.x, 256);
    wrsn_tr_copy(t.next_state + q * S, obs + (size_t)e * S, S, threadIdx.x, 256);
    wrsn_tr_copy(t.action + q * A, t.pend_action + pslot * A, A, threadIdx.x, 256);
    if (threadIdx.x == 0) { t.reward[q] = (float)reward[e]; t.logp[q] = t.pend_logp[pslot]; t.now[q] = now[e]; t.env[q] = e; }
}
