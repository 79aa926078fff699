/*
 * soccer_hip.h — C ABI of libsoccer_hip.so
 *
 * MI355X (gfx950) batched implementation of the step/reset hot path of the
 * Littman-94 grid-soccer Markov game, i.e. of
 *     SoccerSimultaneousEnv.step   (gym_soccer/envs/soccer_simultaneous_env.py:375-408)
 *     SoccerSimultaneousEnv.reset  (gym_soccer/envs/soccer_simultaneous_env.py:410-424)
 * of mimoralea/gym-soccer-littman94, plus the constructor-time rule functions
 * that give those two their meaning (:60-61, :63-109, :146-165, :202-256,
 * :296-362, :364-373).  The reference has no FFI of its own (it is pure
 * Python); these are the entry points a ctypes binding inside that file would
 * call — see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; no exception crosses
 *     the ABI.  soccer_last_error(h) gives the message (h may be NULL for
 *     errors raised before a handle exists).
 *   - "lane" = one environment instance.  A handle owns the resident state of
 *     n_lanes lanes on ONE device, a HIP stream and the staged rule tables.
 *   - unless a parameter says "host", array pointers are DEVICE pointers to
 *     caller-owned buffers of n_lanes elements; all work is enqueued on the
 *     handle's stream and is asynchronous (soccer_sync to wait).
 *   - one handle per host thread; calls on one handle are serialised by its
 *     stream.
 *   - state is structure-of-arrays in HBM: row_a,col_a,row_b,col_b int8[n],
 *     poss uint8[n] (bit0: 0 = A has the ball, 1 = B; bit1: lane needs reset),
 *     t uint8[n] (steps taken in the episode, 0..max_steps).
 *
 * Randomness (replaces the reference's per-env np.random.RandomState,
 * :57-58, consumed once per step and once per reset through gym's
 * categorical_sample, :395, :414):
 *   every batched_reset / batched_step call consumes one "tick" k of the handle (a T-step rollout consumes T).
 *   Four consecutive GLOBAL lanes share Philox4x32-10 blocks:
 *       g = lane_offset + i,  q = g >> 2,  key = (seed & 0xffffffff, seed >> 32)
 *       block(c, purpose) = philox4x32_10(counter = (q & 0xffffffff, q >> 32, c & 0xffffffff, (c >> 32) | purpose << 31), key)
 *   (purpose 0: step/reset, 1: in-kernel action sampling; k < 2^63) and lane g owns word w = block[g & 3].
 *   A uniform is always u = (m + 1/2) * 2^-b for a b-bit integer m — never 0, never on a dyadic threshold:
 *     slip_prob > 0   one block per tick, block(k, 0):
 *         step uniform   u = ((w >> 2) + 1/2) * 2^-30      (30 bits)
 *         reset uniform  u = ((w & 3) + 1/2) / 4           (the ISD has 2 or 4 equiprobable entries, :146-165)
 *     slip_prob == 0  every list probability is 1, 1/2 or 1/4 (:326-360), so a step needs floor(4u) and nothing else:
 *       ONE block serves EIGHT ticks, block(k >> 3, 0); tick k takes nibble number (k & 7) ^ 1 of w, counted from the
 *       least significant end:   nib = (w >> (4 * ((k & 7) ^ 1))) & 15
 *         step uniform   u = ((nib >> 2) + 1/2) / 4
 *         reset uniform  u = ((nib & 3) + 1/2) / 4
 *       (ABI 3.  The fused rollout is bound by vector issue, and a Philox block per step was a third of it.)
 *   So results depend on (seed, global lane id, tick) only — never on the
 *   device count or the launch geometry.  Callers that want to feed their own
 *   uniforms (e.g. the single-env facade, which keeps the reference's MT19937
 *   stream on the host) pass u_step / u_reset arrays instead.
 *   Outcome selection is the reference's: first index whose sequential
 *   float64 running sum of list probabilities exceeds u, index 0 if none does.
 */
#ifndef SOCCER_HIP_H
#define SOCCER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOCCER_ABI_VERSION 3      /* 2: soccer_step_args grew reward_a_f32 / reward_b_f32 / finished
                                     3: the bits -> uniform convention above (half-step offset; eight ticks per block at slip_prob == 0)
                                     (still 3: soccer_trajectory_returns, soccer_comm_* and batched_rollout_ex were ADDED, captured sequences may hold an odd
                                     number of calls, and a caller's u >= 1 on a slip list follows the reference's comparison — nothing a
                                     round-3 caller relied on changed, and checkpoints record this number for the RNG convention alone) */

/* error codes */
#define SOCCER_OK            0
#define SOCCER_E_INVALID    -1   /* bad argument / config (the reference's AssertionError cases) */
#define SOCCER_E_HIP        -2   /* a HIP runtime call failed */
#define SOCCER_E_NOMEM      -3
#define SOCCER_E_STATE      -4   /* call not valid in the handle's current state (e.g. capture) */

/* cfg.flags */
#define SOCCER_F_AUTORESET   1u  /* lanes that terminate/truncate are reset inside the same step */
#define SOCCER_F_NULL_STREAM 2u  /* enqueue on the device's default (null) stream; cfg.stream ignored */
#define SOCCER_F_STEP_STATS  8u  /* batched_step also feeds the episode histogram (soccer_get_stats); off by
                                    default: it costs ~5 % of a launch.  batched_rollout always counts. */
#define SOCCER_F_HOST_MAPPED 4u  /* small handles (single-env facade): state and staging live in pinned host memory
                                    the GPU reads/writes in place, so the *_host calls and state access copy nothing */
#define SOCCER_F_STREAM_ACTIONS 16u /* batched_step reads its action streams with the non-temporal hint: for callers that
                                    walk through action data larger than the Infinity Cache once (long pre-generated
                                    trajectories).  Unset (default): plain loads, which is faster when the buffers a step
                                    reads were written or read a few steps ago (the loop of an RL agent; a short captured
                                    sequence that is replayed) and slower when they stream in from HBM. */

/* actions (soccer_simultaneous_env.py:8-12); moves are (dcol,drow) (:24-30) */
#define SOCCER_NOOP  0
#define SOCCER_NORTH 1
#define SOCCER_SOUTH 2
#define SOCCER_EAST  3
#define SOCCER_WEST  4

typedef struct soccer_handle soccer_handle;
typedef struct soccer_graph  soccer_graph;

/* Constructor arguments: SoccerSimultaneousEnv.__init__(width, height, slip_prob, ..., seed)
 * (soccer_simultaneous_env.py:35) + the batching parameters the reference does not have. */
typedef struct soccer_config {
    uint64_t n_lanes;       /* environments resident on this handle (>=1) */
    int32_t  width;         /* pitch columns WITHOUT the two goal columns, >=5 (:45) */
    int32_t  height;        /* pitch rows, >=4 (:46) */
    double   slip_prob;     /* [0,1] (:50) */
    int32_t  max_steps;     /* truncation limit; the reference hard-codes 100 (:404). 1..250 */
    int32_t  device;        /* HIP device ordinal */
    uint64_t seed;          /* Philox key */
    uint64_t lane_offset;   /* global id of lane 0 (multi-GPU sharding) */
    uint32_t flags;         /* SOCCER_F_* */
    uint32_t envs_per_thread; /* 0 = library default; 1, 4 or 8 force a vector width */
    void*    stream;        /* hipStream_t to enqueue on, or NULL: the handle creates its own */
} soccer_config;

/* batched_step_ex arguments.  Required: act_a, act_b (each unless that player has a fixed policy).
 * Every output may be NULL (skipped).
 * Action bytes: 0..4.  The *_host / *_staged entry points check every byte on the CPU and return
 * SOCCER_E_INVALID for anything else (nothing is launched).  On the device-pointer paths (batched_step,
 * batched_step_ex, batched_rollout) the kernels cannot raise: a byte b executes as the move table[b & 7] with
 * 5..7 = NOOP — so no value can index outside a rule table or leave the pitch — and any byte outside 0..4 sets
 * the sticky SOCCER_MISUSE_ACTION flag (soccer_get_stats / soccer_peek_misuse). */
typedef struct soccer_step_args {
    const int8_t*  act_a;       /* [n] action of player A, 0..4 */
    const int8_t*  act_b;       /* [n] action of player B, 0..4 */
    const double*  u_step;      /* [n] uniforms for outcome selection, or NULL: per-lane Philox */
    const double*  u_reset;     /* [n] uniforms for in-step auto-reset, or NULL: per-lane Philox */
    uint16_t*      obs;         /* [n] observation index after the step (after auto-reset if it fired) */
    int8_t*        reward;      /* [n] player A's reward -1/0/+1; B's is the negation (:400-402) */
    uint8_t*       terminated;  /* [n] done flag of the sampled transition (:403) */
    uint8_t*       truncated;   /* [n] timestep >= max_steps (:404) */
    uint8_t*       prob_code;   /* [n] code of the sampled transition's probability, see soccer_prob_table */
    uint16_t*      final_obs;   /* [n] observation BEFORE auto-reset (equals obs when none fired) */
    int8_t*        last_return; /* [n] A's return of the lane's most recently finished episode; written
                                   only on the step an episode ends (terminated or truncated) */
    /* ABI 2: what a gym-style caller reads every step, written by the same launch instead of by cast kernels of its own
     * (device pointers only; NULL = skipped) */
    float*         reward_a_f32; /* [n] player A's reward as float32 (-1.0 / 0.0 / +1.0), :400 */
    float*         reward_b_f32; /* [n] player B's reward as float32 = 0 - A's (:401-402); zeros are +0.0 */
    uint8_t*       finished;     /* [n] terminated | truncated (the vector env's infos["_final_observation"]) */
} soccer_step_args;

/* batched_rollout arguments: T fused steps with state held in registers.
 * Step j of the rollout is bit-identical to the j-th of T successive batched_step calls. */
typedef struct soccer_rollout_args {
    int32_t        n_steps;     /* T >= 1 */
    int32_t        sample_actions; /* 0: read act_a/act_b; 1: draw uniform-random actions in-kernel
                                      from the lane's word w of the purpose-1 Philox block, 15 bits per
                                      player: da = w & 0x7fff, db = (w >> 16) & 0x7fff;
                                      a = (da*5)>>15, b = (db*5)>>15 */
    const int8_t*  act_a;       /* [T][n] (row stride act_stride) or NULL when sample_actions */
    const int8_t*  act_b;
    int64_t        act_stride;  /* elements between consecutive steps (>= n) */
    uint16_t*      obs;         /* [T][n] trajectories (row stride out_stride) or NULL */
    int8_t*        reward;
    uint8_t*       terminated;
    uint8_t*       truncated;
    int64_t        out_stride;
    int32_t*       return_sum;  /* [n] += sum of A's rewards over the T steps, or NULL */
    int32_t*       episode_count; /* [n] += episodes finished during the T steps, or NULL */
    /* sample_actions only: mixed (stochastic) policies, DEVICE uint16[n_states][4].  Row s holds
     * floor(32768 * cumulative probability) of actions 0..3 at observation s (values 0..32768; action 4
     * takes the rest); the action is the number of thresholds <= the player's 15-bit draw.  NULL:
     * uniform.  This is the self-play rollout of BASELINE config 5. */
    const uint16_t* mix_a;
    const uint16_t* mix_b;
} soccer_rollout_args;

/* batched_rollout_ex: per-step trajectories beyond the four result streams — what gym's vector convention reports next to them
 * every step.  [T][n] with the args' out_stride; NULL = skipped.  With both NULL (or extra == NULL) the call IS batched_rollout. */
typedef struct soccer_rollout_extra {
    uint16_t*      final_obs;   /* observation BEFORE any auto-reset of that step (== obs where none fired); goal tuples -> 0 (:493-494) */
    uint8_t*       prob_code;   /* code of the sampled transition's probability, see soccer_prob_table (info['p'], :405) */
} soccer_rollout_extra;

/* ---- lifetime ------------------------------------------------------------------------- */
int soccer_abi_version(void);
int soccer_device_count(int* count);
/* replaces SoccerSimultaneousEnv.__init__ (:35-144): validates like its asserts (:45-46),
 * derives goal rows/cols (:60-61), classifies and numbers the state tuples (:63-109), builds the
 * ISD (:146-165) and the move/bounds table (:364-373), uploads them, allocates the SoA state.
 * Lanes start in the "needs reset" condition (:140). */
int soccer_create(const soccer_config* cfg, soccer_handle** out);
int soccer_destroy(soccer_handle* h);
const char* soccer_last_error(const soccer_handle* h);
/* np_random.seed(seed) (:411-412): re-key Philox and restart the tick counter at 0. */
int soccer_seed(soccer_handle* h, uint64_t seed);
int soccer_sync(soccer_handle* h);

/* ---- the hot path --------------------------------------------------------------------- */
/* reset (:410-424) for all lanes (mask NULL) or the lanes with mask[i] != 0.
 * u_reset NULL: the lane's Philox word.  obs (nullable) receives every lane's current observation. */
int batched_reset(soccer_handle* h, const uint8_t* mask, const double* u_reset, uint16_t* obs);
/* step (:375-408) for all lanes with per-lane Philox randomness. */
int batched_step(soccer_handle* h, const int8_t* act_a, const int8_t* act_b,
                 uint16_t* obs, int8_t* reward, uint8_t* terminated, uint8_t* truncated,
                 uint8_t* prob_code);
int batched_step_ex(soccer_handle* h, const soccer_step_args* args);
int batched_rollout(soccer_handle* h, const soccer_rollout_args* args);
int batched_rollout_ex(soccer_handle* h, const soccer_rollout_args* args, const soccer_rollout_extra* extra);

/* Host-pointer variants for small batches and numpy callers (the single-env facade): identical
 * semantics, but every array pointer is HOST memory.  The call stages inputs through one pinned
 * block (one copy in, one kernel, one copy out) and returns when the results are in the caller's
 * arrays.  last_return must be NULL. */
int batched_step_host(soccer_handle* h, const soccer_step_args* host_args);
int batched_reset_host(soccer_handle* h, const uint8_t* mask, const double* u_reset, uint16_t* obs);

/* Zero-copy form of the above: the caller fills the input arrays of the handle's pinned staging block
 * (soccer_staging gives their HOST addresses, n_lanes elements each, valid for the handle's lifetime),
 * calls batched_step_staged / batched_reset_staged with the SOCCER_STAGE_* bits of the inputs it filled,
 * and reads the results from the block's output arrays (overwritten by the next staged call). */
#define SOCCER_STAGE_ACT_A   1u
#define SOCCER_STAGE_ACT_B   2u
#define SOCCER_STAGE_U_STEP  4u
#define SOCCER_STAGE_U_RESET 8u
#define SOCCER_STAGE_MASK    16u
typedef struct soccer_staging_view {
    int8_t* act_a; int8_t* act_b; uint8_t* mask; double* u_step; double* u_reset;          /* inputs  */
    uint16_t* obs; uint16_t* final_obs; int8_t* reward; uint8_t* terminated; uint8_t* truncated;
    uint8_t* prob_code;                                                                      /* outputs */
} soccer_staging_view;
int soccer_staging(soccer_handle* h, soccer_staging_view* view);
int batched_step_staged(soccer_handle* h, uint32_t inputs);
int batched_reset_staged(soccer_handle* h, uint32_t inputs);

/* Single-agent mode (reference :54-56, :187-188, :266-279): `player` (0 = player_a, 1 = player_b) follows
 * a fixed policy — HOST int8[n_states], action per observation index — looked up with the observation
 * of the CURRENT tuple before every step; that player's action stream may then be NULL.  Only one
 * side may have a policy (:38).  policy NULL clears it.  Rewards stay player A's (+1 A scores); a
 * learner-B host negates them as the reference's table does (:243-244). */
int soccer_set_policy(soccer_handle* h, int32_t player, const int8_t* policy_host, int32_t n_states);

/* ---- one environment, lowest latency (the single-env facade; reference :375-424) --------- */
/* For handles with n_lanes == 1.  The current tuple, the actions and the uniform go in BY VALUE (kernel
 * arguments), the result comes back through a host-mapped record that the kernel writes with one 16-byte
 * store and the call polls: the GPU reads no host memory and the host enters no stream synchronisation
 * (~2x lower latency than batched_step_host on a mapped handle, tools/labs/latency_lab.hip).
 * soccer_step_scalar: in  = row_a..col_b, poss, t, act_a, act_b (ignored for a side with a fixed policy),
 *                           u_step, u_reset (used only with SOCCER_F_AUTORESET);
 *                     out = the next tuple / poss / t / needs_reset in the same fields, obs, reward (player A's),
 *                           terminated, truncated, prob_code.  needs_reset != 0 on input is the reference's
 *                           "Please reset the environment before taking a step" (SOCCER_E_INVALID, :376);
 *                           tuples outside the pitch or unreachable are SOCCER_E_INVALID.
 * soccer_reset_scalar: in = u_reset (the ISD draw, :414); out = tuple, poss, t = 0, obs.
 * Both consume one tick and leave the lane's resident state equal to the returned one.
 * The record: {seq, results, next tuple, flags | t << 8 | check << 16 | seq << 24} written by ONE global_store_dwordx4
 * to host-mapped memory.  The call accepts it only when word 0 == seq, the top byte of word 3 == seq's low byte AND
 * the check byte of word 3 equals the byte-sum of words 1 and 2 of the same call — so a record whose four dwords
 * did not all land (a torn 16-byte write; not observed on gfx950 / PCIe, but not an architectural promise) is
 * never taken for a complete one: the poll simply continues until they have. */
typedef struct soccer_scalar_io {
    int8_t row_a, col_a, row_b, col_b;
    uint8_t poss, needs_reset, t;
    int8_t act_a, act_b;
    int8_t reward;
    uint8_t terminated, truncated, prob_code;
    uint8_t pad_;
    uint16_t obs;
    double u_step, u_reset;
} soccer_scalar_io;
int soccer_step_scalar(soccer_handle* h, soccer_scalar_io* io);
int soccer_reset_scalar(soccer_handle* h, soccer_scalar_io* io);

/* SOCCER_F_HOST_MAPPED handles only: HOST address of the six state streams (row_a, col_a, row_b, col_b,
 * poss|needs_reset<<1, t; `stride` bytes apart).  Valid to read/write whenever the stream is idle
 * (after a *_host call or soccer_sync); writes bypass the tuple validation of soccer_set_state. */
int soccer_host_view(soccer_handle* h, uint8_t** state, uint64_t* stride);

/* ---- state injection / readback (`env.state = tuple`, tests/test_deterministic...py:43) -- */
/* HOST pointers of n_lanes elements; any pointer may be NULL (field left unchanged / not read).
 * These synchronise the stream. */
int soccer_set_state(soccer_handle* h, const int8_t* row_a, const int8_t* col_a,
                     const int8_t* row_b, const int8_t* col_b, const uint8_t* poss,
                     const uint8_t* t, const uint8_t* needs_reset);
int soccer_get_state(soccer_handle* h, int8_t* row_a, int8_t* col_a, int8_t* row_b, int8_t* col_b,
                     uint8_t* poss, uint8_t* t, uint8_t* needs_reset);

/* ---- rule tables (what the reference exposes as state_space / goal_states / isd) --------- */
/* n_states = nS (:64,:105-106) incl. the terminal index 0; lut_len = H*(W+2)*H*(W+2)*2 */
int soccer_dims(const soccer_handle* h, int32_t* n_states, int32_t* lut_len,
                int32_t* n_isd, int32_t* internal_width);
/* HOST outputs. lut[(((ra*W+ca)*H+rb)*W+cb)*2+p] = observation index, 0 for goal tuples,
 * 0xFFFF for unreachable tuples (:73-88); goal_value = +1/-1 for goal tuples (:94-102), else 0;
 * isd_states[n_isd][5] (:146-165). Any pointer may be NULL. */
int soccer_get_tables(const soccer_handle* h, uint16_t* lut, int8_t* goal_value, int8_t* isd_states);
/* The reference's transition relation P_readable (:167-293), computed on the device by the same rule
 * functions the step kernels use.  HOST outputs, key = lut index * 25 + action_a * 5 + action_b:
 *   count[key]           entries in the list (1..36), -1 for unreachable tuples (the reference has no key)
 *   prob/next_flat/reward/done[key*36 + k]   k-th entry in the reference's list order: probability
 *   (float64, weight * outcome probability, :241), lut index of the next tuple, player A's reward,
 *   done (:235-240). */
int soccer_enumerate_transitions(soccer_handle* h, int32_t* count, double* prob, int32_t* next_flat,
                                 int8_t* reward, uint8_t* done);
/* ---- planners (reference gym_soccer/utils/planners.py) ---------------------------------- */
/* All of them need a single-agent handle (exactly one side has a policy: soccer_set_policy) and work on the
 * learner's tables P[s][a] / Pmat / Rmat exactly as the reference's constructor builds them (:167-293), here
 * assembled from soccer_enumerate_transitions and cached on the handle until the policy changes.  One
 * single-workgroup kernel runs a whole planner; float64 throughout.  Inputs and outputs are HOST arrays
 * (V[n_states], Q[n_states*5], pi[n_states] int32); any output may be NULL.  max_sweeps bounds the total
 * number of sweeps over the state space (SOCCER_E_STATE if it is reached; the outputs hold the last iterate).
 *
 * The list-based planners evaluate  Q[s][a] += prob * (reward + discount_factor * V[next] * (not done))  in
 * list order like the reference: values, greedy policies and iteration counts are the reference's BIT FOR BIT.
 *   soccer_value_iteration     planners.py:4-18   sweeps until max|V - max_a Q| < theta; V is the pre-update
 *                                                 iterate (as the reference returns it), pi the first argmax
 *   soccer_policy_evaluation   planners.py:20-31  V of the deterministic policy pi, from zeros
 *   soccer_policy_improvement  planners.py:33-41  Q from V, new_pi = first argmax
 *   soccer_policy_iteration    planners.py:43-53  from pi0 (the reference draws it with np.random.choice) until
 *                                                 the greedy policy stops changing; V belongs to the last evaluation
 * The dense planners follow the reference's Pmat/Rmat algebra (r + discount_factor * Pmat @ v) with a sequential
 * dot; numpy's BLAS dot associates differently, so they agree with the reference to rounding (~1e-15 relative):
 *   soccer_policy_eval_dense            planners.py:55-70  at most k sweeps of a stochastic policy[n_states*5]
 *                                                          from init (NULL = zeros), stops when the change < theta
 *   soccer_modified_policy_iteration    planners.py:73-87  greedy step + k evaluation sweeps, stops when
 *                                                          |v - greedy_v| <= theta*(1-discount)/(2*discount); V = greedy_v */
int soccer_value_iteration(soccer_handle* h, double theta, double discount_factor, int32_t max_sweeps,
                           double* V, double* Q, int32_t* pi, int32_t* iterations);
int soccer_policy_evaluation(soccer_handle* h, const int32_t* pi, double theta, double discount_factor,
                             int32_t max_sweeps, double* V, int32_t* sweeps);
int soccer_policy_improvement(soccer_handle* h, const double* V, double discount_factor, double* Q, int32_t* new_pi);
int soccer_policy_iteration(soccer_handle* h, const int32_t* pi0, double theta, double discount_factor,
                            int32_t max_sweeps, double* V, double* Q, int32_t* pi, int32_t* iterations);
int soccer_policy_eval_dense(soccer_handle* h, const double* policy, int32_t k, double theta, double discount_factor,
                             int32_t max_sweeps, const double* init, double* v, int32_t* sweeps);
int soccer_modified_policy_iteration(soccer_handle* h, int32_t k, double theta, double discount_factor,
                                     int32_t max_sweeps, double* V, double* Q, int32_t* pi, int32_t* iterations);
/* HOST output: prob[c*3+k] = slip-combination weight c (0: no slip, 1: B slips, 2: A slips,
 * 3: both; :211-222, evaluated left to right in float64) times outcome probability 1, 0.5, 0.25
 * (k = 0,1,2; :326-360).  prob_code values index this table (:241). */
int soccer_prob_table(const soccer_handle* h, double prob[12]);

/* ---- episode statistics ------------------------------------------------------------------ */
/* hist[0..2] = episodes finished with A's return -1, 0, +1 since create / soccer_reset_stats, counted by
 * batched_rollout and — on handles created with SOCCER_F_STEP_STATS — by batched_step;
 * misuse = sticky flags: SOCCER_MISUSE_FROZEN if any lane was stepped while it needed reset (the reference's
 * assert, :376; such lanes are left untouched), SOCCER_MISUSE_ACTION if any action byte on a device-pointer
 * path was outside 0..4 (the reference raises IndexError, :393; see "action bytes" above).
 * Synchronises the stream. HOST outputs. */
#define SOCCER_MISUSE_FROZEN 1u
#define SOCCER_MISUSE_ACTION 2u
int soccer_get_stats(soccer_handle* h, uint64_t hist[3], uint64_t* misuse);
/* the misuse flags as they stand, WITHOUT synchronising (the kernels write them to host-mapped memory): what the
 * launches that have completed so far have raised. */
uint32_t soccer_peek_misuse(const soccer_handle* h);
int soccer_reset_stats(soccer_handle* h);
uint64_t soccer_tick(const soccer_handle* h);
/* Checkpoint / resume.  The reference keeps (state tuple, timestep, needs_reset, RandomState) per env; here
 * the six state streams (soccer_get_state / soccer_set_state) plus (seed, tick) determine every later
 * result of a handle, on any device count. */
uint64_t soccer_get_seed(const soccer_handle* h);
int soccer_set_tick(soccer_handle* h, uint64_t tick);

/* Episode returns from [n_steps][n_lanes] result trajectories — what n_steps batched_step calls or one batched_rollout
 * wrote (DEVICE pointers, row stride `stride` elements): one pass over the three streams.
 *   last_return[i]   (nullable, device int8[n])  player A's return of lane i's most recently finished episode = the reward
 *                    of the last step at which terminated | truncated was set (:235-240, :400-404); 0 if none finished
 *   episode_count[i] (nullable, device int32[n]) episodes lane i finished during the n_steps steps
 *   hist             (nullable, HOST uint64[3])  all finished episodes by A's return -1, 0, +1; when given the call synchronises
 * This is the per-lane value BASELINE configs[3] gathers across GPUs (soccer_comm_all_gather). */
int soccer_trajectory_returns(soccer_handle* h, int32_t n_steps, const int8_t* reward, const uint8_t* terminated,
                              const uint8_t* truncated, int64_t stride, int8_t* last_return,
                              int32_t* episode_count, uint64_t hist[3]);

/* ---- multi-GPU: RCCL over xGMI (the reference has no counterpart) ------------------------------------
 * One process and one handle per GPU; lanes never interact, so stepping needs NO collective.  The only exchange is after a
 * run: an all-gather of per-lane episode returns into global lane order and small reductions (the 3-bin histogram, clocks).
 * librccl is resolved at run time (dlopen), so a single-GPU process never loads it.
 *   soccer_comm_unique_id   rank 0 creates the 128-byte id (ncclGetUniqueId) and hands it to every rank by any host channel
 *                           (gym_soccer_littman94_amd/comm.py: a file next to the launcher, or the caller's own)
 *   soccer_comm_init        ncclCommInitRank on the handle's device; collective over all `world` ranks
 *   soccer_comm_all_gather  recv[r * bytes_per_rank ...] = rank r's send[0 .. bytes_per_rank) (DEVICE pointers; enqueued on
 *                           the handle's stream, asynchronous): equal contiguous shards land in global lane order
 *   soccer_comm_sum_u64 / soccer_comm_max_f64 / soccer_comm_barrier   1..8 HOST values reduced over the ranks in place;
 *                           these synchronise the stream (the barrier is a one-element sum) */
#define SOCCER_COMM_ID_BYTES 128
int soccer_comm_unique_id(uint8_t id[SOCCER_COMM_ID_BYTES]);
int soccer_comm_init(soccer_handle* h, int32_t world, int32_t rank, const uint8_t id[SOCCER_COMM_ID_BYTES]);
int soccer_comm_destroy(soccer_handle* h);
int soccer_comm_all_gather(soccer_handle* h, const void* send, void* recv, uint64_t bytes_per_rank);
int soccer_comm_sum_u64(soccer_handle* h, uint64_t* values, int32_t count);
int soccer_comm_max_f64(soccer_handle* h, double* values, int32_t count);
int soccer_comm_barrier(soccer_handle* h);

/* ---- device memory + timing helpers (so a host without torch can drive the library) ------ */
int soccer_malloc(soccer_handle* h, size_t bytes, void** dptr);
int soccer_free(soccer_handle* h, void* dptr);
int soccer_memcpy_h2d(soccer_handle* h, void* dst, const void* src, size_t bytes); /* sync */
int soccer_memcpy_d2h(soccer_handle* h, void* dst, const void* src, size_t bytes); /* sync */
int soccer_memset(soccer_handle* h, void* dst, int value, size_t bytes);           /* async */
/* HIP events on the handle's stream.  soccer_timer_start / soccer_timer_mark may also be called during a graph
 * capture: the two event records then become nodes of the graph, and after a replay soccer_timer_read returns the
 * device time between them — first captured kernel's start to last one's end, without the replay's start-up latency. */
int soccer_timer_start(soccer_handle* h);
int soccer_timer_mark(soccer_handle* h);                       /* records the closing event */
int soccer_timer_read(soccer_handle* h, float* elapsed_ms);    /* waits for the closing event */
int soccer_timer_stop(soccer_handle* h, float* elapsed_ms);    /* = mark + read */
/* Device clock stamps.  soccer_stamp enqueues — or, inside a capture, records as a graph node — a one-thread kernel that
 * stores the device's constant-rate wall clock into slot `slot` of a host-mapped block; soccer_stamps_read copies slots
 * as they stand, WITHOUT synchronising (a slot reads 0 until its kernel has run, provided soccer_stamps_clear zeroed it
 * while nothing was writing it — but a line the host has just written costs the device a coherence round trip to write:
 * prefer reading the slots after a synchronisation and never clearing them) and reports the clock rate.  Every slot has a
 * 64-byte line of its own.  A captured soccer_timer_start / _mark pair is stamps 0 and 1, and after a single replay
 * soccer_timer_read just watches slot 1 change from the host: no runtime call sits between the end of the region on the
 * device and the host noticing it. */
#define SOCCER_STAMP_SLOTS 256
int soccer_stamp(soccer_handle* h, int32_t slot);
int soccer_stamps_clear(soccer_handle* h, int32_t first, int32_t count);
int soccer_stamps_read(soccer_handle* h, int32_t first, int32_t count, uint64_t* ticks, int32_t* khz);

/* ---- hipGraph capture of a sequence of batched_* calls ------------------------------------ */
/* Calls between begin and end are recorded instead of executed.  Any number of batched_* calls may be recorded (the tick
 * lives in device memory in two alternating slots so that a replay advances it; a sequence with an ODD number of calls
 * gets one extra one-thread node that moves it back to the slot a replay starts from, ~1.5 us per replay). */
int soccer_graph_begin(soccer_handle* h);
int soccer_graph_end(soccer_handle* h, soccer_graph** out);
int soccer_graph_launch(soccer_handle* h, soccer_graph* g, int32_t replays);
int soccer_graph_destroy(soccer_handle* h, soccer_graph* g);

#ifdef __cplusplus
}
#endif
#endif /* SOCCER_HIP_H */
