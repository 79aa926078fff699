// soccer_kernels.hpp — gfx950 (CDNA4 / MI355X) kernels for the batched step / reset / rollout.
//
// What is evaluated per lane, and where the reference states it
// (gym_soccer/envs/soccer_simultaneous_env.py):
//   cell move via the move/bounds table ................ _next_cell          :364-373
//   ordered 5-way collision resolution ................. _get_next_state     :296-362
//   nine slip combinations, float64 weights, zero-skip . :202-227, :241
//   done / reward ...................................... :235-240
//   outcome selection .................................. categorical_sample  :395 (gym 0.26.2)
//   bookkeeping (timestep, truncation, needs_reset) .... :396-406
//   reset from the initial state distribution .......... :410-424
//
// Execution shape: wave64; each thread owns 4 (rollout: E = 1/4/8) consecutive lanes (environments) so
// that every SoA byte stream is read and written with one dword (dwordx2) per thread — 256/512 B per
// wave instruction.  One Philox4x32-10 block serves four consecutive global lanes.
//   step_kernel_hot / step_kernel : one step per launch; rolled lane loop (small code: a launch starts
//                                   with a cold instruction cache); rule tables read through L1/L2
//   rollout_kernel                : T steps per launch with the state in registers; lane loop unrolled;
//                                   rule tables staged in LDS once per workgroup
//   reset_kernel, enumerate_kernel: reset from the ISD; the reference's transition table
// No MFMA: there is no contraction on this path.  By bytes the kernels are HBM-bound (19 B per
// env-step); measured, a single-step launch at 2^20 lanes is bound by launch latency plus ~130 vector
// instructions per lane that do not overlap its load and store phases.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "soccer_swar.hpp"

namespace soccer {

constexpr int kBlock = 256;
#ifndef SOCCER_HOT_UNROLL
#define SOCCER_HOT_UNROLL 4          // lane-loop unroll of step_kernel_hot<false>: see DESIGN.md section 4
#endif
#ifndef SOCCER_HOT_UNROLL_SLIP
#define SOCCER_HOT_UNROLL_SLIP 4
#endif
constexpr int kHistSlots = 16384;        // minimum: >= waves of the largest CAPPED grid (8 blocks x 4 waves x 512 CUs); a handle
                                         // whose byte-parallel step launches more waves than this (n_lanes > 2^22) gets more slots
constexpr int kHistStride = 4;           // u64 per slot: return -1, 0, +1, pad
constexpr int kIsdWords = 16;            // LDS: 4 ISD entries x (A, B, poss|obs<<16, pad)

struct KernelParams {
    // resident state: six byte streams back to back, `state_stride` bytes apart, in the order
    // row_a, col_a, row_b, col_b, poss (bit0 possession, bit1 needs_reset), t
    uint8_t* state;
    unsigned long long state_stride;
    // rule tables in global memory (staged to LDS)
    const uint16_t* lut;                  // [lut_len] observation index per state tuple
    const uint32_t* next_cell;            // [2][H*W][5]: (has_ball, cell, move) -> (row<<8|col)<<16 | (row*W+col) reached
    const uint32_t* isd;                  // [kIsdWords]
    // single-agent mode: the fixed side's action per observation index (int8[nS]), or nullptr
    const int8_t* policy_a; const int8_t* policy_b;
    // randomness
    const unsigned long long* tick_in;    // device tick slot read by this launch
    unsigned long long* tick_out;         // slot written (tick_in + ticks consumed)
    uint32_t key0, key1;
    unsigned long long lane_offset;
    // statistics
    unsigned long long* hist;             // [hist_mask + 1][4]: one private slot per wave of the grid, bins 0..2
    uint32_t hist_mask;                   // slots - 1 (a power of two >= the waves of the largest grid that counts)
    unsigned int* misuse;                 // sticky flag
    // geometry / constants
    unsigned long long first;             // first lane (within the handle) this launch covers
    unsigned long long n;                 // number of lanes this launch covers
    int32_t W, HW, HW5, nc_len, lut_len;   // HW5 = 5*H*W
    int32_t max_steps;
    uint32_t autoreset;
    uint32_t step_stats;                  // batched_step feeds the episode histogram (SOCCER_F_STEP_STATS)
    uint32_t isd_shift;                   // 2 - log2(n_isd): index = two random bits >> isd_shift
    double w[4];                          // slip-combination weights c0..c3 (:211-222)
    // slip fast path: cumulative weight after each ACTIVE (non-zero) combination in reference order
    // (+inf beyond), their count, and their combination ids packed 4 bits each
    double B[9]; uint32_t nb; unsigned long long act_pack;
    // integer form of the same decision for draws that come from a Philox word (u = (m + 1/2) * 2^-30, m < 2^30):
    // running sum t <= u  <=>  m >= ceil(t * 2^30 - 1/2).  Used only when the host has checked, by walking every list
    // shape, that this integer is the same for every shape at every entry position (soccer_slip.hpp): then it is the
    // reference's decision for every draw and the float64 walk is never needed (slip_int = 1).
    uint32_t CB[9]; uint32_t slip_int;    // 0 float64 only, 1 integer only
    const uint4* sub;                     // [9] per active combination: { t1 (2 outcomes), t1, t2, t3 (4 outcomes) }, as integers
};

struct StepIO {
    const int8_t* act_a; const int8_t* act_b;
    const double* u_step; const double* u_reset;
    uint16_t* obs; int8_t* reward; uint8_t* terminated; uint8_t* truncated;
    uint8_t* prob_code; uint16_t* final_obs; int8_t* last_return;
    float* reward_a_f32; float* reward_b_f32; uint8_t* finished;      // the gym surface's float rewards / terminated | truncated
    // the 4-lane groups step_kernel_swar<.., SLIPM = 3, ..> left to the exact float64 walk (a caller's uniform within 2^-40 of a
    // nominal threshold): the per-lane kernel then steps exactly these groups and zeroes the count; nullptr: every group
    const uint32_t* worklist; uint32_t* work_count;
};

struct ResetIO {
    const uint8_t* mask; const double* u_reset; uint16_t* obs;
};

struct RolloutIO {
    int32_t n_steps; int32_t sample_actions;
    const uint16_t* mix_a; const uint16_t* mix_b;   // [nS][4] cumulative thresholds (0..32768) of a mixed policy, or nullptr
    const int8_t* act_a; const int8_t* act_b; long long act_stride;
    uint16_t* obs; int8_t* reward; uint8_t* terminated; uint8_t* truncated; long long out_stride;
    int32_t* return_sum; int32_t* episode_count;
    uint16_t* final_obs; uint8_t* prob_code;      // batched_rollout_ex: per-step [T][n] trajectories (row stride out_stride) or nullptr
};

// ---- Philox4x32-10 (Salmon et al. 2011; Random123 constants) ---------------------------------
struct Philox4 { uint32_t w[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // hi ^ counter ^ key as ONE v_bitop3_b32 (truth table 0x96: three-way xor); the compiler emits two v_xor_b32 for the
        // expression.  -20 of the ~60 vector instructions of a block: +4 % on the fused rollout, +7 % on the self-play rollout
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// the block shared by global lanes 4q .. 4q+3 at `tick`; purpose 0 = step/reset, 1 = sampled actions.
// Callers pass `tick >> 3` for the step/reset block of a slip_prob == 0 handle (one block serves eight ticks).
__device__ __forceinline__ Philox4 lane_block(const KernelParams& P, unsigned long long q,
                                              unsigned long long tick, uint32_t purpose) {
#ifdef SOCCER_LAB_NO_PHILOX     // tools/labs/kernel_lab.hip ablation build only
    return Philox4{{(uint32_t)q * 2654435761u, (uint32_t)tick, (uint32_t)q ^ 0x9E3779B9u, purpose}};
#endif
    return philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)tick,
                         (uint32_t)(tick >> 32) | (purpose << 31), P.key0, P.key1);
}

// One lane's randomness for a step: the uniform as float64 (slip lists), floor(4u) (lists whose
// probabilities are dyadic: slip_prob == 0) and two independent bits for the ISD draw.
struct Draw { double u; uint32_t top2; uint32_t reset2; uint32_t m; };   // m = w >> 2 for word draws

// The RNG convention of include/soccer_hip.h, per lane:
//   SLIP (slip_prob > 0): lane word w of the tick's own block: u = ((w >> 2) + 1/2) * 2^-30, reset bits = w & 3
//   otherwise: the lane's word of the block of tick >> 3; this tick's nibble is number (tick & 7) ^ 1 from the least
//   significant end: its two high bits are floor(4u) (u = (that + 1/2) / 4), its two low bits the reset draw
template <bool SLIP>
__device__ __forceinline__ Draw draw_from_word(uint32_t w, unsigned long long tick) {
    if (SLIP) return Draw{((double)(w >> 2) + 0.5) * 0x1.0p-30, w >> 30, w & 3u, w >> 2};
    const uint32_t nib = (w >> (4u * (((uint32_t)tick & 7u) ^ 1u))) & 15u;
    return Draw{((double)(nib >> 2) + 0.5) * 0.25, nib >> 2, nib & 3u, 0u};
}
// the tick a step/reset block is keyed by
template <bool SLIP>
__device__ __forceinline__ unsigned long long block_tick(unsigned long long tick) { return SLIP ? tick : tick >> 3; }
// A caller-supplied uniform.  Values outside [0,1) (and NaN) make every running sum compare
// "not greater", which categorical_sample resolves to index 0 — same as u = 0.
__device__ __forceinline__ double sane_uniform(double u) { return ((u >= 0.0) && (u < 1.0)) ? u : 0.0; }
// The same for the float64 walk over a slip list (slip_prob > 0), whose total may round to 1 + 2^-52: there the reference's
// argmax(cumsum(p) > u) picks the LAST entry for u == 1.0 (np_random.random() never returns it, but a caller's array may hold
// it), so the upper side is left to the walk — which answers index 0 when no running sum exceeds u — and only negative values
// and NaN are folded onto 0 (every first entry is positive: index 0 either way).
__device__ __forceinline__ double sane_uniform_walk(double u) { return (u >= 0.0) ? u : 0.0; }

// ---- LDS-resident rule tables ------------------------------------------------------------------
struct Tables {
    const uint16_t* lut;    // observation index per tuple
    const uint32_t* nc;     // move/bounds table (LDS in the rollout/reset kernels, global in the step kernel)
    const uint32_t* isd;    // initial-state entries
};

// LDS layout (dwords): [0, kIsdWords) ISD, [kIsdWords, kIsdWords + nc_len) move/bounds table, then
// (LUT_LDS) the observation table
template <bool LUT_LDS>
__device__ __forceinline__ Tables stage_tables(const KernelParams& P, uint32_t* smem) {
    if (threadIdx.x < kIsdWords) smem[threadIdx.x] = P.isd[threadIdx.x];
    uint32_t* nc = smem + kIsdWords;
    for (int i = threadIdx.x; i < P.nc_len; i += kBlock) nc[i] = P.next_cell[i];
    Tables T;
    T.isd = smem; T.nc = nc;
    if (LUT_LDS) {
        uint32_t* lut = nc + P.nc_len;
        const int lut_dw = P.lut_len >> 1;                // lut_len is even
        const uint32_t* src = reinterpret_cast<const uint32_t*>(P.lut);
        for (int i = threadIdx.x; i < lut_dw; i += kBlock) lut[i] = src[i];
        T.lut = reinterpret_cast<const uint16_t*>(lut);
    } else {
        T.lut = P.lut;                                    // global (L1/L2)
    }
    __syncthreads();
    return T;
}

// ---- one lane's state in registers -------------------------------------------------------------
// A player's position is carried as one word: low 16 bits the cell id row*W+col (table / LUT index),
// high 16 bits (row<<8 | col) (what the SoA streams store).  Equality of words == equality of cells.
struct Lane {
    uint32_t A, B;      // positions
    uint32_t p;         // possession 0/1
    uint32_t need;      // needs_reset 0/1
    uint32_t t;
};

struct StepResult {
    uint32_t obs, final_obs;
    int32_t reward;
    uint32_t term, trunc, code;
    uint32_t finished;      // episode ended at this step (before any auto-reset)
};

// a*b + c for operands below 2^24: one full-rate v_mad_u32_u24 (a 32-bit a*b+c is a quarter-rate op)
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }

__device__ __forceinline__ uint32_t make_pos(uint32_t row, uint32_t col, int W) {
    return mad24(row, (uint32_t)W, col) | (((row << 8) | col) << 16);
}
__device__ __forceinline__ uint32_t cell_of(uint32_t pos) { return pos & 0xffffu; }
__device__ __forceinline__ uint32_t col_of(uint32_t pos) { return (pos >> 16) & 0xffu; }

// Observation index of a tuple: one gather from the per-tuple table (goal tuples hold 0, :493-494).
// A closed form exists — obs = 1 + 2*(iA*(NI-1) + iB - (iB > iA)) + p over interior-cell indices, see
// Rules::build, which checks the table against it — but the kernels are VALU-bound and the gather
// rides the memory pipe: the ~12 extra vector instructions measured 4 % slower per step.
__device__ __forceinline__ uint32_t obs_of(const Tables& T, const KernelParams& P, uint32_t A, uint32_t B, uint32_t p) {
#ifdef SOCCER_LAB_NO_GATHER
    return (mad24(cell_of(A), (uint32_t)P.HW, cell_of(B)) << 1) | p;
#endif
    return T.lut[(mad24(cell_of(A), (uint32_t)P.HW, cell_of(B)) << 1) | p];
}

// slipped move of an action: variant 0 intended, 1/2 the two orthogonals (:205-206)
//   NOOP->NOOP,NOOP  NORTH->EAST,WEST  SOUTH->WEST,EAST  EAST->SOUTH,NORTH  WEST->NORTH,SOUTH
__device__ __forceinline__ uint32_t slip_move(uint32_t a, int variant) {
    if (variant == 0) return a;
    const uint32_t tab = variant == 1 ? 0x12430u : 0x21340u;
    return (tab >> (4u * a)) & 7u;
}

enum : uint32_t { K_MOVE = 0, K_FLIP = 1, K_COIN = 2, K_FOUR = 3 };

struct Resolved { uint32_t kind, nA, nB; };

// the cell a player reaches with one move (_next_cell :364-373, via the move/bounds table)
__device__ __forceinline__ uint32_t moved(const Tables& T, const KernelParams& P, uint32_t pos, uint32_t has_ball, uint32_t mv) {
#ifdef SOCCER_LAB_NO_GATHER     // tools/labs/kernel_lab.hip ablation build only
    return pos + (mv == 3u ? 0x00010001u : 0u) * (has_ball & 1u);
#endif
    return T.nc[mad24(has_ball, (uint32_t)P.HW5, mad24(cell_of(pos), 5u, mv))];
}

// _get_next_state (:296-362) for a live tuple, given the cells nA / nB the two (possibly slipped) moves
// reach.  aa/ab are the ORIGINAL actions: the reference's NOOP tests use those, not the moves.
__host__ __device__ __forceinline__ Resolved classify(uint32_t A, uint32_t B, uint32_t nA, uint32_t nB, uint32_t aa, uint32_t ab) {
    const bool e1 = nA == B, e2 = nB == A, sA = nA == A, sB = nB == B;
    const bool swap = e1 & e2;                                                         // :315-322
    const bool stander = (e1 & (ab == 0u)) | (e2 & (aa == 0u));                      // :330-331
    const bool bounce = (sA & (aa != 0u) & e2) | (sB & (ab != 0u) & e1);              // :338-339
    const bool same = nA == nB;                                                        // :347
    const uint32_t kind = (swap | (bounce & !stander)) ? (uint32_t)K_COIN
                        : stander ? (uint32_t)K_FLIP : same ? (uint32_t)K_FOUR : (uint32_t)K_MOVE;
    return Resolved{kind, nA, nB};
}

struct Outcome { uint32_t A, B, p, kcode; };

// outcome k of a resolved collision, in the reference's list order
// (:326-327, :335, :343-344, :352-356, :360); branch-free selects on values
__device__ __forceinline__ Outcome pick(uint32_t A, uint32_t B, uint32_t p, const Resolved& R, uint32_t k) {
    const bool mv = R.kind == K_MOVE, fl = R.kind == K_FLIP, four = R.kind == K_FOUR;
    const bool a_moves = mv | (four & (k >= 2u));
    const bool b_moves = mv | (four & (k < 2u));
    Outcome o;
    o.A = a_moves ? R.nA : A;
    o.B = b_moves ? R.nB : B;
    o.p = mv ? p : (fl ? (p ^ 1u) : (k & 1u));
    o.kcode = (mv | fl) ? 0u : (four ? 2u : 1u);
    return o;
}

// Returns true when the lane was stepped while it needed a reset (left untouched; :376).
// WORD: the draw is known to come from a Philox word (d.m valid), which allows the integer slip decision.
// INT_ONLY: the caller has checked P.slip_int on the host; the float64 decision is compiled out.
template <bool SLIP, bool WORD = false, bool INT_ONLY = false>
__device__ __forceinline__ bool lane_step(const Tables& T, const KernelParams& P, Lane& Lref,
                                          uint32_t aa, uint32_t ab, const Draw& d, StepResult& out) {
    const uint32_t A = Lref.A, B = Lref.B, p = Lref.p, t = Lref.t;
    const uint32_t Wm1 = (uint32_t)(P.W - 1);
    const uint32_t carrier_col = col_of(p ? B : A);
    const bool in_goal = (carrier_col == 0u) | (carrier_col == Wm1);   // goal tuple: absorbing (:300-301)
    Outcome sel;
    uint32_t cls = 0;
    if (!SLIP) {
        // single surviving combination, weight 1.0 (:226-227): list probabilities are 1, .5/.5 or .25x4,
        // so the sampled index is floor(2u) / floor(4u)
        const Resolved R = classify(A, B, moved(T, P, A, p ^ 1u, aa), moved(T, P, B, p, ab), aa, ab);
        const uint32_t k = R.kind == K_COIN ? (d.top2 >> 1) : d.top2;
        sel = pick(A, B, p, R, k);
    } else {
        // VA / VB / CLS of the nine combinations (:209-223), 2 bits each: A's move variant, B's move variant and the
        // weight class (c0..c3).  Bit fields, not arrays: a loop the compiler leaves rolled must not turn them
        // (or the by-value weights P.w) into scratch.
        constexpr uint32_t VA2 = 0u | (0u << 2) | (0u << 4) | (1u << 6) | (2u << 8) | (1u << 10) | (1u << 12) | (2u << 14) | (2u << 16);
        constexpr uint32_t VB2 = 0u | (1u << 2) | (2u << 4) | (0u << 6) | (0u << 8) | (1u << 10) | (2u << 12) | (1u << 14) | (2u << 16);
        constexpr uint32_t CL2 = 0u | (1u << 2) | (1u << 4) | (2u << 6) | (2u << 8) | (3u << 10) | (3u << 12) | (3u << 14) | (3u << 16);
        const double w0 = P.w[0], w1 = P.w[1], w2 = P.w[2], w3 = P.w[3];
#define SOCCER_WEIGHT_OF(cl) (((cl) & 2u) ? (((cl) & 1u) ? w3 : w2) : (((cl) & 1u) ? w1 : w0))
        const bool use_int = INT_ONLY || (WORD && P.slip_int != 0u);     // wave-uniform
        if (use_int) {
            // Integer decision (see KernelParams::CB): combination = number of scaled cumulative weights <= m,
            // outcome within it = number of its scaled thresholds <= m.  No float64, no fallback — and the
            // combination is known before the move table is read, so two reads suffice.
            uint32_t idx = 0u;
#pragma unroll
            for (int i = 0; i < 9; ++i) idx += d.m >= P.CB[i] ? 1u : 0u;
            const uint32_t c_i = (uint32_t)((P.act_pack >> (4u * idx)) & 0xfull);
            const uint32_t va = (VA2 >> (2u * c_i)) & 3u, vb = (VB2 >> (2u * c_i)) & 3u;
            cls = (CL2 >> (2u * c_i)) & 3u;
            const uint32_t as = va == 0u ? aa : (((va == 1u ? 0x12430u : 0x21340u) >> (4u * aa)) & 7u);   // slip_move
            const uint32_t bs = vb == 0u ? ab : (((vb == 1u ? 0x12430u : 0x21340u) >> (4u * ab)) & 7u);
            const uint32_t cA = moved(T, P, A, p ^ 1u, as), cB = moved(T, P, B, p, bs);
            Resolved R = classify(A, B, cA, cB, aa, ab);
            R.kind = in_goal ? (uint32_t)K_MOVE : R.kind;
            const uint4 th = P.sub[idx];
            const bool two = R.kind == K_COIN, four = R.kind == K_FOUR;
            const uint32_t k = (((two & (d.m >= th.x)) | (four & (d.m >= th.y))) ? 1u : 0u) +
                               ((four & (d.m >= th.z)) ? 1u : 0u) + ((four & (d.m >= th.w)) ? 1u : 0u);
            sel = pick(A, B, p, R, k);
        } else if constexpr (!INT_ONLY) {
        // the slipped move of variant v as a run-time value (slip_move with a constant v is the same table)
        auto slip_move_rt = [](uint32_t a, uint32_t v) { return v == 0u ? a : (((v == 1u ? 0x12430u : 0x21340u) >> (4u * a)) & 7u); };
        // (1) Fast decision.  The list's running sums are, up to rounding, the cumulative weights of the
        // active combinations (P.B, summed on the host in the reference's order) plus multiples of the
        // combination's own q; the true float64 sums differ from these nominal values by < 1e-14 (at most
        // 36 additions of terms <= 1).  If u is farther than 2^-40 from every nominal threshold it can be
        // compared against, the entry found from the nominal thresholds IS the entry the sequential sum
        // finds.  Otherwise (u on or next to a threshold, or beyond the last one) the lane takes the exact
        // path (2).
        uint32_t idx = 0u; bool near_thr = false; double S = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const double b = P.B[i];                                    // wave-uniform; +inf past the last active one
            const bool ge = d.u >= b;
            idx += ge ? 1u : 0u;
            S = ge ? b : S;
            near_thr |= fabs(d.u - b) < 0x1.0p-40;
        }
        near_thr |= idx >= P.nb;
        uint32_t sel_c = (uint32_t)((P.act_pack >> (4u * idx)) & 0xfull), sel_k = 0u;
        uint32_t cA_sel, cB_sel;                                        // the cells the SELECTED combination's moves reach
        {
            // the fast decision needs the move table for ONE combination: two reads (round 4; it used to fetch the three distinct
            // moves of both players up front for the walk below, six dependent gathers on every lane: 17.1 us per launch at 2^20 lanes)
            const uint32_t va = (VA2 >> (2u * sel_c)) & 3u, vb = (VB2 >> (2u * sel_c)) & 3u, cl = (CL2 >> (2u * sel_c)) & 3u;
            const uint32_t cA = moved(T, P, A, p ^ 1u, slip_move_rt(aa, va));
            const uint32_t cB = moved(T, P, B, p, slip_move_rt(ab, vb));
            cA_sel = cA; cB_sel = cB;
            const uint32_t kind = in_goal ? (uint32_t)K_MOVE : classify(A, B, cA, cB, aa, ab).kind;
            const uint32_t n = kind == K_COIN ? 2u : (kind == K_FOUR ? 4u : 1u);
            const double wq = SOCCER_WEIGHT_OF(cl);
            const double q = wq * (n == 1u ? 1.0 : (n == 2u ? 0.5 : 0.25));
            const double t1 = S + q, t2 = t1 + q, t3 = t2 + q;
            sel_k = (((n > 1u) & (d.u >= t1)) ? 1u : 0u) + (((n > 2u) & (d.u >= t2)) ? 1u : 0u) +
                    (((n > 2u) & (d.u >= t3)) ? 1u : 0u);
            near_thr |= (n > 1u) & (fabs(d.u - t1) < 0x1.0p-40);
            near_thr |= (n > 2u) & (fabs(d.u - t2) < 0x1.0p-40);
            near_thr |= (n > 2u) & (fabs(d.u - t3) < 0x1.0p-40);
        }
        // (2) Exact decision, exactly as categorical_sample walks the list: sequential float64 running sum
        // over the combinations in reference order, each contributing 1, 2 or 4 equal entries; record
        // WHICH entry (combination c, outcome k) is the first to exceed u.
        if (near_thr) {
            // each player has only three distinct moves (intended + two orthogonals): 6 table reads serve all nine combinations
            uint32_t cellA[3], cellB[3];
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                cellA[v] = moved(T, P, A, p ^ 1u, slip_move(aa, v));
                cellB[v] = moved(T, P, B, p, slip_move(ab, v));
            }
            double acc = 0.0;
            bool found = false;
            int first_c = -1;                                           // wave-uniform (weights are)
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const uint32_t cl_c = (CL2 >> (2 * c)) & 3u;
                const double wgt = SOCCER_WEIGHT_OF(cl_c);
                if (wgt == 0.0) continue;                               // uniform branch (:226-227)
                if (first_c < 0) first_c = c;
                const uint32_t va_c = (VA2 >> (2 * c)) & 3u, vb_c = (VB2 >> (2 * c)) & 3u;
                const uint32_t cA_c = va_c == 0u ? cellA[0] : (va_c == 1u ? cellA[1] : cellA[2]);
                const uint32_t cB_c = vb_c == 0u ? cellB[0] : (vb_c == 1u ? cellB[1] : cellB[2]);
                const uint32_t kind = in_goal ? (uint32_t)K_MOVE : classify(A, B, cA_c, cB_c, aa, ab).kind;
                const uint32_t n = kind == K_COIN ? 2u : (kind == K_FOUR ? 4u : 1u);
                const double q = wgt * (n == 1u ? 1.0 : (n == 2u ? 0.5 : 0.25));   // :241
                const double a1 = acc + q, a2 = a1 + q, a3 = a2 + q, a4 = a3 + q;   // sequential cumsum
                const double end = n == 1u ? a1 : (n == 2u ? a2 : a4);
                const uint32_t k = (d.u >= a1 ? 1u : 0u) + (((n > 1u) & (d.u >= a2)) ? 1u : 0u) +
                                   (((n > 2u) & (d.u >= a3)) ? 1u : 0u);
                const bool here = !found & (end > d.u);
                sel_c = here ? (uint32_t)c : sel_c; sel_k = here ? k : sel_k;
                found |= here;
                acc = end;
            }
            if (!found) { sel_c = (uint32_t)(first_c < 0 ? 0 : first_c); sel_k = 0u; }   // argmax of all-False is 0
            const uint32_t va = (VA2 >> (2u * sel_c)) & 3u, vb = (VB2 >> (2u * sel_c)) & 3u;
            cA_sel = va == 0u ? cellA[0] : (va == 1u ? cellA[1] : cellA[2]);
            cB_sel = vb == 0u ? cellB[0] : (vb == 1u ? cellB[1] : cellB[2]);
        }
#undef SOCCER_WEIGHT_OF
        // the selected combination
        cls = (CL2 >> (2u * sel_c)) & 3u;
        sel = pick(A, B, p, classify(A, B, cA_sel, cB_sel, aa, ab), sel_k);
        }
    }
    sel.A = in_goal ? A : sel.A; sel.B = in_goal ? B : sel.B; sel.p = in_goal ? p : sel.p;
    sel.kcode = in_goal ? 0u : sel.kcode;
    // done / reward (:235-240)
    const uint32_t ncc = col_of(sel.p ? sel.B : sel.A);
    const bool goal_now = (ncc == 0u) | (ncc == Wm1);
    const int32_t reward = (goal_now & !in_goal) ? (ncc == Wm1 ? 1 : -1) : 0;
    const uint32_t tt = t + 1u;                                         // :399
    const uint32_t trunc = tt >= (uint32_t)P.max_steps ? 1u : 0u;      // :404
    const uint32_t done = goal_now ? 1u : 0u;
    const uint32_t need = done | trunc;                                 // :406
    const uint32_t ob_step = obs_of(T, P, sel.A, sel.B, sel.p);      // :397 (goal tuples map to 0)
    out.obs = ob_step; out.final_obs = ob_step; out.reward = reward; out.term = done; out.trunc = trunc;
    out.code = cls * 3u + sel.kcode; out.finished = need;
    Lane L{sel.A, sel.B, sel.p, need, tt};
    if (need && P.autoreset) {                                          // in-step reset (:414-423)
        const uint4 e = *reinterpret_cast<const uint4*>(T.isd + 4u * (d.reset2 >> P.isd_shift));
        L.A = e.x; L.B = e.y; L.p = e.z & 1u; L.t = 0u; L.need = 0u;
        out.obs = e.z >> 16;
    }
    const bool frozen = Lref.need != 0u;
    if (frozen) {                        // rare: a lane that needs reset is left untouched (:376)
        const uint32_t ob = obs_of(T, P, A, B, p);
        out.obs = ob; out.final_obs = ob; out.reward = 0; out.term = in_goal ? 1u : 0u;
        out.trunc = t >= (uint32_t)P.max_steps ? 1u : 0u; out.code = 0u; out.finished = 0u;
        L = Lref;
    }
    Lref = L;
    return frozen;
}

__device__ __forceinline__ void lane_reset(const Tables& T, const KernelParams& P, Lane& L, uint32_t two_bits, uint32_t& ob) {
    const uint4 e = *reinterpret_cast<const uint4*>(T.isd + 4u * (two_bits >> P.isd_shift));
    L.A = e.x; L.B = e.y; L.p = e.z & 1u; L.t = 0u; L.need = 0u;
    ob = e.z >> 16;
}

// ---- E-wide packed byte / halfword vectors -------------------------------------------------------
// E consecutive bytes of one SoA stream, held as E/4 dwords so that element access is a constant
// bit-field extract (no byte arrays: those end up in scratch).  E = 1 is the scalar fallback.
template <int E> struct PackB {
    static constexpr int NW = E / 4;
    uint32_t w[NW];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < NW; ++k) w[k] = 0u;
    }
    __device__ __forceinline__ uint32_t get(int j) const { return (w[j >> 2] >> (8 * (j & 3))) & 0xffu; }
    // v must already fit 8 bits
    __device__ __forceinline__ void put(int j, uint32_t v) { w[j >> 2] |= v << (8 * (j & 3)); }
    __device__ __forceinline__ void load(const void* base, unsigned long long i) {
        const uint8_t* p = static_cast<const uint8_t*>(base) + i;
        if constexpr (E == 4) { w[0] = *reinterpret_cast<const uint32_t*>(p); }
        else { static_assert(E == 8, "E must be 1, 4 or 8");
               const uint2 v = *reinterpret_cast<const uint2*>(p); w[0] = v.x; w[1] = v.y; }
    }
    __device__ __forceinline__ void store(void* base, unsigned long long i) const {
        uint8_t* p = static_cast<uint8_t*>(base) + i;
        if constexpr (E == 4) { *reinterpret_cast<uint32_t*>(p) = w[0]; }
        else { *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]); }
    }
    // streaming variants (trajectory outputs / action inputs: touched once)
    __device__ __forceinline__ void load_nt(const void* base, unsigned long long i) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(base) + i);
#pragma unroll
        for (int k = 0; k < NW; ++k) w[k] = __builtin_nontemporal_load(p + k);
    }
    __device__ __forceinline__ void store_nt(void* base, unsigned long long i) const {
        uint32_t* p = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(base) + i);
#pragma unroll
        for (int k = 0; k < NW; ++k) __builtin_nontemporal_store(w[k], p + k);
    }
};
template <> struct PackB<1> {
    uint32_t b;
    __device__ __forceinline__ void clear() { b = 0u; }
    __device__ __forceinline__ uint32_t get(int) const { return b; }
    __device__ __forceinline__ void put(int, uint32_t v) { b = v; }
    __device__ __forceinline__ void load(const void* base, unsigned long long i) { b = static_cast<const uint8_t*>(base)[i]; }
    __device__ __forceinline__ void store(void* base, unsigned long long i) const { static_cast<uint8_t*>(base)[i] = (uint8_t)b; }
    __device__ __forceinline__ void load_nt(const void* base, unsigned long long i) { load(base, i); }
    __device__ __forceinline__ void store_nt(void* base, unsigned long long i) const { store(base, i); }
};

// action bytes as the kernels execute them (swar::canon4); returns non-zero when a byte was outside 0..4
template <int E> __device__ __forceinline__ uint32_t canon_pack(PackB<E>& a) {
    uint32_t bad = 0u;
    if constexpr (E == 1) { const uint32_t c = swar::canon4(a.b & 0xffu); bad = c ^ (a.b & 0xffu); a.b = c; }
    else {
#pragma unroll
        for (int k = 0; k < PackB<E>::NW; ++k) { const uint32_t c = swar::canon4(a.w[k]); bad |= c ^ a.w[k]; a.w[k] = c; }
    }
    return bad;
}

// E consecutive uint16 of one stream, as E/2 dwords; values must already fit 16 bits
template <int E> struct PackH {
    static constexpr int NW = E / 2;
    uint32_t w[NW];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < NW; ++k) w[k] = 0u;
    }
    __device__ __forceinline__ void put(int j, uint32_t v) { w[j >> 1] |= v << (16 * (j & 1)); }
    __device__ __forceinline__ void store(uint16_t* base, unsigned long long i) const {
        uint16_t* p = base + i;
        if constexpr (E == 4) { *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]); }
        else { *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]); }
    }
    __device__ __forceinline__ void store_nt(uint16_t* base, unsigned long long i) const {
        unsigned long long* p = reinterpret_cast<unsigned long long*>(base + i);
#pragma unroll
        for (int k = 0; k < NW; k += 2) __builtin_nontemporal_store((unsigned long long)w[k] | ((unsigned long long)w[k + 1] << 32), p + (k >> 1));
    }
};
template <> struct PackH<1> {
    uint32_t h;
    __device__ __forceinline__ void clear() { h = 0u; }
    __device__ __forceinline__ void put(int, uint32_t v) { h = v; }
    __device__ __forceinline__ void store(uint16_t* base, unsigned long long i) const { base[i] = (uint16_t)h; }
    __device__ __forceinline__ void store_nt(uint16_t* base, unsigned long long i) const { store(base, i); }
};

// E consecutive int32 accumulators (return_sum / episode_count), read-modify-write
template <int E>
__device__ __forceinline__ void add_words(int32_t* base, unsigned long long i, const int32_t (&d)[E]) {
    if constexpr (E == 1) { base[i] += d[0]; }
    else {
#pragma unroll
        for (int k = 0; k < E; k += 4) {
            int4 v = *reinterpret_cast<const int4*>(base + i + k);
            v.x += d[k]; v.y += d[k + 1]; v.z += d[k + 2]; v.w += d[k + 3];
            *reinterpret_cast<int4*>(base + i + k) = v;
        }
    }
}

// the six state streams of E lanes, still packed as loaded (so the loads can be issued early)
template <int E>
struct RawState {
    PackB<E> ra, ca, rb, cb, ps, tt;
    __device__ __forceinline__ void load(const KernelParams& P, unsigned long long i) {
        const uint8_t* s = P.state;
        ra.load(s, i); ca.load(s + P.state_stride, i); rb.load(s + 2 * P.state_stride, i);
        cb.load(s + 3 * P.state_stride, i); ps.load(s + 4 * P.state_stride, i); tt.load(s + 5 * P.state_stride, i);
    }
};

template <int E>
struct LaneVec {
    Lane L[E];
    __device__ __forceinline__ void unpack(const KernelParams& P, const RawState<E>& r) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            L[j].A = make_pos(r.ra.get(j), r.ca.get(j), P.W);
            L[j].B = make_pos(r.rb.get(j), r.cb.get(j), P.W);
            const uint32_t f = r.ps.get(j);
            L[j].p = f & 1u; L[j].need = (f >> 1) & 1u; L[j].t = r.tt.get(j);
        }
    }
    __device__ __forceinline__ void load(const KernelParams& P, unsigned long long i) {
        RawState<E> r; r.load(P, i); unpack(P, r);
    }
    __device__ __forceinline__ void store(const KernelParams& P, unsigned long long i) const {
        PackB<E> ra, ca, rb, cb, ps, tt;
        ra.clear(); ca.clear(); rb.clear(); cb.clear(); ps.clear(); tt.clear();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            ra.put(j, L[j].A >> 24); ca.put(j, (L[j].A >> 16) & 0xffu);
            rb.put(j, L[j].B >> 24); cb.put(j, (L[j].B >> 16) & 0xffu);
            ps.put(j, L[j].p | (L[j].need << 1)); tt.put(j, L[j].t);
        }
        uint8_t* s = P.state;
        ra.store(s, i); ca.store(s + P.state_stride, i); rb.store(s + 2 * P.state_stride, i);
        cb.store(s + 3 * P.state_stride, i); ps.store(s + 4 * P.state_stride, i); tt.store(s + 5 * P.state_stride, i);
    }
};

// episode histogram: per-thread counts, one DPP wave reduction at exit, one atomic per bin per wave
// on a sharded global array (no LDS, no workgroup barrier)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);   // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true);   // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// EARLY: fetch the wave's slot at kernel entry so the load's latency hides under the whole kernel (the
// single-step kernel, where every microsecond of tail counts); otherwise read-modify-write at exit
// (the rollout kernel, which would pay 6 live VGPRs — and an occupancy step — for nothing).
template <bool EARLY>
struct HistAcc {
    uint32_t fin, pos, neg;     // this thread's finished episodes: all / return +1 / return -1
    ulonglong2 old01; unsigned long long old2;   // EARLY only: the wave's slot as of kernel entry (lane 0)
    __device__ __forceinline__ unsigned long long* slot_at(unsigned long long* base, uint32_t mask) const {
        const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        return base + (size_t)(wave & mask) * kHistStride;
    }
    // Every wave of a launch owns one slot — the host sizes the array to the largest grid that counts (soccer_create) —
    // and launches are stream-ordered, so plain loads and stores accumulate without atomics.
    __device__ __forceinline__ void init_at(unsigned long long* base, uint32_t mask) {
        fin = 0u; pos = 0u; neg = 0u;
        if (EARLY) {
            old01 = make_ulonglong2(0ull, 0ull); old2 = 0ull;
            if ((threadIdx.x & 63u) == 0u) {
                const unsigned long long* h = slot_at(base, mask);
                old01 = *reinterpret_cast<const ulonglong2*>(h); old2 = h[2];
            }
        }
    }
    __device__ __forceinline__ void init(const KernelParams& P) { init_at(P.hist, P.hist_mask); }
    __device__ __forceinline__ void add(uint32_t finished, int32_t reward) {
        fin += finished; pos += reward > 0 ? 1u : 0u; neg += reward < 0 ? 1u : 0u;
    }
    // totals of a whole group at once: `finished` episodes, sum of their rewards and of |reward|
    // (rewards are -1/0/+1, and only the step that ends an episode can carry one)
    __device__ __forceinline__ void add_totals(uint32_t finished, int32_t reward_sum, uint32_t nonzero) {
        fin += finished; pos += (nonzero + (uint32_t)reward_sum) >> 1; neg += (nonzero - (uint32_t)reward_sum) >> 1;
    }
    // Call once at kernel exit, where every lane of the wave is active.
    __device__ __forceinline__ void flush(const KernelParams& P) { flush_at(P.hist, P.hist_mask); }
    __device__ __forceinline__ void flush_at(unsigned long long* base, uint32_t mask) {
        const uint32_t tot = wave_sum(fin), p = wave_sum(pos), n = wave_sum(neg);
        if ((threadIdx.x & 63u) == 0u && tot) {
            unsigned long long* h = slot_at(base, mask);
            if (!EARLY) { old01 = *reinterpret_cast<const ulonglong2*>(h); old2 = h[2]; }
            *reinterpret_cast<ulonglong2*>(h) = make_ulonglong2(old01.x + n, old01.y + (tot - p - n));
            h[2] = old2 + p;
        }
    }
};

__device__ __forceinline__ void publish_tick(const KernelParams& P, unsigned long long tick, unsigned long long used) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *P.tick_out = tick + used;
}

// random words of the E lanes starting at global lane g0 (one Philox block per 4 aligned lanes)
template <int E>
__device__ __forceinline__ void lane_words(const KernelParams& P, unsigned long long g0, unsigned long long tick,
                                           uint32_t purpose, uint32_t (&w)[E]) {
    if (E >= 4 && (g0 & 3ull) == 0ull) {                // wave-uniform: lane_offset % 4 == 0
#pragma unroll
        for (int k = 0; k < E; k += 4) {
            const Philox4 b = lane_block(P, (g0 + k) >> 2, tick, purpose);
#pragma unroll
            for (int j = 0; j < 4 && k + j < E; ++j) w[k + j] = b.w[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const unsigned long long g = g0 + j;
            const Philox4 b = lane_block(P, g >> 2, tick, purpose);
            const uint32_t s = (uint32_t)g & 3u;
            w[j] = s & 2u ? (s & 1u ? b.w[3] : b.w[2]) : (s & 1u ? b.w[1] : b.w[0]);
        }
    }
}

// =================================================================================================
// batched_step
// =================================================================================================
// Each thread owns the 4 consecutive lanes [4g, 4g+4) and walks them in a ROLLED loop: the code of one
// lane step exists once, so a launch — which starts with a cold instruction cache — fetches ~4x less
// code than an unrolled body (measured: 12.8 -> 10.2 us per launch at 2^20 lanes, tools/labs/kernel_lab).
// Bytes are peeled off the packed input dwords by shifting and results are shifted into packed output
// dwords with v_alignbyte, so no per-lane register arrays are needed.  The rule tables are read
// straight from global memory (4.3 KB, L1/L2 resident): with 3 lookups per lane a per-workgroup LDS
// staging pass + barrier costs more than it saves (measured: -0.7 us).
template <bool VEC>
__device__ __forceinline__ uint32_t load4(const void* base, unsigned long long i, int cnt) {
    const uint8_t* p = static_cast<const uint8_t*>(base) + i;
    if (VEC) return *reinterpret_cast<const uint32_t*>(p);
    uint32_t v = 0u;
    for (int k = 0; k < cnt; ++k) v |= (uint32_t)p[k] << (8 * k);
    return v;
}
template <bool VEC>
__device__ __forceinline__ void store4(void* base, unsigned long long i, int cnt, uint32_t v) {
    uint8_t* p = static_cast<uint8_t*>(base) + i;
    if (VEC) { *reinterpret_cast<uint32_t*>(p) = v; return; }
    for (int k = 0; k < cnt; ++k) p[k] = (uint8_t)(v >> (8 * k));
}
template <bool VEC>
__device__ __forceinline__ void store4h(uint16_t* base, unsigned long long i, int cnt, uint32_t lo, uint32_t hi) {
    uint16_t* p = base + i;
    if (VEC) { *reinterpret_cast<uint2*>(p) = make_uint2(lo, hi); return; }
    for (int k = 0; k < cnt; ++k) p[k] = (uint16_t)((k < 2 ? lo : hi) >> (16 * (k & 1)));
}

// VEC:    the launch covers a multiple of 4 lanes starting at a multiple of 4, all streams dword-aligned
//         (the host sends a ragged tail / misaligned buffers to the VEC = false instantiation);
// SHARED: (lane_offset + first) % 4 == 0, so a thread's 4 lanes are exactly one Philox block;
// EXPLICIT_U ("generic"): caller-supplied uniforms (u_step / u_reset) may replace the Philox draw, and
//         a fixed-policy side (single-agent mode, reference :187-188) takes its action from
//         policy[observation of the current tuple] instead of the action stream.
// LEAN:   no prob_code / final_obs / last_return outputs and no step statistics: their code is compiled
//         out (a launch fetches its code into a cold instruction cache: -0.6 us per launch).
// The hot instantiation <SLIP=false, EXPLICIT_U=false, VEC=true, SHARED=true, LEAN=true> carries none
// of the fallback code.
template <bool SLIP, bool EXPLICIT_U, bool VEC, bool SHARED, int UNROLL = 1, int BLOCK = kBlock, bool LEAN = false>
__global__ __launch_bounds__(BLOCK) void step_kernel(const KernelParams P, const StepIO IO) {
    const unsigned long long groups = (P.n + 3) >> 2;
    const unsigned long long stride = (unsigned long long)gridDim.x * BLOCK;
    const unsigned long long tick = *P.tick_in;                 // scalar load; published at the end so that its miss
                                                                // does not sit in front of the first data loads
    // the episode histogram of single steps is opt-in (SOCCER_F_STEP_STATS): counting, the wave
    // reduction and the slot update cost ~0.5 us of a ~9 us launch
    const bool stats = !LEAN && P.step_stats != 0u;
    HistAcc<true> hist; hist.fin = 0u; hist.pos = 0u; hist.neg = 0u; hist.old01 = make_ulonglong2(0ull, 0ull); hist.old2 = 0ull;
    if (stats) hist.init(P);
    Tables T; T.lut = P.lut; T.nc = P.next_cell; T.isd = P.isd;
    bool mis = false;
    uint32_t bad_act = 0u;
    const unsigned long long todo = IO.worklist ? (unsigned long long)*IO.work_count : groups;      // (one workgroup when listed)
    for (unsigned long long k = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; k < todo; k += stride) {
        const unsigned long long g = IO.worklist ? (unsigned long long)IO.worklist[k] : k;
        const unsigned long long i0 = P.first + (g << 2);
        const int cnt = VEC ? 4 : ((P.n - (g << 2)) < 4ull ? (int)(P.n - (g << 2)) : 4);
        const uint8_t* sp = P.state;
        uint32_t ra = load4<VEC>(sp, i0, cnt), ca = load4<VEC>(sp + P.state_stride, i0, cnt);
        uint32_t rb = load4<VEC>(sp + 2 * P.state_stride, i0, cnt), cb = load4<VEC>(sp + 3 * P.state_stride, i0, cnt);
        uint32_t ps = load4<VEC>(sp + 4 * P.state_stride, i0, cnt), tt = load4<VEC>(sp + 5 * P.state_stride, i0, cnt);
        uint32_t aa = 0u, ab = 0u;
        if (!EXPLICIT_U || !P.policy_a) aa = load4<VEC>(IO.act_a, i0, cnt);
        if (!EXPLICIT_U || !P.policy_b) ab = load4<VEC>(IO.act_b, i0, cnt);
        // an action byte executes as table[byte & 7] with 5..7 -> NOOP, so none can index outside a rule table; any
        // byte outside 0..4 is reported (the reference raises IndexError, :393)
        { const uint32_t ca_ = swar::canon4(aa), cb_ = swar::canon4(ab); bad_act |= (ca_ ^ aa) | (cb_ ^ ab); aa = ca_; ab = cb_; }
        // randomness does not depend on the loads above: it is computed while they are in flight
        const bool need_philox = !EXPLICIT_U || (IO.u_step == nullptr) || (P.autoreset && IO.u_reset == nullptr);
        Philox4 blk{{0u, 0u, 0u, 0u}};
        if (SHARED && need_philox) blk = lane_block(P, (P.lane_offset + i0) >> 2, block_tick<SLIP>(tick), 0u);
        uint32_t nra = 0, nca = 0, nrb = 0, ncb = 0, nps = 0, ntt = 0;
        uint32_t o_rew = 0, o_term = 0, o_trunc = 0, o_code = 0, o_lo = 0, o_hi = 0, f_lo = 0, f_hi = 0, fin_mask = 0;
#pragma unroll UNROLL
        for (int j = 0; j < cnt; ++j) {
            uint32_t w = j & 2 ? (j & 1 ? blk.w[3] : blk.w[2]) : (j & 1 ? blk.w[1] : blk.w[0]);
            if (!SHARED && need_philox) {
                const unsigned long long gl = P.lane_offset + i0 + j;
                const Philox4 b1 = lane_block(P, gl >> 2, block_tick<SLIP>(tick), 0u);
                const uint32_t sl = (uint32_t)gl & 3u;
                w = sl & 2u ? (sl & 1u ? b1.w[3] : b1.w[2]) : (sl & 1u ? b1.w[1] : b1.w[0]);
            }
            Draw d = draw_from_word<SLIP>(w, tick);
            if (EXPLICIT_U) {
                // (fetching the group's four uniforms ahead of this rolled loop was tried in round 4: slower on the slip walk,
                // 17.1 -> 20.4 us per launch at 2^20 lanes — sixteen more live registers; slip 0 takes step_kernel_swar<.., EXPL>)
                if (IO.u_step) { const double raw = IO.u_step[i0 + j]; const double u = sane_uniform(raw); d.u = SLIP ? sane_uniform_walk(raw) : u; d.top2 = (uint32_t)(u * 4.0); }
                if (IO.u_reset) d.reset2 = (uint32_t)(sane_uniform(IO.u_reset[i0 + j]) * 4.0);
            }
            // byte j of every packed stream: one v_bfe_u32 each (the offset 8*j is wave-uniform)
            const uint32_t sh = 8u * (uint32_t)j;
            const uint32_t psj = __builtin_amdgcn_ubfe(ps, sh, 8u);
            Lane L;
            L.A = make_pos(__builtin_amdgcn_ubfe(ra, sh, 8u), __builtin_amdgcn_ubfe(ca, sh, 8u), P.W);
            L.B = make_pos(__builtin_amdgcn_ubfe(rb, sh, 8u), __builtin_amdgcn_ubfe(cb, sh, 8u), P.W);
            L.p = psj & 1u; L.need = (psj >> 1) & 1u; L.t = __builtin_amdgcn_ubfe(tt, sh, 8u);
            uint32_t a_now = __builtin_amdgcn_ubfe(aa, sh, 8u), b_now = __builtin_amdgcn_ubfe(ab, sh, 8u);
            if (EXPLICIT_U && (P.policy_a || P.policy_b)) {         // the fixed side acts on the current observation
                const uint32_t s_now = obs_of(T, P, L.A, L.B, L.p);
                if (P.policy_a) a_now = (uint32_t)(uint8_t)P.policy_a[s_now];
                if (P.policy_b) b_now = (uint32_t)(uint8_t)P.policy_b[s_now];
            }
            StepResult R;
            // a caller-supplied uniform is an arbitrary double; without one the draw is the lane's Philox word
            // (fixed-policy handles take this kernel too) and the integer slip decision applies
            if (EXPLICIT_U && IO.u_step) mis |= lane_step<SLIP, false>(T, P, L, a_now, b_now, d, R);
            else mis |= lane_step<SLIP, true>(T, P, L, a_now, b_now, d, R);
            nra = __builtin_amdgcn_alignbyte(L.A >> 24, nra, 1); nca = __builtin_amdgcn_alignbyte((L.A >> 16) & 0xffu, nca, 1);
            nrb = __builtin_amdgcn_alignbyte(L.B >> 24, nrb, 1); ncb = __builtin_amdgcn_alignbyte((L.B >> 16) & 0xffu, ncb, 1);
            nps = __builtin_amdgcn_alignbyte(L.p | (L.need << 1), nps, 1); ntt = __builtin_amdgcn_alignbyte(L.t, ntt, 1);
            o_rew = __builtin_amdgcn_alignbyte((uint32_t)R.reward & 0xffu, o_rew, 1);
            o_term = __builtin_amdgcn_alignbyte(R.term, o_term, 1); o_trunc = __builtin_amdgcn_alignbyte(R.trunc, o_trunc, 1);
            if (!LEAN) o_code = __builtin_amdgcn_alignbyte(R.code, o_code, 1);
            o_lo = __builtin_amdgcn_alignbit(o_hi, o_lo, 16); o_hi = (o_hi >> 16) | (R.obs << 16);
            if (!LEAN) { f_lo = __builtin_amdgcn_alignbit(f_hi, f_lo, 16); f_hi = (f_hi >> 16) | (R.final_obs << 16); }
            if (!LEAN) fin_mask |= R.finished << j;
            if (stats) hist.add(R.finished, R.reward);
        }
        if (!VEC && cnt < 4) {               // ragged tail: the shifted-in bytes sit at the top
            const int sh = 8 * (4 - cnt);
            nra >>= sh; nca >>= sh; nrb >>= sh; ncb >>= sh; nps >>= sh; ntt >>= sh;
            o_rew >>= sh; o_term >>= sh; o_trunc >>= sh; o_code >>= sh;
            for (int k = cnt; k < 4; ++k) {
                o_lo = __builtin_amdgcn_alignbit(o_hi, o_lo, 16); o_hi >>= 16;
                f_lo = __builtin_amdgcn_alignbit(f_hi, f_lo, 16); f_hi >>= 16;
            }
        }
        uint8_t* sw = P.state;
        store4<VEC>(sw, i0, cnt, nra); store4<VEC>(sw + P.state_stride, i0, cnt, nca);
        store4<VEC>(sw + 2 * P.state_stride, i0, cnt, nrb); store4<VEC>(sw + 3 * P.state_stride, i0, cnt, ncb);
        store4<VEC>(sw + 4 * P.state_stride, i0, cnt, nps); store4<VEC>(sw + 5 * P.state_stride, i0, cnt, ntt);
        if (IO.obs) store4h<VEC>(IO.obs, i0, cnt, o_lo, o_hi);
        if (IO.reward) store4<VEC>(IO.reward, i0, cnt, o_rew);
        if (IO.terminated) store4<VEC>(IO.terminated, i0, cnt, o_term);
        if (IO.truncated) store4<VEC>(IO.truncated, i0, cnt, o_trunc);
        if (!LEAN && IO.prob_code) store4<VEC>(IO.prob_code, i0, cnt, o_code);
        if (!LEAN && IO.final_obs) store4h<VEC>(IO.final_obs, i0, cnt, f_lo, f_hi);
        if (!LEAN && IO.last_return && fin_mask) {
            for (int j = 0; j < cnt; ++j)
                if ((fin_mask >> j) & 1u) IO.last_return[i0 + j] = (int8_t)(o_rew >> (8 * j));
        }
        if (!LEAN && (IO.reward_a_f32 || IO.reward_b_f32 || IO.finished)) {
            for (int j = 0; j < cnt; ++j) {
                const float f = (float)(int8_t)(o_rew >> (8 * j));
                if (IO.reward_a_f32) IO.reward_a_f32[i0 + j] = f;
                if (IO.reward_b_f32) IO.reward_b_f32[i0 + j] = 0.0f - f;
                if (IO.finished) IO.finished[i0 + j] = (uint8_t)(((o_term | o_trunc) >> (8 * j)) & 1u);
            }
        }
    }
    if (mis) P.misuse[0] = 1u;
    if (bad_act) P.misuse[1] = 1u;
    if (stats) hist.flush(P);
    if (P.tick_out) publish_tick(P, tick, 1ull);
    if (IO.worklist) {                       // launched as ONE workgroup: everyone has read the count, the list is consumed
        __syncthreads();
        if (threadIdx.x == 0) *IO.work_count = 0u;
    }
}

// The instantiation every Philox-driven, dword-aligned, 4-outputs-only step takes (bench.py's path):
// one group of 4 lanes per thread, no grid-stride loop, no fallback or optional-output code at all.
// Same lane loop as step_kernel; kept separate because a launch starts with a cold instruction cache
// and every instruction that is not fetched counts (-0.4 us per launch against step_kernel<..., LEAN>).
template <bool SLIP, bool INT_ONLY = false, int UNROLL = 1>
__device__ __forceinline__ void hot_group(const KernelParams& P, const StepIO& IO, unsigned long long g,
                                          const unsigned long long* tick_ptr, unsigned long long tick_val) {
    const unsigned long long i0 = P.first + (g << 2);
    Tables T; T.lut = P.lut; T.nc = P.next_cell; T.isd = P.isd;
    const uint8_t* sp = P.state;
#ifndef SOCCER_TEMPORAL_STATE
#define SOCCER_LD(p) __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p))
#else
#define SOCCER_LD(p) (*reinterpret_cast<const uint32_t*>(p))
#endif
    const uint32_t ra = SOCCER_LD(sp + i0);
    const uint32_t ca = SOCCER_LD(sp + P.state_stride + i0);
    const uint32_t rb = SOCCER_LD(sp + 2 * P.state_stride + i0);
    const uint32_t cb = SOCCER_LD(sp + 3 * P.state_stride + i0);
    const uint32_t ps = SOCCER_LD(sp + 4 * P.state_stride + i0);
    const uint32_t tt = SOCCER_LD(sp + 5 * P.state_stride + i0);
#undef SOCCER_LD
    uint32_t aa = 0u, ab = 0u;
#ifndef SOCCER_TEMPORAL_IO
    aa = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(IO.act_a + i0));
    ab = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(IO.act_b + i0));
#else
    aa = *reinterpret_cast<const uint32_t*>(IO.act_a + i0);
    ab = *reinterpret_cast<const uint32_t*>(IO.act_b + i0);
#endif
    // action bytes execute as table[byte & 7] with 5..7 -> NOOP; anything outside 0..4 is reported (:393)
    const uint32_t aa_raw = aa, ab_raw = ab;
    aa = swar::canon4(aa); ab = swar::canon4(ab);
    // The tick comes from device memory (graph replays cannot change kernel arguments).  It is read AFTER the
    // eight data loads above have been issued: read first, its scalar-cache miss (~1 us) sat in front of them.
    const unsigned long long tick = tick_ptr ? *tick_ptr : tick_val;
    if (P.tick_out) publish_tick(P, tick, 1ull);
    // the thread's 4 lanes are exactly one Philox block; computed while the loads are in flight
    const Philox4 blk = lane_block(P, (P.lane_offset + i0) >> 2, block_tick<SLIP>(tick), 0u);
    uint32_t nra = 0, nca = 0, nrb = 0, ncb = 0, nps = 0, ntt = 0, o_rew = 0, o_term = 0, o_trunc = 0, o_lo = 0, o_hi = 0;
    uint32_t posA[4] = {0u, 0u, 0u, 0u}, posB[4] = {0u, 0u, 0u, 0u};   // UNROLL == 4 only
    bool mis = false;
#pragma unroll UNROLL
    for (int j = 0; j < 4; ++j) {
        const uint32_t w = j & 2 ? (j & 1 ? blk.w[3] : blk.w[2]) : (j & 1 ? blk.w[1] : blk.w[0]);
        const uint32_t sh = 8u * (uint32_t)j;
        const uint32_t psj = __builtin_amdgcn_ubfe(ps, sh, 8u);
        Lane L;
        L.A = make_pos(__builtin_amdgcn_ubfe(ra, sh, 8u), __builtin_amdgcn_ubfe(ca, sh, 8u), P.W);
        L.B = make_pos(__builtin_amdgcn_ubfe(rb, sh, 8u), __builtin_amdgcn_ubfe(cb, sh, 8u), P.W);
        L.p = psj & 1u; L.need = (psj >> 1) & 1u; L.t = __builtin_amdgcn_ubfe(tt, sh, 8u);
        StepResult R;
        const uint32_t a_now = __builtin_amdgcn_ubfe(aa, sh, 8u), b_now = __builtin_amdgcn_ubfe(ab, sh, 8u);
        mis |= lane_step<SLIP, true, INT_ONLY>(T, P, L, a_now, b_now, draw_from_word<SLIP>(w, tick), R);
        if constexpr (UNROLL == 4) { posA[j] = L.A; posB[j] = L.B; }    // rows / columns gathered with v_perm after the loop
        else {
            nra = __builtin_amdgcn_alignbyte(L.A >> 24, nra, 1); nca = __builtin_amdgcn_alignbyte((L.A >> 16) & 0xffu, nca, 1);
            nrb = __builtin_amdgcn_alignbyte(L.B >> 24, nrb, 1); ncb = __builtin_amdgcn_alignbyte((L.B >> 16) & 0xffu, ncb, 1);
        }
        nps = __builtin_amdgcn_alignbyte(L.p | (L.need << 1), nps, 1); ntt = __builtin_amdgcn_alignbyte(L.t, ntt, 1);
        o_rew = __builtin_amdgcn_alignbyte((uint32_t)R.reward & 0xffu, o_rew, 1);
        o_term = __builtin_amdgcn_alignbyte(R.term, o_term, 1); o_trunc = __builtin_amdgcn_alignbyte(R.trunc, o_trunc, 1);
        o_lo = __builtin_amdgcn_alignbit(o_hi, o_lo, 16); o_hi = (o_hi >> 16) | (R.obs << 16);
    }
    if constexpr (UNROLL == 4) {
        // the row (byte 3) and column (byte 2) of four position words -> the packed row / column dwords: 4 byte
        // permutes per player instead of a shift + funnel shift per lane and field (v_perm_b32 picks bytes 0-3 from
        // its second operand, 4-7 from its first)
        const uint32_t a01 = __builtin_amdgcn_perm(posA[1], posA[0], 0x07030602u), a23 = __builtin_amdgcn_perm(posA[3], posA[2], 0x07030602u);
        const uint32_t b01 = __builtin_amdgcn_perm(posB[1], posB[0], 0x07030602u), b23 = __builtin_amdgcn_perm(posB[3], posB[2], 0x07030602u);
        nca = __builtin_amdgcn_perm(a23, a01, 0x05040100u); nra = __builtin_amdgcn_perm(a23, a01, 0x07060302u);
        ncb = __builtin_amdgcn_perm(b23, b01, 0x05040100u); nrb = __builtin_amdgcn_perm(b23, b01, 0x07060302u);
    }
    uint8_t* sw = P.state;
    // the state is re-read by the NEXT launch only, i.e. after the kernel-boundary write-back / invalidate of L2:
    // streaming it as well is worth another ~1 % (6.91 -> 6.83 us)
#ifndef SOCCER_TEMPORAL_STATE
#define SOCCER_ST(p, v) __builtin_nontemporal_store((v), reinterpret_cast<uint32_t*>(p))
#else
#define SOCCER_ST(p, v) (*reinterpret_cast<uint32_t*>(p) = (v))
#endif
    SOCCER_ST(sw + i0, nra); SOCCER_ST(sw + P.state_stride + i0, nca);
    SOCCER_ST(sw + 2 * P.state_stride + i0, nrb); SOCCER_ST(sw + 3 * P.state_stride + i0, ncb);
    SOCCER_ST(sw + 4 * P.state_stride + i0, nps); SOCCER_ST(sw + 5 * P.state_stride + i0, ntt);
#undef SOCCER_ST
    // Results are written once and never re-read by these kernels, actions are read once: non-temporal accesses
    // keep them from displacing the resident state in L2 / Infinity Cache (7.66 -> 6.97 us per launch).
#ifndef SOCCER_TEMPORAL_IO
    if (IO.obs) __builtin_nontemporal_store((unsigned long long)o_lo | ((unsigned long long)o_hi << 32),
                                            reinterpret_cast<unsigned long long*>(IO.obs + i0));
    if (IO.reward) __builtin_nontemporal_store(o_rew, reinterpret_cast<uint32_t*>(IO.reward + i0));
    if (IO.terminated) __builtin_nontemporal_store(o_term, reinterpret_cast<uint32_t*>(IO.terminated + i0));
    if (IO.truncated) __builtin_nontemporal_store(o_trunc, reinterpret_cast<uint32_t*>(IO.truncated + i0));
#else
    if (IO.obs) *reinterpret_cast<uint2*>(IO.obs + i0) = make_uint2(o_lo, o_hi);
    if (IO.reward) *reinterpret_cast<uint32_t*>(IO.reward + i0) = o_rew;
    if (IO.terminated) *reinterpret_cast<uint32_t*>(IO.terminated + i0) = o_term;
    if (IO.truncated) *reinterpret_cast<uint32_t*>(IO.truncated + i0) = o_trunc;
#endif
    if (mis) P.misuse[0] = 1u;
    if ((aa ^ aa_raw) | (ab ^ ab_raw)) P.misuse[1] = 1u;
}

// The seven leading scalar arguments (14 dwords) repeat the fields of P / IO that the first loads depend on
// (the hot path always starts at lane 0 of the handle):
// the library is built with -mllvm -amdgpu-kernarg-preload-count=14, so they arrive in SGPRs at wave launch
// and the nine data loads are issued without first waiting for a scalar load of the kernarg segment
// (-0.3 .. -0.6 us per launch, tools/labs/pipeline_lab.hip); the rest of P is fetched while they are in flight.
template <bool SLIP, bool INT_ONLY = false>
__global__ __launch_bounds__(kBlock) void step_kernel_hot(uint8_t* state, unsigned long long state_stride,
                                                          const int8_t* act_a, const int8_t* act_b,
                                                          const unsigned long long* tick_in,
                                                          unsigned long long n, unsigned long long tick_val,
                                                          const KernelParams P, const StepIO IO) {
    // tick_in == nullptr: an eager launch — the host knows the tick and passes it by value (tick_val), which takes the
    // scalar load off the path (-2.4 %); captured launches read the device slot (their arguments are frozen).
    const unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    if ((g << 2) >= n) return;                                      // n is a multiple of 4 here; the first lane is 0
    KernelParams Q = P; Q.state = state; Q.state_stride = state_stride; Q.n = n; Q.first = 0ull;
    StepIO J = IO; J.act_a = act_a; J.act_b = act_b;
    hot_group<SLIP, INT_ONLY, SLIP ? SOCCER_HOT_UNROLL_SLIP : SOCCER_HOT_UNROLL>(Q, J, g, tick_in, tick_val);
}

// =================================================================================================
// batched_step, byte-parallel: the four lanes of a thread stay packed in their dwords (soccer_swar.hpp)
// =================================================================================================
// Same launch shape and memory behaviour as step_kernel_hot (one 4-lane group per thread, eight non-temporal
// dword loads, the Philox block computed while they are in flight, ten non-temporal stores, leading scalar
// arguments preloaded into SGPRs) — but no byte peeling, no per-lane loop and NO rule-table read: ~45 vector
// instructions per env-step instead of ~128 and no dependent gather between the loads and the stores.
// Takes every Philox-driven, dword-aligned step of a slip_prob == 0 handle whose pitch fits the byte arithmetic
// (swar::fits: every golden pitch up to 11x7 does).  GENERAL = false is the steady state of an auto-resetting
// handle (no frozen lane, no lane in a goal tuple); FULL adds final_obs and prob_code (VectorSoccerEnv).
// slip_prob > 0 with the caller's uniforms: the float64 form of the slip decision (SlipTables::B / w / act_pack / nb)
struct SlipF64 { double B[9]; double w[4]; unsigned long long act_pack; uint32_t nb; uint32_t pad_; };
struct SwarParams {
    swar::Consts C;
    uint32_t key0, key1;
    unsigned long long lane_offset;
    unsigned long long first;               // first lane (within the handle) this launch covers; multiple of 4
    unsigned long long* tick_out;
    unsigned int* misuse;                   // [0] a frozen lane was stepped (:376), [1] an action byte outside 0..4 (:393)
    unsigned long long* hist; uint32_t hist_mask;   // OUT == 2: episode histogram slots (SOCCER_F_STEP_STATS), or nullptr
    swar::SlipConsts L; const swar::Quad* sub;   // SLIPM == 1: integer cumulative weights / the nine rows of quarter thresholds
    const uint32_t* slip_lut;               // SLIPM == 2: SlipTables::lut_step (kSlipStepBuckets bytes), then T (kSlipThresholds words)
    uint32_t act_stream;                    // SOCCER_F_STREAM_ACTIONS: the action streams are read with the non-temporal hint
    const int8_t* policy_a; const int8_t* policy_b;   // POLICY: the fixed side's int8[nS] policy (the other is nullptr)
    uint16_t* obs; int8_t* reward; uint8_t* terminated; uint8_t* truncated;
    uint8_t* prob_code; uint16_t* final_obs;                          // OUT == 2
    float* reward_a_f32; float* reward_b_f32; uint8_t* finished; int8_t* last_return;   // OUT >= 1
    const double* u_step; const double* u_reset;   // EXPL: caller-supplied uniforms (16-byte aligned; either may be nullptr: Philox then)
    const SlipF64* f64;                            // SLIPM == 3: the nominal float64 thresholds of the slip list
    uint32_t* worklist; uint32_t* work_count;      // SLIPM == 3: groups left to the exact walk (see StepIO)
};


// OUT — which outputs the instantiation can write (every pointer may still be NULL):
//   0  obs / reward / terminated / truncated: the 8-argument batched_step (19 B per env-step)
//   1  + reward_a_f32 / reward_b_f32 / finished / last_return: what a gym-style loop reads every step, without the `info`
//      extras (VectorSoccerEnv(io="device", info=False): 27 B per env-step when the int8 reward stream is left out)
//   2  + final_obs / prob_code and, when Q.hist is set, the episode histogram (VectorSoccerEnv's info; 31 B)
// Launch shape (tools/labs/swar_sweep.sh, profiles/r02_sweep.md): one 4-lane group per thread with non-temporal dword
// accesses measured best; 8 or 16 lanes per thread (dwordx2 / dwordx4), plain or write-through stores and 512-thread
// workgroups were all equal or slower, and an instantiation without the frozen-lane / goal-tuple code was not faster
// (the kernel is bound by launch + memory latency, not by vector issue any more).
// SLIPM: handles with slip_prob > 0 whose integer slip decision is the reference's for every draw (SlipTables::swar_ok).
//   1  each lane counts the integer cumulative weights and its combination's quarter points below its draw, one by one (the
//      threshold rows are gathered while the state loads are still in flight: they depend on the random word only) — ~30 vector
//      instructions per lane;
//   2  (SlipTables::lut_step_ok: slips within about [0.09, 0.96]) by table, like the rollout: a launch lives for one step, so
//      each WAVE stages what one 16-byte load per lane brings in — 1 024 byte buckets over the draw's top 10 bits — and the
//      threshold list (one entry per lane), issued ahead of the state loads and parked in the wave's own 1 280 bytes of LDS while those
//      are in flight (no workgroup barrier); a lane then needs two LDS reads and two exact compares (~9 instructions).  The table cannot be gathered from global memory instead: a wave's
//      loads return in order, so a gather issued after the state loads waits for all of them, and the 16 KB table of the
//      rollout is two dependent L2 round trips on top (5.15 us per launch, the same as comparing one by one; a 64 KB table with
//      the candidate inlined, one gather, thrashes the 16 KB L1: 6.05 us).
// Without SLIP the thread's block is the one of tick >> 3 and the lanes' draws are this tick's nibbles (swar::rand_nibble).
// POLICY: single-agent handles — the fixed side's action is looked up from its int8[nS] policy by the observation of
// the CURRENT tuple (four byte gathers per thread, behind the state loads); that side's action stream may be NULL.
// EXPL (slip_prob == 0 handles): the caller's own uniforms (batched_step_ex's u_step / u_reset: the reference-RNG replay path,
// e.g. a host that keeps the reference's MT19937 streams) replace the lanes' Philox bits.  Every list probability is 1, 1/2
// or 1/4 and the ISD is uniform over 4 or 2 entries, so floor(4u) IS the reference's first-exceeds decision for any double
// (:395, :414; values outside [0, 1) and NaN select index 0 like argmax of an all-False array): four doubles per stream and
// thread, two 16-byte loads each, issued with the state loads; round 3 sent these calls to the per-lane kernel (11.0 us).
constexpr unsigned long long kSwarLaunchLanes = 1ull << 30;   // 4 bytes per lane (the float rewards) * 2^30 lanes: offsets below 2^32
template <int OUT, int SLIPM = 0, bool POLICY = false, int GEO = 0, bool EXPL = false>
__global__ __launch_bounds__(kBlock) void step_kernel_swar(const uint8_t* state_in, unsigned long long state_stride,
                                                           const int8_t* act_a, const int8_t* act_b,
                                                           const unsigned long long* tick_in,
                                                           unsigned long long n, unsigned long long tick_val,
                                                           const SwarParams Q) {
    constexpr bool FULL = OUT == 2;
    constexpr bool SLIP = SLIPM != 0;
    static_assert((SLIPM == 3) ? EXPL : (!EXPL || SLIPM == 0), "caller-supplied uniforms: SLIPM 0 (dyadic lists) or 3 (float64 slip decision)");
    static_assert(kSlipStepBuckets == 64 * 16 && kSlipThresholds <= 64, "one 16-byte piece of the table per lane of a wave");
    const unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    const bool active = (g << 2) < n;                                // n is a multiple of 4 here
    // SLIPM == 2: this lane's 16 bytes of the bucket table and its entry of the threshold list — the oldest loads
    // of the wave, so the wait for them does not wait for the state.  Every WAVE keeps a copy of its own: no workgroup barrier.
    const uint32_t lane = threadIdx.x & 63u;
    uint4 st_lut = make_uint4(0u, 0u, 0u, 0u); uint32_t st_thr = 0u;
    if (SLIPM == 2) {
        st_lut = reinterpret_cast<const uint4*>(Q.slip_lut)[lane];
        st_thr = Q.slip_lut[kSlipStepBuckets / 4 + lane];                 // (the list is padded to 64 entries)
    }
    if (SLIPM != 2 && !FULL && !active) return;
    HistAcc<true> hist;
    const bool stats = FULL && Q.hist != nullptr;                    // wave-uniform
    if (FULL) { hist.fin = 0u; hist.pos = 0u; hist.neg = 0u; hist.old01 = make_ulonglong2(0ull, 0ull); hist.old2 = 0ull; }
    if (stats) hist.init_at(Q.hist, Q.hist_mask);
    // Byte offsets are 32-bit (the host launches at most kSwarLaunchLanes lanes at a time): a uniform base plus a 32-bit
    // per-thread offset is what the compiler turns into SGPR-base addressing (global_load v, v_off, s[base:base+1]) — no
    // 64-bit vector add per stream (20 vector instructions of about 245 with 64-bit offsets).
    const uint32_t i0 = (uint32_t)Q.first + ((uint32_t)g << 2);
#define AT(base, off) (reinterpret_cast<const uint8_t*>(base) + (off))
    const uint8_t* sp = state_in;
    swar::Group S{0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t aa = 0u, ab = 0u;
    // SLIPM == 2: the state loads are issued by every lane (lanes beyond n re-read the first group) — under a branch the wait for
    // the table loads ahead of them could no longer count on their order and would become a wait for everything
    const bool fetch = SLIPM == 2 ? true : active;
    const uint32_t l0 = SLIPM == 2 ? (active ? i0 : (uint32_t)Q.first) : i0;
    if (fetch) {
        // The action streams first, by plain loads unless the caller asked for the non-temporal hint (include/soccer_hip.h): buffers
        // written or read a few steps ago are served from the Infinity Cache, and a non-temporal load gives that up — 0.2 us per
        // launch at 2^20 lanes — while action data streaming in from HBM is 0.4 us per launch faster with the hint (DESIGN.md
        // 4.3).  Both arms issue the same number of loads, so the waits below still count on the order.
        const bool ld_a = !POLICY || !Q.policy_a, ld_b = !POLICY || !Q.policy_b;
        // (each block gets the offset through an empty asm of its own: instruction selection works a block at a time and only
        // turns base + offset into SGPR-base addressing when it sees the addition in the block of the access)
        if (Q.act_stream) {
            uint32_t la = l0; asm("" : "+v"(la));
            if (ld_a) aa = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(AT(act_a, la)));
            if (ld_b) ab = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(AT(act_b, la)));
        } else {
            // (written as wavefront-scope relaxed atomic loads — plain global_load_dword instructions — because the optimiser
            // merges two arms that differ in nothing but the non-temporal hint, and drops the hint)
            uint32_t la = l0; asm("" : "+v"(la));
            if (ld_a) aa = __hip_atomic_load(reinterpret_cast<const uint32_t*>(AT(act_a, la)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (ld_b) ab = __hip_atomic_load(reinterpret_cast<const uint32_t*>(AT(act_b, la)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        uint32_t ls = l0; asm("" : "+v"(ls));
        S.ra = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(AT(sp, ls)));
        S.ca = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(AT(sp + state_stride, ls)));
        S.rb = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(AT(sp + 2 * state_stride, ls)));
        S.cb = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(AT(sp + 3 * state_stride, ls)));
        S.ps = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(AT(sp + 4 * state_stride, ls)));
        S.tt = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(AT(sp + 5 * state_stride, ls)));
    }
    // EXPL: the four lanes' uniforms, as floor(4u) (two bits each) — behind the state loads, ahead of the Philox block
    uint32_t xq = 0u, xr = 0u;
    double us0 = 0.0, us1 = 0.0, us2 = 0.0, us3 = 0.0;               // SLIPM == 3: the step uniforms themselves
    if (SLIPM == 3 && active) {
        const double2 a = *reinterpret_cast<const double2*>(Q.u_step + i0), b = *reinterpret_cast<const double2*>(Q.u_step + i0 + 2);
        us0 = a.x; us1 = a.y; us2 = b.x; us3 = b.y;
    }
    if (EXPL && active) {
        auto quarters = [&](const double* base) {
            const double2 a = *reinterpret_cast<const double2*>(base + i0), b = *reinterpret_cast<const double2*>(base + i0 + 2);
            return (uint32_t)(sane_uniform(a.x) * 4.0) | ((uint32_t)(sane_uniform(a.y) * 4.0) << 8) |
                   ((uint32_t)(sane_uniform(b.x) * 4.0) << 16) | ((uint32_t)(sane_uniform(b.y) * 4.0) << 24);
        };
        if (SLIPM != 3 && Q.u_step) xq = quarters(Q.u_step);
        if (Q.u_reset) xr = quarters(Q.u_reset);
    }
    // the tick: by value for eager launches, from the device slot for captured ones (read after the data loads are issued)
    const unsigned long long tick = tick_in ? *tick_in : tick_val;
    const unsigned long long q = (Q.lane_offset + i0) >> 2;     // the thread's 4 lanes are exactly one Philox block
    const unsigned long long bt = block_tick<SLIP>(tick);
    Philox4 blk{{0u, 0u, 0u, 0u}};
    if (!EXPL || !Q.u_step || !Q.u_reset)                           // (wave-uniform; both uniforms supplied: no block is needed)
        blk = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)bt, (uint32_t)(bt >> 32), Q.key0, Q.key1);
    const uint8_t* slip_lut = nullptr; const uint32_t* slip_thr = nullptr;
    if (SLIPM == 2) {                                                // park the table: all 64 lanes, whether their lanes exist or not
        __shared__ __attribute__((aligned(16))) uint32_t s_slip[SLIPM == 2 ? kBlock / 64 : 1][SLIPM == 2 ? kSlipStepLdsWords : 4];
        uint32_t* mine = s_slip[threadIdx.x >> 6];
        reinterpret_cast<uint4*>(mine)[lane] = st_lut;
        mine[kSlipStepBuckets / 4 + lane] = st_thr;
        // a wave's LDS operations complete in order; the fences keep the compiler from moving the reads below above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        slip_lut = reinterpret_cast<const uint8_t*>(mine); slip_thr = mine + kSlipStepBuckets / 4;
        if (!FULL && !active) return;
    }
    if (active) {
        if (POLICY) {                                               // the fixed side acts on the current observation (:187-188)
            uint32_t s_lo, s_hi;
            const uint32_t cc0 = swar::bfi(swar::mask_of(S.ps << 7), S.cb, S.ca);
            swar::obs4<true>(Q.C, S.ra, S.ca, S.rb, S.cb, S.ps & swar::K01, swar::is_zero(cc0) | swar::is_zero(cc0 ^ Q.C.Wm1x4), s_lo, s_hi);
            const int8_t* pol = Q.policy_a ? Q.policy_a : Q.policy_b;
            const uint32_t act = (uint32_t)(uint8_t)pol[s_lo & 0xffffu] | ((uint32_t)(uint8_t)pol[s_lo >> 16] << 8) |
                                 ((uint32_t)(uint8_t)pol[s_hi & 0xffffu] << 16) | ((uint32_t)(uint8_t)pol[s_hi >> 16] << 24);
            if (Q.policy_a) aa = act; else ab = act;
        }
        swar::Out o;
        uint32_t sa = 0u, sb = 0u, cls4 = 0u;
        swar::Rand4 rnd;
        bool listed = false;                                             // SLIPM == 3: the group goes to the exact walk of the per-lane kernel
        if (SLIPM == 3) {
            // The caller's uniform against the NOMINAL thresholds of the slip list — the cumulative weights of the active
            // combinations, then the quarter points of the selected one.  The reference's sequential float64 sums differ from
            // these by < 1e-14 whatever the list's shape (lane_step, fast decision), so a uniform farther than 2^-40 from every
            // threshold it is compared with is decided as the reference decides it; a group with a lane inside that margin (or
            // beyond the last threshold) is left to the per-lane kernel's exact walk: listed, nothing stored here.
            const SlipF64& F = *Q.f64;
            const double us[4] = {us0, us1, us2, us3};
            uint32_t c4 = 0u, k4 = 0u; bool near = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double u = sane_uniform_walk(us[j]);
                uint32_t idx = 0u; double S0 = 0.0;
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const double bi = F.B[i];                         // wave-uniform; +inf past the last active one
                    const bool ge = u >= bi;
                    idx += ge ? 1u : 0u; S0 = ge ? bi : S0;
                    near |= fabs(u - bi) < 0x1.0p-40;
                }
                near |= idx >= F.nb;
                const uint32_t c = (uint32_t)((F.act_pack >> (4u * idx)) & 0xfull);
                constexpr uint32_t CL2 = 0u | (1u << 2) | (1u << 4) | (2u << 6) | (2u << 8) | (3u << 10) | (3u << 12) | (3u << 14) | (3u << 16);
                const uint32_t cl = (CL2 >> (2u * c)) & 3u;           // weight class of combination c (:211-222), as in lane_step
                const double wq = (cl & 2u) ? ((cl & 1u) ? F.w[3] : F.w[2]) : ((cl & 1u) ? F.w[1] : F.w[0]);
                const double q = wq * 0.25, t1 = S0 + q, t2 = t1 + q, t3 = t2 + q;
                const uint32_t kq = (u >= t1 ? 1u : 0u) + (u >= t2 ? 1u : 0u) + (u >= t3 ? 1u : 0u);
                near |= fabs(u - t1) < 0x1.0p-40; near |= fabs(u - t2) < 0x1.0p-40; near |= fabs(u - t3) < 0x1.0p-40;
                c4 |= c << (8 * j); k4 |= kq << (8 * j);
            }
            if (near) {
                const uint32_t slot = atomicAdd(Q.work_count, 1u);
                Q.worklist[slot] = (uint32_t)g;
                listed = true;                                            // nothing of this group is stored or counted here
            }
            swar::slip_moves4(c4, swar::canon4(aa), swar::canon4(ab), sa, sb, cls4);
            rnd = swar::Rand4{k4 << 6, Q.u_reset ? (xr >> Q.C.isd_shift) : (swar::pack_byte0(blk.w[0], blk.w[1], blk.w[2], blk.w[3]) >> Q.C.isd_shift)};
        } else if (SLIP) {
            uint32_t k4 = 0u;
            if (SLIPM == 2) {
                const uint32_t p4 = swar::slip_count4_lut<kSlipStepBucketBits, kSlipStepCompares>(slip_lut, slip_thr, blk.w[0], blk.w[1], blk.w[2], blk.w[3]);
                // the counts need the random words and the table only: the empty statement ties the loaded actions to them, so that
                // the wait for the state loads comes after the table reads and not before
                asm volatile("" : "+v"(aa), "+v"(ab) : "v"(p4));
                swar::slip_from_count4(p4, Q.L.c_off, swar::canon4(aa), swar::canon4(ab), sa, sb, k4, cls4);
            }
            else swar::slip_select4(Q.L, Q.sub, swar::canon4(aa), swar::canon4(ab), blk.w[0], blk.w[1], blk.w[2], blk.w[3], sa, sb, k4, cls4);
            rnd = swar::Rand4{k4 << 6, swar::pack_byte0(blk.w[0], blk.w[1], blk.w[2], blk.w[3]) >> Q.C.isd_shift};
        } else {
            rnd = swar::rand_nibble(Q.C.isd_shift, (uint32_t)tick & 7u, blk.w[0], blk.w[1], blk.w[2], blk.w[3]);
            if (EXPL) {                                              // Rand4: the quarter in bits 7, 6 of each byte; the reset draw, shifted
                if (Q.u_step) rnd.kq = xq << 6;
                if (Q.u_reset) rnd.rs = xr >> Q.C.isd_shift;
            }
        }
        if (!listed) {
        // Frozen lanes and goal tuples exist only without auto-reset or after a state injection; a thread none of whose lanes is
        // in either condition (nearly every thread of an auto-resetting handle) takes the step without the code for them —
        // 31 vector instructions fewer, 12 for the test: in this kernel every instruction shows (5.6 ns).
        const uint32_t edge = swar::is_zero(S.ca) | swar::is_zero(S.cb) | swar::is_zero(S.ca ^ Q.C.Wm1x4) | swar::is_zero(S.cb ^ Q.C.Wm1x4);
        const bool special = Q.C.autoreset == 0u || (((S.ps << 6) | edge) & swar::K80) != 0u;
        if (special) swar::step4<true, FULL, SLIP, GEO>(Q.C, S, aa, ab, sa, sb, cls4, rnd, o);
        else swar::step4<false, FULL, SLIP, GEO>(Q.C, S, aa, ab, sa, sb, cls4, rnd, o);
        uint8_t* sw = const_cast<uint8_t*>(sp);
        // the stores' offset is opaque to the optimiser: it would otherwise hoist the 64-bit addresses of the loads above the
        // branch and reuse them (instruction selection works a block at a time and then no longer sees base + offset)
        uint32_t j0 = i0; asm volatile("" : "+v"(j0));
        const uint32_t j0x2 = j0 << 1, j0x4 = j0 << 2;
#define ATW(base, off) (reinterpret_cast<uint8_t*>(base) + (off))
        __builtin_nontemporal_store(S.ra, reinterpret_cast<uint32_t*>(ATW(sw, j0)));
        __builtin_nontemporal_store(S.ca, reinterpret_cast<uint32_t*>(ATW(sw + state_stride, j0)));
        __builtin_nontemporal_store(S.rb, reinterpret_cast<uint32_t*>(ATW(sw + 2 * state_stride, j0)));
        __builtin_nontemporal_store(S.cb, reinterpret_cast<uint32_t*>(ATW(sw + 3 * state_stride, j0)));
        __builtin_nontemporal_store(S.ps, reinterpret_cast<uint32_t*>(ATW(sw + 4 * state_stride, j0)));
        __builtin_nontemporal_store(S.tt, reinterpret_cast<uint32_t*>(ATW(sw + 5 * state_stride, j0)));
        if (Q.obs) __builtin_nontemporal_store((unsigned long long)o.obs_lo | ((unsigned long long)o.obs_hi << 32),
                                               reinterpret_cast<unsigned long long*>(ATW(Q.obs, j0x2)));
        if (Q.reward) __builtin_nontemporal_store(o.rew, reinterpret_cast<uint32_t*>(ATW(Q.reward, j0)));
        if (Q.terminated) __builtin_nontemporal_store(o.term, reinterpret_cast<uint32_t*>(ATW(Q.terminated, j0)));
        if (Q.truncated) __builtin_nontemporal_store(o.trunc, reinterpret_cast<uint32_t*>(ATW(Q.truncated, j0)));
        if (OUT >= 1) {
            if (Q.reward_a_f32 || Q.reward_b_f32) {                 // the rewards as the floats a gym caller reads (:400-402)
                const int32_t r = (int32_t)o.rew;
                const float f0 = (float)((r << 24) >> 24), f1 = (float)((r << 16) >> 24), f2 = (float)((r << 8) >> 24), f3 = (float)(r >> 24);
                typedef float f4 __attribute__((ext_vector_type(4)));
                if (Q.reward_a_f32) { const f4 va = {f0, f1, f2, f3}; __builtin_nontemporal_store(va, reinterpret_cast<f4*>(ATW(Q.reward_a_f32, j0x4))); }
                if (Q.reward_b_f32) { const f4 vb = {0.0f - f0, 0.0f - f1, 0.0f - f2, 0.0f - f3};
                                      __builtin_nontemporal_store(vb, reinterpret_cast<f4*>(ATW(Q.reward_b_f32, j0x4))); }
            }
            if (Q.finished) __builtin_nontemporal_store(o.term | o.trunc, reinterpret_cast<uint32_t*>(ATW(Q.finished, j0)));
            // A's return of the episode that just ended = the reward of its last step (only that step can carry one);
            // lanes whose episode goes on keep what the stream holds.  Rare: one read-modify-write of the thread's own dword.
            if (Q.last_return && (o.finished & swar::K80)) {
                uint32_t* lr = reinterpret_cast<uint32_t*>(ATW(Q.last_return, j0));
                *lr = swar::bfi(swar::mask_of(o.finished), o.rew, *lr);
            }
        }
        if (FULL) {
            if (Q.prob_code) __builtin_nontemporal_store(o.code, reinterpret_cast<uint32_t*>(ATW(Q.prob_code, j0)));
            if (Q.final_obs) __builtin_nontemporal_store((unsigned long long)o.fin_lo | ((unsigned long long)o.fin_hi << 32),
                                                         reinterpret_cast<unsigned long long*>(ATW(Q.final_obs, j0x2)));
            // finished episodes by return: a reward byte is 0x01 / 0xff only on the step that ends the episode
            if (stats) hist.add_totals((uint32_t)__builtin_popcount(o.finished & swar::K80),
                                       (int32_t)__builtin_popcount(o.rew & swar::K01) - 2 * (int32_t)__builtin_popcount(o.rew & swar::K80),
                                       (uint32_t)__builtin_popcount(o.rew & swar::K01));
        }
        if (o.frozen) Q.misuse[0] = 1u;
        if (o.bad_action) Q.misuse[1] = 1u;
        }
        // (published last: a store in flight ahead of the loads' waits would turn them into waits for everything — loads and
        // stores share the wave's counter and complete out of order with respect to each other)
        if (Q.tick_out && blockIdx.x == 0 && threadIdx.x == 0) *Q.tick_out = tick + 1ull;
#undef AT
#undef ATW
    }
    if (stats) hist.flush_at(Q.hist, Q.hist_mask);
}

// =================================================================================================
// batched_reset
// =================================================================================================
template <bool LUT_LDS, bool SLIP>
__global__ __launch_bounds__(kBlock) void reset_kernel(const KernelParams P, const ResetIO IO) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const Tables T = stage_tables<LUT_LDS>(P, smem);
    const unsigned long long tick = *P.tick_in;
    if (P.tick_out) publish_tick(P, tick, 1ull);
    for (unsigned long long k = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; k < P.n;
         k += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long i = P.first + k;                   // the launch covers lanes [first, first + n)
        const bool sel = IO.mask == nullptr || IO.mask[i] != 0;
        uint32_t ob = 0u;
        if (sel) {
            uint32_t bits;
            if (IO.u_reset) bits = (uint32_t)(sane_uniform(IO.u_reset[i]) * 4.0);
            else { uint32_t w[1]; lane_words<1>(P, P.lane_offset + i, block_tick<SLIP>(tick), 0u, w); bits = draw_from_word<SLIP>(w[0], tick).reset2; }
            LaneVec<1> S; lane_reset(T, P, S.L[0], bits, ob);
            S.store(P, i);
        } else if (IO.obs) {
            LaneVec<1> S; S.load(P, i);
            ob = obs_of(T, P, S.L[0].A, S.L[0].B, S.L[0].p);
        }
        if (IO.obs) IO.obs[i] = (uint16_t)ob;
    }
}

// batched_reset, byte-parallel (Philox draws, dword-aligned streams, pitches that fit the byte arithmetic): four lanes per
// thread, 6 dword stores + one 8-byte observation store; MASKED also reads the six state dwords and the mask dword.
struct ResetSwar {
    swar::Consts C;
    uint8_t* state; unsigned long long state_stride;
    unsigned long long n, lane_offset;
    const unsigned long long* tick_in; unsigned long long* tick_out;
    uint32_t key0, key1;
    const uint8_t* mask; uint16_t* obs;
};
template <bool MASKED, bool SLIP>
__global__ __launch_bounds__(kBlock) void reset_kernel_swar(const ResetSwar R) {
    const unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
    const unsigned long long i0 = g << 2;
    if (i0 >= R.n) return;                                          // n is a multiple of 4 here; the first lane is 0
    uint8_t* sp = R.state + i0;
    swar::Group S{0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t mask4 = 0u;
    if (MASKED) {
        S.ra = *reinterpret_cast<const uint32_t*>(sp); S.ca = *reinterpret_cast<const uint32_t*>(sp + R.state_stride);
        S.rb = *reinterpret_cast<const uint32_t*>(sp + 2 * R.state_stride); S.cb = *reinterpret_cast<const uint32_t*>(sp + 3 * R.state_stride);
        S.ps = *reinterpret_cast<const uint32_t*>(sp + 4 * R.state_stride); S.tt = *reinterpret_cast<const uint32_t*>(sp + 5 * R.state_stride);
        mask4 = *reinterpret_cast<const uint32_t*>(R.mask + i0);
    }
    const unsigned long long tick = *R.tick_in;
    if (blockIdx.x == 0 && threadIdx.x == 0) *R.tick_out = tick + 1ull;
    if (!MASKED) {      // what a reset of every lane writes without a draw — both columns, the timestep — leaves before the Philox block
        *reinterpret_cast<uint32_t*>(sp + R.state_stride) = R.C.isd_ca4; *reinterpret_cast<uint32_t*>(sp + 3 * R.state_stride) = R.C.isd_cb4;
        *reinterpret_cast<uint32_t*>(sp + 5 * R.state_stride) = 0u;
    }
    const unsigned long long q = (R.lane_offset + i0) >> 2;
    const unsigned long long bt = block_tick<SLIP>(tick);
    const Philox4 blk = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)bt, (uint32_t)(bt >> 32), R.key0, R.key1);
    uint32_t o_lo, o_hi;
    const swar::Rand4 rnd = SLIP ? swar::rand_words(R.C.isd_shift, blk.w[0], blk.w[1], blk.w[2], blk.w[3])
                                 : swar::rand_nibble(R.C.isd_shift, (uint32_t)tick & 7u, blk.w[0], blk.w[1], blk.w[2], blk.w[3]);
    swar::reset4<MASKED>(R.C, S, mask4, rnd, o_lo, o_hi);
    *reinterpret_cast<uint32_t*>(sp) = S.ra; *reinterpret_cast<uint32_t*>(sp + 2 * R.state_stride) = S.rb;
    *reinterpret_cast<uint32_t*>(sp + 4 * R.state_stride) = S.ps;
    if (MASKED) {
        *reinterpret_cast<uint32_t*>(sp + R.state_stride) = S.ca; *reinterpret_cast<uint32_t*>(sp + 3 * R.state_stride) = S.cb;
        *reinterpret_cast<uint32_t*>(sp + 5 * R.state_stride) = S.tt;
    }
    if (R.obs) *reinterpret_cast<uint2*>(R.obs + i0) = make_uint2(o_lo, o_hi);
}

// =================================================================================================
// one environment, one step or reset, lowest latency (the single-env facade's path; reference :375-424)
// =================================================================================================
// Inputs arrive BY VALUE as kernel arguments and the results leave as ONE 16-byte store to a host-mapped
// record the host polls: the GPU reads no host memory and the host never enters a stream synchronisation
// (tools/labs/latency_lab.hip: 7.9 us for launch + kernel-written flag + poll against 12.6 us for launch +
// hipStreamSynchronize).  The lane's resident state streams are updated too.
struct ScalarIO {
    uint32_t pos;       // row_a | col_a << 8 | row_b << 16 | col_b << 24
    uint32_t misc;      // poss | t << 8 | act_a << 16 | act_b << 24
    uint32_t op;        // 0 step, 1 reset
    uint32_t seq;       // written to record.x last
    double u_step, u_reset;
    uint4* record;      // host-mapped: { seq, obs | (reward & 0xff) << 16 | term << 24 | trunc << 25 | code << 26,
                        //                next pos (as `pos`), poss | needs_reset << 1 | t << 8 | check << 16 | (seq & 0xff) << 24 }
                        //                check = byte-sum of words 1 and 2 (a torn record is never taken for a complete one)
};

template <bool SLIP>
__global__ __launch_bounds__(64) void scalar_kernel(const KernelParams P, const ScalarIO IO) {
    if (threadIdx.x != 0) return;
    Tables T; T.lut = P.lut; T.nc = P.next_cell; T.isd = P.isd;
    Lane L; StepResult R;
    R.obs = 0u; R.final_obs = 0u; R.reward = 0; R.term = 0u; R.trunc = 0u; R.code = 0u; R.finished = 0u;
    if (IO.op == 1u) {
        uint32_t ob = 0u;
        lane_reset(T, P, L, (uint32_t)(sane_uniform(IO.u_reset) * 4.0), ob);
        R.obs = ob;
    } else {
        L.A = make_pos(IO.pos & 0xffu, (IO.pos >> 8) & 0xffu, P.W);
        L.B = make_pos((IO.pos >> 16) & 0xffu, IO.pos >> 24, P.W);
        L.p = IO.misc & 1u; L.need = 0u; L.t = (IO.misc >> 8) & 0xffu;
        uint32_t a_now = (IO.misc >> 16) & 0xffu, b_now = IO.misc >> 24;
        if (P.policy_a || P.policy_b) {                             // the fixed side acts on the current observation
            const uint32_t s_now = obs_of(T, P, L.A, L.B, L.p);
            if (P.policy_a) a_now = (uint32_t)(uint8_t)P.policy_a[s_now];
            if (P.policy_b) b_now = (uint32_t)(uint8_t)P.policy_b[s_now];
        }
        const double u = sane_uniform(IO.u_step);
        const Draw d{SLIP ? sane_uniform_walk(IO.u_step) : u, (uint32_t)(u * 4.0), (uint32_t)(sane_uniform(IO.u_reset) * 4.0), 0u};
        (void)lane_step<SLIP>(T, P, L, a_now, b_now, d, R);
    }
    uint8_t* sw = P.state;
    sw[0] = (uint8_t)(L.A >> 24); sw[P.state_stride] = (uint8_t)(L.A >> 16);
    sw[2 * P.state_stride] = (uint8_t)(L.B >> 24); sw[3 * P.state_stride] = (uint8_t)(L.B >> 16);
    sw[4 * P.state_stride] = (uint8_t)(L.p | (L.need << 1)); sw[5 * P.state_stride] = (uint8_t)L.t;
    const uint32_t res = R.obs | (((uint32_t)R.reward & 0xffu) << 16) | (R.term << 24) | (R.trunc << 25) | (R.code << 26);
    const uint32_t npos = (L.A >> 24) | (((L.A >> 16) & 0xffu) << 8) | ((L.B >> 24) << 16) | (((L.B >> 16) & 0xffu) << 24);
    __threadfence_system();                 // the resident state before the record
    // one 16-byte store = one write transaction; the sequence number opens it, its low byte closes it and word 3 carries a
    // check byte over words 1 and 2, so the host can tell a complete record from a torn one without a second fence (a second fence would put a PCIe
    // round trip on the critical path: +1.6 us per step, measured)
    uint32_t check = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) check += ((res >> (8 * q)) & 0xffu) + ((npos >> (8 * q)) & 0xffu);
    *IO.record = make_uint4(IO.seq, res, npos, L.p | (L.need << 1) | (L.t << 8) | ((check & 0xffu) << 16) | (IO.seq << 24));
    // the call consumes one tick like every batched_* call; nothing above needed its value (the uniforms are the
    // caller's), so its cache miss stays off the path to the record
    *P.tick_out = *P.tick_in + 1ull;
}

// =================================================================================================
// batched_rollout: T fused steps, state in registers, actions streamed in, trajectories streamed out
// =================================================================================================
// DYN = false: both action streams come from memory (the trajectory collector); the code for in-kernel
// sampling, mixed policies and the fixed-policy gather is compiled out.
template <int E, bool SLIP, bool DYN>
__device__ __forceinline__ void rollout_group(const Tables& T, const KernelParams& P, const RolloutIO& IO,
                                              unsigned long long i0, unsigned long long tick0,
                                              HistAcc<false>& hist, bool& any_misuse) {
    LaneVec<E> S; S.load(P, i0);
    int32_t ret[E], eps[E];
    uint32_t nonzero = 0u;                  // number of steps of this thread's lanes that carried a reward
#pragma unroll
    for (int j = 0; j < E; ++j) { ret[j] = 0; eps[j] = 0; }
    PackB<E> aa, ab; aa.clear(); ab.clear();
    const bool sample = DYN && IO.sample_actions;
    uint32_t bad_act = 0u;
    if (!sample) {
        if (!DYN || IO.act_a) aa.load_nt(IO.act_a, i0);
        if (!DYN || IO.act_b) ab.load_nt(IO.act_b, i0);
        bad_act |= canon_pack(aa) | canon_pack(ab);
    }
    // the observation of the current tuple is carried along when an action depends on it
    const bool fixed = DYN && (P.policy_a != nullptr || P.policy_b != nullptr ||    // single-agent mode
                               (sample && (IO.mix_a != nullptr || IO.mix_b != nullptr)));
    uint32_t s_now[E];
#pragma unroll
    for (int j = 0; j < E; ++j) s_now[j] = fixed ? obs_of(T, P, S.L[j].A, S.L[j].B, S.L[j].p) : 0u;
    for (int s = 0; s < IO.n_steps; ++s) {
        const unsigned long long tick = tick0 + (unsigned long long)s;
        PackB<E> naa = aa, nab = ab;
        if (!sample && s + 1 < IO.n_steps) {                            // prefetch the next step's actions
            if (!DYN || IO.act_a) naa.load_nt(IO.act_a + (long long)(s + 1) * IO.act_stride, i0);
            if (!DYN || IO.act_b) nab.load_nt(IO.act_b + (long long)(s + 1) * IO.act_stride, i0);
            bad_act |= canon_pack(naa) | canon_pack(nab);
        }
        uint32_t words[E], awords[E];
        lane_words<E>(P, P.lane_offset + i0, block_tick<SLIP>(tick), 0u, words);
        if (sample) lane_words<E>(P, P.lane_offset + i0, tick, 1u, awords);
        PackB<E> o_rew, o_term, o_trunc, o_code; PackH<E> o_obs, o_fin;
        o_rew.clear(); o_term.clear(); o_trunc.clear(); o_obs.clear(); o_code.clear(); o_fin.clear();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const Draw d = draw_from_word<SLIP>(words[j], tick);
            uint32_t a = aa.get(j), b = ab.get(j);
            if (sample) {                               // two actions from one 32-bit word, 15 bits each
                const uint32_t ha = awords[j] & 0x7fffu, hb = (awords[j] >> 16) & 0x7fffu;
                a = (ha * 5u) >> 15;                    // uniform
                b = (hb * 5u) >> 15;
                if (IO.mix_a) {                         // mixed policy: first action whose cumulative threshold exceeds the draw
                    const uint2 th = *reinterpret_cast<const uint2*>(IO.mix_a + 4u * s_now[j]);
                    a = (ha >= (th.x & 0xffffu)) + (ha >= (th.x >> 16)) + (ha >= (th.y & 0xffffu)) + (ha >= (th.y >> 16));
                }
                if (IO.mix_b) {
                    const uint2 th = *reinterpret_cast<const uint2*>(IO.mix_b + 4u * s_now[j]);
                    b = (hb >= (th.x & 0xffffu)) + (hb >= (th.x >> 16)) + (hb >= (th.y & 0xffffu)) + (hb >= (th.y >> 16));
                }
            }
            if (fixed) {
                if (P.policy_a) a = (uint32_t)(uint8_t)P.policy_a[s_now[j]];
                if (P.policy_b) b = (uint32_t)(uint8_t)P.policy_b[s_now[j]];
            }
            StepResult R;
            any_misuse |= lane_step<SLIP, true>(T, P, S.L[j], a, b, d, R);
            if (DYN) s_now[j] = R.obs;
            o_obs.put(j, R.obs); o_rew.put(j, (uint32_t)R.reward & 0xffu); o_term.put(j, R.term); o_trunc.put(j, R.trunc);
            o_fin.put(j, R.final_obs); o_code.put(j, R.code);
            ret[j] += R.reward; eps[j] += (int32_t)R.finished; nonzero += (uint32_t)R.reward & 1u;
        }
        const long long off = (long long)s * IO.out_stride;
        if (IO.obs) o_obs.store_nt(IO.obs + off, i0);
        if (IO.reward) o_rew.store_nt(IO.reward + off, i0);
        if (IO.terminated) o_term.store_nt(IO.terminated + off, i0);
        if (IO.truncated) o_trunc.store_nt(IO.truncated + off, i0);
        if (IO.final_obs) o_fin.store_nt(IO.final_obs + off, i0);
        if (IO.prob_code) o_code.store_nt(IO.prob_code + off, i0);
        aa = naa; ab = nab;
    }
    S.store(P, i0);
    {
        int32_t rsum = 0; uint32_t fsum = 0u;
#pragma unroll
        for (int j = 0; j < E; ++j) { rsum += ret[j]; fsum += (uint32_t)eps[j]; }
        hist.add_totals(fsum, rsum, nonzero);
    }
    if (IO.return_sum) add_words<E>(IO.return_sum, i0, ret);
    if (IO.episode_count) add_words<E>(IO.episode_count, i0, eps);
    if (bad_act) P.misuse[1] = 1u;
}

template <int E, bool SLIP, bool LUT_LDS, bool DYN>
__global__ __launch_bounds__(kBlock) void rollout_kernel(const KernelParams P, const RolloutIO IO) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    HistAcc<false> hist; hist.init(P);
    const Tables T = stage_tables<LUT_LDS>(P, smem);
    const unsigned long long tick0 = *P.tick_in;
    if (P.tick_out) publish_tick(P, tick0, (unsigned long long)IO.n_steps);   // nullptr: the tail of a launch that already did
    const unsigned long long groups = (P.n + E - 1) / E;        // the launch covers lanes [first, first + n) of the handle
    bool any_misuse = false;
    for (unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; g < groups;
         g += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long rel = g * E, i0 = P.first + rel;
        if (E == 1 || rel + E <= P.n) {
            rollout_group<E, SLIP, DYN>(T, P, IO, i0, tick0, hist, any_misuse);
        } else {
            for (unsigned long long i = rel; i < P.n; ++i)
                rollout_group<1, SLIP, DYN>(T, P, IO, P.first + i, tick0, hist, any_misuse);
        }
    }
    if (any_misuse) P.misuse[0] = 1u;
    hist.flush(P);
}

// =================================================================================================
// batched_rollout, byte-parallel: T fused steps with the four lanes of a thread packed in six registers
// =================================================================================================
// The step of soccer_swar.hpp in a loop: no rule table, no LDS transition table (so every pitch that fits the byte
// arithmetic — all golden ones up to 11x7 — and every slip whose integer decision is exact take the same kernel), no
// state-code conversion on entry / exit, frozen and goal-tuple lanes handled by the step itself.  Per step a thread
// issues one Philox block, two action dwords (prefetched a step ahead) and four result stores.
//   DYN: some action is produced in the kernel — sampled uniformly or from [nS][4] mixed-policy thresholds (config 5),
//        or looked up from a fixed int8[nS] policy (single-agent mode); these are per-lane gathers keyed by the lane's
//        current observation, which the step already produces.  The tables sit in LDS when they fit (`lds_tables`).
struct RolloutSwar {       // everything the kernel needs, and nothing else (KernelParams is twice this: SGPR spills)
    uint8_t* state; unsigned long long state_stride;
    unsigned long long first, n, lane_offset;
    const unsigned long long* tick_in; unsigned long long* tick_out;
    unsigned long long* hist; unsigned int* misuse;
    const int8_t* policy_a; const int8_t* policy_b;
    uint32_t key0, key1;
    swar::Consts C; swar::SlipConsts L; const swar::Quad* sub;
    uint32_t hist_mask;
    int32_t nS; int32_t lds_tables;
    uint32_t act_off;                       // dword offset of the action staging area in dynamic LDS (16 x 256 dwords per workgroup)
    uint32_t tab_off;                       // dword offset of the mixed-policy / fixed-policy tables in dynamic LDS
    const uint32_t* slip_lut;               // SLIPM == 2: SlipTables::lut (kSlipBuckets bytes) followed by SlipTables::T
};

// where the slip selection of a byte-parallel kernel reads its thresholds: SLIPM == 1 the nine rows of quarter points
// (compared one by one, for the slips whose thresholds crowd a table bucket), SLIPM == 2 the bucket table + the ascending
// threshold list (swar::slip_select4_lut)
struct SlipSrc { const swar::Quad* sub; const uint8_t* lut; const uint32_t* T; };
constexpr int kSlipLutWords = 4096 + 40;        // = soccer::kSlipLdsWords (soccer_slip.hpp is host-only)

// A mixed-policy row holds four 16-bit cumulative thresholds t0 <= t1 <= t2 <= t3 (values 0..2^15) as two dwords; the
// action is the number of them that are <= the player's 15-bit draw h.  With `hs` = h in both halves and bit 15 set,
// (h + 0x8000) - t has bit 15 set exactly when h >= t: two packed subtractions put the four answers into the sign bits
// of bytes 1, 3, 5, 7 of an 8-byte pair, which is what v_perm_b32's selectors 8..11 replicate — one permute turns them
// into four 0xff / 0x00 bytes and one population count gives 8 x the action.
__device__ __forceinline__ uint32_t count8_le15(uint32_t hs, uint32_t tx, uint32_t ty) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const uint32_t x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, hs) - __builtin_bit_cast(u16x2, tx));
    const uint32_t y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, hs) - __builtin_bit_cast(u16x2, ty));
    return (uint32_t)__builtin_popcount(swar::perm(y, x, 0x0b0a0908u));
}
// the two 15-bit action draws of a lane's purpose-1 word w (da = w & 0x7fff, db = (w >> 16) & 0x7fff): `wm` = w with
// bits 15 and 31 forced, each half then duplicated by one byte permute
__device__ __forceinline__ uint32_t draw_a15(uint32_t wm) { return swar::perm(0u, wm, 0x01000100u); }
__device__ __forceinline__ uint32_t draw_b15(uint32_t wm) { return swar::perm(0u, wm, 0x03020302u); }

// the T steps of one thread's four lanes.  GENERAL = false: no lane is frozen or in a goal tuple on entry and the handle
// auto-resets, so none ever will be (the steady state): the step's code for those cases is compiled out.
// DYNM — where the actions come from: 0 both from the action streams; 1 both sampled uniformly in the kernel; 2 both
// sampled from mixed-policy tables staged in LDS as one 16-byte row per state (config 5); 4 / 5 player A / B follows its
// fixed policy and the other side's actions are streamed (single-agent mode); 3 anything else (a table on one side only,
// tables too big for LDS, a fixed policy against a sampled side ...: decided by wave-uniform run-time tests).  The common
// shapes are instantiations of their own because every optional pointer that stays live costs scalar registers, and the
// loop of the catch-all form spilled them (60-300 v_readlane_b32 per step).
// Randomness (include/soccer_hip.h): with SLIP one step/reset block per tick; without, one block per EIGHT ticks — the
// thread keeps it transposed (swar::transpose4) in p0..p3, p0 serving the current pair of ticks — which takes the Philox
// rounds from ~45 to ~6 vector instructions per step; sampled actions take the lane's word of the tick's purpose-1 block.
template <int DYNM, int SLIPM, bool GENERAL, int GEO, bool FULL = false>
__device__ __forceinline__ void rollout_swar_group(const RolloutSwar& R, const RolloutIO& IO, const SlipSrc& slip,
                                                   const uint2* mix_a_in, const uint2* mix_b_in, const int8_t* pol_a_in, const int8_t* pol_b_in,
                                                   uint32_t* act_lds,
                                                   uint32_t i0, unsigned long long tick0, swar::Group& S,
                                                   uint32_t& fin_tot, uint32_t& nz_tot, uint32_t& neg_tot,
                                                   uint32_t (&acc)[4], uint32_t& frozen_any, uint32_t& bad_any) {
    constexpr bool DYN = DYNM != 0;
    constexpr bool SLIP = SLIPM != 0;
    constexpr bool STAGED = DYNM == 0 || DYNM == 4 || DYNM == 5;        // action streams staged through LDS, eight steps at a time
    constexpr bool TRUSTED = DYNM == 1 || DYNM == 2;                    // both sides sampled in 0..4 by the kernel itself
    const bool sample = DYNM == 1 || DYNM == 2 || (DYNM == 3 && IO.sample_actions);
    const uint4* mix_ab = DYNM == 2 ? reinterpret_cast<const uint4*>(mix_a_in) : nullptr;   // LDS rows { a: x, y; b: z, w }
    const uint2* mix_a = DYNM == 3 ? mix_a_in : nullptr;
    const uint2* mix_b = DYNM == 3 ? mix_b_in : nullptr;
    const int8_t* pol_a = DYNM == 3 || DYNM == 4 ? pol_a_in : nullptr;
    const int8_t* pol_b = DYNM == 3 || DYNM == 5 ? pol_b_in : nullptr;
    const bool use_pol_a = DYNM == 4 || (DYNM == 3 && pol_a != nullptr), use_pol_b = DYNM == 5 || (DYNM == 3 && pol_b != nullptr);
    const bool use_mix_a = DYNM == 3 && sample && mix_a != nullptr;
    const bool use_mix_b = DYNM == 3 && sample && mix_b != nullptr;
    const bool load_a = DYNM == 0 || DYNM == 5 || (DYNM == 3 && !sample && IO.act_a != nullptr);
    const bool load_b = DYNM == 0 || DYNM == 4 || (DYNM == 3 && !sample && IO.act_b != nullptr);
    const bool lane_acc = IO.return_sum != nullptr || IO.episode_count != nullptr;
    const bool by_obs = DYNM == 2 || DYNM == 4 || DYNM == 5 || (DYNM == 3 && (use_pol_a || use_pol_b || use_mix_a || use_mix_b));
    // Action streams.  A wave's loads and stores share one completion counter and may complete out of order with
    // respect to each other, so waiting for ONE prefetched action dword means waiting for every result store issued
    // before it: with a load per step the wave drained its stores every step and sat out their write latency (the
    // step took 1.5 us of which the SIMD was busy 1.1).  Instead the action dwords of eight steps — the ticks of one
    // Philox block — are fetched a block ahead into registers, parked in the thread's sixteen private LDS dwords at
    // the block boundary (the one wait per eight steps) and read back per step by ds_read, which counts separately.
    // Plain loads, not non-temporal ones: re-read or streamed, the action rows come in faster without the hint (T = 100, 2^20
    // lanes: 7.0 - 7.3 against 6.5 - 6.9 x 10^11 env-steps/s when the 200 MB block is re-read, 6.4 against 6.35 when six blocks are
    // visited in turn; tools/labs/rollout_stream_lab.py) — unlike the single step's (step_kernel_swar, SOCCER_F_STREAM_ACTIONS).
    uint32_t aa = 0u, ab = 0u;
    uint32_t nx[16];                                                    // STAGED: the next block's action dwords, in flight
#pragma unroll
    for (int k = 0; k < 16; ++k) nx[k] = 0u;
    // issue the loads of the block whose tick-0 step is `sb` (steps outside the rollout are clamped: a harmless re-read)
    auto fetch = [&](int sb) {
        uint32_t f0 = i0; asm volatile("" : "+v"(f0));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int st = sb + k; st = st < 0 ? 0 : st; st = st < IO.n_steps ? st : IO.n_steps - 1;
            // The step's rows are held in scalar registers: passed through an empty asm, else the optimiser folds the row offset
            // into the per-thread address and pays a 64-bit vector multiply-add per load (16 per block of eight steps).  What
            // comes out of an asm statement is a generic pointer unless its type says global memory (flat_load otherwise).
            typedef const uint8_t __attribute__((address_space(1)))* gbytes;
            typedef const uint32_t __attribute__((address_space(1)))* gwords;
            gbytes row_a = (gbytes)(IO.act_a + (long long)st * IO.act_stride);
            gbytes row_b = (gbytes)(IO.act_b + (long long)st * IO.act_stride);
            asm volatile("" : "+s"(row_a)); asm volatile("" : "+s"(row_b));
            if (load_a) nx[2 * k] = *(gwords)(row_a + f0);
            if (load_b) nx[2 * k + 1] = *(gwords)(row_b + f0);
        }
    };
    if (STAGED) fetch(-(int)((uint32_t)tick0 & 7u));
    else {
        if (load_a) aa = *reinterpret_cast<const uint32_t*>(IO.act_a + i0);
        if (load_b) ab = *reinterpret_cast<const uint32_t*>(IO.act_b + i0);
    }
    // the observation of the current tuple (goal tuples: 0), carried along when an action depends on it
    uint32_t s_lo = 0u, s_hi = 0u;
    if (by_obs) {
        const uint32_t cc0 = swar::bfi(swar::mask_of(S.ps << 7), S.cb, S.ca);
        swar::obs4<true>(R.C, S.ra, S.ca, S.rb, S.cb, S.ps & swar::K01, swar::is_zero(cc0) | swar::is_zero(cc0 ^ R.C.Wm1x4), s_lo, s_hi);
    }
    const swar::Consts& C = R.C;
    const unsigned long long q = (R.lane_offset + i0) >> 2;
    uint32_t fin_loc = 0u, nz_loc = 0u, neg_loc = 0u;                   // this group's finished episodes / steps with a reward / see below
    uint32_t p0 = 0u, p1 = 0u, p2 = 0u, p3 = 0u;                        // !SLIP: the current eight-tick block, transposed
    if (!SLIP) {
        const unsigned long long bt = tick0 >> 3;
        const Philox4 b = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)bt, (uint32_t)(bt >> 32), R.key0, R.key1);
        swar::transpose4(b.w[0], b.w[1], b.w[2], b.w[3], p0, p1, p2, p3);
        for (uint32_t r = ((uint32_t)tick0 & 7u) >> 1; r != 0u; --r) { p0 = p1; p1 = p2; p2 = p3; }   // wave-uniform
    }
    for (int s = 0; s < IO.n_steps; ++s) {
        const unsigned long long tick = tick0 + (unsigned long long)s;
        const uint32_t t = (uint32_t)tick & 7u;                         // wave-uniform, like everything that steers the blocks below
        const bool new_block = t == 0u || s == 0;
        uint32_t naa = aa, nab = ab;
        if (STAGED) {
            if (new_block) {                                            // park this block's actions, fetch the next block's
#pragma unroll
                for (int k = 0; k < 16; ++k) if ((k & 1) ? load_b : load_a) act_lds[k * kBlock] = nx[k];
                if (s + 8 - (int)t < IO.n_steps) fetch(s + 8 - (int)t);
            }
            if (load_a) aa = act_lds[(2u * t) * kBlock];
            if (load_b) ab = act_lds[(2u * t + 1u) * kBlock];
        } else if (s + 1 < IO.n_steps) {                                // DYNM == 3: prefetch the next step's actions
            if (load_a) naa = *reinterpret_cast<const uint32_t*>(IO.act_a + (long long)(s + 1) * IO.act_stride + i0);
            if (load_b) nab = *reinterpret_cast<const uint32_t*>(IO.act_b + (long long)(s + 1) * IO.act_stride + i0);
        }
        uint32_t a4 = aa, b4 = ab;
        if (DYN) {
            uint32_t aw[4] = {0u, 0u, 0u, 0u};
            if (sample) {
                const Philox4 ab_blk = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)tick, (uint32_t)(tick >> 32) | 0x80000000u, R.key0, R.key1);
                aw[0] = ab_blk.w[0]; aw[1] = ab_blk.w[1]; aw[2] = ab_blk.w[2]; aw[3] = ab_blk.w[3];
                a4 = 0u; b4 = 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t ob = ((j & 2 ? s_hi : s_lo) >> (16 * (j & 1))) & 0xffffu;
                if (DYNM == 2) {                    // both sides from their tables: one 16-byte LDS row per lane
                    const uint4 th = mix_ab[ob];
                    const uint32_t wm = aw[j] | 0x80008000u;
                    a4 |= count8_le15(draw_a15(wm), th.x, th.y) << (8 * j);      // 8 x the action; divided after the loop
                    b4 |= count8_le15(draw_b15(wm), th.z, th.w) << (8 * j);
                } else if (sample) {                // two actions from one 32-bit word, 15 bits each
                    const uint32_t ha = aw[j] & 0x7fffu, hb = (aw[j] >> 16) & 0x7fffu;
                    uint32_t a = (ha * 5u) >> 15, b = (hb * 5u) >> 15;          // uniform
                    const uint32_t wm = aw[j] | 0x80008000u;
                    if (use_mix_a) { const uint2 th = mix_a[ob]; a = count8_le15(draw_a15(wm), th.x, th.y) >> 3; }
                    if (use_mix_b) { const uint2 th = mix_b[ob]; b = count8_le15(draw_b15(wm), th.x, th.y) >> 3; }
                    a4 |= a << (8 * j); b4 |= b << (8 * j);
                }
                if (use_pol_a) a4 = (a4 & ~(0xffu << (8 * j))) | ((uint32_t)(uint8_t)pol_a[ob] << (8 * j));
                if (use_pol_b) b4 = (b4 & ~(0xffu << (8 * j))) | ((uint32_t)(uint8_t)pol_b[ob] << (8 * j));
            }
            if (DYNM == 2) { a4 >>= 3; b4 >>= 3; }  // every byte held 8 x (0..4): no bit crosses a byte
        }
        swar::Out o;
        uint32_t sa = 0u, sb = 0u, cls4 = 0u;
        swar::Rand4 rnd;
        if (SLIP) {
            const Philox4 blk = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)tick, (uint32_t)(tick >> 32), R.key0, R.key1);
            uint32_t k4 = 0u;
            const uint32_t ca4 = TRUSTED ? a4 : swar::canon4(a4), cb4 = TRUSTED ? b4 : swar::canon4(b4);
            if (SLIPM == 2) swar::slip_select4_lut(slip.lut, slip.T, R.L.c_off, ca4, cb4, blk.w[0], blk.w[1], blk.w[2], blk.w[3], sa, sb, k4, cls4);
            else swar::slip_select4(R.L, slip.sub, ca4, cb4, blk.w[0], blk.w[1], blk.w[2], blk.w[3], sa, sb, k4, cls4);
            rnd = swar::Rand4{k4 << 6, swar::pack_byte0(blk.w[0], blk.w[1], blk.w[2], blk.w[3]) >> C.isd_shift};
        } else {
            if (t == 0u && s != 0) {
                const unsigned long long bt = tick >> 3;
                // (the key through an empty asm: the ten round keys are then derived here, by scalar adds every eighth step, instead
                // of living in twenty scalar registers across the loop — which spilled to vector lanes and came back by v_readlane)
                uint32_t k0 = R.key0, k1 = R.key1; asm volatile("" : "+s"(k0), "+s"(k1));
                const Philox4 b = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)bt, (uint32_t)(bt >> 32), k0, k1);
                swar::transpose4(b.w[0], b.w[1], b.w[2], b.w[3], p0, p1, p2, p3);
            }
            rnd = swar::rand_pair(C.isd_shift, t, p0);
            // odd tick: the pair is used up.  The empty asm keeps this a (scalar) branch around three moves; as selects it
            // was three v_cndmask_b32 every step, each several times the cost of a move (tools/labs/valu_rate_lab.hip).
            if (t & 1u) { asm volatile(""); p0 = p1; p1 = p2; p2 = p3; }
        }
        swar::step4<GENERAL, FULL, SLIP, GEO, TRUSTED>(C, S, a4, b4, sa, sb, cls4, rnd, o);
        s_lo = o.obs_lo; s_hi = o.obs_hi;
        // the step's row of every stream as a uniform base (scalar registers) + this thread's 32-bit byte offset: stores of the
        // form v_off, data, s[base] (the offset passes through an empty asm per step, else the optimiser keeps one 64-bit
        // per-thread address per stream across the loop and adds the row to it with vector instructions)
        const long long row = (long long)s * IO.out_stride;
        uint32_t j0 = i0; asm volatile("" : "+v"(j0));
        if (IO.obs) __builtin_nontemporal_store((unsigned long long)o.obs_lo | ((unsigned long long)o.obs_hi << 32),
                                                reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(IO.obs + row) + (j0 << 1)));
        if (IO.reward) __builtin_nontemporal_store(o.rew, reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(IO.reward + row) + j0));
        if (IO.terminated) __builtin_nontemporal_store(o.term, reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(IO.terminated + row) + j0));
        if (IO.truncated) __builtin_nontemporal_store(o.trunc, reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(IO.truncated + row) + j0));
        if (FULL) {     // batched_rollout_ex: what gym's vector convention reports per step next to the four streams
            if (IO.final_obs) __builtin_nontemporal_store((unsigned long long)o.fin_lo | ((unsigned long long)o.fin_hi << 32),
                                                          reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(IO.final_obs + row) + (j0 << 1)));
            if (IO.prob_code) __builtin_nontemporal_store(o.code, reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(IO.prob_code + row) + j0));
        }
        // finished episodes by return: a reward byte is 0x01 / 0xff only on the step that ends an episode
        fin_loc += (uint32_t)__builtin_popcount(o.finished & swar::K80);
        if (GENERAL) { nz_loc += (uint32_t)__builtin_popcount(o.rew & swar::K01); neg_loc += (uint32_t)__builtin_popcount(o.rew & swar::K80); }
        else {
            // without frozen / goal-tuple lanes a step terminates exactly when it carries a reward, so the clean 0 / 1 bytes of
            // `terminated` count the rewards and the bits of the reward bytes (0x01 / 0xff) count (+1) + 8 x (-1): two
            // population counts without a mask
            nz_loc += (uint32_t)__builtin_popcount(o.term); neg_loc += (uint32_t)__builtin_popcount(o.rew);
        }
        if (lane_acc) {                                             // wave-uniform
            // reward bytes sign-extended to int16 pairs (v_perm_b32's sign selectors), finished flags to 0 / 1
            acc[0] = swar::pk_add(acc[0], swar::perm(o.rew << 8, o.rew, 0x08010a00u));
            acc[1] = swar::pk_add(acc[1], swar::perm(o.rew << 8, o.rew, 0x09030b02u));
            const uint32_t f01 = swar::one_of(o.finished);
            acc[2] += swar::perm(0u, f01, 0x0c010c00u); acc[3] += swar::perm(0u, f01, 0x0c030c02u);   // <= 4096 < 2^16: no carry
        }
        if (GENERAL) frozen_any |= o.frozen;
        if (!TRUSTED) bad_any |= o.bad_action;
        if (!STAGED) { aa = naa; ab = nab; }
    }
    fin_tot += fin_loc; nz_tot += nz_loc;
    neg_tot += GENERAL ? neg_loc : (neg_loc - nz_loc) / 7u;             // (pos + 8 neg) - (pos + neg) = 7 neg
}

// FULL: also the per-step final_obs / prob_code trajectories (batched_rollout_ex; +3 B per env-step and the second observation index)
template <int DYNM, int SLIPM, int GEO = 0, bool FULL = false>
__global__ __launch_bounds__(kBlock) void rollout_swar_kernel(const RolloutSwar R, const RolloutIO IO) {
    constexpr bool SLIP = SLIPM != 0;
    constexpr bool DYN = DYNM >= 2;          // the forms that look something up by the observation
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    HistAcc<false> hist; hist.init_at(R.hist, R.hist_mask);
    // LDS: the slip thresholds — SLIPM == 1: dynamic [0, 36) the nine rows; SLIPM == 2: a STATIC array (its address is an
    // immediate of the ds_read, not an add per lane) holding the bucket table and the ascending list — then, dynamic from
    // R.tab_off (DYN, when they fit), the mixed-policy rows — DYNM == 2: one 16-byte row { a's
    // four thresholds, b's four } per state; DYNM == 3: mix_a rows, then mix_b rows (8 B per state) — and the two fixed
    // policies (1 B per state), then from R.act_off the action staging area
    SlipSrc slip{R.sub, nullptr, nullptr};
    const uint2* mix_a = reinterpret_cast<const uint2*>(IO.mix_a);
    const uint2* mix_b = reinterpret_cast<const uint2*>(IO.mix_b);
    const int8_t* pol_a = DYNM == 3 || DYNM == 4 ? R.policy_a : nullptr; const int8_t* pol_b = DYNM == 3 || DYNM == 5 ? R.policy_b : nullptr;
    const bool sample = DYNM == 2 || (DYNM == 3 && IO.sample_actions);
    if (SLIP || (DYN && R.lds_tables)) {
        if (SLIPM == 1) { if (threadIdx.x < 36) smem[threadIdx.x] = reinterpret_cast<const uint32_t*>(R.sub)[threadIdx.x];
                          slip.sub = reinterpret_cast<const swar::Quad*>(smem); }
        if (SLIPM == 2) {
            __shared__ __attribute__((aligned(16))) uint32_t s_slip[SLIPM == 2 ? kSlipLutWords : 4];
            for (int i = threadIdx.x; i < kSlipLutWords; i += kBlock) s_slip[i] = R.slip_lut[i];
            slip.lut = reinterpret_cast<const uint8_t*>(s_slip); slip.T = s_slip + kSlipLutWords - 40;
        }
        if (DYNM == 2) {                     // the host picks this shape only when the rows fit
            uint4* lab = reinterpret_cast<uint4*>(smem + R.tab_off);
            for (int i = threadIdx.x; i < R.nS; i += kBlock) { const uint2 xa = mix_a[i], xb = mix_b[i]; lab[i] = make_uint4(xa.x, xa.y, xb.x, xb.y); }
            mix_a = reinterpret_cast<const uint2*>(lab); mix_b = nullptr;
        } else if (DYN && R.lds_tables) {
            uint2* la = reinterpret_cast<uint2*>(smem + R.tab_off); uint2* lb = la + R.nS;
            int8_t* pa = reinterpret_cast<int8_t*>(lb + R.nS); int8_t* pb = pa + ((R.nS + 15) & ~15);
            if (sample && mix_a) { for (int i = threadIdx.x; i < R.nS; i += kBlock) la[i] = mix_a[i]; mix_a = la; }
            if (sample && mix_b) { for (int i = threadIdx.x; i < R.nS; i += kBlock) lb[i] = mix_b[i]; mix_b = lb; }
            if (pol_a) { for (int i = threadIdx.x; i < R.nS; i += kBlock) pa[i] = pol_a[i]; pol_a = pa; }
            if (pol_b) { for (int i = threadIdx.x; i < R.nS; i += kBlock) pb[i] = pol_b[i]; pol_b = pb; }
        }
        __syncthreads();
    }
    const unsigned long long tick0 = *R.tick_in;
    if (R.tick_out && blockIdx.x == 0 && threadIdx.x == 0) *R.tick_out = tick0 + (unsigned long long)IO.n_steps;
    const unsigned long long groups = R.n >> 2;                      // the launch covers a multiple of 4 lanes
    uint32_t frozen_any = 0u, bad_any = 0u;
    uint32_t* act_lds = smem + R.act_off + threadIdx.x;              // this thread's sixteen dwords, kBlock apart
    for (unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; g < groups;
         g += (unsigned long long)gridDim.x * kBlock) {
        // 32-bit byte offsets: the host hands the kernel at most kSwarLaunchLanes lanes at a time (like step_kernel_swar)
        const uint32_t i0 = (uint32_t)R.first + ((uint32_t)g << 2);
        const uint8_t* sp = R.state + i0;
        swar::Group S;
        S.ra = *reinterpret_cast<const uint32_t*>(sp); S.ca = *reinterpret_cast<const uint32_t*>(sp + R.state_stride);
        S.rb = *reinterpret_cast<const uint32_t*>(sp + 2 * R.state_stride); S.cb = *reinterpret_cast<const uint32_t*>(sp + 3 * R.state_stride);
        S.ps = *reinterpret_cast<const uint32_t*>(sp + 4 * R.state_stride); S.tt = *reinterpret_cast<const uint32_t*>(sp + 5 * R.state_stride);
        uint32_t fin_tot = 0u, nz_tot = 0u, neg_tot = 0u;
        uint32_t acc[4] = {0u, 0u, 0u, 0u};     // per lane: int16 return (two pairs), uint16 finished episodes (two pairs); T <= 4096
        // any lane frozen, any player in a goal column (= a goal tuple), or no auto-reset: the general step
        const uint32_t edge = swar::is_zero(S.ca) | swar::is_zero(S.cb) | swar::is_zero(S.ca ^ R.C.Wm1x4) | swar::is_zero(S.cb ^ R.C.Wm1x4);
        const bool special = R.C.autoreset == 0u || (((S.ps << 6) | edge) & swar::K80) != 0u;
        if (special) rollout_swar_group<DYNM, SLIPM, true, GEO, FULL>(R, IO, slip, mix_a, mix_b, pol_a, pol_b, act_lds, i0, tick0, S, fin_tot, nz_tot, neg_tot, acc, frozen_any, bad_any);
        else rollout_swar_group<DYNM, SLIPM, false, GEO, FULL>(R, IO, slip, mix_a, mix_b, pol_a, pol_b, act_lds, i0, tick0, S, fin_tot, nz_tot, neg_tot, acc, frozen_any, bad_any);
        uint8_t* sw = R.state + i0;
        *reinterpret_cast<uint32_t*>(sw) = S.ra; *reinterpret_cast<uint32_t*>(sw + R.state_stride) = S.ca;
        *reinterpret_cast<uint32_t*>(sw + 2 * R.state_stride) = S.rb; *reinterpret_cast<uint32_t*>(sw + 3 * R.state_stride) = S.cb;
        *reinterpret_cast<uint32_t*>(sw + 4 * R.state_stride) = S.ps; *reinterpret_cast<uint32_t*>(sw + 5 * R.state_stride) = S.tt;
        hist.add_totals(fin_tot, (int32_t)nz_tot - 2 * (int32_t)neg_tot, nz_tot);
        if (IO.return_sum != nullptr || IO.episode_count != nullptr) {
            int32_t ret[4] = {(int32_t)(int16_t)(acc[0] & 0xffffu), (int32_t)(int16_t)(acc[0] >> 16), (int32_t)(int16_t)(acc[1] & 0xffffu), (int32_t)(int16_t)(acc[1] >> 16)};
            int32_t eps[4] = {(int32_t)(acc[2] & 0xffffu), (int32_t)(acc[2] >> 16), (int32_t)(acc[3] & 0xffffu), (int32_t)(acc[3] >> 16)};
            if (IO.return_sum) add_words<4>(IO.return_sum, i0, ret);
            if (IO.episode_count) add_words<4>(IO.episode_count, i0, eps);
        }
    }
    if (frozen_any) R.misuse[0] = 1u;
    if (bad_any) R.misuse[1] = 1u;
    hist.flush_at(R.hist, R.hist_mask);
}

// =================================================================================================
// transition-table export: what the reference's constructor materialises as P_readable (:167-293)
// =================================================================================================
constexpr int kMaxOutcomes = 36;         // 9 slip combinations x up to 4 collision outcomes

struct EnumIO {
    int32_t* count;        // [n_tuples*25]   entries in the list, -1 for unreachable tuples (no key)
    double* prob;          // [n_tuples*25*36]
    int32_t* next;         // [n_tuples*25*36] flat tuple index of the next state
    int8_t* reward;        // [n_tuples*25*36] player A's reward
    uint8_t* done;         // [n_tuples*25*36]
    int32_t n_tuples, H;
};

// One thread per (state tuple, joint action): the ordered outcome list exactly as the reference builds
// it — combinations in order, zero weights dropped, collision outcomes in order, p = weight * outcome
// probability — using the same rule functions (moved / classify / pick) as the step kernels.
__global__ __launch_bounds__(kBlock) void enumerate_kernel(const KernelParams P, const EnumIO IO) {
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (long long)IO.n_tuples * 25) return;
    const int f = (int)(gid / 25), ja = (int)(gid % 25);
    const uint32_t aa = (uint32_t)(ja / 5), ab = (uint32_t)(ja % 5);
    int r = f;
    const uint32_t p = r & 1; r >>= 1;
    const uint32_t cb = r % P.W; r /= P.W;
    const uint32_t rb = r % IO.H; r /= IO.H;
    const uint32_t ca = r % P.W; const uint32_t ra = r / P.W;
    const uint32_t lut = P.lut[f];
    if (lut == 0xFFFFu) { IO.count[gid] = -1; return; }            // unreachable: the reference has no key (:179-180)
    Tables T; T.lut = P.lut; T.nc = P.next_cell; T.isd = P.isd;
    const uint32_t A = make_pos(ra, ca, P.W), B = make_pos(rb, cb, P.W);
    const bool in_goal = lut == 0u;                                  // goal tuple (:300-301)
    constexpr int VA[9] = {0, 0, 0, 1, 2, 1, 1, 2, 2};
    constexpr int VB[9] = {0, 1, 2, 0, 0, 1, 2, 1, 2};
    constexpr int CLS[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
    const uint32_t Wm1 = (uint32_t)(P.W - 1);
    int n = 0;
    const long long base = gid * kMaxOutcomes;
    for (int c = 0; c < 9; ++c) {
        const double wgt = P.w[CLS[c]];
        if (wgt == 0.0) continue;                                    // :226-227
        if (in_goal) {                                               // absorbing self-loop, done, reward 0 (:235-236)
            IO.prob[base + n] = wgt * 1.0; IO.next[base + n] = f; IO.reward[base + n] = 0; IO.done[base + n] = 1; ++n;
            continue;
        }
        const uint32_t nA = moved(T, P, A, p ^ 1u, slip_move(aa, VA[c])), nB = moved(T, P, B, p, slip_move(ab, VB[c]));
        const Resolved R = classify(A, B, nA, nB, aa, ab);
        const int cnt = R.kind == K_COIN ? 2 : (R.kind == K_FOUR ? 4 : 1);
        const double q = cnt == 1 ? 1.0 : (cnt == 2 ? 0.5 : 0.25);
        for (int k = 0; k < cnt; ++k) {
            const Outcome o = pick(A, B, p, R, (uint32_t)k);
            const uint32_t ncc = col_of(o.p ? o.B : o.A);
            const bool goal = (ncc == 0u) | (ncc == Wm1);
            const int nf = (int)((((o.A >> 24) * (uint32_t)P.W + ((o.A >> 16) & 0xffu)) * (uint32_t)IO.H + (o.B >> 24)) * (uint32_t)P.W +
                                 ((o.B >> 16) & 0xffu)) * 2 + (int)o.p;
            IO.prob[base + n] = wgt * q;                             // :241
            IO.next[base + n] = nf;
            IO.reward[base + n] = goal ? (ncc == Wm1 ? 1 : -1) : 0;  // :237-240
            IO.done[base + n] = goal ? 1 : 0;
            ++n;
        }
    }
    IO.count[gid] = n;
}

// =================================================================================================
// planners on the single-agent transition lists (reference gym_soccer/utils/planners.py:4-87)
// =================================================================================================
enum PlanMode : int32_t { kPlanVI = 0, kPlanEval = 1, kPlanImprove = 2, kPlanPI = 3, kPlanMPI = 4, kPlanEvalDense = 5 };

// one list entry, 16 bytes = one dwordx4 load.  Lists are padded to a multiple of kPlanPad entries with
// (prob 0, next 0, done) entries, which add an exact +-0 to the running sum, so the loops below can fetch
// kPlanPad entries per wait without changing a bit of the result.
struct __attribute__((aligned(16))) PlanEntry { double prob; int32_t next_done; float reward; };   // next | done << 31
constexpr int kPlanPad = 4;

struct PlanIO {
    // P[s][a] lists in the reference's order (:167-293), CSR by (state, learner action): offset[nS*5 + 1];
    // reward is the learner's (+-1, +-0)
    const int32_t* offset; const PlanEntry* list;
    // rows of Pmat / Rmat (:280-291): per (state, action) the next states in ascending index with their
    // accumulated probability (the dense dot's order), and the expected reward
    const int32_t* m_offset; const PlanEntry* m_list; const double* m_R;
    double* V; double* newV; double* Q; int32_t* pi;
    int32_t* counters;          // [0] outer iterations, [1] sweeps, [2] 1 = stopped by max_sweeps
    int32_t nS, mode, max_sweeps, k;
    double theta, gamma, threshold;
};

// Q += prob * (reward + discount_factor * V[next_state] * (not done)), summed in list order (planners.py:12,28,39)
__device__ __forceinline__ double list_backup(const PlanIO& IO, const double* V, int s, int a) {
    double q = 0.0;
    const int end = IO.offset[s * 5 + a + 1];
    for (int e = IO.offset[s * 5 + a]; e < end; e += kPlanPad) {
        PlanEntry x[kPlanPad];
#pragma unroll
        for (int j = 0; j < kPlanPad; ++j) x[j] = IO.list[e + j];
#pragma unroll
        for (int j = 0; j < kPlanPad; ++j) {
            const double cont = (IO.gamma * V[x[j].next_done & 0x7fffffff]) * (x[j].next_done < 0 ? 0.0 : 1.0);
            q = q + x[j].prob * ((double)x[j].reward + cont);
        }
    }
    return q;
}

// dot(Pmat[s, :, a], v) with a sequential sum over the non-zero entries in ascending next-state index
__device__ __forceinline__ double dense_dot(const PlanIO& IO, const double* V, int s, int a) {
    double acc = 0.0;
    const int end = IO.m_offset[s * 5 + a + 1];
    for (int e = IO.m_offset[s * 5 + a]; e < end; e += kPlanPad) {
        PlanEntry x[kPlanPad];
#pragma unroll
        for (int j = 0; j < kPlanPad; ++j) x[j] = IO.m_list[e + j];
#pragma unroll
        for (int j = 0; j < kPlanPad; ++j) acc = acc + x[j].prob * V[x[j].next_done];
    }
    return acc;
}

// Rmat[s, a] + discount_factor * dot(Pmat[s, :, a], v)   (planners.py:62-65, :80)
__device__ __forceinline__ double dense_backup(const PlanIO& IO, const double* V, int s, int a) {
    return IO.m_R[s * 5 + a] + IO.gamma * dense_dot(IO, V, s, a);
}

// maximum of a non-negative double over the workgroup (such doubles order like their bit patterns)
__device__ __forceinline__ double block_max(double d, unsigned long long* slot) {
    if (threadIdx.x == 0) *slot = 0ull;
    __syncthreads();
    atomicMax(slot, (unsigned long long)__double_as_longlong(d));
    __syncthreads();
    const double r = __longlong_as_double((long long)*slot);
    __syncthreads();
    return r;
}

// One workgroup runs a whole planner: the problem is nS x 5 short lists, a launch per sweep would be pure
// launch latency.  Synchronous sweeps in float64 with V in LDS; the list-based planners (value iteration,
// policy evaluation / improvement / iteration) evaluate exactly the reference's expressions in the
// reference's order, so values, greedy policies and iteration counts are the reference's bit for bit;
// modified policy iteration follows the reference's dense Pmat/Rmat algebra with a sequential dot (numpy's
// BLAS dot associates differently: equal to ~1e-15 relative, see tests/test_planner.py).
__global__ __launch_bounds__(1024) void planner_kernel(const PlanIO IO) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    double* V = reinterpret_cast<double*>(smem);                 // [nS]
    __shared__ unsigned long long s_slot;
    const int nS = IO.nS, tid = threadIdx.x, nt = blockDim.x;
    int outer = 0, sweeps = 0, capped = 0;

    // greedy step over the lists: Q, first maximising action; returns max |V - max_a Q| over own states
    auto greedy_lists = [&](bool* changed) {
        double dmax = 0.0;
        for (int s = tid; s < nS; s += nt) {
            double best = 0.0; int arg = 0;
            for (int a = 0; a < 5; ++a) {
                const double q = list_backup(IO, V, s, a);
                IO.Q[s * 5 + a] = q;
                if (a == 0 || q > best) { best = q; arg = a; }            // np.argmax: first maximum
            }
            IO.newV[s] = best;
            if (changed && IO.pi[s] != arg) *changed = true;
            IO.pi[s] = arg;
            dmax = fmax(dmax, fabs(V[s] - best));
        }
        return dmax;
    };
    // policy_evaluation (planners.py:20-31) of IO.pi from zeros; leaves the result in IO.newV
    auto evaluate = [&]() {
        for (int s = tid; s < nS; s += nt) V[s] = 0.0;
        __syncthreads();
        for (;;) {
            double dmax = 0.0;
            for (int s = tid; s < nS; s += nt) {
                const double v = list_backup(IO, V, s, IO.pi[s]);
                IO.newV[s] = v;
                dmax = fmax(dmax, fabs(V[s] - v));
            }
            const double delta = block_max(dmax, &s_slot);
            ++sweeps;
            if (delta < IO.theta) break;
            if (sweeps >= IO.max_sweeps) { capped = 1; break; }
            for (int s = tid; s < nS; s += nt) V[s] = IO.newV[s];
            __syncthreads();
        }
    };

    if (IO.mode == kPlanVI) {                                   // planners.py:4-18
        for (int s = tid; s < nS; s += nt) V[s] = 0.0;
        __syncthreads();
        for (;;) {
            const double delta = block_max(greedy_lists(nullptr), &s_slot);
            ++outer; ++sweeps;
            if (delta < IO.theta) break;
            if (sweeps >= IO.max_sweeps) { capped = 1; break; }
            for (int s = tid; s < nS; s += nt) V[s] = IO.newV[s];
            __syncthreads();
        }
        for (int s = tid; s < nS; s += nt) IO.V[s] = V[s];       // the reference returns the pre-update V
    } else if (IO.mode == kPlanEval) {                          // planners.py:20-31
        evaluate();
        outer = sweeps;
        for (int s = tid; s < nS; s += nt) IO.V[s] = IO.newV[s];
    } else if (IO.mode == kPlanImprove) {                       // planners.py:33-41
        for (int s = tid; s < nS; s += nt) V[s] = IO.V[s];
        __syncthreads();
        (void)greedy_lists(nullptr);
        outer = 1;
    } else if (IO.mode == kPlanPI) {                            // planners.py:43-53
        for (;;) {
            evaluate();
            __syncthreads();
            for (int s = tid; s < nS; s += nt) { V[s] = IO.newV[s]; IO.V[s] = IO.newV[s]; }
            __syncthreads();
            bool changed = false;
            (void)greedy_lists(&changed);
            ++outer;
            const double any = block_max(changed ? 1.0 : 0.0, &s_slot);
            if (any == 0.0 || capped) break;
        }
    } else if (IO.mode == kPlanEvalDense) {                     // policy_eval, planners.py:55-70 (policy[s, a] in IO.Q)
        for (int s = tid; s < nS; s += nt) V[s] = IO.V[s];
        __syncthreads();
        for (int i = 0; i < IO.k; ++i) {
            double d2 = 0.0;
            for (int s = tid; s < nS; s += nt) {
                double r_pi = 0.0, p_pi = 0.0;
                for (int a = 0; a < 5; ++a) {
                    const double w = IO.Q[s * 5 + a];
                    const double acc = dense_dot(IO, V, s, a);
                    r_pi = r_pi + w * IO.m_R[s * 5 + a];
                    p_pi = p_pi + acc * w;
                }
                const double v = r_pi + IO.gamma * p_pi;
                IO.newV[s] = v;
                d2 = fmax(d2, fabs(v - V[s]));
            }
            const double delta = block_max(d2, &s_slot);
            for (int s = tid; s < nS; s += nt) V[s] = IO.newV[s];
            __syncthreads();
            ++sweeps;
            if (delta < IO.theta) break;
            if (sweeps >= IO.max_sweeps) { capped = 1; break; }
        }
        outer = sweeps;
        for (int s = tid; s < nS; s += nt) IO.V[s] = V[s];
    } else {                                                    // modified_policy_iteration, planners.py:73-87
        for (int s = tid; s < nS; s += nt) V[s] = 0.0;
        __syncthreads();
        for (;;) {
            double dmax = 0.0;
            for (int s = tid; s < nS; s += nt) {
                double best = 0.0; int arg = 0;
                for (int a = 0; a < 5; ++a) {
                    const double q = dense_backup(IO, V, s, a);
                    IO.Q[s * 5 + a] = q;
                    if (a == 0 || q > best) { best = q; arg = a; }
                }
                IO.newV[s] = best; IO.pi[s] = arg;
                dmax = fmax(dmax, fabs(V[s] - best));
            }
            const double gap = block_max(dmax, &s_slot);
            ++sweeps;
            if (gap <= IO.threshold) break;                       // returns greedy_v, q, counter (:83-84)
            if (sweeps >= IO.max_sweeps) { capped = 1; break; }
            for (int s = tid; s < nS; s += nt) V[s] = IO.newV[s];  // policy_eval(init = greedy_v), :55-70
            __syncthreads();
            for (int i = 0; i < IO.k; ++i) {
                double d2 = 0.0;
                for (int s = tid; s < nS; s += nt) {
                    const double v = dense_backup(IO, V, s, IO.pi[s]);
                    IO.newV[s] = v;
                    d2 = fmax(d2, fabs(v - V[s]));
                }
                const double delta = block_max(d2, &s_slot);
                for (int s = tid; s < nS; s += nt) V[s] = IO.newV[s];
                __syncthreads();
                ++sweeps;
                if (delta < IO.theta) break;
                if (sweeps >= IO.max_sweeps) { capped = 1; break; }
            }
            ++outer;
            if (capped) break;
        }
        for (int s = tid; s < nS; s += nt) IO.V[s] = IO.newV[s];
    }
    if (tid == 0) { IO.counters[0] = outer; IO.counters[1] = sweeps; IO.counters[2] = capped; }
}

// =================================================================================================
// episode returns from result trajectories (soccer_trajectory_returns)
// =================================================================================================
// What the caller of T batched_step calls (or of one batched_rollout) holds afterwards is [T][n] reward / terminated / truncated
// streams; what BASELINE config 4 gathers over xGMI is ONE value per lane — player A's return of the lane's most recently
// finished episode (= the reward of the step that ended it: only that step can carry one, :235-240) — plus the 3-bin histogram
// of all finished episodes.  One pass over the three streams, 3 B per env-step read, 1 (+4) B per lane written: HBM-bound.
// VEC: four lanes per thread by dword (streams 4-aligned, stride % 4 == 0); otherwise a lane per thread by byte.
struct TrajIO {
    const int8_t* reward; const uint8_t* terminated; const uint8_t* truncated;
    long long stride; int32_t n_steps; unsigned long long n;
    int8_t* last_return; int32_t* episode_count;       // nullable
    unsigned long long* hist; uint32_t slot0;           // device u64[slots][4]: per-workgroup counts of returns -1, 0, +1, from slot `slot0`
};
template <bool VEC>
__global__ __launch_bounds__(kBlock) void trajectory_returns_kernel(const TrajIO IO) {
    uint32_t fin_t = 0u, nz_t = 0u, neg_t = 0u;
    const unsigned long long units = VEC ? (IO.n >> 2) : IO.n;
    for (unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; g < units;
         g += (unsigned long long)gridDim.x * kBlock) {
        if (VEC) {
            const unsigned long long i0 = g << 2;
            uint32_t last = 0u, c8 = 0u, cnt[4] = {0u, 0u, 0u, 0u};
            // eight rows in flight per thread (24 independent dword loads, streamed once: non-temporal; the scheduling barrier
            // keeps them ahead of the arithmetic): a load per row and wait ran at 1.2 TB/s (profiles/r04_a: 174 us for T = 64)
            constexpr int U = 8;
            auto row_of = [&](uint32_t r, uint32_t f) {
                const uint32_t nz = ((f | ((f & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u) >> 7;     // 0 / 1 per byte
                const uint32_t m = nz * 255u, rm = r & m;
                last = (last & ~m) | rm;
                c8 += nz;
                fin_t += (uint32_t)__builtin_popcount(nz); nz_t += (uint32_t)__builtin_popcount(rm & 0x01010101u);
                neg_t += (uint32_t)__builtin_popcount(rm & 0x80808080u);
            };
            int s0 = 0;
            for (; s0 + U <= IO.n_steps; s0 += U) {
                uint32_t r[U], ft[U], fr[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const long long row = (long long)(s0 + k) * IO.stride;
                    r[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(IO.reward + row + i0));
                    ft[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(IO.terminated + row + i0));
                    fr[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(IO.truncated + row + i0));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < U; ++k) row_of(r[k], ft[k] | fr[k]);
                if ((s0 & 127) == 120) { for (int j = 0; j < 4; ++j) cnt[j] += (c8 >> (8 * j)) & 0xffu; c8 = 0u; }   // every 128 rows: no byte overflows
            }
            for (; s0 < IO.n_steps; ++s0) {                                              // the last n_steps % 8 rows (c8 gains < 8 here)
                const long long row = (long long)s0 * IO.stride;
                row_of(*reinterpret_cast<const uint32_t*>(IO.reward + row + i0),
                       *reinterpret_cast<const uint32_t*>(IO.terminated + row + i0) | *reinterpret_cast<const uint32_t*>(IO.truncated + row + i0));
            }
            for (int j = 0; j < 4; ++j) cnt[j] += (c8 >> (8 * j)) & 0xffu;
            if (IO.last_return) *reinterpret_cast<uint32_t*>(IO.last_return + i0) = last;
            if (IO.episode_count) *reinterpret_cast<int4*>(IO.episode_count + i0) = make_int4((int)cnt[0], (int)cnt[1], (int)cnt[2], (int)cnt[3]);
        } else {
            int8_t last = 0; uint32_t cnt = 0u;
            for (int s = 0; s < IO.n_steps; ++s) {
                const long long row = (long long)s * IO.stride;
                const int8_t r = IO.reward[row + g];
                if (IO.terminated[row + g] | IO.truncated[row + g]) {
                    last = r; ++cnt; ++fin_t; nz_t += r != 0 ? 1u : 0u; neg_t += r < 0 ? 1u : 0u;
                }
            }
            if (IO.last_return) IO.last_return[g] = last;
            if (IO.episode_count) IO.episode_count[g] = (int32_t)cnt;
        }
    }
    // one private slot per workgroup, summed by the host (atomics of 4 096 waves on three words of one line were 120 of the
    // 170 us this kernel took at T = 64: profiles/r04_a_kernel_stats_other.csv)
    __shared__ uint32_t part[kBlock / 64][3];
    const uint32_t tot = wave_sum(fin_t), nzs = wave_sum(nz_t), neg = wave_sum(neg_t);
    if ((threadIdx.x & 63u) == 0u) { part[threadIdx.x >> 6][0] = neg; part[threadIdx.x >> 6][1] = tot - nzs; part[threadIdx.x >> 6][2] = nzs - neg; }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long v = 0ull;
        for (int wv = 0; wv < kBlock / 64; ++wv) v += part[wv][threadIdx.x];
        IO.hist[(size_t)(IO.slot0 + blockIdx.x) * 4 + threadIdx.x] = v;
    }
}

}  // namespace soccer
