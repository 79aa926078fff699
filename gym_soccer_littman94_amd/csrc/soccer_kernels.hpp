// soccer_kernels.hpp — gfx950 (CDNA4 / MI355X) kernels for the batched step / reset / rollout.
//
// What is evaluated per lane, and where the reference states it
// (gym_soccer/envs/soccer_simultaneous_env.py):
//   cell move via the LDS move/bounds table ............ _next_cell          :364-373
//   ordered 5-way collision resolution ................. _get_next_state     :296-362
//   nine slip combinations, float64 weights, zero-skip . :202-227, :241
//   done / reward ...................................... :235-240
//   outcome selection .................................. categorical_sample  :395 (gym 0.26.2)
//   bookkeeping (timestep, truncation, needs_reset) .... :396-406
//   reset from the initial state distribution .......... :410-424
//
// Execution shape: wave64; each thread owns E consecutive lanes (environments) so that every SoA
// byte stream is read and written with one 4/8/16-byte access per thread (coalesced 256 B .. 1 KiB
// per wave instruction).  Rule tables (observation LUT, move/bounds table) are staged into LDS once
// per workgroup; the grid is sized so each workgroup stages once and then grid-strides.
// No MFMA: there is no contraction on this path.  The kernel is HBM/issue bound.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace soccer {

constexpr int kBlock = 256;
constexpr int kHistShards = 64;          // per-shard stride of 16 u64 (128 B) to keep atomics apart
constexpr int kHistStride = 16;

struct KernelParams {
    // resident state, structure-of-arrays
    int8_t* row_a; int8_t* col_a; int8_t* row_b; int8_t* col_b;
    uint8_t* poss;                        // bit0 possession, bit1 needs_reset
    uint8_t* t;
    // rule tables in global memory (staged to LDS)
    const uint16_t* lut;                  // [lut_len]
    const uint16_t* next_cell;            // [2*H*W*5]
    // randomness
    const unsigned long long* tick_in;    // device tick slot read by this launch
    unsigned long long* tick_out;         // slot written (tick_in + ticks consumed)
    uint32_t key0, key1;
    unsigned long long lane_offset;
    // statistics
    unsigned long long* hist;             // [kHistShards][kHistStride], bins 0..2 used
    unsigned int* misuse;                 // sticky flag
    // geometry / constants
    unsigned long long n;
    int32_t H, W, HW, lut_len, nc_len;
    int32_t max_steps;
    uint32_t autoreset;
    uint32_t isd_shift;                   // 2 - log2(n_isd): index = top2 >> isd_shift
    uint32_t isd_pos[4];                  // row_a | col_a<<8 | row_b<<16 | col_b<<24
    uint32_t isd_poss_obs[4];             // poss | obs<<16
    double w[4];                          // slip-combination weights c0..c3 (:211-222)
};

struct StepIO {
    const int8_t* act_a; const int8_t* act_b;
    const double* u_step; const double* u_reset;
    uint16_t* obs; int8_t* reward; uint8_t* terminated; uint8_t* truncated;
    uint8_t* prob_code; uint16_t* final_obs; int8_t* last_return;
};

struct ResetIO {
    const uint8_t* mask; const double* u_reset; uint16_t* obs;
};

struct RolloutIO {
    int32_t n_steps; int32_t sample_actions;
    const int8_t* act_a; const int8_t* act_b; long long act_stride;
    uint16_t* obs; int8_t* reward; uint8_t* terminated; uint8_t* truncated; long long out_stride;
    int32_t* return_sum; int32_t* episode_count;
};

// ---- Philox4x32-10 (Salmon et al. 2011; Random123 constants) ---------------------------------
struct Philox4 { uint32_t w0, w1, w2, w3; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// One uniform in the two forms the rules need: the float64 value (slip lists) and floor(4u)
// (every list whose probabilities are dyadic: slip_prob == 0 and the ISD).
struct Draw { double u; uint32_t top2; };

__device__ __forceinline__ Draw draw_from_words(uint32_t lo, uint32_t hi) {
    const uint64_t m = (((uint64_t)hi << 32) | lo) >> 11;          // 53 bits
    return Draw{(double)m * 0x1.0p-53, hi >> 30};
}
// A caller-supplied uniform.  Values outside [0,1) (and NaN) make every running sum compare
// "not greater", which categorical_sample resolves to index 0 — same as u = 0.
__device__ __forceinline__ Draw draw_from_double(double u) {
    const bool ok = (u >= 0.0) && (u < 1.0);
    const double v = ok ? u : 0.0;
    return Draw{v, (uint32_t)(v * 4.0)};
}

// ---- LDS-resident rule tables ------------------------------------------------------------------
struct Tables {
    const uint16_t* lut;    // LDS (or global when it does not fit)
    const uint16_t* nc;     // LDS
};

template <bool LUT_LDS>
__device__ __forceinline__ Tables stage_tables(const KernelParams& P, uint16_t* smem) {
    // copy as dwords; both tables are padded to an even element count by the host
    uint32_t* dst = reinterpret_cast<uint32_t*>(smem);
    const int nc_dw = P.nc_len >> 1;
    const uint32_t* src_nc = reinterpret_cast<const uint32_t*>(P.next_cell);
    for (int i = threadIdx.x; i < nc_dw; i += kBlock) dst[i] = src_nc[i];
    Tables T;
    T.nc = smem;
    if (LUT_LDS) {
        const int lut_dw = P.lut_len >> 1;
        const uint32_t* src_lut = reinterpret_cast<const uint32_t*>(P.lut);
        for (int i = threadIdx.x; i < lut_dw; i += kBlock) dst[nc_dw + i] = src_lut[i];
        T.lut = smem + P.nc_len;
    } else {
        T.lut = P.lut;
    }
    __syncthreads();
    return T;
}

// ---- one lane's state in registers -------------------------------------------------------------
struct Lane {
    uint32_t A, B;      // position of each player packed (row<<8 | col)
    uint32_t p;         // possession 0/1
    uint32_t need;      // needs_reset 0/1
    uint32_t t;
};

struct StepResult {
    uint32_t obs, final_obs;
    int32_t reward;
    uint32_t term, trunc, code;
    uint32_t finished;      // episode ended at this step (before any auto-reset)
    uint32_t misused;
};

__device__ __forceinline__ uint32_t cell_of(uint32_t pos, int W) { return (pos >> 8) * W + (pos & 0xffu); }

__device__ __forceinline__ uint32_t obs_of(const Tables& T, const KernelParams& P, uint32_t A, uint32_t B, uint32_t p) {
    const uint32_t f = ((cell_of(A, P.W) * (uint32_t)P.HW) + cell_of(B, P.W)) * 2u + p;
    return T.lut[f];
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

// _get_next_state (:296-362) for a live tuple with players at A / B (row<<8|col) and possession p.
// aa/ab: ORIGINAL actions (the NOOP tests); mvA/mvB: the (possibly slipped) moves.
__device__ __forceinline__ Resolved resolve(const Tables& T, const KernelParams& P, uint32_t A, uint32_t B,
                                            uint32_t p, uint32_t aa, uint32_t ab, uint32_t mvA, uint32_t mvB) {
    const uint32_t ballA = p ^ 1u, ballB = p;
    const uint32_t nA = T.nc[(ballA * P.HW + cell_of(A, P.W)) * 5u + mvA];
    const uint32_t nB = T.nc[(ballB * P.HW + cell_of(B, P.W)) * 5u + mvB];
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

template <bool SLIP>
__device__ __forceinline__ void lane_step(const Tables& T, const KernelParams& P, Lane& Lref,
                                          uint32_t aa, uint32_t ab, const Draw& ds, const Draw& dr,
                                          StepResult& out) {
    const uint32_t A = Lref.A, B = Lref.B, p = Lref.p, t = Lref.t, need_in = Lref.need;
    const uint32_t Wm1 = (uint32_t)(P.W - 1);
    const uint32_t carrier_col = (p ? B : A) & 0xffu;
    const bool in_goal = (carrier_col == 0u) | (carrier_col == Wm1);   // goal tuple: absorbing (:300-301)
    Outcome sel;
    uint32_t cls = 0;
    if (!SLIP) {
        // single surviving combination, weight 1.0 (:226-227): list probabilities are 1, .5/.5 or .25x4,
        // so the sampled index is floor(2u) / floor(4u)
        const Resolved R = resolve(T, P, A, B, p, aa, ab, aa, ab);
        const uint32_t k = R.kind == K_COIN ? (ds.top2 >> 1) : ds.top2;
        sel = pick(A, B, p, R, k);
    } else {
        constexpr int VA[9] = {0, 0, 0, 1, 2, 1, 1, 2, 2};
        constexpr int VB[9] = {0, 1, 2, 0, 0, 1, 2, 1, 2};
        constexpr int CLS[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
        double acc = 0.0;
        bool found = false, have_first = false;
        Outcome first = Outcome{A, B, p, 0u}; uint32_t first_cls = 0;
        sel = first;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const double wgt = P.w[CLS[c]];
            if (wgt == 0.0) continue;                                   // uniform branch (:226-227)
            const Resolved R = resolve(T, P, A, B, p, aa, ab, slip_move(aa, VA[c]), slip_move(ab, VB[c]));
            const uint32_t kind = in_goal ? (uint32_t)K_MOVE : R.kind;
            const uint32_t n = kind == K_COIN ? 2u : (kind == K_FOUR ? 4u : 1u);
            const double q = wgt * (n == 1u ? 1.0 : (n == 2u ? 0.5 : 0.25));   // :241
            const double a1 = acc + q, a2 = a1 + q, a3 = a2 + q, a4 = a3 + q;   // sequential cumsum
            const double end = n == 1u ? a1 : (n == 2u ? a2 : a4);
            const uint32_t k = (ds.u >= a1 ? 1u : 0u) + (((n > 1u) & (ds.u >= a2)) ? 1u : 0u) +
                               (((n > 2u) & (ds.u >= a3)) ? 1u : 0u);
            const bool here = !found & (end > ds.u);
            const Outcome o = pick(A, B, p, R, here ? k : 0u);
            const bool take_first = !have_first;
            sel.A = here ? o.A : sel.A; sel.B = here ? o.B : sel.B; sel.p = here ? o.p : sel.p;
            sel.kcode = here ? o.kcode : sel.kcode; cls = here ? (uint32_t)CLS[c] : cls;
            first.A = take_first ? o.A : first.A; first.B = take_first ? o.B : first.B;
            first.p = take_first ? o.p : first.p; first.kcode = take_first ? o.kcode : first.kcode;
            first_cls = take_first ? (uint32_t)CLS[c] : first_cls;
            found |= here; have_first = true;
            acc = end;
        }
        if (!found) { sel = first; cls = first_cls; }                   // argmax of all-False is 0
    }
    sel.A = in_goal ? A : sel.A; sel.B = in_goal ? B : sel.B; sel.p = in_goal ? p : sel.p;
    sel.kcode = in_goal ? 0u : sel.kcode;
    // done / reward (:235-240)
    const uint32_t ncc = (sel.p ? sel.B : sel.A) & 0xffu;
    const bool goal_now = (ncc == 0u) | (ncc == Wm1);
    const int32_t reward = (goal_now & !in_goal) ? (ncc == Wm1 ? 1 : -1) : 0;
    const uint32_t tt = t + 1u;                                         // :399
    const uint32_t trunc = tt >= (uint32_t)P.max_steps ? 1u : 0u;      // :404
    const uint32_t done = goal_now ? 1u : 0u;
    const uint32_t need = done | trunc;                                 // :406
    const uint32_t ob_step = obs_of(T, P, sel.A, sel.B, sel.p);         // :397 (goal tuples map to 0)
    // in-step auto-reset (:414-423)
    const uint32_t i = dr.top2 >> P.isd_shift;
    const uint32_t rpos = i & 2u ? (i & 1u ? P.isd_pos[3] : P.isd_pos[2]) : (i & 1u ? P.isd_pos[1] : P.isd_pos[0]);
    const uint32_t rpo = i & 2u ? (i & 1u ? P.isd_poss_obs[3] : P.isd_poss_obs[2])
                                : (i & 1u ? P.isd_poss_obs[1] : P.isd_poss_obs[0]);
    const bool do_reset = (need != 0u) & (P.autoreset != 0u);
    const uint32_t rA = ((rpos & 0xffu) << 8) | ((rpos >> 8) & 0xffu);
    const uint32_t rB = (((rpos >> 16) & 0xffu) << 8) | (rpos >> 24);
    // a lane that needs reset is left untouched; the reference asserts (:376)
    const bool frozen = need_in != 0u;
    const uint32_t ob_frozen = frozen ? obs_of(T, P, A, B, p) : 0u;
    Lref.A = frozen ? A : (do_reset ? rA : sel.A);
    Lref.B = frozen ? B : (do_reset ? rB : sel.B);
    Lref.p = frozen ? p : (do_reset ? (rpo & 1u) : sel.p);
    Lref.t = frozen ? t : (do_reset ? 0u : tt);
    Lref.need = frozen ? 1u : (do_reset ? 0u : need);
    out.obs = frozen ? ob_frozen : (do_reset ? (rpo >> 16) : ob_step);
    out.final_obs = frozen ? ob_frozen : ob_step;
    out.reward = frozen ? 0 : reward;
    out.term = frozen ? (in_goal ? 1u : 0u) : done;
    out.trunc = frozen ? (t >= (uint32_t)P.max_steps ? 1u : 0u) : trunc;
    out.code = frozen ? 0u : (cls * 3u + sel.kcode);
    out.finished = frozen ? 0u : need;
    out.misused = frozen ? 1u : 0u;
}

__device__ __forceinline__ void lane_reset(const KernelParams& P, Lane& L, const Draw& dr, uint32_t& ob) {
    const uint32_t i = dr.top2 >> P.isd_shift;
    const uint32_t pos = i & 2u ? (i & 1u ? P.isd_pos[3] : P.isd_pos[2]) : (i & 1u ? P.isd_pos[1] : P.isd_pos[0]);
    const uint32_t po = i & 2u ? (i & 1u ? P.isd_poss_obs[3] : P.isd_poss_obs[2])
                               : (i & 1u ? P.isd_poss_obs[1] : P.isd_poss_obs[0]);
    L.A = ((pos & 0xffu) << 8) | ((pos >> 8) & 0xffu);
    L.B = (((pos >> 16) & 0xffu) << 8) | (pos >> 24);
    L.p = po & 1u; L.t = 0u; L.need = 0u;
    ob = po >> 16;
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
    __device__ __forceinline__ void put(int j, uint32_t v) { w[j >> 2] |= (v & 0xffu) << (8 * (j & 3)); }
    __device__ __forceinline__ void load(const void* base, unsigned long long i) {
        const uint8_t* p = static_cast<const uint8_t*>(base) + i;
        if constexpr (E == 4) { w[0] = *reinterpret_cast<const uint32_t*>(p); }
        else if constexpr (E == 8) { const uint2 v = *reinterpret_cast<const uint2*>(p); w[0] = v.x; w[1] = v.y; }
        else { static_assert(E == 16, "E must be 1, 4, 8 or 16");
               const uint4 v = *reinterpret_cast<const uint4*>(p); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    }
    __device__ __forceinline__ void store(void* base, unsigned long long i) const {
        uint8_t* p = static_cast<uint8_t*>(base) + i;
        if constexpr (E == 4) { *reinterpret_cast<uint32_t*>(p) = w[0]; }
        else if constexpr (E == 8) { *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]); }
        else { *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]); }
    }
};
template <> struct PackB<1> {
    uint32_t b;
    __device__ __forceinline__ void clear() { b = 0u; }
    __device__ __forceinline__ uint32_t get(int) const { return b; }
    __device__ __forceinline__ void put(int, uint32_t v) { b = v & 0xffu; }
    __device__ __forceinline__ void load(const void* base, unsigned long long i) { b = static_cast<const uint8_t*>(base)[i]; }
    __device__ __forceinline__ void store(void* base, unsigned long long i) const { static_cast<uint8_t*>(base)[i] = (uint8_t)b; }
};

// E consecutive uint16 of one stream, as E/2 dwords
template <int E> struct PackH {
    static constexpr int NW = E / 2;
    uint32_t w[NW];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < NW; ++k) w[k] = 0u;
    }
    __device__ __forceinline__ void put(int j, uint32_t v) { w[j >> 1] |= (v & 0xffffu) << (16 * (j & 1)); }
    __device__ __forceinline__ void store(uint16_t* base, unsigned long long i) const {
        uint16_t* p = base + i;
        if constexpr (E == 4) { *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]); }
        else if constexpr (E == 8) { *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]); }
        else {
            *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4*>(p + 8) = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
};
template <> struct PackH<1> {
    uint32_t h;
    __device__ __forceinline__ void clear() { h = 0u; }
    __device__ __forceinline__ void put(int, uint32_t v) { h = v; }
    __device__ __forceinline__ void store(uint16_t* base, unsigned long long i) const { base[i] = (uint16_t)h; }
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

template <int E>
struct LaneVec {
    Lane L[E];
    __device__ __forceinline__ void load(const KernelParams& P, unsigned long long i) {
        PackB<E> ra, ca, rb, cb, ps, tt;
        ra.load(P.row_a, i); ca.load(P.col_a, i); rb.load(P.row_b, i); cb.load(P.col_b, i);
        ps.load(P.poss, i); tt.load(P.t, i);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            L[j].A = (ra.get(j) << 8) | ca.get(j);
            L[j].B = (rb.get(j) << 8) | cb.get(j);
            const uint32_t f = ps.get(j);
            L[j].p = f & 1u; L[j].need = (f >> 1) & 1u; L[j].t = tt.get(j);
        }
    }
    __device__ __forceinline__ void store(const KernelParams& P, unsigned long long i) const {
        PackB<E> ra, ca, rb, cb, ps, tt;
        ra.clear(); ca.clear(); rb.clear(); cb.clear(); ps.clear(); tt.clear();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            ra.put(j, L[j].A >> 8); ca.put(j, L[j].A); rb.put(j, L[j].B >> 8); cb.put(j, L[j].B);
            ps.put(j, L[j].p | (L[j].need << 1)); tt.put(j, L[j].t);
        }
        ra.store(P.row_a, i); ca.store(P.col_a, i); rb.store(P.row_b, i); cb.store(P.col_b, i);
        ps.store(P.poss, i); tt.store(P.t, i);
    }
};

// per-workgroup episode histogram: LDS bins, flushed to a sharded global array at exit
struct HistAcc {
    unsigned int* bins;     // LDS [4]
    __device__ __forceinline__ void init(unsigned int* lds) {
        bins = lds;
        if (threadIdx.x < 4) bins[threadIdx.x] = 0u;
    }
    __device__ __forceinline__ void add(uint32_t finished, int32_t reward) {
        if (finished) atomicAdd(&bins[reward + 1], 1u);
    }
    __device__ __forceinline__ void flush(const KernelParams& P) {
        __syncthreads();
        if (threadIdx.x < 3) {
            const unsigned int v = bins[threadIdx.x];
            if (v) atomicAdd(&P.hist[(blockIdx.x % kHistShards) * kHistStride + threadIdx.x], (unsigned long long)v);
        }
    }
};

__device__ __forceinline__ void publish_tick(const KernelParams& P, unsigned long long tick, unsigned long long used) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *P.tick_out = tick + used;
}

// =================================================================================================
// batched_step
// =================================================================================================
__device__ __forceinline__ void lane_draws(const KernelParams& P, unsigned long long lane, unsigned long long tick,
                                           bool need_philox, const double* u_step, const double* u_reset,
                                           Draw& ds, Draw& dr) {
    ds = Draw{0.0, 0u}; dr = Draw{0.0, 0u};
    if (need_philox) {
        const unsigned long long gid = P.lane_offset + lane;
        const Philox4 r = philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)tick,
                                        (uint32_t)(tick >> 32), P.key0, P.key1);
        ds = draw_from_words(r.w0, r.w1); dr = draw_from_words(r.w2, r.w3);
    }
    if (u_step) ds = draw_from_double(u_step[lane]);
    if (u_reset) dr = draw_from_double(u_reset[lane]);
}

// one group of E consecutive lanes starting at i0 (all in range)
template <int E, bool SLIP>
__device__ __forceinline__ void step_group(const Tables& T, const KernelParams& P, const StepIO& IO,
                                           unsigned long long i0, unsigned long long tick, bool need_philox,
                                           HistAcc& hist, uint32_t& any_misuse) {
    LaneVec<E> S; S.load(P, i0);
    PackB<E> aa, ab; aa.load(IO.act_a, i0); ab.load(IO.act_b, i0);
    PackB<E> o_rew, o_term, o_trunc, o_code; PackH<E> o_obs, o_fin;
    o_rew.clear(); o_term.clear(); o_trunc.clear(); o_code.clear(); o_obs.clear(); o_fin.clear();
    uint32_t fin_mask = 0u;
#pragma unroll
    for (int j = 0; j < E; ++j) {
        Draw ds, dr;
        lane_draws(P, i0 + j, tick, need_philox, IO.u_step, IO.u_reset, ds, dr);
        StepResult R;
        lane_step<SLIP>(T, P, S.L[j], aa.get(j), ab.get(j), ds, dr, R);
        o_obs.put(j, R.obs); o_fin.put(j, R.final_obs);
        o_rew.put(j, (uint32_t)R.reward); o_term.put(j, R.term); o_trunc.put(j, R.trunc); o_code.put(j, R.code);
        hist.add(R.finished, R.reward);
        fin_mask |= R.finished << j;
        any_misuse |= R.misused;
    }
    S.store(P, i0);
    if (IO.obs) o_obs.store(IO.obs, i0);
    if (IO.reward) o_rew.store(IO.reward, i0);
    if (IO.terminated) o_term.store(IO.terminated, i0);
    if (IO.truncated) o_trunc.store(IO.truncated, i0);
    if (IO.prob_code) o_code.store(IO.prob_code, i0);
    if (IO.final_obs) o_fin.store(IO.final_obs, i0);
    if (IO.last_return && fin_mask) {
#pragma unroll
        for (int j = 0; j < E; ++j)
            if ((fin_mask >> j) & 1u) IO.last_return[i0 + j] = (int8_t)o_rew.get(j);
    }
}

template <int E, bool SLIP, bool LUT_LDS>
__global__ __launch_bounds__(kBlock) void step_kernel(const KernelParams P, const StepIO IO) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    __shared__ unsigned int s_hist[4];
    HistAcc hist; hist.init(s_hist);
    const Tables T = stage_tables<LUT_LDS>(P, smem);
    const unsigned long long tick = *P.tick_in;
    publish_tick(P, tick, 1ull);
    const bool need_philox = (IO.u_step == nullptr) || (P.autoreset && IO.u_reset == nullptr);
    const unsigned long long groups = (P.n + E - 1) / E;
    uint32_t any_misuse = 0u;
    for (unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; g < groups;
         g += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long i0 = g * E;
        if (E == 1 || i0 + E <= P.n) {
            step_group<E, SLIP>(T, P, IO, i0, tick, need_philox, hist, any_misuse);
        } else {
            // ragged tail: fewer than E lanes left, one at a time
            for (unsigned long long i = i0; i < P.n; ++i)
                step_group<1, SLIP>(T, P, IO, i, tick, need_philox, hist, any_misuse);
        }
    }
    if (any_misuse) *P.misuse = 1u;
    hist.flush(P);
}

// =================================================================================================
// batched_reset
// =================================================================================================
template <bool LUT_LDS>
__global__ __launch_bounds__(kBlock) void reset_kernel(const KernelParams P, const ResetIO IO) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const Tables T = stage_tables<LUT_LDS>(P, smem);
    const unsigned long long tick = *P.tick_in;
    publish_tick(P, tick, 1ull);
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < P.n;
         i += (unsigned long long)gridDim.x * kBlock) {
        const bool sel = IO.mask == nullptr || IO.mask[i] != 0;
        uint32_t ob = 0u;
        if (sel) {
            Draw ds, dr;
            lane_draws(P, i, tick, IO.u_reset == nullptr, nullptr, IO.u_reset, ds, dr);
            LaneVec<1> S; lane_reset(P, S.L[0], dr, ob);
            S.store(P, i);
        } else if (IO.obs) {
            LaneVec<1> S; S.load(P, i);
            ob = obs_of(T, P, S.L[0].A, S.L[0].B, S.L[0].p);
        }
        if (IO.obs) IO.obs[i] = (uint16_t)ob;
    }
}

// =================================================================================================
// batched_rollout: T fused steps, state in registers, actions streamed in, trajectories streamed out
// =================================================================================================
template <int E, bool SLIP>
__device__ __forceinline__ void rollout_group(const Tables& T, const KernelParams& P, const RolloutIO& IO,
                                              unsigned long long i0, unsigned long long tick0,
                                              HistAcc& hist, uint32_t& any_misuse) {
    LaneVec<E> S; S.load(P, i0);
    int32_t ret[E], eps[E];
#pragma unroll
    for (int j = 0; j < E; ++j) { ret[j] = 0; eps[j] = 0; }
    PackB<E> aa, ab; aa.clear(); ab.clear();
    if (!IO.sample_actions) { aa.load(IO.act_a, i0); ab.load(IO.act_b, i0); }
    for (int s = 0; s < IO.n_steps; ++s) {
        const unsigned long long tick = tick0 + (unsigned long long)s;
        PackB<E> naa = aa, nab = ab;
        if (!IO.sample_actions && s + 1 < IO.n_steps) {                 // prefetch the next step's actions
            naa.load(IO.act_a + (long long)(s + 1) * IO.act_stride, i0);
            nab.load(IO.act_b + (long long)(s + 1) * IO.act_stride, i0);
        }
        PackB<E> o_rew, o_term, o_trunc; PackH<E> o_obs;
        o_rew.clear(); o_term.clear(); o_trunc.clear(); o_obs.clear();
#pragma unroll
        for (int j = 0; j < E; ++j) {
            Draw ds, dr;
            lane_draws(P, i0 + j, tick, true, nullptr, nullptr, ds, dr);
            uint32_t a = aa.get(j), b = ab.get(j);
            if (IO.sample_actions) {
                const unsigned long long gid = P.lane_offset + i0 + j;
                const Philox4 q = philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)tick,
                                                (uint32_t)(tick >> 32) | 0x80000000u, P.key0, P.key1);
                a = (uint32_t)(((uint64_t)q.w0 * 5u) >> 32);
                b = (uint32_t)(((uint64_t)q.w1 * 5u) >> 32);
            }
            StepResult R;
            lane_step<SLIP>(T, P, S.L[j], a, b, ds, dr, R);
            o_obs.put(j, R.obs); o_rew.put(j, (uint32_t)R.reward); o_term.put(j, R.term); o_trunc.put(j, R.trunc);
            ret[j] += R.reward; eps[j] += (int32_t)R.finished;
            hist.add(R.finished, R.reward);
            any_misuse |= R.misused;
        }
        const long long off = (long long)s * IO.out_stride;
        if (IO.obs) o_obs.store(IO.obs + off, i0);
        if (IO.reward) o_rew.store(IO.reward + off, i0);
        if (IO.terminated) o_term.store(IO.terminated + off, i0);
        if (IO.truncated) o_trunc.store(IO.truncated + off, i0);
        aa = naa; ab = nab;
    }
    S.store(P, i0);
    if (IO.return_sum) add_words<E>(IO.return_sum, i0, ret);
    if (IO.episode_count) add_words<E>(IO.episode_count, i0, eps);
}

template <int E, bool SLIP, bool LUT_LDS>
__global__ __launch_bounds__(kBlock) void rollout_kernel(const KernelParams P, const RolloutIO IO) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    __shared__ unsigned int s_hist[4];
    HistAcc hist; hist.init(s_hist);
    const Tables T = stage_tables<LUT_LDS>(P, smem);
    const unsigned long long tick0 = *P.tick_in;
    publish_tick(P, tick0, (unsigned long long)IO.n_steps);
    const unsigned long long groups = (P.n + E - 1) / E;
    uint32_t any_misuse = 0u;
    for (unsigned long long g = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; g < groups;
         g += (unsigned long long)gridDim.x * kBlock) {
        const unsigned long long i0 = g * E;
        if (E == 1 || i0 + E <= P.n) {
            rollout_group<E, SLIP>(T, P, IO, i0, tick0, hist, any_misuse);
        } else {
            for (unsigned long long i = i0; i < P.n; ++i)
                rollout_group<1, SLIP>(T, P, IO, i, tick0, hist, any_misuse);
        }
    }
    if (any_misuse) *P.misuse = 1u;
    hist.flush(P);
}

}  // namespace soccer
