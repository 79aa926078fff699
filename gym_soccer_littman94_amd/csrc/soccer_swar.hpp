// soccer_swar.hpp — one step of FOUR environments at once, on the packed SoA dwords as they are loaded.
//
// The state streams (row_a, col_a, row_b, col_b, poss, t) and the action streams hold one byte per lane, so the
// dword a thread loads from each stream carries four consecutive lanes.  Instead of peeling the bytes apart,
// stepping lane by lane through the rule tables and re-packing the results, everything below works on the four
// bytes of a dword in parallel (SIMD within a register): byte-wise table lookups by v_perm_b32 with the DATA as
// the selector, carry-free byte arithmetic (every value is < 128, bit 7 of each byte is the guard), truth values
// carried in bit 7 of each byte ("flag words": the other 7 bits are don't-care, so AND / OR / BFI / XOR work
// bitwise), and byte masks made from flag words with the sign-replicating selectors of v_perm_b32.  No rule
// table is read from memory at all: cell moves, the ordered collision test, the outcome pick, done / reward, the
// reset from the initial state distribution and the observation index (its closed form, see Rules::build) are
// arithmetic.  Per env-step this is ~45 vector instructions instead of ~128, and no dependent gather.
//
// What is evaluated, and where the reference states it (gym_soccer/envs/soccer_simultaneous_env.py):
//   cell move .............................. _next_cell          :364-373
//   ordered 5-way collision resolution ..... _get_next_state     :296-362
//   slip: the selected combination's moves . :202-227 (the combination is chosen per lane by the caller)
//   done / reward .......................... :235-240
//   bookkeeping (timestep, truncation) ..... :396-406
//   reset from the ISD ..................... :410-424 (:146-165)
//   observation index ...................... :63-109, :487-497 (closed form of the enumeration order)
//
// The same source compiles for the device (builtins) and for the host (plain C++ restatements of the four
// builtins), so tests/test_swar_host.py checks it on the CPU against the oracle for EVERY tuple x joint action x
// outcome draw of every golden pitch — no GPU needed to know the rules are right.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SOCCER_HD __host__ __device__ __forceinline__
#else
#define SOCCER_HD inline
#endif

namespace soccer {
// Slip selection by table (SlipTables in soccer_slip.hpp, swar::slip_select4_lut below): the ascending threshold list has
// kSlipThresholds entries (4 per combination, padded).
constexpr int kSlipThresholds = 40;
// The single step's table: a launch lives for one step, so its waves can stage only what one load per lane brings in — 64 lanes x
// 16 bytes = 1 024 byte buckets over the draw's top 10 bits; up to kSlipStepCompares thresholds may lie inside a bucket and are
// compared exactly.
constexpr int kSlipStepBucketBits = 10, kSlipStepBuckets = 1 << kSlipStepBucketBits, kSlipStepCompares = 2;
constexpr int kSlipStepLdsWords = kSlipStepBuckets / 4 + 64;     // the table, then the threshold list padded to one entry per lane

namespace swar {

// ---- the four machine operations the byte-parallel code leans on ----------------------------------------------
// v_perm_b32: result byte j = byte sel[j] of the 8 bytes {hi: 7..4, lo: 3..0}; sel 8..11 = 0x00/0xff by the sign of
// byte 1 / 3 / 5 / 7; sel 12 = 0x00; sel >= 13 = 0xff.
SOCCER_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    const uint64_t in = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int j = 0; j < 4; ++j) {
        const uint32_t s = (sel >> (8 * j)) & 0xffu;
        uint32_t b;
        if (s >= 13u) b = 0xffu;
        else if (s == 12u) b = 0u;
        else if (s >= 8u) b = ((in >> (8 * (2 * (s - 8u) + 1) + 7)) & 1u) ? 0xffu : 0u;
        else b = (uint32_t)(in >> (8 * s)) & 0xffu;
        r |= b << (8 * j);
    }
    return r;
#endif
}
// v_bfi_b32: bits of x where m is set, bits of y elsewhere
// (as v_bitop3_b32, truth table 0xca: with register operands it issues like a plain logic op, v_bfi_b32 does not —
// tools/labs/valu_rate_lab.hip)
SOCCER_HD uint32_t bfi(uint32_t m, uint32_t x, uint32_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(m, x, y, 0xca);
#else
    return (m & x) | (~m & y);
#endif
}
// v_pk_mad_u16: a * b + c on the two 16-bit halves, wrapping
SOCCER_HD uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 r = __builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b) + __builtin_bit_cast(u16x2, c);
    return __builtin_bit_cast(uint32_t, r);
#else
    const uint32_t lo = ((a & 0xffffu) * (b & 0xffffu) + (c & 0xffffu)) & 0xffffu;
    const uint32_t hi = ((a >> 16) * (b >> 16) + (c >> 16)) & 0xffffu;
    return lo | (hi << 16);
#endif
}

// v_pk_add_u16: a + b on the two 16-bit halves, wrapping
SOCCER_HD uint32_t pk_add(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 r = __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, r);
#else
    return ((a + b) & 0xffffu) | ((((a >> 16) + (b >> 16)) & 0xffffu) << 16);
#endif
}

// v_alignbit_b32 with both inputs equal: rotate right
SOCCER_HD uint32_t rotr(uint32_t x, uint32_t r) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(x, x, r);
#else
    r &= 31u;
    return r ? (x >> r) | (x << (32u - r)) : x;
#endif
}

constexpr uint32_t K80 = 0x80808080u, K01 = 0x01010101u, K7F = 0x7f7f7f7fu;

// flag word (truth in bit 7 of each byte) -> byte mask 0xff / 0x00: the sign selectors of v_perm_b32 reach the odd
// bytes of its 8-byte input, so the word goes in once as it is and once shifted up by a byte
SOCCER_HD uint32_t mask_of(uint32_t flag) { return perm(flag << 8, flag, 0x090b080au); }
// flag word -> 0x01 / 0x00 per byte
SOCCER_HD uint32_t one_of(uint32_t flag) { return (flag >> 7) & K01; }
// bytes < 128: flag = (byte == 0)
SOCCER_HD uint32_t is_zero(uint32_t x) { return K80 - x; }

// Wave-uniform constants of a handle; built once on the host (make_consts) and passed by value.
struct Consts {
    uint32_t Wx2;          // W | W << 16: cell id = row * W + col, two byte pairs at a time
    uint32_t Wm2x2;        // (W - 2) in both halves: interior-cell index + 1 = row * (W - 2) + col
    uint32_t Hp1x4;        // (H + 1) in every byte
    uint32_t Wm1x4;        // (W - 1) in every byte: B's goal line (column 0 is A's)
    uint32_t gr_lo_add;    // (0x80 - goal_lo) in every byte: bit 7 of row + this  <=>  row >= goal_lo   (:60)
    uint32_t gr_hi_add;    // (0x7f - goal_hi) in every byte: bit 7 of row + this  <=>  row >  goal_hi
    uint32_t trunc_add;    // (0x80 - max_steps) in every byte: bit 7 of t + this   <=>  t >= max_steps   (:404)
    uint32_t obs_mul;      // 2 * (NI - 1) in both halves, NI = H * (W - 2) interior cells
    uint32_t isd_ra, isd_rb, isd_p;   // byte k = row of A / row of B / possession of ISD entry k (:146-165)
    uint32_t isd_ca4, isd_cb4;        // the entries' columns (2 and W - 3) in every byte
    uint32_t isd_shift, isd_mask;     // entry index = (two random bits >> shift), 4 or 2 entries
    uint32_t autoreset;
    // Small pitches (H <= 6 and W <= 8, e.g. the reference's default 5x4 and 6x4): the clamps and the goal tests as 8-entry
    // byte tables — one v_perm_b32 instead of ~8 instructions each.  The kernels are instantiated for both geometries
    // (template GEO: 0 arithmetic, 1 tables) because deciding per launch costs what the tables save (run-time flags:
    // 5.2e11 env-steps/s in the rollout, compile-time: 5.7e11).
    uint32_t small;                   // != 0: the tables below are valid, take the GEO = 1 instantiations
    uint32_t row_lo, row_hi;          // [u] = clamp(u - 1, 0, H - 1) for u = row + 1 + drow in 0 .. H + 1
    uint32_t col_lo, col_hi;          // [c] = clamp(c, 1, W - 2): a step into a goal column undone
    uint32_t grow_lo, grow_hi;        // flag byte 0x80 for the goal rows (:60)
    uint32_t rew_lo, rew_hi;          // player A's reward by the carrier's column: 0xff at column 0, 0x01 at W - 1 (:237-240)
    uint32_t term_lo, term_hi;        // 0x01 at the two goal columns
};

// A pitch qualifies when every byte quantity stays below 128 (bit 7 is the guard bit):
// cell ids (H * W), 2 * NI + 2 (the observation's low term) and max_steps.
inline bool fits(int H, int W, int max_steps) {
    const int NI = H * (W - 2);
    return H >= 4 && W >= 7 && H * W <= 128 && 2 * NI + 2 <= 255 && max_steps >= 1 && max_steps <= 127;
}

inline Consts make_consts(int H, int W, int goal_lo, int goal_hi, int max_steps, int n_isd, const int8_t (*isd)[5],
                          bool autoreset) {
    Consts C{};
    auto splat = [](uint32_t b) { return (b & 0xffu) * K01; };
    C.Wx2 = (uint32_t)W | ((uint32_t)W << 16);
    C.Wm2x2 = (uint32_t)(W - 2) | ((uint32_t)(W - 2) << 16);
    C.Hp1x4 = splat((uint32_t)H + 1u);
    C.Wm1x4 = splat((uint32_t)W - 1u);
    C.gr_lo_add = splat(0x80u - (uint32_t)goal_lo);
    C.gr_hi_add = splat(0x7fu - (uint32_t)goal_hi);
    C.trunc_add = splat(0x80u - (uint32_t)max_steps);
    const uint32_t m = 2u * (uint32_t)(H * (W - 2) - 1);
    C.obs_mul = m | (m << 16);
    for (int k = 0; k < 4; ++k) {
        const int e = k < n_isd ? k : 0;
        C.isd_ra |= (uint32_t)(uint8_t)isd[e][0] << (8 * k);
        C.isd_rb |= (uint32_t)(uint8_t)isd[e][2] << (8 * k);
        C.isd_p |= (uint32_t)(uint8_t)isd[e][4] << (8 * k);
    }
    C.isd_ca4 = splat((uint32_t)(uint8_t)isd[0][1]);
    C.isd_cb4 = splat((uint32_t)(uint8_t)isd[0][3]);
    C.isd_shift = n_isd == 4 ? 0u : 1u;
    C.isd_mask = n_isd == 4 ? 0x03030303u : K01;
    C.autoreset = autoreset ? 1u : 0u;
    auto table = [](int n, int lo, int hi, int bias, uint32_t& tlo, uint32_t& thi) {
        uint64_t t = 0;
        for (int u = 0; u < 8; ++u) {
            int v = (u < n ? u : n - 1) - bias;
            v = v < lo ? lo : (v > hi ? hi : v);
            t |= (uint64_t)(uint8_t)v << (8 * u);
        }
        tlo = (uint32_t)t; thi = (uint32_t)(t >> 32);
    };
    C.small = (H + 2 <= 8 && W <= 8) ? 1u : 0u;
    if (C.small) {
        table(H + 2, 0, H - 1, 1, C.row_lo, C.row_hi);
        table(W, 1, W - 2, 0, C.col_lo, C.col_hi);
        auto flags = [](int n, auto f, uint32_t& tlo, uint32_t& thi) {
            uint64_t t = 0;
            for (int u = 0; u < 8 && u < n; ++u) t |= (uint64_t)(uint8_t)f(u) << (8 * u);
            tlo = (uint32_t)t; thi = (uint32_t)(t >> 32);
        };
        flags(H, [&](int r) { return r >= goal_lo && r <= goal_hi ? 0x80 : 0; }, C.grow_lo, C.grow_hi);
        flags(W, [&](int c) { return c == 0 ? 0xff : (c == W - 1 ? 0x01 : 0); }, C.rew_lo, C.rew_hi);
        flags(W, [&](int c) { return c == 0 || c == W - 1 ? 0x01 : 0; }, C.term_lo, C.term_hi);
    }
    return C;
}

// ---- the random bits of four lanes for one tick, as step4 / reset4 consume them ---------------------------------------
//   kq: bits 7 and 6 of byte j = bits 1 and 0 of lane j's QUARTER draw floor(4u) (a flag word and its left shift);
//       lists of probabilities 1, .5/.5 or .25 x 4 are sampled with floor(2u) / floor(4u)
//   rs: byte j, after `& Consts::isd_mask`, = lane j's reset draw (index into the ISD, :414) — the caller has already
//       shifted it right by Consts::isd_shift; the other bits are don't-care
// Where the bits come from is the RNG convention of include/soccer_hip.h:
//   slip_prob == 0  one Philox block serves a group of four lanes for EIGHT ticks: lane j of the group owns word j, tick k
//                   uses its nibble (k & 7) ^ 1 (counted from the least significant): the nibble's two high bits are the
//                   quarter draw, its two low bits the reset draw  (rand_nibble; the rollout walks the block with
//                   transpose4 / rand_pair instead of rotating four words per tick)
//   slip_prob > 0   one block per tick: lane j's word w gives m = w >> 2 (slip_select4 turns it into combination and
//                   quarter) and the reset draw w & 3  (rand_words gives rs; kq = quarter << 6)
struct Rand4 { uint32_t kq, rs; };

// byte j of the result = byte 0 of word j
SOCCER_HD uint32_t pack_byte0(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    return perm(perm(w3, w2, 0x0c0c0400u), perm(w1, w0, 0x0c0c0400u), 0x05040100u);
}
// byte j of the result = byte 3 of word j
SOCCER_HD uint32_t pack_byte3(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    return perm(perm(w3, w2, 0x0c0c0703u), perm(w1, w0, 0x0c0c0703u), 0x05040100u);
}
// one word per lane and tick: quarter draw = the word's two top bits, reset draw = its two low bits
SOCCER_HD Rand4 rand_words(const uint32_t isd_shift, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    return Rand4{pack_byte3(w0, w1, w2, w3), pack_byte0(w0, w1, w2, w3) >> isd_shift};
}
// tick t (= tick & 7) of an eight-tick block: nibble t ^ 1 of each lane's word, rotated into the high nibble of byte 0
SOCCER_HD Rand4 rand_nibble(const uint32_t isd_shift, uint32_t t, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    const uint32_t r = (4u * (t ^ 1u) + 28u) & 31u;
    const uint32_t x = pack_byte0(rotr(w0, r), rotr(w1, r), rotr(w2, r), rotr(w3, r));
    return Rand4{x, x >> (4u + isd_shift)};
}
// the 128 bits of a block as four dwords p[i], byte j of p[i] = byte i of word j: p[i] serves ticks 2i (high nibbles)
// and 2i + 1 (low nibbles) of the block's eight
SOCCER_HD void transpose4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t& p0, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    const uint32_t a = perm(w1, w0, 0x05010400u), b = perm(w1, w0, 0x07030602u);
    const uint32_t c = perm(w3, w2, 0x05010400u), d = perm(w3, w2, 0x07030602u);
    p0 = perm(c, a, 0x05040100u); p1 = perm(c, a, 0x07060302u); p2 = perm(d, b, 0x05040100u); p3 = perm(d, b, 0x07060302u);
}
// the draws of tick t from the transposed dword that holds its pair of ticks: even ticks read the high nibbles
SOCCER_HD Rand4 rand_pair(const uint32_t isd_shift, uint32_t t, uint32_t p) {
    const uint32_t sh = (t & 1u) << 2;
    return Rand4{p << sh, p >> (4u - sh + isd_shift)};
}

// four lanes of the six state streams
struct Group { uint32_t ra, ca, rb, cb, ps, tt; };

struct Out {
    uint32_t obs_lo, obs_hi;       // observation after any auto-reset: lanes 0,1 / 2,3 as uint16 pairs
    uint32_t fin_lo, fin_hi;       // FULL: observation before the reset (gym's final_observation)
    uint32_t rew, term, trunc;     // bytes
    uint32_t code;                 // FULL: prob_code = slip class * 3 + outcome class
    uint32_t finished;             // flag word: the lane's episode ended at this step
    uint32_t frozen;               // flag word: lanes that needed a reset and were left untouched (:376)
    uint32_t bad_action;           // non-zero: some action byte was outside 0..4
};

// byte-wise lookup of an action in an 8-entry table (entries 0..3 in `lo`, 4..7 in `hi`)
SOCCER_HD uint32_t lut8(uint32_t hi, uint32_t lo, uint32_t a) { return perm(hi, lo, a); }

// the action tables: entry = action 0 NOOP, 1 NORTH, 2 SOUTH, 3 EAST, 4 WEST (:8-31); 5..7 behave as NOOP
constexpr uint32_t T_DR1_LO = 0x01020001u, T_DR1_HI = 0x01010101u;   // row delta + 1
constexpr uint32_t T_E_LO = 0x01000000u, T_E_HI = 0u;                  // 1 for EAST
constexpr uint32_t T_W_LO = 0u, T_W_HI = 0x00000001u;                  // 1 for WEST
constexpr uint32_t T_NZ_LO = 0x80808000u, T_NZ_HI = 0x00000080u;       // flag: the action is not NOOP
// an action byte as the kernels execute it: table[byte & 7] with 5..7 -> NOOP, so no byte value can index outside a
// rule table; canon4(x) != x exactly when some byte of x is outside 0..4 (the reference raises IndexError, :393)
constexpr uint32_t T_CANON_LO = 0x03020100u, T_CANON_HI = 0x00000004u;
SOCCER_HD uint32_t canon4(uint32_t x) { return perm(T_CANON_HI, T_CANON_LO, x & 0x07070707u); }
// slipped move of an action (:205-206): NOOP->NOOP,NOOP NORTH->EAST,WEST SOUTH->WEST,EAST EAST->SOUTH,NORTH WEST->NORTH,SOUTH
constexpr uint32_t T_SLIP1_LO = 0x02040300u, T_SLIP1_HI = 0x00000001u;
constexpr uint32_t T_SLIP2_LO = 0x01030400u, T_SLIP2_HI = 0x00000002u;

// one player's tentative cell (_next_cell :364-373) for four lanes: rows clamped to the pitch, a step into a goal
// column reverted unless the player holds the ball and stands in a goal row.  `mv` = the (possibly slipped) move,
// `score` = flag word "holds the ball and is in a goal row" (an EAST / WEST move never changes the row).
template <int GEO>
SOCCER_HD void move4(const Consts& C, uint32_t r, uint32_t c, uint32_t mv, uint32_t score, uint32_t& nr, uint32_t& nc) {
    const uint32_t u = r + lut8(T_DR1_HI, T_DR1_LO, mv);                   // row + 1 + drow, in 0 .. H + 1
    if (GEO == 1) {
        nr = perm(C.row_hi, C.row_lo, u);                                  // clamp (:365) by table
    } else {
        const uint32_t at_top = one_of(is_zero(u));                        // stepped north off row 0
        const uint32_t at_bot = one_of(is_zero(u ^ C.Hp1x4));              // stepped south off row H - 1
        nr = u + at_top + 0xFEFEFEFFu - at_bot;                            // (u + at_top) >= 1 in every byte
    }
    const uint32_t ct = c + lut8(T_E_HI, T_E_LO, mv) - lut8(T_W_HI, T_W_LO, mv);   // :366
    if (GEO == 1) {
        // a step into a goal column that is not a score is undone (:369-372): from column 1 / W - 2 that is a clamp
        nc = bfi(mask_of(score), ct, perm(C.col_hi, C.col_lo, ct));
    } else {
        const uint32_t edge = is_zero(ct) | is_zero(ct ^ C.Wm1x4);        // the target is a goal column (:369)
        const uint32_t revert = bfi(score, 0u, edge);                      // ... and this is not a score (:370-372)
        nc = bfi(mask_of(revert), c, ct);
    }
}

// ---- slip: which of the nine combinations a lane's draw selects, and where in it the draw falls -------------------
// (:202-227, :241, :395) for draws that come from a Philox word, u = (m + 1/2) * 2^-30 with m = w >> 2: the integer form of the
// first-exceeds rule over the list's running sums (soccer_slip.hpp builds and checks the scaled thresholds).
struct alignas(16) Quad { uint32_t x, y, z, w; };    // per combination: integer thresholds { mid-point, quarter points 1, 2, 3 } of its mass
struct SlipConsts {
    uint32_t CB[9];        // scaled cumulative weight after each active combination (0xFFFFFFFF beyond)
    uint32_t c_off;        // id of the first active combination (0; 5 when slip_prob == 1)
};
// combination tables indexed by 8 - id (so that id 0 lands on v_perm_b32's "sign of byte 1" selector, which yields 0):
// A's / B's move variant (0 intended, 1 / 2 the orthogonals) and the weight class, for ids 8 .. 1
constexpr uint32_t T_VA_LO = 0x01010202u, T_VA_HI = 0x00000102u;
constexpr uint32_t T_VB_LO = 0x01020102u, T_VB_HI = 0x01020000u;
constexpr uint32_t T_CL_LO = 0x03030303u, T_CL_HI = 0x01010202u;

// combination ids of four lanes (bytes of c4) -> the slipped moves of the two players and the weight class
SOCCER_HD void slip_moves4(uint32_t c4, uint32_t aa, uint32_t ab, uint32_t& sa, uint32_t& sb, uint32_t& cls4) {
    const uint32_t sel = 0x08080808u - c4;
    const uint32_t va = perm(T_VA_HI, T_VA_LO, sel), vb = perm(T_VB_HI, T_VB_LO, sel);
    cls4 = perm(T_CL_HI, T_CL_LO, sel);
    // slipped move: variant 0 the action itself, 1 / 2 its orthogonals (:205-206)
    sa = bfi(perm(0u, 0x0000ff00u, va), perm(T_SLIP1_HI, T_SLIP1_LO, aa), bfi(perm(0u, 0x00ff0000u, va), perm(T_SLIP2_HI, T_SLIP2_LO, aa), aa));
    sb = bfi(perm(0u, 0x0000ff00u, vb), perm(T_SLIP1_HI, T_SLIP1_LO, ab), bfi(perm(0u, 0x00ff0000u, vb), perm(T_SLIP2_HI, T_SLIP2_LO, ab), ab));
}

// aa / ab: canonical actions (canon4).  sub: the 9 threshold rows (global or LDS).  Per lane: combination = number of
// integer cumulative weights <= m; quarter = number of that combination's quarter points <= m.  The rest is byte-parallel.
SOCCER_HD void slip_select4(const SlipConsts& L, const Quad* sub, uint32_t aa, uint32_t ab,
                            uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3,
                            uint32_t& sa, uint32_t& sb, uint32_t& k4, uint32_t& cls4) {
    const uint32_t w[4] = {w0, w1, w2, w3};
    uint32_t c4 = 0u; k4 = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 4; ++j) {
        const uint32_t m = w[j] >> 2;
        uint32_t idx = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 9; ++i) idx += m >= L.CB[i] ? 1u : 0u;
        const Quad th = sub[idx];
        const uint32_t q = (m >= th.y ? 1u : 0u) + (m >= th.z ? 1u : 0u) + (m >= th.w ? 1u : 0u);
        c4 |= (idx + L.c_off) << (8 * j); k4 |= q << (8 * j);
    }
    slip_moves4(c4, aa, ab, sa, sb, cls4);
}

// The same selection by table (SlipTables::lut / T, staged in LDS): the draw's top BITS bits give the number of thresholds that
// are surely at or below it, the at most NCMP thresholds that can lie inside the bucket are compared exactly, and the count
// p = 4 * position + quarter splits into both answers for all four lanes at once.  ~7 instead of ~30 vector instructions per lane
// (rollouts: 14 bits, one compare — SlipTables::lut; single steps: 10 bits, two compares — SlipTables::lut_step).
template <int BITS = 14, int NCMP = 1>
SOCCER_HD uint32_t slip_count4_lut(const uint8_t* lut, const uint32_t* T, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    const uint32_t w[4] = {w0, w1, w2, w3};
    uint32_t p4 = 0u;                                                      // per lane: the number of thresholds <= its draw
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 4; ++j) {
        const uint32_t m = w[j] >> 2;
        const uint32_t below = lut[w[j] >> (32 - BITS)];                   // bucket = m >> (30 - BITS)
        uint32_t p = below;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < NCMP; ++c) p += m >= T[below + c] ? 1u : 0u;
        p4 |= p << (8 * j);
    }
    return p4;
}
SOCCER_HD void slip_from_count4(uint32_t p4, uint32_t c_off, uint32_t aa, uint32_t ab, uint32_t& sa, uint32_t& sb, uint32_t& k4, uint32_t& cls4) {
    k4 = p4 & 0x03030303u;
    slip_moves4(((p4 >> 2) & 0x3f3f3f3fu) + c_off * K01, aa, ab, sa, sb, cls4);
}
template <int BITS = 14, int NCMP = 1>
SOCCER_HD void slip_select4_lut(const uint8_t* lut, const uint32_t* T, uint32_t c_off, uint32_t aa, uint32_t ab,
                                uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3,
                                uint32_t& sa, uint32_t& sb, uint32_t& k4, uint32_t& cls4) {
    slip_from_count4(slip_count4_lut<BITS, NCMP>(lut, T, w0, w1, w2, w3), c_off, aa, ab, sa, sb, k4, cls4);
}

// Observation index of four tuples (ra, ca, rb, cb, p): 1 + 2 * (iA * (NI - 1) + iB - (iB > iA)) + p over interior-cell
// indices i = row * (W - 2) + col - 1 (Rules::build checks the table against this closed form of the reference's
// enumeration order, :63-109), evaluated as iA * 2(NI - 1) + (2 * (iB - gt) + 1 + p) on two 16-bit halves at a time.
// j = i + 1 comes straight out of one packed multiply-add per player; every byte stays within 0..255 and never borrows:
// 2 * jB + p <= 2 * NI + 3 <= 255 (swar::fits), and jB >= 2 whenever iB > iA.  ZERO: lanes flagged in `zero7` (goal
// tuples, :493-494) get index 0 (their bytes above are meaningless but still borrow-free: a carrier in a goal column
// stands in a goal row >= 1, so j >= W - 2).
template <bool ZERO>
SOCCER_HD void obs4(const Consts& C, uint32_t r_a, uint32_t c_a, uint32_t r_b, uint32_t c_b, uint32_t p01, uint32_t zero7,
                    uint32_t& lo, uint32_t& hi) {
    const uint32_t jA = pk_mad(r_a, C.Wm2x2, c_a), jB = pk_mad(r_b, C.Wm2x2, c_b);
    const uint32_t gt2 = (((jB | K80) - jA) >> 6) & 0x02020202u;           // 2 * (iB > iA)
    const uint32_t iA = jA - K01;
    const uint32_t q = (jB << 1) + p01 - gt2 - K01;                         // 2 * (iB - gt) + 1 + p
    lo = pk_mad(perm(0u, iA, 0x0c010c00u), C.obs_mul, perm(0u, q, 0x0c010c00u));
    hi = pk_mad(perm(0u, iA, 0x0c030c02u), C.obs_mul, perm(0u, q, 0x0c030c02u));
    if (ZERO) {
        const uint32_t z = mask_of(zero7);
        lo = bfi(perm(0u, z, 0x01010000u), 0u, lo); hi = bfi(perm(0u, z, 0x03030202u), 0u, hi);
    }
}

// reset (:410-424) of four lanes: every lane (MASKED = false: nothing is read) or the lanes whose mask byte is non-zero.
// The ISD entry is drawn with the lane's reset draw (Rand4::rs), as in the in-step reset; lanes that are not
// selected keep their state and report the observation of the tuple they hold (goal tuples: 0).
template <bool MASKED>
SOCCER_HD void reset4(const Consts& C, Group& S, uint32_t mask4, const Rand4& R, uint32_t& obs_lo, uint32_t& obs_hi) {
    const uint32_t idx = R.rs & C.isd_mask;
    const uint32_t ira = perm(0u, C.isd_ra, idx), irb = perm(0u, C.isd_rb, idx), ip = perm(0u, C.isd_p, idx);
    uint32_t zero7 = 0u;
    if (MASKED) {
        const uint32_t m = mask_of(((mask4 & K7F) + K7F) | mask4);         // byte != 0
        S.ra = bfi(m, ira, S.ra); S.ca = bfi(m, C.isd_ca4, S.ca); S.rb = bfi(m, irb, S.rb); S.cb = bfi(m, C.isd_cb4, S.cb);
        S.ps = bfi(m, ip, S.ps); S.tt = bfi(m, 0u, S.tt);                  // needs_reset cleared, timestep 0 (:422-423)
        const uint32_t cc = bfi(mask_of(S.ps << 7), S.cb, S.ca);
        zero7 = is_zero(cc) | is_zero(cc ^ C.Wm1x4);
    } else {
        S.ra = ira; S.ca = C.isd_ca4; S.rb = irb; S.cb = C.isd_cb4; S.ps = ip; S.tt = 0u;
    }
    obs4<MASKED>(C, S.ra, S.ca, S.rb, S.cb, S.ps & K01, zero7, obs_lo, obs_hi);
}

// GENERAL = false: the steady state of an auto-resetting handle — no lane is frozen or stands in a goal tuple on
//   entry (so none ever will): the code for those cases is compiled out.  The host tracks when that holds.
// FULL: also produce final_obs and prob_code (VectorSoccerEnv's info / final_observation).
// SLIP: `sa` / `sb` are the slipped moves of the combination the caller selected per lane (slip_select4) and `k4`
//   (one byte per lane) the QUARTER of that combination's probability mass the draw fell into: the outcome index of a
//   four-way list, and twice the outcome index of a two-way list (the host checks that the scaled mid-point threshold
//   of a combination equals its second quarter threshold); `cls4` its weight class for prob_code.  Otherwise the moves
//   are the actions and the quarter is the top two bits of each lane's random word (slip_prob == 0: list
//   probabilities are 1, .5/.5 or .25 x 4, i.e. floor(2u) / floor(4u)).
// R: the lanes' random bits of this tick (Rand4); with SLIP its kq is the quarter from slip_select4, k4 << 6.
// TRUSTED: the action bytes were produced by the kernel itself (sampled in 0..4): no canonical form, no misuse report.
// GEO: 0 the pitch geometry by arithmetic (any pitch that fits), 1 by byte tables (Consts::small).
template <bool GENERAL, bool FULL, bool SLIP, int GEO = 0, bool TRUSTED = false>
SOCCER_HD void step4(const Consts& C, Group& S, uint32_t aa_raw, uint32_t ab_raw, uint32_t sa, uint32_t sb,
                     uint32_t cls4, const Rand4& R, Out& o) {
    const uint32_t ra = S.ra, ca = S.ca, rb = S.rb, cb = S.cb, ps = S.ps, t = S.tt;
    // ---- actions: the low three bits select the move, 5..7 are NOOP; any byte outside 0..4 is reported ---------------
    uint32_t aa = aa_raw, ab = ab_raw;
    if (TRUSTED) o.bad_action = 0u;
    else { aa = canon4(aa_raw); ab = canon4(ab_raw); o.bad_action = (aa ^ aa_raw) | (ab ^ ab_raw); }
    const uint32_t p7 = ps << 7;                                           // flag: B has the ball
    const uint32_t pm = mask_of(p7);
    uint32_t frz7 = 0u, hold_m = 0u, frz_m = 0u;
    if (GENERAL) {
        // lanes that are left as they are: frozen (needs_reset, :376) or in a goal tuple (absorbing, :300-301)
        frz7 = ps << 6;
        const uint32_t cc0 = bfi(pm, cb, ca);                              // the carrier's column
        const uint32_t hold7 = frz7 | is_zero(cc0) | is_zero(cc0 ^ C.Wm1x4);
        hold_m = mask_of(hold7); frz_m = mask_of(frz7);
        aa = bfi(hold_m, 0u, aa); ab = bfi(hold_m, 0u, ab);                // NOOP / NOOP leaves the tuple unchanged
        if (SLIP) { sa = bfi(hold_m, 0u, sa); sb = bfi(hold_m, 0u, sb); }
    }
    if (!SLIP) { sa = aa; sb = ab; } else { sa &= 0x07070707u; sb &= 0x07070707u; }
    // ---- tentative cells (:307-312) ---------------------------------------------------------------------------------
    uint32_t gra, grb;                                                     // flag: row in the goal rows
    if (GEO == 1) { gra = perm(C.grow_hi, C.grow_lo, ra); grb = perm(C.grow_hi, C.grow_lo, rb); }
    else { gra = bfi(ra + C.gr_hi_add, 0u, ra + C.gr_lo_add); grb = bfi(rb + C.gr_hi_add, 0u, rb + C.gr_lo_add); }
    uint32_t nra, nca, nrb, ncb;
    move4<GEO>(C, ra, ca, sa, bfi(p7, 0u, gra), nra, nca);                 // A holds the ball when p == 0
    move4<GEO>(C, rb, cb, sb, grb & p7, nrb, ncb);
    // ---- ordered collision resolution (:315-360) on cell ids ---------------------------------------------------------
    const uint32_t A = pk_mad(ra, C.Wx2, ca), B = pk_mad(rb, C.Wx2, cb);
    const uint32_t nA = pk_mad(nra, C.Wx2, nca), nB = pk_mad(nrb, C.Wx2, ncb);
    const uint32_t e1 = is_zero(nA ^ B), e2 = is_zero(nB ^ A);
    const uint32_t sA = is_zero(nA ^ A), sB = is_zero(nB ^ B), same = is_zero(nA ^ nB);
    const uint32_t nzA = lut8(T_NZ_HI, T_NZ_LO, aa), nzB = lut8(T_NZ_HI, T_NZ_LO, ab);   // the ORIGINAL actions (:330-339)
    const uint32_t swap = e1 & e2;                                         // :315-322
    const uint32_t stander = bfi(nzB, 0u, e1) | bfi(nzA, 0u, e2);          // :330-331
    const uint32_t bounce = ((sA & nzA) & e2) | ((sB & nzB) & e1);         // :338-339
    const uint32_t coin = swap | bfi(stander, 0u, bounce);                 // two outcomes .5/.5 (:326-327, :343-344)
    const uint32_t flip = bfi(coin, 0u, stander);                          // possession changes hands (:335)
    const uint32_t cs = coin | stander;
    const uint32_t four = bfi(cs, 0u, same);                               // four outcomes .25 each (:352-356)
    const uint32_t mv = ~(cs | same);                                      // both move (:360)
    // ---- the outcome draw ---------------------------------------------------------------------------------------------
    // kb1 / kb0: flag words of bit 1 / bit 0 of the outcome index k (coin lists use k = bit 1 of the two-bit draw)
    // quarter 0..3 of the draw: a coin list takes bit 1 (floor(2u)), a four-way list both bits (floor(4u))
    const uint32_t kb1 = R.kq, kb0c = R.kq, kb0f = R.kq << 1;              // kb0 for coin lists / for four-way lists
    const uint32_t amv = (four & kb1) | mv;                                // A takes its cell: k >= 2 of a four-way tie
    const uint32_t bmv = bfi(kb1, mv, four | mv);                          // B takes its cell: k < 2
    const uint32_t am = mask_of(amv), bm = mask_of(bmv);
    uint32_t fra = bfi(am, nra, ra), fca = bfi(am, nca, ca), frb = bfi(bm, nrb, rb), fcb = bfi(bm, ncb, cb);
    const uint32_t p7n = bfi(coin | four, bfi(coin, kb0c, kb0f), p7 ^ flip);
    // ---- done / reward (:235-240), bookkeeping (:399-406) -------------------------------------------------------------
    const uint32_t pmn = mask_of(p7n);
    const uint32_t cc = bfi(pmn, fcb, fca);                                // the carrier's column after the step
    uint32_t goal7, rew, term01;                                           // +1 into B's goal line, -1 (0xff) into A's
    if (GEO == 1) {
        rew = perm(C.rew_hi, C.rew_lo, cc); term01 = perm(C.term_hi, C.term_lo, cc); goal7 = term01 << 7;
    } else {
        const uint32_t g0 = is_zero(cc), gW = is_zero(cc ^ C.Wm1x4);
        goal7 = g0 | gW; rew = one_of(gW) | mask_of(g0); term01 = one_of(goal7);
    }
    if (GENERAL) rew = bfi(hold_m, 0u, rew);                               // from a goal tuple: reward 0 (:235-236)
    uint32_t tt = t + (GENERAL ? bfi(frz_m, 0u, K01) : K01);               // :399 (a frozen lane keeps its timestep)
    const uint32_t trunc7 = tt + C.trunc_add;                              // :404
    const uint32_t need7 = goal7 | trunc7;                                 // :406
    const uint32_t fin7 = GENERAL ? bfi(frz7, 0u, need7) : need7;          // episodes that ended at this step
    o.rew = rew; o.term = term01; o.trunc = one_of(trunc7);
    o.finished = fin7; o.frozen = frz7 & K80;
    if (FULL) {
        // outcome class of the sampled entry: 0 single, 1 one of two, 2 one of four; slip class from the caller
        uint32_t code = one_of(coin) | ((four >> 6) & 0x02020202u);
        if (SLIP) code += cls4 + (cls4 << 1);
        if (GENERAL && SLIP) code = bfi(frz_m, 0u, code);
        o.code = code;
    }
    uint32_t p01 = one_of(p7n);
    if (FULL) obs4<true>(C, fra, fca, frb, fcb, p01, goal7, o.fin_lo, o.fin_hi);      // before any reset (goal tuples -> 0)
    // ---- in-step reset (:410-424): lanes whose episode ended draw an ISD entry with their two low random bits --------
    uint32_t need_out7 = GENERAL ? (need7 | frz7) : need7;
    uint32_t obs_zero7 = goal7;
    if (!GENERAL || C.autoreset) {                                         // wave-uniform
        const uint32_t rm = mask_of(fin7);
        const uint32_t idx = R.rs & C.isd_mask;                            // lane j's reset draw
        fra = bfi(rm, perm(0u, C.isd_ra, idx), fra); fca = bfi(rm, C.isd_ca4, fca);
        frb = bfi(rm, perm(0u, C.isd_rb, idx), frb); fcb = bfi(rm, C.isd_cb4, fcb);
        p01 = bfi(rm, perm(0u, C.isd_p, idx), p01);
        tt = bfi(rm, 0u, tt);
        need_out7 = frz7;
        obs_zero7 = bfi(fin7, 0u, goal7);                                  // only a frozen lane can still sit in a goal tuple
    }
    // the observation of the tuple the lane now holds; without frozen lanes no tuple is a goal tuple after the reset
    obs4<GENERAL>(C, fra, fca, frb, fcb, p01, obs_zero7, o.obs_lo, o.obs_hi);
    S.ra = fra; S.ca = fca; S.rb = frb; S.cb = fcb; S.tt = tt;
    S.ps = p01 | ((need_out7 >> 6) & 0x02020202u);
}

}  // namespace swar
}  // namespace soccer
