// soccer_slip.hpp — host-side construction of the slip-combination weights and thresholds (reference
// gym_soccer/envs/soccer_simultaneous_env.py:202-227, :241).  Host-only; shared by libsoccer_hip.so (soccer_create) and
// the CPU test harness of the byte-parallel step (tests/host/swar_host.cpp).
//
// What the kernels need to know about a slip_prob > 0 handle is where a draw falls in the (state, joint action) list the
// reference samples from (:395): the list holds, for each of the nine slip combinations with a non-zero weight (:209-227)
// in order, 1, 2 or 4 entries of probability weight * {1, .5, .25} (:241), and gym's categorical_sample picks the first
// entry whose SEQUENTIAL float64 running sum exceeds u.  A Philox draw is u = (m + 1/2) * 2^-30 with an integer m < 2^30
// (include/soccer_hip.h), so "running sum t <= u"  <=>  t * 2^30 - 1/2 <= m  <=>  m >= ceil(t * 2^30 - 1/2), and both the
// scaling and the subtraction are exact in float64: the decision is an integer comparison with c(t) = ceil(t * 2^30 - 1/2).
// The running sum t at a given entry depends on the SHAPE of the list before it (a combination that contributes four
// quarter entries does not always round like one that contributes a single entry), so build_slip_tables() walks every
// shape — at most 3^8 prefixes, a few dozen distinct running sums in practice — and accepts the integer decision only if
// c(t) is the same for every shape at every entry position.  It then IS the reference's decision for every draw of every
// lane; nothing is left to a margin.  (Round 2 compared u = m * 2^-30 against nominal thresholds with a 2^-40 safety
// margin: decimal slips such as 0.1 put mathematically dyadic thresholds like 27/32 exactly on a draw, where the float64
// rounding of the particular list decides — those handles needed a float64 walk inside the kernels.  With the half-step
// offset no draw can sit on such a threshold, and of 9 999 slips k / 10 000 and 2 016 fractions n / d, d <= 64, none is
// left ambiguous; 8 and 85 were.)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "soccer_swar.hpp"

namespace soccer {

constexpr int kSlipBucketBits = 14, kSlipBuckets = 1 << kSlipBucketBits;   // (kSlipThresholds = 40 and the single step's table: soccer_swar.hpp)
constexpr int kSlipLdsWords = kSlipBuckets / 4 + kSlipThresholds;       // what a kernel stages: the table, then the thresholds

struct SlipTables {
    double w[4];                 // weights c0..c3 of the four combination classes, float64 exactly as :211-222 evaluates them
    double B[9];                 // cumulative weight after each ACTIVE (non-zero) combination in reference order, +inf beyond
                                 // (nominal thresholds of the float64 fast path that serves caller-supplied uniforms)
    uint32_t nb;                 // number of active combinations
    unsigned long long act_pack; // their ids, 4 bits each
    // Integer form of the decision for Philox draws (see above): entry passed  <=>  m >= c
    uint32_t CB[9];              // c at the end of each active combination (0xFFFFFFFF beyond the active ones)
    swar::Quad sub[9];           // per active combination: { c of the first entry of a two-way list, c of entries 1, 2, 3 of a four-way list }
    uint32_t slip_int;           // 1: the integer decision is the reference's for every draw and every list shape; 0: it is not
                                 // (some entry's c depends on the shape, or a draw could fall beyond the last entry): float64 walk
    uint32_t c_off;              // id of the first active combination when they are consecutive (0, or 5 when slip_prob == 1)
    bool swar_ok;                // the byte-parallel kernels may use the integer decision: slip_int == 1, the active combinations
                                 // are consecutive ids, and a two-way list's first entry ends where a four-way list's second does
                                 // (one quarter index then serves both)
    // Table form of the same decision (swar::slip_select4_lut).  Every active combination owns four consecutive
    // thresholds — its quarter points 1, 2, 3 and its end — so with all of them in one ascending list T the number p of
    // thresholds <= m is 4 * (combination position) + quarter.  `lut` gives, for each 2^16-wide bucket of m, how many
    // thresholds lie at or below the bucket's first draw; the at most ONE that lies inside the bucket is compared exactly.
    uint32_t T[kSlipThresholds]; // ascending, 0xFFFFFFFF beyond the 4 * nb real ones
    uint8_t lut[kSlipBuckets];
    bool lut_ok;                 // swar_ok, T ascending, and no bucket holds two thresholds (needs s^2 / 16 and (1 - s)^2 / 4 >= 2^-14:
                                 // slips within about [0.032, 0.984]; the others compare threshold by threshold)
    uint8_t lut_step[kSlipStepBuckets];   // the same over 2^20-wide buckets (step_kernel_swar<.., 2, ..>)
    bool lut_step_ok;            // swar_ok, T ascending, at most kSlipStepCompares thresholds inside any bucket (slips within about
                                 // [0.09, 0.96])
};

inline SlipTables build_slip_tables(double slip_prob) {
    SlipTables T{};
    static const int cls[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
    {
        volatile double s = slip_prob;        // volatile: no reassociation / contraction
        volatile double one_minus = 1 - s;
        volatile double c0 = one_minus * one_minus;
        volatile double c1a = one_minus * s;  volatile double c1 = c1a * 0.5;
        volatile double c2a = s * one_minus;  volatile double c2 = c2a * 0.5;
        volatile double c3a = s * s;          volatile double c3 = c3a * 0.25;
        T.w[0] = c0; T.w[1] = c1; T.w[2] = c2; T.w[3] = c3;
        volatile double acc = 0.0;
        T.nb = 0; T.act_pack = 0;
        for (int c = 0; c < 9; ++c) T.B[c] = __builtin_inf();
        for (int c = 0; c < 9; ++c) {
            const double wc = T.w[cls[c]];
            if (wc == 0.0) continue;                                    // :226-227
            acc = acc + wc;
            T.B[T.nb] = acc; T.act_pack |= (unsigned long long)c << (4 * T.nb); ++T.nb;
        }
    }
    for (int i = 0; i < 9; ++i) { T.CB[i] = 0xFFFFFFFFu; T.sub[i] = swar::Quad{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}; }
    bool ok = slip_prob != 0.0 && T.nb >= 1;
    // c(t) = ceil(t * 2^30 - 1/2), clamped to 2^30 ("no draw reaches this entry's end": every m is below it)
    constexpr uint32_t kNone = 0xFFFFFFFEu;
    auto c_of = [](double t) -> uint32_t {
        volatile double x = t * 0x1.0p30;                               // exact
        volatile double y = x - 0.5;                                    // exact (see the header comment)
        const double c = __builtin_ceil(y);
        if (!(c > 0.0)) return 0u;
        return c >= 0x1.0p30 ? (1u << 30) : (uint32_t)c;
    };
    auto merge = [&](uint32_t& slot, uint32_t c) { if (slot == kNone) slot = c; else if (slot != c) ok = false; };
    std::vector<double> prefix{0.0};                                    // distinct running sums in front of combination i
    for (uint32_t i = 0; i < T.nb && i < 9; ++i) {
        const double wc = T.w[cls[(T.act_pack >> (4 * i)) & 0xf]];
        uint32_t end = kNone, two0 = kNone, four[3] = {kNone, kNone, kNone};
        std::vector<double> next;
        for (const double S : prefix) {
            for (int n = 1; n <= 4; n <<= 1) {                          // the combination contributes 1, 2 or 4 equal entries
                volatile double q = wc * (n == 1 ? 1.0 : (n == 2 ? 0.5 : 0.25));   // :241
                volatile double acc = S;
                for (int k = 0; k < n; ++k) {
                    acc = acc + q;                                      // the sequential cumsum of categorical_sample
                    const uint32_t c = c_of(acc);
                    if (k == n - 1) merge(end, c);
                    else if (n == 2) merge(two0, c);
                    else merge(four[k], c);
                }
                next.push_back((double)acc);
            }
        }
        std::sort(next.begin(), next.end());
        next.erase(std::unique(next.begin(), next.end()), next.end());
        prefix.swap(next);
        T.CB[i] = end; T.sub[i] = swar::Quad{two0, four[0], four[1], four[2]};
    }
    // a draw beyond the last entry's end would make categorical_sample return index 0 (argmax of all-False)
    if (ok && T.CB[T.nb - 1] < (1u << 30)) ok = false;
    T.slip_int = ok ? 1u : 0u;
    bool consecutive = T.nb >= 1;
    T.c_off = (uint32_t)(T.act_pack & 0xf);
    for (uint32_t i = 0; i < T.nb; ++i) consecutive = consecutive && ((T.act_pack >> (4 * i)) & 0xf) == T.c_off + i;
    bool mid = true;
    for (uint32_t i = 0; i < T.nb; ++i) mid = mid && T.sub[i].x == T.sub[i].z;
    T.swar_ok = T.slip_int == 1u && consecutive && mid;
    // the table form
    for (int j = 0; j < kSlipThresholds; ++j) T.T[j] = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < T.nb && i < 9; ++i) { T.T[4 * i] = T.sub[i].y; T.T[4 * i + 1] = T.sub[i].z; T.T[4 * i + 2] = T.sub[i].w; T.T[4 * i + 3] = T.CB[i]; }
    bool lut_ok = T.swar_ok;
    for (uint32_t j = 1; j < 4 * T.nb; ++j) lut_ok = lut_ok && T.T[j - 1] <= T.T[j];
    // lut[b] = how many thresholds lie at or below the first draw of bucket b; returns the largest number strictly inside one bucket
    auto fill = [&](int bits, uint8_t* lut) -> uint32_t {
        const int shift = 30 - bits;
        uint32_t worst = 0;
        for (int b = 0; b < (1 << bits); ++b) {
            const uint32_t lo = (uint32_t)b << shift, hi = lo + (1u << shift);
            uint32_t below = 0, inside = 0;
            for (uint32_t j = 0; j < 4 * T.nb; ++j) { if (T.T[j] <= lo) ++below; else if (T.T[j] < hi) ++inside; }
            lut[b] = (uint8_t)below;
            worst = std::max(worst, inside);
        }
        return worst;
    };
    const bool ascending = lut_ok;
    if (fill(kSlipBucketBits, T.lut) > 1u) lut_ok = false;
    T.lut_step_ok = ascending && fill(kSlipStepBucketBits, T.lut_step) <= (uint32_t)kSlipStepCompares;
    T.lut_ok = lut_ok;
    return T;
}

}  // namespace soccer
