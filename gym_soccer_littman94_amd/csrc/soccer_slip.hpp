// soccer_slip.hpp — host-side construction of the slip-combination weights and thresholds (reference
// gym_soccer/envs/soccer_simultaneous_env.py:202-227, :241).  Host-only; shared by libsoccer_hip.so (soccer_create) and
// the CPU test harness of the byte-parallel step (tests/host/swar_host.cpp).
#pragma once
#include <cmath>
#include <cstdint>

#include "soccer_swar.hpp"

namespace soccer {

struct SlipTables {
    double w[4];                 // weights c0..c3 of the four combination classes, float64 exactly as :211-222 evaluates them
    double B[9];                 // cumulative weight after each ACTIVE (non-zero) combination in reference order, +inf beyond
    uint32_t nb;                 // number of active combinations
    unsigned long long act_pack; // their ids, 4 bits each
    // Integer form of the decision for draws that come from a Philox word (u = m * 2^-30, m < 2^30): scaling a float64
    // threshold by 2^30 is exact, so u >= b <=> m >= ceil(b * 2^30).
    uint32_t CB[9];              // scaled B (0xFFFFFFFF beyond the active ones)
    swar::Quad sub[9];           // per active combination: { mid-point of a two-way list, quarter points of a four-way list }, scaled
    uint32_t slip_int;           // 0: float64 only; 1: the integer decision is exact for every draw; 2: exact except for draws in `danger`
    uint32_t danger[4];
    uint32_t c_off;              // id of the first active combination when they are consecutive (0, or 5 when slip_prob == 1)
    bool swar_ok;                // the byte-parallel kernels may use the integer decision: slip_int == 1 (or 2: then a thread that
                                 // draws a dangerous integer leaves the byte-parallel path for that step), the active combinations are
                                 // consecutive ids, and every mid-point threshold equals the second quarter point
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
        // nominal thresholds of the slip fast path: running sum of the active weights in list order
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
    // Allowed only if no scaled threshold lies within 2^-10 of an integer (then no draw can be within 2^-40 of a
    // threshold and the nominal decision is the exact one) and the last cumulative weight exceeds every possible draw.
    bool ok = slip_prob != 0.0 && T.nb >= 1;
    uint32_t danger[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}; int n_danger = 0;
    // Dyadic slips (0.5, 0.25, 0.75, 1.0 ...): every weight is a short binary fraction and every float64 sum of the
    // lists is EXACT, so the nominal thresholds ARE the running sums and the integer comparison is the reference's
    // comparison even when a draw sits exactly on a threshold.  Checked with error-free sums.
    bool exact = true;
    auto add_exact = [&](double a, double b) {                          // Fast2Sum: the rounding error of a + b
        volatile double sum = a + b; volatile double bb = sum - a; volatile double err = (a - (sum - bb)) + (b - bb);
        if (err != 0.0) exact = false;
        return (double)sum;
    };
    {
        double acc = 0.0;
        for (int c = 0; c < 9; ++c) {
            const double wc = T.w[cls[c]];
            if (wc == 0.0) continue;
            double t2 = acc; for (int j = 0; j < 2; ++j) t2 = add_exact(t2, wc * 0.5);     // .5/.5 lists
            double t4 = acc; for (int j = 0; j < 4; ++j) t4 = add_exact(t4, wc * 0.25);    // .25 x 4 lists
            acc = add_exact(acc, wc);
            if (t2 != acc || t4 != acc) exact = false;
        }
        volatile double s1 = slip_prob; volatile double om = 1 - s1;
        if (add_exact(om, s1) != 1.0) exact = false;
        // the weight products themselves must be exact too: compare with long double
        const long double S = slip_prob, O = 1.0L - S;
        if ((long double)T.w[0] != O * O || (long double)T.w[1] != O * S * 0.5L || (long double)T.w[2] != S * O * 0.5L ||
            (long double)T.w[3] != S * S * 0.25L || (long double)(double)O != O) exact = false;
    }
    auto scaled = [&](double t, uint32_t& out) {
        const double x = t * 0x1.0p30;                                  // exact
        if (!(x >= 0.0) || x > 0x1.0p31) { ok = false; out = 0xFFFFFFFFu; return; }
        const double r = __builtin_nearbyint(x);
        // a draw m = r (< 2^30) could sit on / next to the threshold: only safe when the sums are exact;
        // otherwise remember r — a lane that draws it walks the float64 sums (slip_int = 2)
        if (!exact && __builtin_fabs(x - r) < 0x1.0p-10 && r < 0x1.0p30) {
            const uint32_t ri = (uint32_t)r;
            bool seen = false;
            for (int q = 0; q < n_danger && q < 4; ++q) seen = seen || danger[q] == ri;
            if (!seen) { if (n_danger < 4) danger[n_danger] = ri; ++n_danger; }
        }
        out = (uint32_t)__builtin_ceil(x);
    };
    for (int i = 0; i < 9; ++i) { T.CB[i] = 0xFFFFFFFFu; T.sub[i] = swar::Quad{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}; }
    for (uint32_t i = 0; i < T.nb && i < 9; ++i) {
        scaled(T.B[i], T.CB[i]);
        const int c = (int)((T.act_pack >> (4 * i)) & 0xf);
        volatile double S = i ? T.B[i - 1] : 0.0;
        volatile double q2 = T.w[cls[c]] * 0.5, q4 = T.w[cls[c]] * 0.25;
        volatile double a1 = S + q2;                                     // two outcomes: t1
        volatile double b1 = S + q4; volatile double b2 = b1 + q4; volatile double b3 = b2 + q4;   // four: t1, t2, t3
        scaled(a1, T.sub[i].x); scaled(b1, T.sub[i].y); scaled(b2, T.sub[i].z); scaled(b3, T.sub[i].w);
    }
    if (ok && T.CB[T.nb - 1] < (1u << 30)) ok = false;                 // some draw would fall beyond the last entry
    T.slip_int = !ok || n_danger > 4 ? 0u : (n_danger ? 2u : 1u);
    for (int q = 0; q < 4; ++q) T.danger[q] = danger[q];
    // what the byte-parallel kernels additionally rely on
    bool consecutive = T.nb >= 1;
    T.c_off = (uint32_t)(T.act_pack & 0xf);
    for (uint32_t i = 0; i < T.nb; ++i) consecutive = consecutive && ((T.act_pack >> (4 * i)) & 0xf) == T.c_off + i;
    // mid-point == second quarter point, or they differ by one around a dangerous integer (then the only draw that tells
    // them apart is that integer, which never takes the integer decision)
    auto is_danger = [&](uint32_t m) { bool d = false; for (int q = 0; q < 4; ++q) d = d || T.danger[q] == m; return d; };
    bool mid = true;
    for (uint32_t i = 0; i < T.nb; ++i) {
        const uint32_t x = T.sub[i].x, z = T.sub[i].z;
        mid = mid && (x == z || (x + 1 == z && is_danger(x)) || (z + 1 == x && is_danger(z)));
    }
    T.swar_ok = (T.slip_int == 1u || T.slip_int == 2u) && consecutive && mid;
    return T;
}

}  // namespace soccer
