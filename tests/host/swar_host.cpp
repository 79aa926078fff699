// Host build of the product's byte-parallel step (gym_soccer_littman94_amd/csrc/soccer_swar.hpp) for the CPU test
// tests/test_swar_host.py: the four GPU builtins it uses are restated in plain C++ inside that header, everything
// else is the code the kernels run.  Test infrastructure; not part of libsoccer_hip.so.
#include <cstdint>
#include <cstring>
#include <string>

#include "../../gym_soccer_littman94_amd/csrc/soccer_rules.hpp"
#include "../../gym_soccer_littman94_amd/csrc/soccer_swar.hpp"
#include "../../gym_soccer_littman94_amd/csrc/soccer_slip.hpp"

using namespace soccer;

static uint32_t ld4(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
static void st4(uint8_t* p, uint32_t v) { std::memcpy(p, &v, 4); }

// n must be a multiple of 4.  state: six byte streams of n lanes (in / out).  words: one uint32 per lane.
// slip_prob > 0: the combination and the quarter are selected from the words by swar::slip_select4 with the tables of
// soccer_slip.hpp (what soccer_create builds).
// Returns 0, -1 when the pitch does not qualify for the byte-parallel path, -3 when the slip does not.
// The words are taken the one-word-per-lane-and-tick way (swar::rand_words: quarter draw = the two top bits, reset draw =
// the two low bits) at every slip_prob, so the rules are exercised with every draw independently of the RNG convention;
// swar_draws_host below checks the eight-ticks-per-block extraction of slip_prob == 0 handles.
static int geo = -1;     // -1: as the library (tables on small pitches); 0: force the arithmetic geometry
extern "C" void swar_set_geo(int g) { geo = g; }
static int sel_mode = 1; // slip selection: 1 threshold by threshold (slip_select4), 2 by the rollout's table (slip_select4_lut: 14 bits,
                         // one compare), 3 by the single step's (10 bits, two compares); -4 if the slip has no such table
extern "C" void swar_set_slip_select(int m) { sel_mode = m; }

extern "C" int swar_step_host(int width, int height, int max_steps, int autoreset, int general, int full, long n,
                              uint8_t* ra, uint8_t* ca, uint8_t* rb, uint8_t* cb, uint8_t* ps, uint8_t* tt,
                              const uint8_t* act_a, const uint8_t* act_b, const uint32_t* words, double slip_prob,
                              uint16_t* obs, uint16_t* final_obs, uint8_t* rew, uint8_t* term, uint8_t* trunc,
                              uint8_t* code, uint8_t* finished, uint8_t* frozen, uint8_t* bad) {
    Rules R;
    if (!R.build(width, height).empty()) return -2;
    if (!swar::fits(R.H, R.W, max_steps)) return -1;
    const swar::Consts C = swar::make_consts(R.H, R.W, R.goal_lo, R.goal_hi, max_steps, R.n_isd, R.isd, autoreset != 0);
    const bool slip = slip_prob != 0.0;
    SlipTables ST{}; swar::SlipConsts L{};
    if (slip) {
        ST = build_slip_tables(slip_prob);
        if (!ST.swar_ok) return -3;
        if (sel_mode == 2 && !ST.lut_ok) return -4;
        if (sel_mode == 3 && !ST.lut_step_ok) return -4;
        for (int i = 0; i < 9; ++i) L.CB[i] = ST.CB[i];
        L.c_off = ST.c_off;
    }
    for (long i = 0; i < n; i += 4) {
        swar::Group S{ld4(ra + i), ld4(ca + i), ld4(rb + i), ld4(cb + i), ld4(ps + i), ld4(tt + i)};
        swar::Out o{};
        const uint32_t a = ld4(act_a + i), b = ld4(act_b + i);
        const uint32_t* w = words + i;
        uint32_t s_a = 0u, s_b = 0u, k4 = 0u, c4 = 0u;
        if (slip && sel_mode == 2) swar::slip_select4_lut(ST.lut, ST.T, L.c_off, swar::canon4(a), swar::canon4(b), w[0], w[1], w[2], w[3], s_a, s_b, k4, c4);
        else if (slip && sel_mode == 3) swar::slip_select4_lut<kSlipStepBucketBits, kSlipStepCompares>(ST.lut_step, ST.T, L.c_off, swar::canon4(a), swar::canon4(b),
                                                                                                    w[0], w[1], w[2], w[3], s_a, s_b, k4, c4);
        else if (slip) swar::slip_select4(L, ST.sub, swar::canon4(a), swar::canon4(b), w[0], w[1], w[2], w[3], s_a, s_b, k4, c4);
        swar::Rand4 rnd = swar::rand_words(C.isd_shift, w[0], w[1], w[2], w[3]);
        if (slip) rnd.kq = k4 << 6;
        // geometry: byte tables where the pitch allows (what the library picks), arithmetic otherwise or when geo == 0 is forced
#define CALL(G, F, SL) do { if (C.small && geo != 0) swar::step4<G, F, SL, 1>(C, S, a, b, s_a, s_b, c4, rnd, o); \
                            else swar::step4<G, F, SL, 0>(C, S, a, b, s_a, s_b, c4, rnd, o); } while (0)
        if (slip) { if (general) { if (full) CALL(true, true, true); else CALL(true, false, true); }
                    else { if (full) CALL(false, true, true); else CALL(false, false, true); } }
        else { if (general) { if (full) CALL(true, true, false); else CALL(true, false, false); }
               else { if (full) CALL(false, true, false); else CALL(false, false, false); } }
#undef CALL
        st4(ra + i, S.ra); st4(ca + i, S.ca); st4(rb + i, S.rb); st4(cb + i, S.cb); st4(ps + i, S.ps); st4(tt + i, S.tt);
        obs[i] = (uint16_t)o.obs_lo; obs[i + 1] = (uint16_t)(o.obs_lo >> 16); obs[i + 2] = (uint16_t)o.obs_hi; obs[i + 3] = (uint16_t)(o.obs_hi >> 16);
        if (full) {
            final_obs[i] = (uint16_t)o.fin_lo; final_obs[i + 1] = (uint16_t)(o.fin_lo >> 16);
            final_obs[i + 2] = (uint16_t)o.fin_hi; final_obs[i + 3] = (uint16_t)(o.fin_hi >> 16);
            st4(code + i, o.code);
        }
        st4(rew + i, o.rew); st4(term + i, o.term); st4(trunc + i, o.trunc);
        st4(finished + i, (o.finished >> 7) & 0x01010101u); st4(frozen + i, (o.frozen >> 7) & 0x01010101u);
        bad[i >> 2] = o.bad_action != 0u;
    }
    return 0;
}

// batched_reset of the byte-parallel kernel: mask nullable (all lanes); words as above (reset draw = word & 3)
extern "C" int swar_reset_host(int width, int height, long n, uint8_t* ra, uint8_t* ca, uint8_t* rb, uint8_t* cb, uint8_t* ps,
                               uint8_t* tt, const uint8_t* mask, const uint32_t* words, uint16_t* obs) {
    Rules R;
    if (!R.build(width, height).empty()) return -2;
    if (!swar::fits(R.H, R.W, 100)) return -1;
    const swar::Consts C = swar::make_consts(R.H, R.W, R.goal_lo, R.goal_hi, 100, R.n_isd, R.isd, true);
    for (long i = 0; i < n; i += 4) {
        swar::Group S{ld4(ra + i), ld4(ca + i), ld4(rb + i), ld4(cb + i), ld4(ps + i), ld4(tt + i)};
        uint32_t lo, hi;
        const uint32_t* w = words + i;
        const swar::Rand4 rnd = swar::rand_words(C.isd_shift, w[0], w[1], w[2], w[3]);
        if (mask) swar::reset4<true>(C, S, ld4(mask + i), rnd, lo, hi);
        else swar::reset4<false>(C, S, 0u, rnd, lo, hi);
        st4(ra + i, S.ra); st4(ca + i, S.ca); st4(rb + i, S.rb); st4(cb + i, S.cb); st4(ps + i, S.ps); st4(tt + i, S.tt);
        obs[i] = (uint16_t)lo; obs[i + 1] = (uint16_t)(lo >> 16); obs[i + 2] = (uint16_t)hi; obs[i + 3] = (uint16_t)(hi >> 16);
    }
    return 0;
}

// The draws of slip_prob == 0 handles (include/soccer_hip.h, ABI 3): one block (four words, one per lane of the group)
// serves eight ticks.  For every tick t = 0..7 this reports what the kernels extract, both ways they do it:
//   out[(t * 2 + 0) * 4 + j] = quarter draw | reset draw << 4 of lane j by swar::rand_nibble (single steps, resets)
//   out[(t * 2 + 1) * 4 + j] = the same by swar::transpose4 + swar::rand_pair with the rotation the rollout loop applies
// isd_shift = 0: the reset draw is reported as the full two bits.
extern "C" void swar_draws_host(const uint32_t* w, uint8_t* out) {
    uint32_t p0, p1, p2, p3;
    swar::transpose4(w[0], w[1], w[2], w[3], p0, p1, p2, p3);
    for (uint32_t t = 0; t < 8; ++t) {
        const swar::Rand4 a = swar::rand_nibble(0u, t, w[0], w[1], w[2], w[3]);
        const swar::Rand4 b = swar::rand_pair(0u, t, p0);
        if (t & 1u) { p0 = p1; p1 = p2; p2 = p3; }
        for (int j = 0; j < 4; ++j) {
            out[(t * 2 + 0) * 4 + j] = (uint8_t)(((a.kq >> (8 * j + 6)) & 3u) | (((a.rs >> (8 * j)) & 3u) << 4));
            out[(t * 2 + 1) * 4 + j] = (uint8_t)(((b.kq >> (8 * j + 6)) & 3u) | (((b.rs >> (8 * j)) & 3u) << 4));
        }
    }
}

// what soccer_create derives from slip_prob (soccer_slip.hpp), for the CPU test of the integer slip decision
extern "C" void swar_slip_tables(double slip_prob, uint32_t* cb9, uint32_t* sub36, uint32_t* flags, double* w4) {
    const SlipTables T = build_slip_tables(slip_prob);
    for (int i = 0; i < 9; ++i) { cb9[i] = T.CB[i]; sub36[4 * i] = T.sub[i].x; sub36[4 * i + 1] = T.sub[i].y; sub36[4 * i + 2] = T.sub[i].z; sub36[4 * i + 3] = T.sub[i].w; }
    flags[0] = T.slip_int; flags[1] = T.swar_ok ? 1u : 0u; flags[2] = T.nb; flags[3] = T.c_off; flags[4] = T.lut_ok ? 1u : 0u; flags[5] = T.lut_step_ok ? 1u : 0u;
    for (int i = 0; i < 4; ++i) w4[i] = T.w[i];
}
