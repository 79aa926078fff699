/*
 * soccer_oracle.c — CPU restatement of the reference's step/reset path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The shipped path (libsoccer_hip.so) neither links
 * nor calls anything in oracle/.
 *
 * It restates, in plain C, what mimoralea/gym-soccer-littman94 does in
 *   gym_soccer/envs/soccer_simultaneous_env.py
 * and deliberately keeps the reference's *shape*: the constructor enumerates every state tuple and
 * joint action and materialises ordered transition lists (:167-293); step() is a list lookup plus
 * one categorical sample (:393-396).  (The HIP kernel instead evaluates the rules arithmetically
 * per lane, so the two are independent statements of the same rules.)
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks this file row-for-row — list order and
 * float64 probabilities included — against tests/golden/table_*.npz, replay_*.npz, reset_*.npz and
 * traj_*.npz, which tests/golden/make_golden.py dumped from the real reference in the build
 * container.
 *
 * Third-party arithmetic: gym 0.26.2 gym/envs/toy_text/utils.py categorical_sample (not vendored
 * by the reference; call sites :395, :414) = argmax(cumsum(asarray(p)) > np_random.random()):
 * sequential float64 running sum, first index that exceeds u, index 0 when none does.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_OUTCOMES 36   /* 9 slip combinations x up to 4 collision outcomes (:209-223, :352-356) */

typedef struct {
    int32_t n;
    double  p[MAX_OUTCOMES];
    int32_t ns[MAX_OUTCOMES];     /* flat tuple index of the next state */
    int8_t  r[MAX_OUTCOMES];
    uint8_t d[MAX_OUTCOMES];
    uint8_t combo[MAX_OUTCOMES];  /* which of the 9 slip combinations produced the entry */
    uint8_t nsp_code[MAX_OUTCOMES]; /* 0: 1.0, 1: 0.5, 2: 0.25 */
} trans_list;

typedef struct soc_oracle {
    int32_t H, W;                 /* W is the internal width = pitch width + 2 (:48) */
    double  slip;
    int32_t n_goal_rows;
    int32_t goal_rows[3];
    int32_t n_tuples;
    uint8_t* kind;                /* 0 unreachable, 1 live, 2 goal */
    uint16_t* lut;                /* observation index; 0 for goal tuples; 0xFFFF unreachable */
    int8_t* goal_value;
    int32_t nS;
    int32_t n_isd;
    int32_t isd[4];               /* flat tuple indices */
    double  isd_p[4];
    trans_list* P;                /* [n_tuples][25], only filled for kind != 0 */
    int32_t max_steps;
} soc_oracle;

/* action -> (dcol, drow)  (:24-30) */
static const int MOVE_DC[5] = {0, 0, 0, 1, -1};
static const int MOVE_DR[5] = {0, -1, 1, 0, 0};

static int in_goal_rows(const soc_oracle* o, int x) {
    for (int i = 0; i < o->n_goal_rows; ++i) if (o->goal_rows[i] == x) return 1;
    return 0;
}
static int in_goal_cols(const soc_oracle* o, int y) { return y == 0 || y == o->W - 1; }

static int32_t flat(const soc_oracle* o, int xa, int ya, int xb, int yb, int p) {
    return (((xa * o->W + ya) * o->H + xb) * o->W + yb) * 2 + p;
}
static void unflat(const soc_oracle* o, int32_t f, int* xa, int* ya, int* xb, int* yb, int* p) {
    *p = f & 1; f >>= 1;
    *yb = f % o->W; f /= o->W;
    *xb = f % o->H; f /= o->H;
    *ya = f % o->W; f /= o->W;
    *xa = f;
}

/* _next_cell (:364-373) */
static void next_cell(const soc_oracle* o, int x, int y, int dc, int dr, int has_ball, int* nx, int* ny) {
    int tx = x + dr;
    if (tx > o->H - 1) tx = o->H - 1;
    if (tx < 0) tx = 0;
    int ty = y + dc;
    int xoob = (ty == 0 || ty == o->W - 1);
    int goal = xoob && in_goal_rows(o, tx) && has_ball;
    if (xoob && !goal) ty = y;
    *nx = tx; *ny = ty;
}

typedef struct { int n; double p[4]; int32_t ns[4]; } outcome_list;

/* _get_next_state (:296-362); aa/ab are the ORIGINAL action ints, (dca,dra)/(dcb,drb) the
 * possibly slipped moves. */
static void get_next_state(const soc_oracle* o, int32_t st, int aa, int ab,
                           int dca, int dra, int dcb, int drb, outcome_list* out) {
    int xa, ya, xb, yb, p;
    unflat(o, st, &xa, &ya, &xb, &yb, &p);
    if (o->kind[st] == 2) {                       /* :300-301 */
        out->n = 1; out->p[0] = 1.0; out->ns[0] = st; return;
    }
    int nxa, nya, nxb, nyb;
    next_cell(o, xa, ya, dca, dra, p == 0, &nxa, &nya);
    next_cell(o, xb, yb, dcb, drb, p == 1, &nxb, &nyb);
    if ((xa == xb && abs(ya - yb) == 1 && nya == yb && nyb == ya) ||
        (ya == yb && abs(xa - xb) == 1 && nxa == xb && nxb == xa)) {            /* :315-327 */
        out->n = 2;
        out->p[0] = 0.5; out->ns[0] = flat(o, xa, ya, xb, yb, 0);
        out->p[1] = 0.5; out->ns[1] = flat(o, xa, ya, xb, yb, 1);
    } else if ((nxa == xb && nya == yb && ab == 0) || (nxb == xa && nyb == ya && aa == 0)) { /* :330-335 */
        out->n = 1; out->p[0] = 1.0; out->ns[0] = flat(o, xa, ya, xb, yb, 1 - p);
    } else if ((xa == nxa && ya == nya && aa != 0 && nxb == xa && nyb == ya) ||
               (xb == nxb && yb == nyb && ab != 0 && nxa == xb && nya == yb)) {  /* :338-344 */
        out->n = 2;
        out->p[0] = 0.5; out->ns[0] = flat(o, xa, ya, xb, yb, 0);
        out->p[1] = 0.5; out->ns[1] = flat(o, xa, ya, xb, yb, 1);
    } else if (nxa == nxb && nya == nyb) {                                      /* :347-356 */
        out->n = 4;
        out->p[0] = 0.25; out->ns[0] = flat(o, xa, ya, nxb, nyb, 0);
        out->p[1] = 0.25; out->ns[1] = flat(o, xa, ya, nxb, nyb, 1);
        out->p[2] = 0.25; out->ns[2] = flat(o, nxa, nya, xb, yb, 0);
        out->p[3] = 0.25; out->ns[3] = flat(o, nxa, nya, xb, yb, 1);
    } else {                                                                    /* :357-360 */
        out->n = 1; out->p[0] = 1.0; out->ns[0] = flat(o, nxa, nya, nxb, nyb, p);
    }
}

/* the transition list of one (state, joint action) (:196-256) */
static void build_list(const soc_oracle* o, int32_t st, int aa, int ab, trans_list* tl) {
    const double s = o->slip;
    /* intended moves and their two orthogonals (:203-206): [(-m[1], m[0]), (m[1], -m[0])] */
    int ma_c = MOVE_DC[aa], ma_r = MOVE_DR[aa];
    int mb_c = MOVE_DC[ab], mb_r = MOVE_DR[ab];
    int mas_c[2] = {-ma_r, ma_r}, mas_r[2] = {ma_c, -ma_c};
    int mbs_c[2] = {-mb_r, mb_r}, mbs_r[2] = {mb_c, -mb_c};
    /* the nine combinations in reference order with float64 weights as written (:209-223) */
    int cac[9], car[9], cbc[9], cbr[9]; double cw[9];
    cac[0] = ma_c;     car[0] = ma_r;     cbc[0] = mb_c;     cbr[0] = mb_r;     cw[0] = (1 - s) * (1 - s);
    cac[1] = ma_c;     car[1] = ma_r;     cbc[1] = mbs_c[0]; cbr[1] = mbs_r[0]; cw[1] = (1 - s) * s * 0.5;
    cac[2] = ma_c;     car[2] = ma_r;     cbc[2] = mbs_c[1]; cbr[2] = mbs_r[1]; cw[2] = (1 - s) * s * 0.5;
    cac[3] = mas_c[0]; car[3] = mas_r[0]; cbc[3] = mb_c;     cbr[3] = mb_r;     cw[3] = s * (1 - s) * 0.5;
    cac[4] = mas_c[1]; car[4] = mas_r[1]; cbc[4] = mb_c;     cbr[4] = mb_r;     cw[4] = s * (1 - s) * 0.5;
    cac[5] = mas_c[0]; car[5] = mas_r[0]; cbc[5] = mbs_c[0]; cbr[5] = mbs_r[0]; cw[5] = s * s * 0.25;
    cac[6] = mas_c[0]; car[6] = mas_r[0]; cbc[6] = mbs_c[1]; cbr[6] = mbs_r[1]; cw[6] = s * s * 0.25;
    cac[7] = mas_c[1]; car[7] = mas_r[1]; cbc[7] = mbs_c[0]; cbr[7] = mbs_r[0]; cw[7] = s * s * 0.25;
    cac[8] = mas_c[1]; car[8] = mas_r[1]; cbc[8] = mbs_c[1]; cbr[8] = mbs_r[1]; cw[8] = s * s * 0.25;
    tl->n = 0;
    for (int c = 0; c < 9; ++c) {
        if (cw[c] == 0) continue;                                  /* :226-227 */
        outcome_list ol;
        get_next_state(o, st, aa, ab, cac[c], car[c], cbc[c], cbr[c], &ol);
        for (int k = 0; k < ol.n; ++k) {
            int32_t ns = ol.ns[k];
            int d; int r;
            if (st == ns && o->kind[st] == 2)      { d = 1; r = 0; }                  /* :235-236 */
            else if (st != ns && o->kind[ns] == 2) { d = 1; r = o->goal_value[ns]; }  /* :237-238 */
            else                                   { d = 0; r = 0; }                  /* :239-240 */
            int i = tl->n++;
            tl->p[i] = cw[c] * ol.p[k];                                               /* :241 */
            tl->ns[i] = ns; tl->r[i] = (int8_t)r; tl->d[i] = (uint8_t)d;
            tl->combo[i] = (uint8_t)c;
            tl->nsp_code[i] = ol.p[k] == 1.0 ? 0 : (ol.p[k] == 0.5 ? 1 : 2);
        }
    }
}

soc_oracle* soc_oracle_create(int width, int height, double slip) {
    if (width < 5 || height < 4) return NULL;                       /* :45-46 */
    soc_oracle* o = (soc_oracle*)calloc(1, sizeof(soc_oracle));
    o->W = width + 2; o->H = height; o->slip = slip; o->max_steps = 100;
    if (height % 2 == 0) {                                          /* :60 */
        o->n_goal_rows = 2; o->goal_rows[0] = (height - 1) / 2; o->goal_rows[1] = height / 2;
    } else {
        o->n_goal_rows = 3; o->goal_rows[0] = height / 2 - 1; o->goal_rows[1] = height / 2;
        o->goal_rows[2] = height / 2 + 1;
    }
    o->n_tuples = o->H * o->W * o->H * o->W * 2;
    o->kind = (uint8_t*)calloc(o->n_tuples, 1);
    o->lut = (uint16_t*)malloc(sizeof(uint16_t) * o->n_tuples);
    o->goal_value = (int8_t*)calloc(o->n_tuples, 1);
    o->nS = 1;                                                      /* terminal state is 0 (:64-65) */
    for (int xa = 0; xa < o->H; ++xa) for (int ya = 0; ya < o->W; ++ya)
    for (int xb = 0; xb < o->H; ++xb) for (int yb = 0; yb < o->W; ++yb)
    for (int p = 0; p < 2; ++p) {
        int32_t f = flat(o, xa, ya, xb, yb, p);
        o->lut[f] = 0xFFFF;
        int a_goal_cell = in_goal_rows(o, xa) && in_goal_cols(o, ya);
        int b_goal_cell = in_goal_rows(o, xb) && in_goal_cols(o, yb);
        if ((in_goal_cols(o, ya) && !in_goal_rows(o, xa)) ||
            (in_goal_cols(o, yb) && !in_goal_rows(o, xb))) continue;            /* :74-77 */
        if ((a_goal_cell && p != 0) || (b_goal_cell && p != 1)) continue;       /* :80-83 */
        if (xa == xb && ya == yb) continue;                                     /* :86-88 */
        if ((a_goal_cell && p == 0) || (b_goal_cell && p == 1)) {               /* :91-103 */
            int ga = (p == 0 && in_goal_rows(o, xa) && ya == o->W - 1) ||
                     (p == 1 && in_goal_rows(o, xb) && yb == o->W - 1);
            o->kind[f] = 2; o->lut[f] = 0; o->goal_value[f] = ga ? 1 : -1;
            continue;
        }
        o->kind[f] = 1; o->lut[f] = (uint16_t)o->nS; o->nS++;                   /* :105-106 */
    }
    /* _generate_isd (:146-165) */
    {
        int col_a = 2, col_b = o->W - 3;
        if (o->n_goal_rows % 2 == 0) {
            int mid = o->n_goal_rows / 2;
            int opt[2] = {o->goal_rows[mid - 1], o->goal_rows[mid]};
            o->n_isd = 0;
            for (int i = 0; i < 2; ++i) {
                int row_a = opt[i];
                int row_b = (row_a == opt[0]) ? opt[1] : opt[0];
                for (int poss = 0; poss < 2; ++poss) {
                    o->isd[o->n_isd] = flat(o, row_a, col_a, row_b, col_b, poss);
                    o->isd_p[o->n_isd] = 0.25; o->n_isd++;
                }
            }
        } else {
            int mr = o->goal_rows[o->n_goal_rows / 2];
            o->n_isd = 2;
            for (int poss = 0; poss < 2; ++poss) {
                o->isd[poss] = flat(o, mr, col_a, mr, col_b, poss);
                o->isd_p[poss] = 0.5;
            }
        }
    }
    /* _initialize_transition_dynamics (:167-293), multi-agent mode */
    o->P = (trans_list*)calloc((size_t)o->n_tuples * 25, sizeof(trans_list));
    for (int32_t st = 0; st < o->n_tuples; ++st) {
        if (o->kind[st] == 0) continue;                                         /* :179-180 */
        for (int aa = 0; aa < 5; ++aa) for (int ab = 0; ab < 5; ++ab)
            build_list(o, st, aa, ab, &o->P[(size_t)st * 25 + aa * 5 + ab]);
    }
    return o;
}

void soc_oracle_destroy(soc_oracle* o) {
    if (!o) return;
    free(o->kind); free(o->lut); free(o->goal_value); free(o->P); free(o);
}

void soc_oracle_set_max_steps(soc_oracle* o, int m) { o->max_steps = m; }
int soc_oracle_ns(const soc_oracle* o) { return o->nS; }
int soc_oracle_n_tuples(const soc_oracle* o) { return o->n_tuples; }
int soc_oracle_internal_width(const soc_oracle* o) { return o->W; }
int soc_oracle_n_isd(const soc_oracle* o) { return o->n_isd; }

void soc_oracle_tables(const soc_oracle* o, uint16_t* lut, uint8_t* kind, int8_t* goal_value,
                       int8_t* isd_states, double* isd_p) {
    if (lut) memcpy(lut, o->lut, sizeof(uint16_t) * o->n_tuples);
    if (kind) memcpy(kind, o->kind, o->n_tuples);
    if (goal_value) memcpy(goal_value, o->goal_value, o->n_tuples);
    for (int i = 0; i < o->n_isd; ++i) {
        int xa, ya, xb, yb, p; unflat(o, o->isd[i], &xa, &ya, &xb, &yb, &p);
        if (isd_states) { int8_t* s = isd_states + 5 * i; s[0] = xa; s[1] = ya; s[2] = xb; s[3] = yb; s[4] = p; }
        if (isd_p) isd_p[i] = o->isd_p[i];
    }
}

/* one transition list, for comparison with the golden table. Returns the entry count, -1 if the
 * tuple is unreachable (the reference has no such key). */
int soc_oracle_transitions(const soc_oracle* o, const int8_t st[5], int aa, int ab,
                           double* p, int8_t* ns5, int8_t* r, uint8_t* d) {
    int32_t f = flat(o, st[0], st[1], st[2], st[3], st[4]);
    if (o->kind[f] == 0) return -1;
    const trans_list* tl = &o->P[(size_t)f * 25 + aa * 5 + ab];
    for (int i = 0; i < tl->n; ++i) {
        int xa, ya, xb, yb, pp; unflat(o, tl->ns[i], &xa, &ya, &xb, &yb, &pp);
        p[i] = tl->p[i]; r[i] = tl->r[i]; d[i] = tl->d[i];
        ns5[5 * i + 0] = xa; ns5[5 * i + 1] = ya; ns5[5 * i + 2] = xb; ns5[5 * i + 3] = yb; ns5[5 * i + 4] = pp;
    }
    return tl->n;
}

/* The whole table in the canonical order of tests/golden/make_golden.py (table_digest): ascending flat tuple index,
 * aa, ab, list position k — the reference's own P_readable iteration order (:167-293).  rows: int8[n][15] =
 * xa,ya,xb,yb,p, aa,ab, k, nxa,nya,nxb,nyb,np, reward, done.  Returns the row count; with rows == NULL only counts. */
int64_t soc_oracle_dump_table(const soc_oracle* o, int8_t* rows, double* prob, int64_t capacity) {
    int64_t n = 0;
    for (int32_t f = 0; f < o->n_tuples; ++f) {
        if (o->kind[f] == 0) continue;
        int xa, ya, xb, yb, pp; unflat(o, f, &xa, &ya, &xb, &yb, &pp);
        for (int ja = 0; ja < 25; ++ja) {
            const trans_list* tl = &o->P[(size_t)f * 25 + ja];
            for (int k = 0; k < tl->n; ++k, ++n) {
                if (!rows) continue;
                if (n >= capacity) return -1;
                int8_t* r = rows + 15 * n;
                int nxa, nya, nxb, nyb, np_; unflat(o, tl->ns[k], &nxa, &nya, &nxb, &nyb, &np_);
                r[0] = xa; r[1] = ya; r[2] = xb; r[3] = yb; r[4] = pp; r[5] = ja / 5; r[6] = ja % 5; r[7] = k;
                r[8] = nxa; r[9] = nya; r[10] = nxb; r[11] = nyb; r[12] = np_; r[13] = tl->r[k]; r[14] = (int8_t)tl->d[k];
                prob[n] = tl->p[k];
            }
        }
    }
    return n;
}

/* categorical_sample (gym 0.26.2): argmax(cumsum(p) > u) */
static int categorical_sample(const double* p, int n, double u) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) { acc += p[i]; if (acc > u) return i; }
    return 0;
}

/* ---- Philox4x32-10 (Salmon et al., SC'11; Random123 constants) ---------------------------- */
void soc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* The product's per-lane randomness (include/soccer_hip.h, ABI 3): four consecutive global lanes share Philox blocks,
 * q = g >> 2, counter = (q_lo, q_hi, c_lo, c_hi | purpose << 31); lane g owns word g & 3.  Every uniform is
 * (m + 1/2) * 2^-b.
 *   slip_prob > 0 : block c = tick;       step m = w >> 2 (b = 30), reset m = w & 3 (b = 2)
 *   slip_prob == 0: block c = tick >> 3;  nib = (w >> (4 * ((tick & 7) ^ 1))) & 15;  step m = nib >> 2, reset m = nib & 3 (b = 2)
 * Sampled actions (purpose 1) always use the tick's own block. */
static uint32_t lane_word(uint64_t seed, uint64_t g, uint64_t c, uint32_t purpose) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint64_t q = g >> 2;
    uint32_t ctr[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)c, (uint32_t)(c >> 32) | (purpose << 31)};
    uint32_t w[4]; soc_philox4x32_10(ctr, key, w);
    return w[g & 3];
}
typedef struct { double u_step, u_reset; } lane_draw;
static lane_draw draw_of(const soc_oracle* o, uint64_t seed, uint64_t g, uint64_t tick) {
    lane_draw d;
    if (o->slip != 0.0) {
        uint32_t w = lane_word(seed, g, tick, 0);
        d.u_step = ((double)(w >> 2) + 0.5) * (1.0 / 1073741824.0);
        d.u_reset = ((double)(w & 3u) + 0.5) * 0.25;
    } else {
        uint32_t w = lane_word(seed, g, tick >> 3, 0);
        uint32_t nib = (w >> (4u * (((uint32_t)tick & 7u) ^ 1u))) & 15u;
        d.u_step = ((double)(nib >> 2) + 0.5) * 0.25;
        d.u_reset = ((double)(nib & 3u) + 0.5) * 0.25;
    }
    return d;
}

/* ---- batched semantics of the product (include/soccer_hip.h), lane by lane ------------------ */
typedef struct {
    int8_t* row_a; int8_t* col_a; int8_t* row_b; int8_t* col_b;
    uint8_t* poss;   /* bit0 possession, bit1 needs_reset */
    uint8_t* t;
} soc_state;

static uint16_t obs_of(const soc_oracle* o, int32_t f) { return o->lut[f]; }

static void do_reset_lane(const soc_oracle* o, soc_state* s, int64_t i, double u) {
    int k = categorical_sample(o->isd_p, o->n_isd, u);               /* :414 */
    int xa, ya, xb, yb, p; unflat(o, o->isd[k], &xa, &ya, &xb, &yb, &p);
    s->row_a[i] = xa; s->col_a[i] = ya; s->row_b[i] = xb; s->col_b[i] = yb;
    s->poss[i] = (uint8_t)p;                                        /* needs_reset cleared (:422) */
    s->t[i] = 0;                                                    /* :423 */
}

/* reset (:410-424).  u_reset NULL -> the lane's Philox word at this tick. */
int soc_oracle_batched_reset(const soc_oracle* o, int64_t n, soc_state* s, const uint8_t* mask,
                             const double* u_reset, uint64_t seed, uint64_t lane_offset,
                             uint64_t tick, uint16_t* obs) {
    for (int64_t i = 0; i < n; ++i) {
        if (!mask || mask[i]) {
            double u = u_reset ? u_reset[i] : draw_of(o, seed, lane_offset + (uint64_t)i, tick).u_reset;
            do_reset_lane(o, s, i, u);
        }
        if (obs) {
            int32_t f = flat(o, s->row_a[i], s->col_a[i], s->row_b[i], s->col_b[i], s->poss[i] & 1);
            obs[i] = obs_of(o, f);
        }
    }
    return 0;
}

/* step (:375-408) + the vector-env auto-reset.  prob (nullable) receives the UNROUNDED float64
 * probability of the sampled transition (the reference rounds it for info["p"], :405).
 * hist (nullable) [3] += finished episodes by A's return -1/0/+1.  Returns the number of lanes
 * that needed reset (the reference's assert, :376); those lanes are left untouched. */
int64_t soc_oracle_batched_step(const soc_oracle* o, int64_t n, soc_state* s,
                                const int8_t* act_a, const int8_t* act_b,
                                const double* u_step, const double* u_reset,
                                uint64_t seed, uint64_t lane_offset, uint64_t tick, int autoreset,
                                uint16_t* obs, int8_t* reward, uint8_t* terminated,
                                uint8_t* truncated, double* prob, uint8_t* prob_code,
                                uint16_t* final_obs, uint64_t* hist) {
    int64_t misuse = 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t f = flat(o, s->row_a[i], s->col_a[i], s->row_b[i], s->col_b[i], s->poss[i] & 1);
        if (s->poss[i] & 2) {                                        /* :376 */
            ++misuse;
            if (obs) obs[i] = obs_of(o, f);
            if (final_obs) final_obs[i] = obs_of(o, f);
            if (reward) reward[i] = 0;
            if (terminated) terminated[i] = o->kind[f] == 2;
            if (truncated) truncated[i] = s->t[i] >= o->max_steps;
            if (prob) prob[i] = 0.0;
            if (prob_code) prob_code[i] = 0;
            continue;
        }
        lane_draw dr = {0.0, 0.0};
        if (!u_step || (autoreset && !u_reset)) dr = draw_of(o, seed, lane_offset + (uint64_t)i, tick);
        double u = u_step ? u_step[i] : dr.u_step;
        const trans_list* tl = &o->P[(size_t)f * 25 + act_a[i] * 5 + act_b[i]];   /* :394 */
        int k = categorical_sample(tl->p, tl->n, u);                              /* :395 */
        int32_t ns = tl->ns[k];                                                   /* :396 */
        int xa, ya, xb, yb, p; unflat(o, ns, &xa, &ya, &xb, &yb, &p);
        s->row_a[i] = xa; s->col_a[i] = ya; s->row_b[i] = xb; s->col_b[i] = yb;
        int tt = s->t[i] + 1;                                                     /* :399 */
        int done = tl->d[k];
        int trunc = tt >= o->max_steps;                                           /* :404 */
        int need = done || trunc;                                                 /* :406 */
        s->t[i] = (uint8_t)tt;
        s->poss[i] = (uint8_t)(p | (need ? 2 : 0));
        uint16_t ob = obs_of(o, ns);                                              /* :397 */
        if (final_obs) final_obs[i] = ob;
        if (reward) reward[i] = tl->r[k];
        if (terminated) terminated[i] = (uint8_t)done;
        if (truncated) truncated[i] = (uint8_t)trunc;
        if (prob) prob[i] = tl->p[k];
        if (prob_code) {
            int c = tl->combo[k];
            int cls = c == 0 ? 0 : (c <= 2 ? 1 : (c <= 4 ? 2 : 3));
            prob_code[i] = (uint8_t)(cls * 3 + tl->nsp_code[k]);
        }
        if (need && hist) hist[tl->r[k] + 1] += 1;
        if (need && autoreset) {
            double ur = u_reset ? u_reset[i] : dr.u_reset;
            do_reset_lane(o, s, i, ur);
            int32_t f2 = flat(o, s->row_a[i], s->col_a[i], s->row_b[i], s->col_b[i], s->poss[i] & 1);
            ob = obs_of(o, f2);
        }
        if (obs) obs[i] = ob;
    }
    return misuse;
}

/* uniform-random joint action of the in-kernel sampler: the lane's word of the purpose-1 block;
 * 15 bits per player: a = ((w & 0x7fff) * 5) >> 15, b = (((w >> 16) & 0x7fff) * 5) >> 15
 * (include/soccer_hip.h, soccer_rollout_args). */
void soc_oracle_sample_actions(int64_t n, uint64_t seed, uint64_t lane_offset, uint64_t tick,
                               int8_t* act_a, int8_t* act_b) {
    for (int64_t i = 0; i < n; ++i) {
        uint32_t w = lane_word(seed, lane_offset + (uint64_t)i, tick, 1);
        act_a[i] = (int8_t)(((w & 0x7fffu) * 5u) >> 15);
        act_b[i] = (int8_t)((((w >> 16) & 0x7fffu) * 5u) >> 15);
    }
}

/* mixed-policy action sampling of batched_rollout (mix_a / mix_b): 15-bit draw per player from the
 * lane's purpose-1 word; action = number of cumulative thresholds <= draw; NULL table = uniform. */
void soc_oracle_sample_actions_mixed(int64_t n, uint64_t seed, uint64_t lane_offset, uint64_t tick,
                                     const uint16_t* obs_now, const uint16_t* mix_a, const uint16_t* mix_b,
                                     int8_t* act_a, int8_t* act_b) {
    for (int64_t i = 0; i < n; ++i) {
        uint32_t w = lane_word(seed, lane_offset + (uint64_t)i, tick, 1);
        uint32_t ha = w & 0x7fffu, hb = (w >> 16) & 0x7fffu;
        int a = (int)((ha * 5u) >> 15), b = (int)((hb * 5u) >> 15);
        if (mix_a) { const uint16_t* t = mix_a + 4 * (size_t)obs_now[i]; a = (ha >= t[0]) + (ha >= t[1]) + (ha >= t[2]) + (ha >= t[3]); }
        if (mix_b) { const uint16_t* t = mix_b + 4 * (size_t)obs_now[i]; b = (hb >= t[0]) + (hb >= t[1]) + (hb >= t[2]) + (hb >= t[3]); }
        act_a[i] = (int8_t)a; act_b[i] = (int8_t)b;
    }
}
