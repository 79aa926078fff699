// soccer_rules.hpp — host-side construction of the small rule tables the kernels stage in LDS.
//
// Restates the constructor-time rules of the reference
// (gym_soccer/envs/soccer_simultaneous_env.py) as table builders:
//   goal rows / cols            :60-61
//   tuple classification + ids  :63-109   -> obs_lut / goal_value / kind
//   initial state distribution  :146-165  -> isd
//   single-player cell move     :364-373  -> next_cell  (the "move/bounds table")
// Nothing here runs per step; the per-step rules (collisions :296-362, slip list :202-256,
// bookkeeping :393-406) are evaluated per lane in soccer_kernels.hpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace soccer {

// action -> (dcol, drow)   (:24-30)
static const int kMoveDc[5] = {0, 0, 0, 1, -1};
static const int kMoveDr[5] = {0, -1, 1, 0, 0};

struct Rules {
    int H = 0, W = 0;            // W = pitch width + 2 goal columns (:48)
    int goal_lo = 0, goal_hi = 0; // goal rows are the contiguous range [goal_lo, goal_hi] (:60)
    int nS = 0;                  // number of observation indices incl. terminal 0
    int n_isd = 0;
    int8_t isd[4][5];            // (row_a, col_a, row_b, col_b, poss)
    uint16_t isd_obs[4];
    std::vector<uint16_t> lut;   // [(((ra*W+ca)*H+rb)*W+cb)*2+p] -> obs index; 0 goal; 0xFFFF unreachable
    std::vector<int8_t> goal_value;
    std::vector<uint8_t> kind;   // 0 unreachable, 1 live, 2 goal
    std::vector<uint32_t> next_cell; // [(has_ball*H*W + row*W+col)*5 + move] -> pos word of the cell reached
    uint32_t isd_words[16];      // 4 x (pos A, pos B, poss | obs<<16, 0) for the kernels

    // position word: low 16 bits the cell id row*W+col, high 16 bits (row<<8 | col)
    uint32_t pos_word(int r, int c) const { return uint32_t(r * W + c) | (uint32_t((r << 8) | c) << 16); }

    bool goal_row(int r) const { return r >= goal_lo && r <= goal_hi; }
    bool goal_col(int c) const { return c == 0 || c == W - 1; }
    int flat(int ra, int ca, int rb, int cb, int p) const {
        return (((ra * W + ca) * H + rb) * W + cb) * 2 + p;
    }

    // returns "" on success, else the message of the reference's failed assert
    std::string build(int width, int height) {
        if (width < 5) return "Width must be at least 5 columns.";     // :45
        if (height < 4) return "Height must be at least 4 rows.";      // :46
        if (height > 120 || width > 120) return "pitch too large (rows/cols must fit int8)";
        H = height; W = width + 2;
        if (H % 2 == 0) { goal_lo = (H - 1) / 2; goal_hi = H / 2; }   // :60
        else { goal_lo = H / 2 - 1; goal_hi = H / 2 + 1; }
        const long n_tuples = 2L * H * W * H * W;
        if (n_tuples > (1L << 24)) return "pitch too large (tuple table)";
        lut.assign(n_tuples, 0xFFFF); goal_value.assign(n_tuples, 0); kind.assign(n_tuples, 0);
        long next_id = 1;                                               // 0 is the terminal state (:64-65)
        for (int ra = 0; ra < H; ++ra) for (int ca = 0; ca < W; ++ca)
        for (int rb = 0; rb < H; ++rb) for (int cb = 0; cb < W; ++cb)
        for (int p = 0; p < 2; ++p) {
            const int f = flat(ra, ca, rb, cb, p);
            const bool a_in_mouth = goal_row(ra) && goal_col(ca);
            const bool b_in_mouth = goal_row(rb) && goal_col(cb);
            if ((goal_col(ca) && !goal_row(ra)) || (goal_col(cb) && !goal_row(rb))) continue; // :74-77
            if ((a_in_mouth && p != 0) || (b_in_mouth && p != 1)) continue;                   // :80-83
            if (ra == rb && ca == cb) continue;                                                // :86-88
            if (a_in_mouth || b_in_mouth) {              // the carrier stands in a goal mouth (:91-103)
                const int carrier_col = p == 0 ? ca : cb;
                kind[f] = 2; lut[f] = 0;
                goal_value[f] = carrier_col == W - 1 ? 1 : -1;          // :94-102
                continue;
            }
            kind[f] = 1; lut[f] = static_cast<uint16_t>(next_id); ++next_id;                  // :105-106
            if (next_id > 0xFFFE) return "pitch too large (observation index must fit uint16)";
        }
        nS = static_cast<int>(next_id);
        // Self-check against the closed form of the numbering: a tuple is live exactly when both players
        // stand on distinct interior cells; with i = row*(W-2) + col-1 and NI = H*(W-2) interior cells,
        // index = 1 + 2*(iA*(NI-1) + iB - (iB > iA)) + p.
        {
            const int NI = H * (W - 2);
            if (nS != NI * (NI - 1) * 2 + 1) return "internal error: state numbering";
            for (int ra = 0; ra < H; ++ra) for (int ca = 1; ca < W - 1; ++ca)
            for (int rb = 0; rb < H; ++rb) for (int cb = 1; cb < W - 1; ++cb) for (int p = 0; p < 2; ++p) {
                if (ra == rb && ca == cb) continue;
                const int ia = ra * (W - 2) + ca - 1, ib = rb * (W - 2) + cb - 1;
                if (lut[flat(ra, ca, rb, cb, p)] != 1 + 2 * (ia * (NI - 1) + ib - (ib > ia ? 1 : 0)) + p)
                    return "internal error: state numbering";
            }
        }
        // initial state distribution (:146-165): A two columns from its goal line, B likewise
        const int col_a = 2, col_b = W - 3;
        const int n_goal_rows = goal_hi - goal_lo + 1;
        if (n_goal_rows % 2 == 0) {
            const int opt[2] = {goal_lo + n_goal_rows / 2 - 1, goal_lo + n_goal_rows / 2};
            n_isd = 0;
            for (int i = 0; i < 2; ++i) for (int poss = 0; poss < 2; ++poss) {
                const int8_t s[5] = {int8_t(opt[i]), int8_t(col_a), int8_t(opt[1 - i]), int8_t(col_b), int8_t(poss)};
                for (int k = 0; k < 5; ++k) isd[n_isd][k] = s[k];
                ++n_isd;
            }
        } else {
            const int mr = goal_lo + n_goal_rows / 2;
            n_isd = 2;
            for (int poss = 0; poss < 2; ++poss) {
                const int8_t s[5] = {int8_t(mr), int8_t(col_a), int8_t(mr), int8_t(col_b), int8_t(poss)};
                for (int k = 0; k < 5; ++k) isd[poss][k] = s[k];
            }
        }
        for (int i = 0; i < n_isd; ++i)
            isd_obs[i] = lut[flat(isd[i][0], isd[i][1], isd[i][2], isd[i][3], isd[i][4])];
        for (int i = 0; i < 4; ++i) {
            const int k = i < n_isd ? i : 0;
            isd_words[4 * i + 0] = pos_word(isd[k][0], isd[k][1]);
            isd_words[4 * i + 1] = pos_word(isd[k][2], isd[k][3]);
            isd_words[4 * i + 2] = uint32_t(isd[k][4]) | (uint32_t(isd_obs[k]) << 16);
            isd_words[4 * i + 3] = 0;
        }
        // move/bounds table (:364-373)
        next_cell.assign(2 * H * W * 5, 0);
        for (int ball = 0; ball < 2; ++ball) for (int r = 0; r < H; ++r) for (int c = 0; c < W; ++c)
        for (int m = 0; m < 5; ++m) {
            int nr = r + kMoveDr[m];
            nr = nr < 0 ? 0 : (nr > H - 1 ? H - 1 : nr);               // :365
            int nc = c + kMoveDc[m];                                    // :366
            const bool edge = (nc == 0 || nc == W - 1);                // :369
            const bool scores = edge && goal_row(nr) && ball;           // :370
            if (edge && !scores) nc = c;                                // :371-372
            if (nc < 0 || nc > W - 1) nc = c;   // only from a goal-mouth cell; those tuples are absorbing
            next_cell[(ball * H * W + r * W + c) * 5 + m] = pos_word(nr, nc);
        }
        return "";
    }
};

}  // namespace soccer
