#!/usr/bin/env python3
"""Search the handle's Philox stream for draws that land ON or NEXT TO a slip threshold.

A lane's step uniform at slip_prob > 0 is u = (m + 1/2) * 2^-30 with m = word >> 2 (include/soccer_hip.h, ABI 3).
The slip kernels decide "running sum <= u" on integers (m >= c, c = ceil(sum * 2^30 - 1/2)); this script finds
(global lane, tick) pairs whose m is such a c or the integer below it, for tests/test_gpu_parity.py to replay against
the oracle's float64 cumsum.  Pure numpy (vectorised Philox4x32-10, the spec's own definition); writes
tests/golden/threshold_draws.json.

    tools/find_threshold_draws.py                 rebuild the file for the default slips
    tools/find_threshold_draws.py 0.15 0.4 ...    search these slips and MERGE their hits into the existing file

(The entries of 0.1 / 0.2 / 0.3 / 0.5 date from ABI 2, u = m * 2^-30 and c' = ceil(sum * 2^30): m in {c' - 1, c'}.  As
c is c' or c' - 1 they still sit on or next to a threshold — within {c - 1, c, c + 1}.  Entries marked "danger" are draws
exactly ON a mathematically dyadic threshold such as 0.1's 27/32: the ones that needed a float64 walk in the kernels then.)
"""
import json
import os
import sys

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox_blocks(q, tick, seed):
    """words[4, len(q)] of block (q, tick) under key seed — purpose 0."""
    c0 = (q & 0xFFFFFFFF).astype(np.uint64); c1 = (q >> 32).astype(np.uint64)
    c2 = np.full_like(c0, tick & 0xFFFFFFFF); c3 = np.full_like(c0, (tick >> 32) & 0x7FFFFFFF)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c0; p1 = M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c1 = p1 & mask; c3 = p0 & mask; c0 = n0; c2 = n2
        k0 = (k0 + W0) & 0xFFFFFFFF; k1 = (k1 + W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3]).astype(np.uint32)


def thresholds(slip):
    """The 9 combination ends and the 9 x 4 within-combination thresholds, float64 as the kernels form them."""
    s = np.float64(slip); om = np.float64(1) - s
    w = [om * om, (om * s) * 0.5, (s * om) * 0.5, (s * s) * 0.25]
    cls = [0, 1, 1, 2, 2, 3, 3, 3, 3]
    out, acc = [], np.float64(0)
    for c in range(9):
        wc = w[cls[c]]
        if wc == 0:
            continue
        S = acc; acc = acc + wc
        out.append(("end", c, float(acc)))
        t = S + wc * 0.5; out.append(("two", c, float(t)))
        t = S + wc * 0.25; out.append(("four1", c, float(t)))
        t = t + wc * 0.25; out.append(("four2", c, float(t)))
        t = t + wc * 0.25; out.append(("four3", c, float(t)))
    return out


def main():
    seed, n_lanes, ticks = 20241004, 1 << 22, 96
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "threshold_draws.json")
    slips = [float(x) for x in sys.argv[1:]]
    res = {"seed": seed, "hits": []}
    if slips:
        res = json.load(open(out)); assert res["seed"] == seed
        res["hits"] = [h for h in res["hits"] if h["slip"] not in slips]
    q = np.arange(n_lanes // 4, dtype=np.uint64)
    for slip in (slips or (0.2, 0.3, 0.5, 0.1, 0.15, 0.4, 2.0 / 3.0)):
        th = thresholds(slip)
        targets = {}
        for name, c, t in th:
            cb = int(np.ceil(t * 2.0 ** 30 - 0.5))
            for m in (cb - 1, cb):
                if 0 <= m < (1 << 30):
                    targets.setdefault(m, []).append((name, c))
        tarr = np.array(sorted(targets), np.uint32)
        found = 0
        for tick in range(ticks):
            m = philox_blocks(q, tick, seed) >> 2
            for j, qi in zip(*np.nonzero(np.isin(m, tarr))):
                res["hits"].append({"slip": slip, "lane": int(qi) * 4 + int(j), "tick": tick, "m": int(m[j, qi]),
                                    "thresholds": [list(x) for x in targets[int(m[j, qi])]]})
                found += 1
        print("slip %.4f: %d draws on/next to a threshold in %d lanes x %d ticks" % (slip, found, n_lanes, ticks), file=sys.stderr)
        # thresholds that are dyadic rationals mathematically (27/32 at slip 0.1 ...) although the float64 sums that form
        # them are not exact: under ABI 2 a draw exactly ON such an integer needed the float64 walk in the kernels
        # ("danger"); such draws are searched for specifically (p = 2^-30 each)
        danger = {}
        for name, c, t in th:
            x = t * 2.0 ** 30; r = round(x)
            if abs(x - r) < 2.0 ** -10 and r < (1 << 30) and slip not in (0.5,):
                danger.setdefault(int(r), []).append((name, c))
        if danger:
            darr = np.array(sorted(danger), np.uint32); got = 0
            for tick in range(ticks, ticks + 1400):
                m = philox_blocks(q, tick, seed) >> 2
                for j, qi in zip(*np.nonzero(np.isin(m, darr))):
                    res["hits"].append({"slip": slip, "lane": int(qi) * 4 + int(j), "tick": tick, "m": int(m[j, qi]),
                                        "thresholds": [list(x) for x in danger[int(m[j, qi])]], "danger": True})
                    got += 1
                if got >= 3:
                    break
            print("slip %.4f: %d draws exactly on the dyadic integer(s) %s" % (slip, got, sorted(danger)), file=sys.stderr)
    with open(out, "w") as f:
        json.dump(res, f, indent=0)
    print("wrote", os.path.normpath(out), len(res["hits"]), "hits")


if __name__ == "__main__":
    main()
