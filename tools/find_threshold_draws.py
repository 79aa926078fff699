#!/usr/bin/env python3
"""Search the handle's Philox stream for draws that land ON or NEXT TO a slip threshold.

A lane's step uniform is u = m * 2^-30 with m = word >> 2 (include/soccer_hip.h).  The slip kernels decide
"u >= threshold" on integers (m >= ceil(threshold * 2^30)); this script finds (global lane, tick) pairs whose m
equals such a scaled threshold or the integer below it, for tests/test_gpu_parity.py to replay against the
oracle's float64 cumsum.  Pure numpy (vectorised Philox4x32-10, the spec's own definition); writes
tests/golden/threshold_draws.json.
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
    res = {"seed": seed, "hits": []}
    for slip in (0.2, 0.3, 0.5, 0.1):   # 0.5: dyadic, every float64 sum exact, thresholds ARE integers after scaling;
                                        # 0.1: one scaled threshold within 2^-10 of an integer (searched for specifically below)
        th = thresholds(slip)
        targets = {}
        for name, c, t in th:
            x = t * 2.0 ** 30
            cb = int(np.ceil(x))
            for m in (cb - 1, cb):
                if 0 <= m < (1 << 30):
                    targets.setdefault(m, []).append((name, c))
        tarr = np.array(sorted(targets), np.uint32)
        q = np.arange(n_lanes // 4, dtype=np.uint64)
        found = 0
        for tick in range(ticks):
            w = philox_blocks(q, tick, seed)
            m = w >> 2
            hit = np.isin(m, tarr)
            for j, qi in zip(*np.nonzero(hit)):
                g = int(qi) * 4 + int(j)
                res["hits"].append({"slip": slip, "lane": g, "tick": tick, "m": int(m[j, qi]),
                                    "thresholds": [list(x) for x in targets[int(m[j, qi])]]})
                found += 1
        print("slip %.2f: %d draws on/next to a threshold in %d lanes x %d ticks" % (slip, found, n_lanes, ticks), file=sys.stderr)
        # thresholds whose scaled value is (almost) an integer although the sums are not exact: the kernels send the
        # lane that draws exactly that integer down the float64 path — find such draws (p = 2^-30 each)
        danger = {}
        for name, c, t in th:
            x = t * 2.0 ** 30; r = round(x)
            if abs(x - r) < 2.0 ** -10 and r < (1 << 30) and slip not in (0.5,):
                danger.setdefault(int(r), []).append((name, c))
        if danger:
            darr = np.array(sorted(danger), np.uint32); got = 0
            for tick in range(ticks, ticks + 1400):
                w = philox_blocks(q, tick, seed); m = w >> 2
                hit = np.isin(m, darr)
                for j, qi in zip(*np.nonzero(hit)):
                    res["hits"].append({"slip": slip, "lane": int(qi) * 4 + int(j), "tick": tick, "m": int(m[j, qi]),
                                        "thresholds": [list(x) for x in danger[int(m[j, qi])]], "danger": True})
                    got += 1
                if got >= 3:
                    break
            print("slip %.2f: %d draws exactly on the dangerous integer(s) %s" % (slip, got, sorted(danger)), file=sys.stderr)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "threshold_draws.json")
    with open(out, "w") as f:
        json.dump(res, f, indent=0)
    print("wrote", os.path.normpath(out), len(res["hits"]), "hits")


if __name__ == "__main__":
    main()
