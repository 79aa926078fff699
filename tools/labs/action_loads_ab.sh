#!/bin/bash
# Runs ON THE GPU BOX: batched_step's action streams by plain loads (the library default) against non-temporal ones
# (SOCCER_F_STREAM_ACTIONS), in the driver's shape (K = 20: the 40 MB action trajectory stays in the Infinity Cache between
# replays) and at K = 1000 (2 GB: it streams in from HBM).
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  for v in nt plain; do
    a=$(python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --rollout 0 --no-vector-env --action-loads $v 2>/dev/null | tail -1)
    b=$(python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline --rollout 0 --no-vector-env --action-loads $v 2>/dev/null | tail -1)
    python3 - "$v" "$a" "$b" <<'PY'
import json, sys
a, b = json.loads(sys.argv[2]), json.loads(sys.argv[3])
print("%-6s K=20: wall %.1f us  device %.1f us  value %.4g  frac %.3f   K=1000: launch_us %.3f  frac %.3f" % (sys.argv[1], a["ms_per_step"] * 20e3, a["roofline"]["device_region_us"], a["value"], a["roofline"]["frac"], b["roofline"]["launch_us"], b["roofline"]["frac"]))
PY
  done
done
