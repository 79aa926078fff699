#!/bin/bash
# Runs ON THE GPU BOX: like lib_ab.sh, but each library variant is timed in the driver's shape (--steps 20 --warmup 5) and at
# --steps 1000.  Usage: tools/labs/lib_ab20.sh base variant ...
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do
  for v in "$@"; do
    cp build/lib_$v.so gym_soccer_littman94_amd/libsoccer_hip.so
    a=$(python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --rollout 0 --no-vector-env 2>/dev/null | tail -1)
    b=$(python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline --rollout 0 --no-vector-env 2>/dev/null | tail -1)
    python3 - "$v" "$a" "$b" <<'PY'
import json, sys
a, b = json.loads(sys.argv[2]), json.loads(sys.argv[3])
print("%-14s K=20: wall %.1f us  device %.1f us  value %.4g   K=1000: launch_us %.3f" % (sys.argv[1], a["ms_per_step"] * 20e3, a["roofline"]["device_region_us"], a["value"], b["roofline"]["launch_us"]))
PY
  done
done
cp build/lib_base.so gym_soccer_littman94_amd/libsoccer_hip.so
