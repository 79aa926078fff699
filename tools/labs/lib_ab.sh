#!/bin/bash
# Runs ON THE GPU BOX: A/B of library variants built into build/lib_<name>.so (each is copied over the package's
# libsoccer_hip.so of this scratch copy, then bench.py's step-only command runs).  Usage: tools/labs/lib_ab.sh base prio1 ... [-- bench flags]
cd "$(dirname "$0")/../.."
NAMES=(); EXTRA=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; EXTRA=("$@"); break; fi; NAMES+=("$1"); shift; done
for rep in 1 2 3; do
  for v in "${NAMES[@]}"; do
    cp build/lib_$v.so gym_soccer_littman94_amd/libsoccer_hip.so
    if [ ${#EXTRA[@]} -eq 0 ]; then EXTRA=(--rollout 0 --no-vector-env); fi
    out=$(python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline "${EXTRA[@]}" 2>/dev/null | tail -1)
    python3 - "$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
x = ""
if "fused_rollout" in d: x += "  rollout %.3g" % d["fused_rollout"]["env_steps_per_s"]
if "selfplay_rollout_config5" in d: x += "  selfplay %.3g" % d["selfplay_rollout_config5"]["env_steps_per_s"]
if "vector_env_device" in d: x += "  venv_us %.2f" % d["vector_env_device"]["us_per_step"]
print("%-12s launch_us %.3f  frac %.3f%s" % (sys.argv[1], d["roofline"]["launch_us"], d["roofline"]["frac"], x))
PY
  done
done
cp build/lib_base.so gym_soccer_littman94_amd/libsoccer_hip.so
