#!/bin/bash
# HISTORICAL (results: profiles/r02_sweep.md): the SOCCER_SWAR_* / SOCCER_FORCE_GENERAL knobs this script sets existed only
# in commit 4aef6d3..f61d5bd's library and were removed once the sweep had picked the shipped shape.
# A/B sweep of the byte-parallel step kernel's launch shape (run on the GPU box): lanes per thread (G x 4), store policy
# (0 nt, 1 plain, 2 write-through), block size, steady-state vs general instantiation.  Prints HIP-event us per launch.
cd "$(dirname "$0")/../.."
run() {
  local tag="$1"; shift
  local out
  out=$(env "$@" python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline --rollout 0 --no-vector-env 2>/dev/null | tail -1)
  python3 - "$tag" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print("%-44s launch_us %.3f  wall_us/step %.3f  frac %.3f" % (sys.argv[1], d["roofline"]["launch_us"], d["ms_per_step"] * 1e3, d["roofline"]["frac"]))
PY
}
run "G1 nt b256 steady (shipped)" X=1
run "G1 nt b256 steady (repeat)" X=1
run "G1 nt b256 general" SOCCER_FORCE_GENERAL=1
run "G1 plain b256" SOCCER_SWAR_STORE=1
run "G1 wt b256" SOCCER_SWAR_STORE=2
run "G1 nt b512" SOCCER_SWAR_BLOCK=512
run "G2 nt b256" SOCCER_SWAR_G=2
run "G2 plain b256" SOCCER_SWAR_G=2 SOCCER_SWAR_STORE=1
run "G2 wt b256" SOCCER_SWAR_G=2 SOCCER_SWAR_STORE=2
run "G2 nt b512" SOCCER_SWAR_G=2 SOCCER_SWAR_BLOCK=512
run "G4 nt b256" SOCCER_SWAR_G=4
run "G4 plain b256" SOCCER_SWAR_G=4 SOCCER_SWAR_STORE=1
run "G4 wt b256" SOCCER_SWAR_G=4 SOCCER_SWAR_STORE=2
run "G4 nt b512" SOCCER_SWAR_G=4 SOCCER_SWAR_BLOCK=512
run "G4 wt b512" SOCCER_SWAR_G=4 SOCCER_SWAR_STORE=2 SOCCER_SWAR_BLOCK=512
