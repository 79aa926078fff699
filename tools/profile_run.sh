#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the rocprofv3 passes behind profiles/<tag>_*.  Usage: tools/profile_run.sh r02_a
#   kt       kernel trace + stats of the bench's timed region (step kernel only; the bench line of the same run is kept)
#   kt_full  the same for the default command (rollouts, self-play, VectorSoccerEnv kernels)
#   fetch / write   FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md, rocprofv3 PMC slots)
#   sq1 / sq2       SQ instruction / wait counters
#   kt_slip / sq_slip   kernel trace and instruction counters of the same commands at slip 0.2
#   kt_k20   kernel trace of the driver's command (--steps 20 --warmup 5)
#   kt_venv  kernel trace of the vector-env leg (the torch child of bench.py, here run as the profiled program itself)
#   kt_other kernel trace of tools/profile_others.py: batched_reset, the per-lane fallback kernels, soccer_trajectory_returns
# Counters are collected with --kernel-trace only (never with the runtime / hip trace domains).
set -e
TAG=${1:-r02}
ROOTDIR="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOTDIR/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOTDIR/bench.py"
# kt: the step kernel alone (the bench's timed region; nothing else in the process for the profiler to interleave with)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o runc -- python3 "$B" --steps 1000 --warmup 50 --no-cpu-baseline --rollout 0 --no-vector-env > "$OUT/bench.json" 2> "$OUT/kt.err"
echo "kt done"
# kt_full: the default command (rollouts, self-play, VectorSoccerEnv): the other kernels' durations
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_full" -o runc -- python3 "$B" --steps 1000 --warmup 50 --no-cpu-baseline --no-vector-env > "$OUT/bench_full.json" 2> "$OUT/kt_full.err"
echo "kt_full done"
PMC_ARGS="--steps 100 --warmup 10 --no-cpu-baseline --rollout 20 --no-vector-env"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o runc -- python3 "$B" $PMC_ARGS > "$OUT/fetch.json" 2> "$OUT/fetch.err"
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o runc -- python3 "$B" $PMC_ARGS > "$OUT/write.json" 2> "$OUT/write.err"
echo "write done"
SQ_ARGS="--steps 60 --warmup 10 --no-cpu-baseline --rollout 20 --no-vector-env"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq1" -o runc -- python3 "$B" $SQ_ARGS > "$OUT/sq1.json" 2> "$OUT/sq1.err"
echo "sq1 done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d "$OUT/sq2" -o runc -- python3 "$B" $SQ_ARGS > "$OUT/sq2.json" 2> "$OUT/sq2.err"
echo "sq2 done"
# slip 0.2: the step kernel's table-selection instantiation (kernel trace, then its instruction counters) + its un-profiled line
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_slip" -o runc -- python3 "$B" --slip 0.2 --steps 1000 --warmup 50 --no-cpu-baseline --no-vector-env > "$OUT/bench_slip.json" 2> "$OUT/kt_slip.err"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d "$OUT/sq_slip" -o runc -- python3 "$B" --slip 0.2 $SQ_ARGS > "$OUT/sq_slip.json" 2> "$OUT/sq_slip.err"
python3 "$B" --slip 0.2 --steps 1000 --warmup 50 --no-cpu-baseline --no-vector-env > "$OUT/bench_slip_unprofiled.json" 2>/dev/null
echo "slip done"
# the driver's own command (--steps 20 --warmup 5: plain action loads, a 146 MB working set) under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_k20" -o runc -- python3 "$B" --steps 20 --warmup 5 --no-cpu-baseline --rollout 0 --no-vector-env > "$OUT/bench_k20_profiled.json" 2> "$OUT/kt_k20.err"
echo "kt_k20 done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_venv" -o runc -- python3 "$B" --leg vector-env --steps 1000 > "$OUT/venv_profiled.json" 2> "$OUT/kt_venv.err"
echo "kt_venv done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_other" -o runc -- python3 "$ROOTDIR/tools/profile_others.py" > "$OUT/others.json" 2> "$OUT/kt_other.err"
echo "kt_other done"
# the two-rank rehearsal of `python bench.py --gpus 2` on this one GPU (host-file communicator; ranks take the region in turn)
python3 "$B" --gpus 2 --comm host --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_2rank_rehearsal.json" 2>/dev/null
# un-profiled reference run of the same command as kt (a profiled run clocks lower)
python3 "$B" --steps 1000 --warmup 50 --no-cpu-baseline > "$OUT/bench_unprofiled.json" 2>/dev/null
python3 "$B" --steps 20 --warmup 5 > "$OUT/bench_driver_shape.json" 2>/dev/null
# the same driver-shape command on the runtime PyTorch bundles (what round 3's ranks ran on): same-box A/B of the host's share
python3 "$B" --steps 20 --warmup 5 --hip-runtime torch --no-vector-env --no-cpu-baseline --rollout 0 > "$OUT/bench_driver_shape_torch_runtime.json" 2>/dev/null
python3 "$B" --steps 20 --warmup 5 --no-vector-env --no-cpu-baseline --rollout 0 > "$OUT/bench_driver_shape_again.json" 2>/dev/null
echo "all done"
