#!/bin/bash
# Runs ON THE GPU BOX: the driver's bench shape (--steps 20 --warmup 5) under different HIP wait settings.
cd "$(dirname "$0")/../.."
pick() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('   value %.3g  ms_per_step %.5f  launch_us %.2f' % (j['value'], j['ms_per_step'], j['roofline']['launch_us']))
"; }
for i in 1 2 3; do echo "default"; python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --rollout 0 --no-vector-env 2>/dev/null | pick; done
for i in 1 2 3; do echo "ROC_ACTIVE_WAIT_TIMEOUT=2000"; ROC_ACTIVE_WAIT_TIMEOUT=2000 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --rollout 0 --no-vector-env 2>/dev/null | pick; done
for i in 1 2 3; do echo "HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --rollout 0 --no-vector-env 2>/dev/null | pick; done
for i in 1 2 3; do echo "both"; HIP_FORCE_DEV_KERNARG=1 ROC_ACTIVE_WAIT_TIMEOUT=2000 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --rollout 0 --no-vector-env 2>/dev/null | pick; done
