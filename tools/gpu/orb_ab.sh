#!/bin/bash
# ORB-mode A/B of the two resize kernels in one session: parity tests, then the ORB bench (overlap and stream order) with
# the row-streaming kernel and with the LDS-staged one (SVO_ORB_RESIZE_STAGED=1).
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity_orb.py -m gpu -q -x > gpurun_out/orb_tests.log 2>&1 || { tail -30 gpurun_out/orb_tests.log; exit 1; }
tail -2 gpurun_out/orb_tests.log
CACHE=/tmp/s0_frames_c2.pt
BARGS="--mode orb --cpu-pairs 0 --no-secondary --no-legs --self-check-pairs 16 --chunks 2 --frames-cache $CACHE"
python bench.py --steps 2 --warmup 1 $BARGS > gpurun_out/bench_cache.log 2>&1; echo "cache exit=$?"
for V in ${ORB_AB_VARIANTS:-stream staged stream2 staged2}; do
  for O in "" "--no-overlap"; do
    case $V in staged*) export SVO_ORB_RESIZE_STAGED=1;; *) unset SVO_ORB_RESIZE_STAGED;; esac
    python bench.py --steps 12 --warmup 3 $BARGS $O > gpurun_out/orb_ab_$V$O.json 2> gpurun_out/orb_ab_$V$O.err || { tail -5 gpurun_out/orb_ab_$V$O.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/orb_ab_$V$O.json"))
print("$V $O", d["ms_per_step"], d["value"], d["config"]["stage_ms_per_step"], d.get("self_check", {}).get("ok"))
PY
  done
done
