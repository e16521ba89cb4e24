#!/bin/bash
# ORB-mode bench under different environment settings in one session: ORB_ENV_AB="NAME=VALUE[,NAME=VALUE] ..." (a "-" = none)
mkdir -p gpurun_out
CACHE=/tmp/s0_frames_c2.pt
BARGS="--mode orb --cpu-pairs 0 --no-secondary --no-legs --self-check-pairs 16 --chunks 2 --frames-cache $CACHE"
python bench.py --steps 2 --warmup 1 $BARGS > gpurun_out/bench_cache.log 2>&1; echo "cache exit=$?"
i=0
for V in $ORB_ENV_AB; do
  for O in "" "--no-overlap"; do
    ( if [ "$V" != "-" ]; then for kv in ${V//,/ }; do export "$kv"; done; fi
      python bench.py --steps 12 --warmup 3 $BARGS $O > gpurun_out/orb_env_$i$O.json 2> gpurun_out/orb_env_$i$O.err ) || { tail -5 gpurun_out/orb_env_$i$O.err; exit 1; }
    python - <<PY
import json
d = json.load(open("gpurun_out/orb_env_$i$O.json"))
print("$V $O", d["ms_per_step"], d["value"], d["config"]["stage_ms_per_step"], d.get("self_check", {}).get("ok"))
PY
  done
  i=$((i+1))
done
