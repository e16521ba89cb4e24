#!/bin/bash
# LK-mode bench (the default workload, no legs) under different environment settings in one session: LK_ENV_AB="- NAME=VALUE ..."
B="--no-legs --no-secondary --cpu-pairs 0 --steps 20 --self-check-pairs 32 $LK_ENV_AB_ARGS"      # e.g. LK_ENV_AB_ARGS="--lk-accum sse2"
for V in $LK_ENV_AB; do
  ( if [ "$V" != "-" ]; then for kv in ${V//,/ }; do export "$kv"; done; fi
    python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V', d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d.get('self_check', {}).get('ok'))" )
done
