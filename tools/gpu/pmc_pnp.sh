#!/bin/bash
mkdir -p gpurun_out
CACHE=/tmp/s0_frames_c1.pt
python bench.py --steps 1 --warmup 1 --cpu-pairs 0 --no-secondary --chunks 1 --mode orb --batch 256 --frames-cache $CACHE > gpurun_out/bench_cache.log 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
: > $R/gpurun_out/pmc_pnp.txt
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-include-regex "svo::pnp" --output-format csv -d /tmp/pmc_pnp$i -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-pairs 0 --no-secondary --chunks 1 --mode orb --batch 256 --no-timing-marks --no-overlap --frames-cache $CACHE > $R/gpurun_out/pmc_pnp$i.log 2>&1
  f=$(find /tmp/pmc_pnp$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/gpu/pmc_summary.py "$f" >> $R/gpurun_out/pmc_pnp.txt
done
