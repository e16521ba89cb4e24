#!/bin/bash
# rocprofv3 kernel stats of the ORB-mode bench (bench.py --mode orb, 256 pairs/step) -> gpurun_out/prof_orb_kernel_stats.csv
mkdir -p gpurun_out
CACHE=/tmp/s0_frames_c1.pt
python bench.py --steps 1 --warmup 1 --cpu-pairs 0 --no-secondary --chunks 1 --mode orb --batch 256 --frames-cache $CACHE > gpurun_out/bench_cache.log 2>&1; echo "cache exit=$?"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_orb; rocprofv3 --kernel-trace --stats --kernel-include-regex "svo::" --output-format csv -d /tmp/prof_orb -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --no-secondary --chunks 1 --mode orb --batch 256 --no-timing-marks --frames-cache $CACHE > $R/gpurun_out/prof_orb.log 2>&1; echo "stats exit=$?"
f=$(find /tmp/prof_orb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/prof_orb_kernel_stats.csv && cat "$f"
