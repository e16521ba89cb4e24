#!/bin/bash
# rocprofv3 kernel stats of the ORB-mode bench (config #3), pose stage in stream order (--no-overlap) and overlapped.
mkdir -p gpurun_out
CACHE=/tmp/s0_frames_c2.pt
BARGS="--mode orb --cpu-pairs 0 --no-secondary --no-self-check --chunks 2 --frames-cache $CACHE"
python bench.py --steps 2 --warmup 1 $BARGS > gpurun_out/bench_cache.log 2>&1; echo "cache exit=$?"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for V in no_overlap overlap; do
  X=""; [ $V = no_overlap ] && X="--no-overlap"
  rm -rf /tmp/prof_orb_$V; rocprofv3 --kernel-trace --stats --kernel-include-regex "svo::" --output-format csv -d /tmp/prof_orb_$V -- python3 $R/bench.py --steps 10 --warmup 2 $BARGS $X > $R/gpurun_out/prof_orb_$V.log 2>&1; echo "$V exit=$?"
  f=$(find /tmp/prof_orb_$V -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/prof_orb_kernel_stats_$V.csv && cut -c1-130 "$f" | head -24
  grep '^{' $R/gpurun_out/prof_orb_$V.log | tail -1 > $R/gpurun_out/prof_orb_bench_line_$V.json
done
