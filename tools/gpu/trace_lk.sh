#!/bin/bash
# kernel timeline (start/end) of the LK-mode bench: -> gpurun_out/lk_trace.csv (svo:: kernels only)
mkdir -p gpurun_out
CACHE=/tmp/s0_frames_c1.pt
python bench.py --steps 1 --warmup 1 --cpu-pairs 0 --no-secondary --chunks 1  --frames-cache $CACHE > gpurun_out/bench_cache.log 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_lk; rocprofv3 --kernel-trace --kernel-include-regex "svo::" --output-format csv -d /tmp/trace_lk -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --no-secondary --chunks 1  --no-timing-marks --frames-cache $CACHE > $R/gpurun_out/trace_lk.log 2>&1
f=$(find /tmp/trace_lk -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/lk_trace.csv; ls -la $R/gpurun_out/lk_trace.csv
