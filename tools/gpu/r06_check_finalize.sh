#!/bin/bash
# Round 6, item 2: the new ordering / host tests, then the sse2 kernel timeline again (finalize_chain_kernel as one wave on a
# high-priority side stream) and the same with SVO_SIDE_PRIORITY=0 -> gpurun_out/sse2_trace_{prio,noprio}.csv
set -o pipefail
python -m pytest tests/test_gpu_stream_order.py tests/test_host_api.py tests/test_gpu_stream.py -x -q -m gpu > gpurun_out/r06_t1.log 2>&1 || { tail -30 gpurun_out/r06_t1.log; exit 1; }
tail -3 gpurun_out/r06_t1.log
R=$GRAFT_REPO_ROOT
CACHE=/tmp/s0_frames_c2.pt
python bench.py --steps 2 --warmup 1 --cpu-pairs 0 --no-secondary --no-self-check --chunks 2 --frames-cache $CACHE > gpurun_out/bench_cache.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
for P in prio noprio; do
  if [ $P = noprio ]; then export SVO_SIDE_PRIORITY=0; fi
  rm -rf /tmp/trace_$P; rocprofv3 --kernel-trace --kernel-include-regex "svo::" --output-format csv -d /tmp/trace_$P -- python3 $R/bench.py --steps 6 --warmup 1 --cpu-pairs 0 --no-secondary --no-self-check --chunks 2 --no-timing-marks --lk-accum sse2 --frames-cache $CACHE > $R/gpurun_out/trace_sse2_$P.log 2>&1 || exit 1
  f=$(find /tmp/trace_$P -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/sse2_trace_$P.csv
  grep '^{' $R/gpurun_out/trace_sse2_$P.log | tail -1 | cut -c1-200
  # throughput A/B, unprofiled: default (exact) and ORB
  python3 $R/bench.py --steps 20 --warmup 3 --cpu-pairs 0 --no-secondary --no-self-check --frames-cache /tmp/s0_frames_c4.pt 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$P exact', d['value'], d['ms_per_step'])"
  python3 $R/bench.py --mode orb --steps 20 --warmup 3 --cpu-pairs 0 --no-secondary --no-self-check --frames-cache /tmp/s0_frames_c4.pt 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$P orb', d['value'], d['ms_per_step'])"
done
