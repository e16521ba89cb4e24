#!/bin/bash
# Round 6, item 1(a): kernel stats + PMC passes of the two float-order LK legs that had no profile (sse2_legacy, simd128),
# plus a kernel timeline of the sse2 leg (finalize_chain_kernel against the next lk_sse2_kernel launch).
# -> gpurun_out/prof_legacy_*, prof_simd128_*, sse2_trace.csv
set -o pipefail
PROF_KERNEL=lk_sse2_kernel PROF_TAG=legacy bash tools/gpu/prof.sh --lk-accum sse2_legacy || exit 1
PROF_KERNEL=lk_sse2_kernel PROF_TAG=simd128 bash tools/gpu/prof.sh --lk-accum simd128 || exit 1
R=$GRAFT_REPO_ROOT
CACHE=/tmp/s0_frames_c2.pt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_sse2; rocprofv3 --kernel-trace --kernel-include-regex "svo::" --output-format csv -d /tmp/trace_sse2 -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-pairs 0 --no-secondary --no-self-check --chunks 2 --no-timing-marks --lk-accum sse2 --frames-cache $CACHE > $R/gpurun_out/trace_sse2.log 2>&1; echo "trace exit=$?"
f=$(find /tmp/trace_sse2 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/sse2_trace.csv; ls -la $R/gpurun_out/sse2_trace.csv
