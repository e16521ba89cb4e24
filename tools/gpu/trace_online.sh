#!/bin/bash
# kernel timeline of the online path (one svo_add_frame per pair): -> gpurun_out/online_trace_<mode>.csv
MODE=${1:-lk}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --kernel-include-regex "svo::" --output-format csv -d /tmp/trace_online_$MODE -- python3 $R/tools/gpu/online_loop.py $MODE 12 > $R/gpurun_out/trace_online_$MODE.log 2>&1
f=$(find /tmp/trace_online_$MODE -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/online_trace_$MODE.csv; ls -la $R/gpurun_out/online_trace_$MODE.csv
