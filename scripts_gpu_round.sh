#!/bin/bash
# One GPU-box round: parity tests, bench, rocprofv3 kernel stats.  Output under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/tests.log 2>&1; echo "tests exit=$?" | tee -a gpurun_out/tests.log
tail -5 gpurun_out/tests.log
python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit=$?"
tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-pairs 0 > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo "prof exit=$?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*kernel_stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
