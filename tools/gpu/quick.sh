#!/bin/bash
# quick GPU check: parity tests + bench (no profiler)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/tests.log 2>&1; echo "tests exit=$?" | tee -a gpurun_out/tests.log
tail -15 gpurun_out/tests.log
python bench.py --steps 10 --warmup 2 --cpu-pairs 0 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit=$?"
tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
python bench.py --steps 10 --warmup 2 --cpu-pairs 0 --mode orb > gpurun_out/bench_orb.log 2> gpurun_out/bench_orb.err; echo "bench orb exit=$?"
tail -3 gpurun_out/bench_orb.log; tail -5 gpurun_out/bench_orb.err
