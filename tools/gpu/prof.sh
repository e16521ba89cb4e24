#!/bin/bash
# rocprofv3 round for the default bench command (LK mode, 256 pairs/step): kernel stats + PMC passes
# (each counter group in its own run, never together with a trace domain).
# Summaries -> gpurun_out/prof_*; tools/gpu/lk_pmc_json.py turns the lk_kernel rows into
# profiles/rNN_lk_pmc.json (what bench.py's roofline object reads); copy what is to be judged into profiles/.
# Usage: tools/gpu/prof.sh [extra bench.py flags, e.g. --lk-accum sse2] ; PROF_KERNEL=lk_sse2_kernel PROF_TAG=sse2 for the variant
mkdir -p gpurun_out
CACHE=/tmp/s0_frames_c2.pt
KERNEL=${PROF_KERNEL:-lk_kernel}
TAG=${PROF_TAG:-lk}
BARGS="--cpu-pairs 0 --no-secondary --no-self-check --chunks 2 --frames-cache $CACHE $*"
python bench.py --steps 2 --warmup 1 $BARGS > gpurun_out/bench_cache.log 2>&1; echo "cache exit=$?"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats; rocprofv3 --kernel-trace --stats --kernel-include-regex "svo::" --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 $BARGS > $R/gpurun_out/prof_${TAG}_stats.log 2>&1; echo "stats exit=$?"
grep '^{' $R/gpurun_out/prof_${TAG}_stats.log | tail -1 > $R/gpurun_out/prof_${TAG}_bench_line.json
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/prof_${TAG}_kernel_stats.csv && cat "$f" | cut -c1-160
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/prof_pmc$i; rocprofv3 --pmc $PMC --kernel-include-regex "svo::" --output-format csv -d /tmp/prof_pmc$i -- python3 $R/bench.py --steps 2 --warmup 1 $BARGS --no-timing-marks --no-overlap > $R/gpurun_out/prof_${TAG}_pmc$i.log 2>&1; echo "pmc$i exit=$?"
  f=$(find /tmp/prof_pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/gpu/pmc_summary.py "$f" > $R/gpurun_out/prof_${TAG}_pmc$i.txt
done
cat $R/gpurun_out/prof_${TAG}_pmc[0-9].txt > $R/gpurun_out/prof_${TAG}_pmc_summary.txt
grep $KERNEL $R/gpurun_out/prof_${TAG}_pmc_summary.txt
python3 $R/tools/gpu/lk_pmc_json.py $R/gpurun_out/prof_${TAG}_pmc_summary.txt $R/gpurun_out/prof_${TAG}_kernel_stats.csv $KERNEL > $R/gpurun_out/${TAG}_pmc.json; cat $R/gpurun_out/${TAG}_pmc.json
