#!/bin/bash
# PMC passes for the ORB-mode bench (bench.py --mode orb --batch 256); summaries -> gpurun_out/pmc_orb*.txt
mkdir -p gpurun_out
CACHE=/tmp/s0_frames_c1.pt
python bench.py --steps 1 --warmup 1 --cpu-pairs 0 --no-secondary --chunks 1 --mode orb --batch 256 --frames-cache $CACHE > gpurun_out/bench_cache.log 2>&1; echo "cache exit=$?"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-include-regex "svo::orb" --output-format csv -d /tmp/pmc_orb$i -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-pairs 0 --no-secondary --chunks 1 --mode orb --batch 256 --no-timing-marks --no-overlap --frames-cache $CACHE > $R/gpurun_out/pmc_orb$i.log 2>&1; echo "pmc$i exit=$?"
  f=$(find /tmp/pmc_orb$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/gpu/pmc_summary.py "$f" > $R/gpurun_out/pmc_orb$i.txt
done
cat $R/gpurun_out/pmc_orb[0-9].txt > $R/gpurun_out/pmc_orb_summary.txt
# kernel durations of the same command, then the JSON bench.py --mode orb reads (profiles/rNN_orb_pmc.json)
rm -rf /tmp/orb_stats; rocprofv3 --kernel-trace --stats --kernel-include-regex "svo::" --output-format csv -d /tmp/orb_stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-pairs 0 --no-secondary --chunks 1 --mode orb --batch 256 --no-overlap --frames-cache $CACHE > $R/gpurun_out/orb_stats.log 2>&1; echo "stats exit=$?"
f=$(find /tmp/orb_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/orb_kernel_stats.csv
python3 $R/tools/gpu/orb_pmc_json.py $R/gpurun_out/pmc_orb_summary.txt $R/gpurun_out/orb_kernel_stats.csv > $R/gpurun_out/orb_pmc.json; cat $R/gpurun_out/orb_pmc.json
