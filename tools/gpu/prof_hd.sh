#!/bin/bash
# rocprofv3 round for the bench's hd leg (BASELINE config #4: 1920x1080, exactly 2000 corners, 128 pairs per step), run alone
# by tools/gpu/hd_leg.py: kernel stats + the PMC passes (each counter group in its own run, the program directly after `--`).
# -> gpurun_out/hd_lk_pmc.json (copy to profiles/rNN_hd_lk_pmc.json: what bench.py's hd.roofline reads for --hd-batch 128)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hd_stats; rocprofv3 --kernel-trace --stats --kernel-include-regex "svo::" --output-format csv -d /tmp/hd_stats -- python3 $R/tools/gpu/hd_leg.py 128 6 > $R/gpurun_out/prof_hd_stats.log 2>&1; echo "stats exit=$?"
f=$(find /tmp/hd_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/prof_hd_kernel_stats.csv && cut -c1-150 "$f" | head -6
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/hd_pmc$i; rocprofv3 --pmc $PMC --kernel-include-regex "svo::lk" --output-format csv -d /tmp/hd_pmc$i -- python3 $R/tools/gpu/hd_leg.py 128 2 > $R/gpurun_out/prof_hd_pmc$i.log 2>&1; echo "pmc$i exit=$?"
  f=$(find /tmp/hd_pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/gpu/pmc_summary.py "$f" > $R/gpurun_out/prof_hd_pmc$i.txt
done
cat $R/gpurun_out/prof_hd_pmc[0-9].txt > $R/gpurun_out/prof_hd_pmc_summary.txt
python3 $R/tools/gpu/lk_pmc_json.py $R/gpurun_out/prof_hd_pmc_summary.txt $R/gpurun_out/prof_hd_kernel_stats.csv lk_kernel \
  "tools/gpu/prof_hd.sh: rocprofv3 --pmc <one group per run> --kernel-include-regex svo::lk -- python3 tools/gpu/hd_leg.py 128 2 (the bench's hd leg alone: 128 pairs of 1920x1080 frames per launch, exactly 2000 corners per frame, overlap on)" \
  > $R/gpurun_out/hd_lk_pmc.json; cat $R/gpurun_out/hd_lk_pmc.json
