#!/bin/bash
# PMC counters of the online path's kernels (one pair per call): -> gpurun_out/pmc_online_<mode>.txt
MODE=${1:-lk}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
: > $R/gpurun_out/pmc_online_$MODE.txt
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-include-regex "svo::" --output-format csv -d /tmp/pmc_online_$MODE$i -- python3 $R/tools/gpu/online_loop.py $MODE 12 > $R/gpurun_out/pmc_online_$MODE$i.log 2>&1; echo "pmc$i exit=$?"
  f=$(find /tmp/pmc_online_$MODE$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/gpu/pmc_summary.py "$f" >> $R/gpurun_out/pmc_online_$MODE.txt
done
grep -E "lk_kernel|pnp_hyp|pnp_refit|triangulate" $R/gpurun_out/pmc_online_$MODE.txt
