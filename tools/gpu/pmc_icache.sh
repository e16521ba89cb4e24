#!/bin/bash
# instruction-fetch counters of the online path's lone-wave kernels: -> gpurun_out/pmc_icache_<mode>.txt
MODE=${1:-orb}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc_list.txt 2>&1
i=0
: > $R/gpurun_out/pmc_icache_$MODE.txt
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_IFETCH" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH_LEVEL" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_ic_$MODE$i
  rocprofv3 --pmc $PMC --kernel-include-regex "svo::" --output-format csv -d /tmp/pmc_ic_$MODE$i -- python3 $R/tools/gpu/online_loop.py $MODE 12 > $R/gpurun_out/pmc_ic_$MODE$i.log 2>&1; echo "pmc$i exit=$?"
  f=$(find /tmp/pmc_ic_$MODE$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $R/tools/gpu/pmc_summary.py "$f" >> $R/gpurun_out/pmc_icache_$MODE.txt
done
cat $R/gpurun_out/pmc_icache_$MODE.txt
