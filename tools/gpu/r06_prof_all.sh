#!/bin/bash
# Round 6: every profile bench.py's roofline objects read, on the round's final kernel sources (hash-guarded):
#   LK exact / sse2 / simd128 / sse2_legacy (prof.sh), the hd leg (prof_hd.sh), ORB (prof_orb.sh + pmc_orb.sh)
# -> gpurun_out/{lk,sse2,simd128,legacy}_pmc.json, prof_*_kernel_stats.csv, hd_lk_pmc.json, orb_pmc.json, ...
bash tools/gpu/prof.sh > gpurun_out/r06_prof_lk.log 2>&1; tail -2 gpurun_out/r06_prof_lk.log
PROF_KERNEL=lk_sse2_kernel PROF_TAG=sse2 bash tools/gpu/prof.sh --lk-accum sse2 > gpurun_out/r06_prof_sse2.log 2>&1; echo sse2 done
PROF_KERNEL=lk_sse2_kernel PROF_TAG=simd128 bash tools/gpu/prof.sh --lk-accum simd128 > gpurun_out/r06_prof_simd128.log 2>&1; echo simd128 done
PROF_KERNEL=lk_sse2_kernel PROF_TAG=legacy bash tools/gpu/prof.sh --lk-accum sse2_legacy > gpurun_out/r06_prof_legacy.log 2>&1; echo legacy done
bash tools/gpu/prof_hd.sh > gpurun_out/r06_prof_hd.log 2>&1; echo hd done
bash tools/gpu/prof_orb.sh > gpurun_out/r06_prof_orb.log 2>&1; echo orb stats done
bash tools/gpu/pmc_orb.sh > gpurun_out/r06_pmc_orb.log 2>&1; echo orb pmc done
