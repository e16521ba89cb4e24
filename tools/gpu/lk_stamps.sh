#!/bin/bash
# Where a wave of lk_kernel spends its cycles: rebuilds the library ON THE GPU BOX with -DSVO_LK_STAMP=k for one
# section k at a time (s_memtime stamps around that section of lk_call4 only; 8 = the whole call), runs 256 S0 pairs
# through svo_track_batch and prints the section's share of a wave's life.  -> gpurun_out/lk_stamps.txt
# (the product build in the repo is not touched: the box's copy is scratch)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
: > $R/gpurun_out/lk_stamps.txt
for k in 8 0 1 2 4 5 6; do
  cd $R/stereo-visual-odometry_amd/csrc && touch lk.hip && make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DSVO_LK_STAMP=$k" > /dev/null 2>&1
  cd $R && python3 tools/gpu/lk_stamps.py $k | tee -a gpurun_out/lk_stamps.txt
done
