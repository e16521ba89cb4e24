#!/bin/bash
# per-section s_memtime ticks of orb_cellfast_kernel's first wave, a few workgroups of image 2, batched ORB bench (one step)
mkdir -p gpurun_out
cd stereo-visual-odometry_amd/csrc && touch orb.hip && make EXTRA=-DSVO_CF_STAMP > /dev/null 2>&1; cd ../..
python3 bench.py --mode orb --steps 1 --warmup 0 --batch 8 --chunks 1 --cpu-pairs 0 --no-secondary --no-self-check > gpurun_out/cf_stamps.txt 2>&1; grep "^cf " gpurun_out/cf_stamps.txt | sort | uniq -c | head -40
# (rebuild the library without the stamps before any other measurement: touch orb.hip && make -C stereo-visual-odometry_amd/csrc)
