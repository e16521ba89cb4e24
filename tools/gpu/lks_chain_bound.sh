#!/bin/bash
# Round 6: what can a fast path for lk_sse2_kernel's serial b chains buy at most?  Builds the library with the timing experiments
# compiled in (-DSVO_LKS_EXPERIMENTS, on the GPU box's scratch copy only) and runs the sse2 bench leg with
#   SVO_LKS_EXP unset: the product kernel | 1: s_setprio(3) around the chain | 2: the terms added as a TREE, no serial chain (an
#   upper bound: what a 100 % hit rate at zero guard cost would give) | 3: tree + guard in the chain lanes + serial chain on a miss
# results: profiles/r06_lk_sse2_chain_bound.json
touch stereo-visual-odometry_amd/csrc/lk_sse2.hip
make -C stereo-visual-odometry_amd/csrc EXTRA=-DSVO_LKS_EXPERIMENTS > /dev/null || exit 1
LK_ENV_AB_ARGS="--lk-accum sse2 --no-self-check" LK_ENV_AB="- SVO_LKS_EXP=1 SVO_LKS_EXP=2 SVO_LKS_EXP=3 -" bash tools/gpu/lk_env_ab.sh
