#!/bin/bash
# per-section s_memtime ticks of the node-parallel quadtree kernel, (image 0, level 0) instance, online ORB loop
mkdir -p gpurun_out
cd stereo-visual-odometry_amd/csrc && touch orb.hip && make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-result -DSVO_QT_STAMP" > /dev/null 2>&1; cd ../..
python3 tools/gpu/online_loop.py orb 6 > gpurun_out/qt_stamps.txt 2>&1; tail -8 gpurun_out/qt_stamps.txt
# (rebuild the library without the stamps before any other measurement: `make -C stereo-visual-odometry_amd/csrc clean all`)
