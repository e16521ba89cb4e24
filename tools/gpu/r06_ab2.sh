#!/bin/bash
# side-stream priority A/B (ORB and LK modes, overlap on) + the new pose-latency test
python -m pytest tests/test_gpu_stream.py -x -q -m gpu -k "poses_of_a_batch" 2>&1 | tail -5
for P in 1 0 1 0; do
  export SVO_SIDE_PRIORITY=$P
  python bench.py --mode orb --steps 30 --warmup 3 --cpu-pairs 0 --no-secondary --no-self-check --frames-cache /tmp/s0_frames_c4.pt 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio=$P orb', d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'])"
  python bench.py --steps 30 --warmup 3 --cpu-pairs 0 --no-secondary --no-self-check --frames-cache /tmp/s0_frames_c4.pt 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio=$P lk', d['value'], d['ms_per_step'])"
done
