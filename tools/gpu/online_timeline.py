#!/usr/bin/env python3
"""Kernel timeline of the median pair of an online trace (tools/gpu/trace_online.sh):
python3 tools/gpu/online_timeline.py gpurun_out/online_trace_orb.csv > profiles/rNN_online_kernel_timeline_orb.csv
A pair's kernels end with chain_kernel."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "svo::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a pair ends with chain_kernel; the pair printed is the one with the MEDIAN span (a lone pair's trace catches the odd
# 50-100 us hiccup of the host or of the profiler)
ends_at = [i for i, r in enumerate(rows) if "chain_kernel" in r["Kernel_Name"]]
pairs = []
for a, b in zip([-1] + ends_at[:-1], ends_at):
    seg = rows[a + 1:b + 1]
    pairs.append((int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"]), a + 1, b + 1))
pairs = sorted(pairs[1:] if len(pairs) > 2 else pairs)      # (the first pair after start-up is not typical)
_, lo, hi = pairs[len(pairs) // 2]
last = rows[lo:hi]
t0 = int(last[0]["Start_Timestamp"])
print("kernel,start_us,duration_us,grid,workgroup")
for r in last:
    name = r["Kernel_Name"].split("svo::")[1].split("(")[0]
    print(f'{name},{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f},{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.1f},'
          f'{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]},{r["Workgroup_Size_X"]}')
print(f'# span_us,{(int(last[-1]["End_Timestamp"]) - t0) / 1e3:.1f}')
