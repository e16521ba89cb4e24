#!/usr/bin/env python3
"""ORB-mode kernels of a PMC summary (tools/gpu/pmc_summary.py output of pmc_orb.sh) + their kernel-trace durations
-> the JSON bench.py --mode orb reads (profiles/rNN_orb_pmc.json): per kernel the HBM traffic per launch as
MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in KiB, gfx950's FETCH_SIZE x2 correction), instruction
counts and the launch time; `cellfast_traffic_bytes` (per STEP: all cell-FAST launches of a step) is the figure of bench.py's roofline object."""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

vals = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    parts = line.split()
    if len(parts) < 3 or "svo::" not in line:
        continue
    kern = line.split("(")[0].replace("void ", "").strip().replace("svo::", "")      # templated kernels print as "void svo::name<..>("
    name = [p for p in parts if p.isupper() or p.endswith("_sum")]
    if not name:
        continue
    mean = [p for p in parts if p.startswith("mean=")][0]
    mean = float(mean[5:]) if len(mean) > 5 else float(parts[parts.index("mean=") + 1])
    vals[kern][name[0]] = mean
dur = {}
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    for row in csv.DictReader(open(sys.argv[2])):
        dur[row["Name"].split("(")[0].replace("void ", "").strip().replace("svo::", "")] = (float(row["AverageNs"]) * 1e-6, int(row["Calls"]))
out = {"command": "tools/gpu/pmc_orb.sh: rocprofv3 --pmc <one group per run> -- python3 bench.py --mode orb --batch 256 --steps 1 --warmup 1 "
                  "--no-overlap (256 S0 pairs per step: 514 images through the extractor)", "kernels": {}}
for k, v in sorted(vals.items()):
    f, w = v.get("FETCH_SIZE"), v.get("WRITE_SIZE")
    e = {"avg_launch_ms": round(dur[k][0], 4) if k in dur else None,
         "traffic_bytes": int((2 * f + w) * 1024) if f is not None and w is not None else None,
         "fetch_size_kb": f, "write_size_kb": w, "valu_wave_instructions": v.get("SQ_INSTS_VALU"),
         "salu_wave_instructions": v.get("SQ_INSTS_SALU"), "lds_wave_instructions": v.get("SQ_INSTS_LDS"), "waves": v.get("SQ_WAVES"),
         "lds_bank_conflict_cycles": v.get("SQ_LDS_BANK_CONFLICT"), "lds_idx_active_cycles": v.get("SQ_LDS_IDX_ACTIVE"),
         "tcc_hit": v.get("TCC_HIT_sum"), "tcc_miss": v.get("TCC_MISS_sum"), "grbm_gui_active": v.get("GRBM_GUI_ACTIVE")}
    # a kernel launched several times per step (round 6: cell FAST runs one launch per run of levels of one occupancy class):
    # launches per step = its calls / the blur kernel's (one launch per step); the per-launch means above x that = per step
    if k in dur and "orb_blur_kernel" in dur and dur["orb_blur_kernel"][1] > 0:
        lps = dur[k][1] / dur["orb_blur_kernel"][1]
        e["launches_per_step"] = round(lps, 3)
        e["per_step_ms"] = round(dur[k][0] * lps, 4)
    out["kernels"][k] = e
h = hashlib.sha256()
for f in ("orb.hip", "orb_pattern.h", "svo_device.h", "svo_kernels.h"):             # what the ORB kernels are built from (bench.py orb_source_hash)
    h.update(open(os.path.join(ROOT, "stereo-visual-odometry_amd", "csrc", f), "rb").read())
out["source_sha256_16"] = h.hexdigest()[:16]
cf = out["kernels"].get("orb_cellfast_kernel", {})
out["cellfast_traffic_bytes"] = int(cf["traffic_bytes"] * cf.get("launches_per_step", 1)) if cf.get("traffic_bytes") is not None else None   # per STEP
print(json.dumps(out, indent=1))
