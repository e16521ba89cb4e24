#!/usr/bin/env python3
"""lk_kernel rows of a PMC summary (tools/gpu/pmc_summary.py output of the passes in prof.sh) ->
the JSON bench.py's roofline object reads (profiles/rNN_lk_pmc.json): HBM traffic per launch as
MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE are KiB counters; gfx950's FETCH_SIZE
under-reports wide reads by 2x: the correction is applied, which makes the figure an upper estimate
for this kernel's 4-byte gathers), VALU wave-instructions per launch, and the hash of the kernel
source the numbers belong to."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "lk_kernel"          # or lk_sse2_kernel (bench.py --lk-accum sse2)
COMMAND = sys.argv[4] if len(sys.argv) > 4 else ("tools/gpu/prof.sh: rocprofv3 --pmc <one group per run> --kernel-include-regex svo:: -- python3 bench.py --steps 2 "
                                                 "--warmup 1 --cpu-pairs 0 --no-secondary --no-timing-marks --no-overlap (256 S0 pairs per launch)")
vals = {}
for line in open(sys.argv[1]):
    if KERNEL not in line:
        continue
    parts = line.split()
    name = [p for p in parts if p.isupper() or p.endswith("_sum")][0]
    mean = float([p for p in parts if p.startswith("mean=")][0][5:] or parts[parts.index("mean=") + 1]) if any(
        p.startswith("mean=") and len(p) > 5 for p in parts) else float(parts[parts.index("mean=") + 1])
    vals[name] = mean
avg_ns = None
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    import csv
    for row in csv.DictReader(open(sys.argv[2])):
        if KERNEL in row["Name"]:
            avg_ns = float(row["AverageNs"])
h = hashlib.sha256()
for f in ("lk.hip", "lk_common.h", "svo_device.h", "svo_kernels.h") + (("lk_sse2.hip",) if KERNEL != "lk_kernel" else ()):
    h.update(open(os.path.join(ROOT, "stereo-visual-odometry_amd", "csrc", f), "rb").read())
fetch_kb, write_kb = vals.get("FETCH_SIZE"), vals.get("WRITE_SIZE")
out = {
    "kernel": "svo::" + KERNEL,
    "command": COMMAND,
    "source_sha256_16": h.hexdigest()[:16],
    "kernel_trace_avg_launch_ms": round(avg_ns * 1e-6, 4) if avg_ns else None,
    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
    "traffic_bytes": int((2 * fetch_kb + write_kb) * 1024) if fetch_kb is not None and write_kb is not None else None,
    "valu_wave_instructions": vals.get("SQ_INSTS_VALU"),
    "salu_wave_instructions": vals.get("SQ_INSTS_SALU"),
    "lds_wave_instructions": vals.get("SQ_INSTS_LDS"),
    "sq_active_inst_valu_quadcycles": vals.get("SQ_ACTIVE_INST_VALU"),
    "sq_busy_cycles": vals.get("SQ_BUSY_CYCLES"), "sq_waves": vals.get("SQ_WAVES"),
    "tcc_hit": vals.get("TCC_HIT_sum"), "tcc_miss": vals.get("TCC_MISS_sum"),
    "lds_bank_conflict_cycles": vals.get("SQ_LDS_BANK_CONFLICT"), "lds_idx_active_cycles": vals.get("SQ_LDS_IDX_ACTIVE"),
    "wait_inst_any_wave_cycles": vals.get("SQ_WAIT_INST_ANY"), "wave_cycles": vals.get("SQ_WAVE_CYCLES"),
    "grbm_gui_active": vals.get("GRBM_GUI_ACTIVE"),
    # shader clock during the launch: GRBM_GUI_ACTIVE sums the busy cycles of the 8 XCDs
    "clock_ghz_measured": (round(vals["GRBM_GUI_ACTIVE"] / 8 / avg_ns, 4) if vals.get("GRBM_GUI_ACTIVE") and avg_ns else None),
    "valu_peak_wave_instr_per_cycle_per_simd": 0.246,
    "valu_peak_source": "profiles/r02_valu_roof.txt: v_dot2_i32_i16, v_perm_b32, v_alignbyte_b32, v_pk_*, DPP, v_cndmask, v_readlane "
                        "all issue one wave-instruction per 4.06-4.2 cycles per SIMD at 2-8 waves per SIMD (plain 32-bit add/and/shift: 2.03-2.3)",
}
print(json.dumps(out, indent=1))
