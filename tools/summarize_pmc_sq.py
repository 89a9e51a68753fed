#!/usr/bin/env python3
"""Summarise tools/pmc_sq.sh (SQ counters of the update kernel) into profiles/<tag>_pmc_sq_<cfg>.csv.
usage: tools/summarize_pmc_sq.py r02 cfg4 [kernel-name substring, default k_update]"""
import collections, csv, glob, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
kname = sys.argv[3] if len(sys.argv) > 3 else "k_update"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc, dur, full_name = collections.defaultdict(list), [], None
for p in (f"pmc_sq_a_{cfg}", f"pmc_sq_b_{cfg}", f"pmc_sq_c_{cfg}"):
    files = sorted(glob.glob(os.path.join(root, "gpurun_out", p, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        continue
    seen = set()
    for r in csv.DictReader(open(files[-1])):
        if kname not in r["Kernel_Name"]:
            continue
        full_name = full_name or r["Kernel_Name"]
        if r["Kernel_Name"] != full_name:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if p.startswith("pmc_sq_a") and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
avg = {k: sum(v) / len(v) for k, v in acc.items()}
d = sum(dur) / len(dur)
n_simd, n_se = 1024, 32
g = lambda k: avg.get(k, float("nan"))
derived = {
    "launches": len(dur), "avg_duration_us": d,
    "valu_instructions_per_wave": g("SQ_INSTS_VALU") / g("SQ_WAVES"),
    "cycles_per_valu_instruction": 4 * g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU"),
    "busy_cycles_per_shader_engine": g("SQ_BUSY_CYCLES") / n_se,
    "effective_clock_ghz": g("SQ_BUSY_CYCLES") / n_se / d / 1e3,
    "valu_busy_fraction": 4 * g("SQ_ACTIVE_INST_VALU") / n_simd / (g("SQ_BUSY_CYCLES") / n_se),
    "lds_busy_fraction": 4 * g("SQ_ACTIVE_INST_LDS") / n_simd / (g("SQ_BUSY_CYCLES") / n_se),
    "wave_cycles_issuing": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
    "wave_cycles_waiting_for_issue": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
    "wave_cycles_parked_waitcnt_barrier": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
    "wave_cycles_waiting_for_lds_issue": g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"),
    "lds_instructions_per_wave": g("SQ_INSTS_LDS") / g("SQ_WAVES"),
    "vmem_read_instructions_per_wave": g("SQ_INSTS_VMEM_RD") / g("SQ_WAVES"),
    "salu_instructions_per_wave": g("SQ_INSTS_SALU") / g("SQ_WAVES"),
    "lds_conflict_fraction_of_lds_cycles": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"),
    "mean_waves_in_flight_per_simd": g("SQ_WAVE_CYCLES") * 4 / n_simd / (g("SQ_BUSY_CYCLES") / n_se),
}
out = os.path.join(root, "profiles", f"{tag}_pmc_sq_{cfg}.csv")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["name", "value", "note"])
    for k in sorted(avg):
        w.writerow([k, "%.6g" % avg[k], f"average per {full_name} launch (SQ_WAVE_CYCLES / WAIT / ACTIVE_INST in quad-cycles)"])
    for k, v in derived.items():
        w.writerow([k, "%.6g" % v, "derived"])
print(full_name)
for k, v in derived.items():
    print(k, "%.4g" % v)
