#!/usr/bin/env python3
"""Summarise tools/pmc_sq.sh (SQ counters of k_update) into profiles/<tag>_pmc_sq_cfg2.csv.  usage: tools/summarize_pmc_sq.py r01"""
import collections, csv, glob, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc, dur = collections.defaultdict(list), []
for p in ("pmc_sq_a", "pmc_sq_b"):
    f = sorted(glob.glob(os.path.join(root, "gpurun_out", p, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
    seen = set()
    for r in csv.DictReader(open(f)):
        if "k_update<" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if p == "pmc_sq_a" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
avg = {k: sum(v) / len(v) for k, v in acc.items()}
d = sum(dur) / len(dur)
n_simd, n_se = 1024, 32
derived = {
    "launches": len(dur), "avg_duration_us": d,
    "valu_instructions_per_wave": avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"],
    "cycles_per_valu_instruction": 4 * avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_INSTS_VALU"],
    "busy_cycles_per_shader_engine": avg["SQ_BUSY_CYCLES"] / n_se,
    "effective_clock_ghz": avg["SQ_BUSY_CYCLES"] / n_se / d / 1e3,
    "valu_busy_fraction": 4 * avg["SQ_ACTIVE_INST_VALU"] / n_simd / (avg["SQ_BUSY_CYCLES"] / n_se),
    "wave_cycles_issuing": avg["SQ_ACTIVE_INST_ANY"] / avg["SQ_WAVE_CYCLES"],
    "wave_cycles_waiting_for_issue": avg["SQ_WAIT_INST_ANY"] / avg["SQ_WAVE_CYCLES"],
    "wave_cycles_parked_waitcnt_barrier": avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"],
    "lds_conflict_fraction_of_lds_cycles": avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"],
}
out = os.path.join(root, "profiles", f"{tag}_pmc_sq_cfg2.csv")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["name", "value", "note"])
    for k in sorted(avg):
        w.writerow([k, "%.6g" % avg[k], "average per k_update<1,1,1,0> launch (SQ_WAVE_CYCLES / WAIT / ACTIVE_INST in quad-cycles)"])
    for k, v in derived.items():
        w.writerow([k, "%.6g" % v, "derived"])
for k, v in derived.items():
    print(k, "%.4g" % v)
