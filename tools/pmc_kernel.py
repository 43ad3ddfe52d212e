"""Reduce rocprofv3 --pmc passes (one directory per counter group, as tools/pmc_*.sh write them) to one JSON record per
launch of the kernels whose name contains `substr`.   python tools/pmc_kernel.py ROOT SUBSTR [ALGORITHMIC_BYTES]"""
import csv, glob, json, sys
from collections import defaultdict
root, substr = sys.argv[1], sys.argv[2]
alg = int(sys.argv[3]) if len(sys.argv) > 3 else None
vals, names = defaultdict(list), set()
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if substr in r["Kernel_Name"]:
            names.add(r["Kernel_Name"].split("(")[0])
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
per = {k: sum(v) / len(v) for k, v in vals.items()}
gui = per.get("GRBM_GUI_ACTIVE", 0.0)
fetch_kb, write_kb = per.get("FETCH_SIZE", 0.0), per.get("WRITE_SIZE", 0.0)
rec = {"kernel": ", ".join(sorted(names)), "launches_averaged": {k: len(v) for k, v in vals.items()},
       "per_launch": {k: per[k] for k in sorted(per)},
       "derived": {
           "mfma_busy_fraction": per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0) if gui else None,
           "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs",
           "tcc_hit_rate": per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"]) if "TCC_HIT_sum" in per else None,
           "lds_bank_conflict_fraction": per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"] if per.get("SQ_LDS_IDX_ACTIVE") else None,
           "wave_cycles_waiting_on_lds": per["SQ_WAIT_INST_LDS"] / per["SQ_WAVE_CYCLES"] if per.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_LDS" in per else None,
       },
       "corrections": "gfx950: FETCH_SIZE tallies 128-B requests of wide (16 B/lane) reads at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); "
                      "WRITE_SIZE as reported.  Memory-side (fabric) bytes incl. Infinity-Cache hits: an upper bound on HBM traffic.",
       "traffic_bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024), "algorithmic_bytes_per_launch": alg}
print(json.dumps(rec, indent=2))
