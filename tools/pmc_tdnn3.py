"""Reduce the rocprofv3 --pmc passes of tools/pmc_tdnn3.sh to one JSON record per launch of the tdnn3 forward contraction."""
import csv, glob, json, sys
from collections import defaultdict
root = sys.argv[1]
vals = defaultdict(list)
kernel_names = set()
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    # the forward (BIAS_RELU) instantiation of whatever stream-K kind the launcher picked (round 3: kind 8 for tdnn3 at batch 64)
    rows = [r for r in csv.DictReader(open(f)) if "conv_gemm_streamk_kernel<1, " in r["Kernel_Name"]]
    kernel_names.update(r["Kernel_Name"].split("(")[0] for r in rows[-4:])
    by_counter = defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for name, lst in by_counter.items():
        lst.sort()
        # layer_bench first runs one full pass (tdnn2..5 forward use this kernel too), then 1 warm + 4 timed launches of
        # tdnn3 forward: keep the last four dispatches
        vals[name] += [v for _, v in lst[-4:]]
per = {k: sum(v) / len(v) for k, v in vals.items()}
fetch_kb, write_kb = per.get("FETCH_SIZE", 0.0), per.get("WRITE_SIZE", 0.0)
traffic = int((2.0 * fetch_kb + write_kb) * 1024)
gui = per.get("GRBM_GUI_ACTIVE", 0.0)
rec = {
    "kernel": "%s: tdnn3 forward (B=64: M=17280 N=512 K=3584); round 3 = kind 8, 256x128 tile, eight 64x64 computing waves + four staging waves" % ", ".join(sorted(kernel_names)),
    "command": "bash tools/pmc_tdnn3.sh  (rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 tools/layer_bench.py "
               "--layers 3 --iters 4 --repeats 1; one pass per group: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE | SQ_*)",
    "launches_averaged": {k: len(v) for k, v in vals.items()},
    "per_launch": {k: per[k] for k in sorted(per)},
    "derived": {
        "mfma_busy_fraction": per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0) if gui else None,
        "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs",
        "tcc_hit_rate": per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"]) if "TCC_HIT_sum" in per else None,
    },
    "corrections": "gfx950: FETCH_SIZE tallies 128-B requests of wide (16 B/lane) reads at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); "
                   "WRITE_SIZE as reported (uncalibrated). L2 memory-side (fabric) bytes: Infinity-Cache hits are included, so this is an upper "
                   "bound on HBM traffic.",
    "traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": 81100000,
}
print(json.dumps(rec, indent=2))
