"""Condenses the rocprofv3 outputs of scripts/gpu_check.sh (gpurun_out/prof, gpurun_out/pmc*) into one text summary."""
import collections, csv, glob, json
for f in glob.glob("gpurun_out/prof/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", f)
    for r in csv.DictReader(open(f)):
        print(f"  {r['Name'][:90]:90s} calls={r['Calls']} avg_ns={float(r['AverageNs']):.0f} min={r['MinNs']} max={r['MaxNs']} pct={r['Percentage']}")
tot = {}
for d in sorted(glob.glob("gpurun_out/pmc*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "aba_" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            tot[k] = sum(v) / len(v)
print("== PMC (per ABA kernel launch, averaged):")
for k in sorted(tot):
    print(f"  {k:24s} {tot[k]:16.1f}")
if "SQ_WAVES" in tot:
    w = tot["SQ_WAVES"]
    print(f"  per wave: VALU insts {tot.get('SQ_INSTS_VALU', 0)/w:.0f}, LDS insts {tot.get('SQ_INSTS_LDS', 0)/w:.0f}, SALU {tot.get('SQ_INSTS_SALU', 0)/w:.0f}, "
          f"wave quad-cycles {tot.get('SQ_WAVE_CYCLES', 0)/w:.0f}")
if "FETCH_SIZE" in tot:
    # FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM) —
    # the x2 correction is applied as an upper bound; narrow 8-byte accesses are uncalibrated.
    fetch, write = tot["FETCH_SIZE"] * 1024, tot.get("WRITE_SIZE", 0) * 1024
    print(f"  HBM traffic per launch: FETCH {fetch:.0f} B (x2 corrected: {2*fetch:.0f} B), WRITE {write:.0f} B")
    json.dump({"fetch_bytes_raw": fetch, "fetch_bytes_x2": 2 * fetch, "write_bytes": write}, open("gpurun_out/pmc_traffic.json", "w"))
