"""Condenses what scripts/gpu_measure.sh collected under gpurun_out/measure into the files that are committed under profiles/, one set PER LEG (a leg = one timed
entry point of one BASELINE config, profiled in a run of its own): r06_kernel_stats_<leg>.csv (rocprofv3 --kernel-trace --stats, our kernels),
r06_pmc_<leg>.txt (PMC per launch and per wavefront) and r06_pmc_traffic.json (HBM bytes per step from FETCH_SIZE / WRITE_SIZE, keyed like bench.py looks it
up, with the kernel-source hash)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (CONFIGS, kernel_source_hash)

OUT = os.path.join(ROOT, "gpurun_out", "measure")
OURS = ("aba_", "rnea_", "crba_", "chol_", "loop_", "mk_", "kin_", "momentum_", "emit_", "pack_", "big_", "jac_", "mom_", "energy_", "com_")


def short(name):
    n = name.split("(")[0]
    for pre in ("void rbd::", "rbd::", "void "):
        if n.startswith(pre):
            n = n[len(pre):]
    return n[:80]


def pmc(dirname):
    """kernel -> counter -> mean per dispatch (rows of one dispatch summed first)"""
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if len(files) > 1:  # gpurun MERGES into gpurun_out/: an earlier measurement left in place would be summed with this one
        sys.exit(f"{dirname}: {len(files)} counter files — remove gpurun_out/measure before running scripts/gpu_measure.sh again")
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith(OURS):
                per[k][r["Counter_Name"]][r.get("Dispatch_Id", "0")] += float(r["Counter_Value"])
    return {k: {c: (sum(v.values()) / len(v), len(v)) for c, v in d.items()} for k, d in per.items()}


LEGS = {"c2": (2, "dynamics", ""), "c2id": (2, "inverse_dynamics", ""), "c3": (3, "mass_matrix_solve", ""), "c3noM": (3, "mass_matrix_solve", "_noM"), "c3pk": (3, "mass_matrix_solve", "_packed"),
        "c4": (4, "dynamics", ""), "c5": (5, "dynamics", ""), "c2big": (2, "dynamics", "", 65536), "c2idb": (2, "inverse_dynamics", "_bodies", 65536),
        # round 6: the fixed-base Atlas, the RK4 `simulate` step (a leg runs 10 + 60 + 2 steps: warm-up, timed, parity) and the kinematics by-products (a leg
        # runs the set of three calls 2 x 70 times: together, then each on its own) — (config, op, suffix, batch, model, dtype, steps the leg runs)
        "axf": (2, "dynamics", "", 4096, "atlas_fixed"), "rmech": (2, "dynamics", "", 65536, "randmech1"),
        "sim64": (2, "simulate", "", 4096, "atlas_floating", "f64", 72), "sim64b": (2, "simulate", "", 65536, "atlas_floating", "f64", 72),
        "sim32": (2, "simulate", "", 65536, "atlas_floating", "f32", 72),
        "kin": (2, "kinematics", "", 65536, "atlas_floating", "f64", 140), "kin4k": (2, "kinematics", "", 4096, "atlas_floating", "f64", 140)}


def main():
    legs = sys.argv[1:] or list(LEGS)
    path = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")
    h = bench.kernel_source_hash()
    rec = {}
    if os.path.exists(path):
        old = json.load(open(path))
        if old.get("source_hash") == h:
            rec = old
    rec["source_hash"] = h
    rec["_note"] = ("HBM bytes per bench step (all kernels of one step) from rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE, collected in separate passes by "
                    "scripts/gpu_measure.sh; FETCH_SIZE (KiB) doubled for gfx950 as MI355X_MICROARCH.md (HBM) prescribes, which is exact for wide coalesced "
                    "reads and an upper bound for the 8-byte strided ones; WRITE_SIZE is uncalibrated.  Infinity-Cache hits are counted, so with inputs "
                    "re-read every step this is fabric traffic, an upper bound of DRAM traffic.")
    rec.setdefault("detail", {})
    for leg in legs:
        C, op, sfx = LEGS[leg][:3]
        cfg = bench.CONFIGS[C]
        L = LEGS[leg]
        key = f"{L[4] if len(L) > 4 else cfg['model']}_{L[5] if len(L) > 5 else cfg['dtype']}_B{L[3] if len(L) > 3 else cfg['batch']}_{op}{sfx}"
        leg_steps = L[6] if len(L) > 6 else None  # legs whose step is more than one launch of a kernel: launches per step = calls / steps
        # --- kernel stats
        rows = []
        for f in glob.glob(os.path.join(OUT, f"stats_{leg}", "**", "*kernel_stats.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        with open(os.path.join(ROOT, "profiles", f"r06_kernel_stats_{leg}.csv"), "w") as fo:
            fo.write(f"# leg {leg}: rocprofv3 --kernel-trace --stats -- python bench.py (scripts/gpu_measure.sh leg_args {leg}) --no-cpu-baseline --no-extra-legs --no-other-configs --steps 60 --warmup 10 ; sources {h}\n")
            fo.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
            for r in rows:
                fo.write(",".join(['"' + short(r["Name"]) + '"'] + [r[c] for c in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")]) + "\n")
        calls = {short(r["Name"]): int(r["Calls"]) for r in rows}
        avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in rows}
        # --- PMC
        fetch, write = pmc(os.path.join(OUT, f"pmc_fetch_{leg}")), pmc(os.path.join(OUT, f"pmc_write_{leg}"))
        sq = pmc(os.path.join(OUT, f"pmc_sq1_{leg}"))
        for k, d in pmc(os.path.join(OUT, f"pmc_sq2_{leg}")).items():
            sq.setdefault(k, {}).update(d)
        step_kernels = [k for k, n in calls.items() if k.startswith(OURS) and n >= 60]
        total, detail = 0.0, {}
        with open(os.path.join(ROOT, "profiles", f"r06_pmc_{leg}.txt"), "w") as fo:
            fo.write(f"# leg {leg} (config {C}): {key}; sources {h}; rocprofv3 --pmc (scripts/gpu_measure.sh); per launch, averaged over the launches of the run\n")
            for k in sorted(set(fetch) | set(write) | set(sq)):
                fr = fetch.get(k, {}).get("FETCH_SIZE", (0.0, 0))[0] * 1024
                wr = write.get(k, {}).get("WRITE_SIZE", (0.0, 0))[0] * 1024
                fo.write(f"{k}  calls={calls.get(k)} avg_ns={avg_ns.get(k)}\n")
                fo.write(f"  FETCH_SIZE {fr:14.0f} B raw, {2 * fr:14.0f} B with the gfx950 x2 correction; WRITE_SIZE {wr:14.0f} B\n")
                d = sq.get(k, {})
                w = d.get("SQ_WAVES", (0, 0))[0]
                for c in sorted(d):
                    fo.write(f"  {c:24s} {d[c][0]:16.0f}" + (f"   per wave {d[c][0] / w:12.1f}" if w else "") + "\n")
                if k in step_kernels:
                    mult = calls[k] / leg_steps if leg_steps else 1.0
                    fr, wr = fr * mult, wr * mult
                    total += 2 * fr + wr
                    detail[k] = {"fetch_bytes_raw": fr, "fetch_bytes_x2": 2 * fr, "write_bytes": wr, "avg_ns": avg_ns.get(k), "launches_per_step": mult,
                                 "valu_insts_per_wave": (d["SQ_INSTS_VALU"][0] / w) if w and "SQ_INSTS_VALU" in d else None}
        raw = sum(d["fetch_bytes_raw"] + d["write_bytes"] for d in detail.values())
        sig = {k: float(r["StdDev"]) / float(r["AverageNs"]) for r in rows for k in [short(r["Name"])] if k in step_kernels and float(r["AverageNs"]) > 0}
        rec[key] = {"bytes_per_step_fetch_x2": total, "bytes_per_step_raw": raw, "fetch_raw_bytes": sum(d["fetch_bytes_raw"] for d in detail.values()),
                    "kernels": sorted(detail), "stddev_over_mean": sig} if detail else None
        rec["detail"][key] = detail
        print(f"leg {leg} ({key}): step kernels {step_kernels}; traffic {total:.0f} B/step with FETCH x2, {raw:.0f} raw; sigma/mean {sig}")
    json.dump(rec, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
