#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_*) into the small files committed under profiles/.

  python profiles/summarize.py <tag>      e.g. r01a
Reads  gpurun_out/prof_stats/<tag>_kernel_stats.csv        (rocprofv3 --kernel-trace --stats)
       gpurun_out/prof_fetch/<tag>_counter_collection.csv  (rocprofv3 --pmc FETCH_SIZE)
       gpurun_out/prof_write/<tag>_counter_collection.csv  (rocprofv3 --pmc WRITE_SIZE)
Writes profiles/<tag>_kernel_stats.csv (verbatim), profiles/<tag>_pmc.json, and profiles/pmc_traffic.json
(the per-burst HBM traffic bench.py reports in roofline.traffic).

HBM bytes follow MI355X_MICROARCH.md "HBM": bytes = counter * 1024; on gfx950 FETCH_SIZE reports half of
the bytes of a coalesced streaming read, so the read side is doubled (checked here against the known
2500 B/burst input: 2 x FETCH = 1.02 x algorithmic); WRITE_SIZE is used as is (1.02 x algorithmic).
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
sys.path.insert(0, ROOT)
from osmo_trx_amd.srchash import hot_kernel_source_hash   # noqa: E402

KERNEL = "nb_pull4_kernel"                                 # the hot kernel (round 6: the normal-burst kernel)
LIST_KERNEL = "burst_pull4_kernel<false, false, true, true>"   # the general kernel over the list it leaves behind


def counter(path, name):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    return vals


def main(tag, bursts=1 << 20):
    shutil.copyfile(os.path.join(G, "prof_stats", f"{tag}_kernel_stats.csv"),
                    os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    stats, list_stats = {}, {}
    with open(os.path.join(G, "prof_stats", f"{tag}_kernel_stats.csv")) as f:
        for row in csv.DictReader(f):
            if KERNEL in row["Name"]:
                stats = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "min_ns": int(row["MinNs"]),
                         "max_ns": int(row["MaxNs"])}
            if LIST_KERNEL in row["Name"]:
                list_stats = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "min_ns": int(row["MinNs"]),
                              "max_ns": int(row["MaxNs"])}
    fetch = counter(os.path.join(G, "prof_fetch", f"{tag}_counter_collection.csv"), "FETCH_SIZE")
    write = counter(os.path.join(G, "prof_write", f"{tag}_counter_collection.csv"), "WRITE_SIZE")
    fetch_b = 2.0 * 1024.0 * sum(fetch) / len(fetch)        # gfx950 half-count correction
    write_b = 1024.0 * sum(write) / len(write)
    out = {
        "tag": tag, "kernel": KERNEL, "bursts_per_launch": bursts, "kernel_trace": stats, "list_kernel_trace": list_stats,
        "source_hash": hot_kernel_source_hash(),
        "FETCH_SIZE_KB_raw_avg": sum(fetch) / len(fetch), "WRITE_SIZE_KB_avg": sum(write) / len(write),
        "hbm_read_bytes_per_launch": fetch_b, "hbm_write_bytes_per_launch": write_b,
        "hbm_bytes_per_burst": (fetch_b + write_b) / bursts,
        "algorithmic_bytes_per_burst": 3132,
        "note": "read side = 2 x FETCH_SIZE x 1024 (gfx950 half-count, MI355X_MICROARCH.md HBM section); "
                "separate --pmc passes for FETCH_SIZE and WRITE_SIZE",
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
    json.dump({"tag": tag, "source_hash": out["source_hash"], "hbm_bytes_per_burst": out["hbm_bytes_per_burst"]},
              open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"))
    print(json.dumps(out, indent=1))


def sq_counters(tag, bursts=1 << 20, waves_per_simd=4, waves_per_cu=16):
    """profiles/<tag>_sq_counters.json from the three SQ passes of tools/run_profiles.sh (per-burst averages)."""
    per = {}
    for i in (1, 2, 3):
        path = os.path.join(G, f"prof_sq{i}", f"{tag}_counter_collection.csv")
        if not os.path.exists(path):
            continue
        acc = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                if KERNEL in row["Kernel_Name"]:
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in acc.items():
            per[k] = round(sum(v) / len(v) / bursts, 2)
    if not per:
        return
    d = {}
    n_simd, n_cu, n_xcd = 1024, 256, 8
    if "SQ_WAVE_CYCLES" in per:
        d["wave_cycles_per_burst"] = round(4 * per["SQ_WAVE_CYCLES"], 1)              # quad-cycles -> cycles
    if "GRBM_GUI_ACTIVE" in per and "SQ_WAVE_CYCLES" in per:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: kernel duration in shader cycles = value / 8 (and / wall time = the
        # effective clock, MI355X_MICROARCH.md "DVFS give-back")
        kcyc = per["GRBM_GUI_ACTIVE"] * bursts / n_xcd
        d["kernel_cycles"] = round(kcyc)
        d["mean_resident_waves_per_simd"] = round(4 * per["SQ_WAVE_CYCLES"] * bursts / (kcyc * n_simd), 2)
        if "SQ_ACTIVE_INST_VALU" in per:
            d["valu_busy"] = round(4 * per["SQ_ACTIVE_INST_VALU"] * bursts / (kcyc * n_simd), 3)      # of every SIMD, whole kernel
        if "SQ_ACTIVE_INST_SCA" in per:
            d["salu_busy"] = round(4 * per["SQ_ACTIVE_INST_SCA"] * bursts / (kcyc * n_simd), 3)
        if "SQ_LDS_IDX_ACTIVE" in per:
            d["lds_busy"] = round(per["SQ_LDS_IDX_ACTIVE"] * bursts / (kcyc * n_cu), 3)                 # cycles, per CU
    if "SQ_WAVE_CYCLES" in per:
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if k in per:
                d["wave_time_share_" + k[3:].lower()] = round(per[k] / per["SQ_WAVE_CYCLES"], 3)
    out = {"tag": tag, "kernel": KERNEL, "bursts_per_launch": bursts, "per_burst": per, "derived": d,
           "note": "rocprofv3 --pmc, three separate passes (tools/run_profiles.sh), bench.py --main-only --steps 2; "
                   "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles (MI355X_MICROARCH.md); busy fractions are over the "
                   "kernel's duration on all 1024 SIMDs / 256 CUs (GRBM_GUI_ACTIVE / 8 XCDs), so they include the time a SIMD holds "
                   "fewer than its 4 waves (mean_resident_waves_per_simd)"}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters.json"), "w"), indent=1)
    print(json.dumps(d, indent=1))
    # what bench.py quotes as roofline.compute (the vector-issue bound of the kernel), next to the HBM traffic
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tp) and "valu_busy" in d:
        t = json.load(open(tp))
        t["compute"] = {"valu_busy": d.get("valu_busy"), "salu_busy": d.get("salu_busy"), "lds_busy": d.get("lds_busy"),
                        "valu_insts_per_burst": per.get("SQ_INSTS_VALU"), "salu_insts_per_burst": per.get("SQ_INSTS_SALU"),
                        "lds_insts_per_burst": per.get("SQ_INSTS_LDS"),
                        "mean_resident_waves_per_simd": d.get("mean_resident_waves_per_simd")}
        json.dump(t, open(tp, "w"))


if __name__ == "__main__":
    main(sys.argv[1])
    sq_counters(sys.argv[1])
    bj = os.path.join(G, f"{sys.argv[1]}_bench.json")
    if os.path.exists(bj):
        shutil.copyfile(bj, os.path.join(ROOT, "profiles", f"{sys.argv[1]}_bench.json"))
