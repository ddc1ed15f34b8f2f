#!/usr/bin/env python3
"""Condense tools/run_profiles_aux.sh output into profiles/<tag>_aux_kernel_stats.csv (rocprofv3 --kernel-trace --stats rows
of the library's own kernels, verbatim) and profiles/<tag>_aux.json (per workload: HIP-event time of the run, algorithmic
bytes, achieved GB/s and fraction of the 8 TB/s HBM roofline, PMC HBM traffic).

  python profiles/summarize_aux.py r02

HBM bytes follow MI355X_MICROARCH.md: counter x 1024, FETCH_SIZE doubled on gfx950 (half-count of coalesced streaming
reads), WRITE_SIZE as is; separate --pmc passes.  In the PMC passes bench_aux.py runs every workload twice (one warm-up,
one "timed" launch, REPS=1): dispatch k of a kernel belongs to workload k // 2 of that kernel, in program order."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
OURS = ("burst_pull", "pack_trxd", "va_demod", "channelize_kernel", "resample_kernel", "convolve_kernel", "convert_short_float",
        "delay_vector", "energy_detect", "vector_slicer", "sch_detect", "save_")
PEAK = 8000.0


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0]


def main(tag):
    stats_in = os.path.join(G, "prof_aux_stats", f"{tag}_aux_kernel_stats.csv")
    rows = [r for r in csv.DictReader(open(stats_in)) if any(k in r["Name"] for k in OURS)]
    with open(os.path.join(ROOT, "profiles", f"{tag}_aux_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_ALL)
        w.writeheader()
        w.writerows(rows)
    stats = {short(r["Name"]): {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": int(r["MinNs"]),
                                "max_ns": int(r["MaxNs"])} for r in rows}

    def pmc(kind, cname):
        per = {}
        path = os.path.join(G, f"prof_aux_{kind}", f"{tag}_aux_counter_collection.csv")
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == cname:
                per.setdefault(short(r["Kernel_Name"]), []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        return {k: [v for _, v in sorted(vs)] for k, vs in per.items()}
    fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")

    out, seen = [], {}
    for line in open(os.path.join(G, f"{tag}_aux.jsonl")):
        if not line.startswith("{"):
            continue
        j = json.loads(line)
        kerns = j["kernel"].split("+")
        gbs = j["algo_bytes"] / (j["event_ms"] * 1e-3) / 1e9
        e = dict(j)
        e["algo_GBps"] = round(gbs, 1)
        e["hbm_roofline_frac"] = round(gbs / PEAK, 4)
        e["per_unit"] = f'{j["units"] / j["event_ms"] / 1e3:.1f} M{j["unit"]}/s'
        rd = wr = 0.0
        ok = True
        for k in kerns:
            i = seen.get(k, 0)
            seen[k] = i + 1
            try:
                rd += 2.0 * 1024.0 * fetch[k][2 * i + 1]
                wr += 1024.0 * write[k][2 * i + 1]
            except (KeyError, IndexError):
                ok = False
        if ok:
            e["pmc_hbm_read_bytes"] = rd
            e["pmc_hbm_write_bytes"] = wr
            e["pmc_over_algorithmic"] = round((rd + wr) / j["algo_bytes"], 3)
        e["kernel_trace"] = {k: stats.get(k) for k in kerns}
        out.append(e)
    json.dump({"tag": tag, "peak_GBps": PEAK, "workloads": out,
               "note": "event_ms = HIP-event average inside the rocprofv3 --kernel-trace run; kernel_trace = that kernel's row of "
                       "the stats file (all its launches in the program, several workloads for burst_pull4_kernel<false, false>)"},
              open(os.path.join(ROOT, "profiles", f"{tag}_aux.json"), "w"), indent=1)
    print(f'{"kernel":36s} {"workload":52s} {"ms":>8s} {"rate":>18s} {"GB/s":>7s} {"frac":>6s} {"pmc/algo":>8s}')
    for e in out:
        print(f'{e["kernel"][:36]:36s} {e["what"][:52]:52s} {e["event_ms"]:8.3f} {e["per_unit"]:>18s} {e["algo_GBps"]:7.0f} '
              f'{e["hbm_roofline_frac"]:6.3f} {e.get("pmc_over_algorithmic", float("nan")):8.2f}')


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")
