#!/usr/bin/env python
"""Turn rocprofv3 CSV output (gpurun_out/..., scratch) into the small summaries kept under profiles/ (tracked).

    python profiles/parse_rocprof.py stats   <kernel_stats.csv>  <out.csv>  [steps]
    python profiles/parse_rocprof.py traffic <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
    python profiles/parse_rocprof.py pmc     <out.json> <counter_collection.csv> [<counter_collection.csv> ...]

`traffic` follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from separate --pmc passes, both are in
KiB, and on gfx950 FETCH_SIZE under-reports wide coalesced reads by exactly 2x, so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.
"""
import collections
import csv
import json
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()


def stats(src, dst, steps=None):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"] + (["ms_per_step"] if steps else []))
        for r in rows:
            tot = float(r["TotalDurationNs"]) / 1e6
            line = [short(r["Name"]), r["Calls"], f"{tot:.3f}", f"{float(r['AverageNs']) / 1e3:.2f}", r["Percentage"]]
            if steps:
                line.append(f"{tot / float(steps):.3f}")
            w.writerow(line)


def traffic(fetch_csv, write_csv, dst):
    def per_kernel(path, counter):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        return agg
    fe, wr = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for k in fe:
        if k not in wr:
            continue
        f = sum(fe[k]) / len(fe[k])
        w = sum(wr[k]) / len(wr[k])
        out[k] = {"launches_sampled": len(fe[k]), "fetch_size_kib_raw": round(f, 1), "write_size_kib": round(w, 1),
                  "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                         "(gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md); average over the launches of the profiled command",
               "kernels": out}, open(dst, "w"), indent=1)


def pmc(dst, *srcs):
    """per-kernel averages of every counter of one or more `--pmc` passes (counter_collection.csv): {kernel: {counter: mean per launch}}.
    rocprofv3 sums a counter over its hardware instances (SQ: all SEs/XCDs; GRBM_GUI_ACTIVE: the 8 XCDs)."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in srcs:
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(path)):
            key = (path, r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[(path, r["Dispatch_Id"])] = short(r["Kernel_Name"])
            per_dispatch[(path, r["Dispatch_Id"], "duration_us")] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
        for (pth, disp, ctr), v in per_dispatch.items():
            agg[names[(pth, disp)]][ctr].append(v)
    out = {k: dict(launches=max(len(v) for v in c.values()), **{ctr: round(sum(v) / len(v), 1) for ctr, v in sorted(c.items())})
           for k, c in agg.items()}
    for k, c in out.items():      # derived: effective clock (MI355X_MICROARCH.md, DVFS give-back) and MFMA-pipe busy share of all SIMD cycles
        if c.get("GRBM_GUI_ACTIVE") and c.get("duration_us"):
            c["derived_clock_mhz"] = round(c["GRBM_GUI_ACTIVE"] / 8 / c["duration_us"], 1)
            if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                c["derived_mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    json.dump({"method": "rocprofv3 --pmc <counters> (own passes, no tracing); per-launch mean of each counter summed over its hardware instances",
               "kernels": out}, open(dst, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    elif sys.argv[1] == "pmc":
        pmc(sys.argv[2], *sys.argv[3:])
    else:
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
