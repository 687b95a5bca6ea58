#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace --stats, and --pmc passes) into a small text summary
that is committed under profiles/.  Usage: summarize_profile.py <dir> [--json out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name
    for a, b in (("rt::", ""), ("void ", "")):
        n = n.replace(a, b)
    return n[:70]


def kernel_stats(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append(r)
    return rows


def kernel_trace(d):
    per = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per


def pmc(d):
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    base = sys.argv[1]
    res = {}
    print("== rocprofv3 --kernel-trace --stats (%s/trace) ==" % base)
    st = kernel_stats(os.path.join(base, "trace"))
    for r in sorted(st, key=lambda r: -float(r.get("TotalDurationNs", 0) or 0)):
        print("%-72s calls %6s  total %10.1f us  avg %9.2f us  %5s%%" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                                       float(r["AverageNs"]) / 1e3, r.get("Percentage", "")))
    tr = kernel_trace(os.path.join(base, "trace"))
    res["avg_us"] = {short(k): sum(v) / len(v) for k, v in tr.items()}
    # the FULL-SIZE launches of the two traversal kernels: what profiles/latest_profile.json carries and bench.py's roofline.rocprof uses
    # (a context's first frame also launches the same instantiations on its small later bounces; those are left out: < half the longest)
    print("== full-size launches of the traversal kernels (the averages latest_profile.json carries) ==")
    for k, v in sorted(tr.items(), key=lambda kv: -sum(kv[1])):
        if "k_trace<" not in k and "k_tile<" not in k and "k_beam" not in k:
            continue
        durs = sorted(v)
        big = [x for x in durs if x >= 0.5 * durs[-1]]
        print("%-72s full-size launches %4d of %4d  avg %9.2f us  (all launches: avg %9.2f us)" % (short(k), len(big), len(durs), sum(big) / len(big), sum(durs) / len(durs)))
    for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_l2", None)):
        p = pmc(os.path.join(base, tag))
        if not p:
            continue
        print("== PMC pass %s ==" % tag)
        for k, cs in p.items():
            line = "%-72s" % short(k)
            for c, v in cs.items():
                # a context's first frame also launches the traversal kernels on its small later bounces: the per-launch figure
                # is the mean over the full-size launches (>= half the largest value)
                big = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v
                line += "  %s avg %.4g (n=%d of %d launches)" % (c, sum(big) / len(big), len(big), len(v))
                res.setdefault("pmc", {}).setdefault(short(k), {})[c] = sum(big) / len(big)
            print(line)
    # HBM traffic per launch of the closest-hit traversal kernel (gfx950 correction of
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE is in KiB of 64-B requests and reads HALF the bytes of a wide
    # coalesced stream; WRITE_SIZE is exact) — both bounds are reported.
    pm = res.get("pmc", {})

    def is_closest(k):
        return "k_trace<0, false, false" in k or "k_trace4<0, false, false" in k or "k_packet<0, false, false" in k or "k_beam(" in k or k.endswith("k_beam")

    def is_shadow(k):
        return "k_trace<1, true, false" in k or "k_trace4<1, true, false" in k or "k_packet<1, true, false" in k or "k_beam_shadow" in k

    def traffic(pred, label, key):
        # the instantiation that does the work (entry records: k_trace<..., true>); a context's first frame also launches the plain one on its small later bounces
        cands = [k for k in pm if pred(k) and "FETCH_SIZE" in pm[k]]
        if not cands:
            return
        k = max(cands, key=lambda c: pm[c]["FETCH_SIZE"])
        fetch_kib = pm[k]["FETCH_SIZE"]
        write_kib = pm[k].get("WRITE_SIZE", 0.0)
        lo = (fetch_kib + write_kib) * 1024.0
        hi = (2.0 * fetch_kib + write_kib) * 1024.0
        res[key + "_uncorrected"] = lo
        res[key] = hi
        print("%s (%s): FETCH_SIZE %.1f KiB, WRITE_SIZE %.1f KiB per launch -> HBM bytes/launch %.3e (x2 read correction) / %.3e (raw)" % (label, k, fetch_kib, write_kib, hi, lo))
    traffic(is_closest, "closest-hit traversal", "hbm_bytes_per_launch")
    traffic(is_shadow, "shadow traversal", "hbm_bytes_per_launch_shadow")
    if "--json" in sys.argv:
        json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    # profiles/latest_profile.json: what bench.py attaches to its roofline block — only when its tag matches the run
    if "--latest" in sys.argv:
        out = sys.argv[sys.argv.index("--latest") + 1]
        tag = json.loads(os.environ.get("RT_PROFILE_TAG", "{}"))
        def big_launches(pred):
            cands = [sorted(v) for k, v in tr.items() if pred(k)]
            if not cands:
                return []
            durs = max(cands, key=lambda d: sum(d))     # the instantiation that does the work
            return [x for x in durs if x >= 0.5 * durs[-1]]   # its full-size launches (a context's first frame also launches small later bounces)
        big = big_launches(lambda k: is_closest(short(k)) or is_closest(k))
        big_sh = big_launches(lambda k: is_shadow(short(k)) or is_shadow(k))
        latest = {"tag": tag, "source": os.environ.get("RT_PROFILE_SOURCE", base),
                  "k_trace_closest_avg_ms": (sum(big) / len(big) / 1e3) if big else None, "k_trace_closest_launches": len(big),
                  "k_trace_shadow_avg_ms": (sum(big_sh) / len(big_sh) / 1e3) if big_sh else None, "k_trace_shadow_launches": len(big_sh),
                  "hbm_bytes_per_launch_shadow": res.get("hbm_bytes_per_launch_shadow"),
                  "hbm_bytes_per_launch": res.get("hbm_bytes_per_launch"), "hbm_bytes_per_launch_uncorrected": res.get("hbm_bytes_per_launch_uncorrected"),
                  "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernels serialised by the profiler), read side doubled per MI355X_MICROARCH.md (HBM)",
                  "limiter": os.environ.get("RT_PROFILE_LIMITER")}
        json.dump(latest, open(out, "w"), indent=1)
        print("wrote", out, json.dumps(latest)[:300])


if __name__ == "__main__":
    main()
