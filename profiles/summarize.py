#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (gpurun_out/prof/<pass>/...) into profiles/<round>/ and refresh
profiles/measured.json -- the recorded per-launch figures (fabric traffic, L2 hit rate, MFMA utilisation) that
bench.py quotes, WITH their source file and the hash of the kernel sources they were measured on, for the sweep
entries and as the fallback when its own live counter passes are unavailable.

  python profiles/summarize.py gpurun_out/prof_round/reddit_d128 profiles/r01 reddit_d128

Passes expected under the input dir: trace (--kernel-trace --stats), fetch (--pmc FETCH_SIZE),
write (--pmc WRITE_SIZE), l2 (--pmc TCC_HIT_sum TCC_MISS_sum), mfma (--pmc SQ_VALU_MFMA_BUSY_CYCLES
GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES) -- each its own run (tools/profile_round.sh), as
MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
Corrections applied (same guide, section HBM): counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact for 16-B stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def counters(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    src, dst, key = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    out = {"key": key, "kernels": {}}
    for f in glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, "%s_kernel_stats.csv" % key))
        for r in csv.DictReader(open(f)):
            if "hcspmm" in r["Name"]:
                out["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                             "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
    pmc = {}
    for p in ("fetch", "write", "l2", "mfma"):
        for (k, c), v in counters(os.path.join(src, p)).items():
            if "hcspmm" in k:
                pmc.setdefault(k, {})[c] = v
    out["pmc_per_launch"] = pmc
    main_k = [k for k in pmc if "hybrid" in k]
    if main_k:
        m = pmc[main_k[0]]
        # a step = the hybrid launch + (low-degree graphs) the tiny-task launch + the fix-up: traffic and L2 requests are
        # summed over all of them (as bench.py's live passes do); MFMA counters are the hybrid kernel's
        step = [v for k, v in pmc.items() if "hcspmm::" in k]
        fetch_b = 2.0 * sum(v.get("FETCH_SIZE", 0.0) for v in step) * 1024.0
        write_b = sum(v.get("WRITE_SIZE", 0.0) for v in step) * 1024.0
        m = dict(m, TCC_HIT_sum=sum(v.get("TCC_HIT_sum", 0.0) for v in step), TCC_MISS_sum=sum(v.get("TCC_MISS_sum", 0.0) for v in step)) \
            if "TCC_HIT_sum" in m else m
        out["traffic_bytes_per_launch"] = fetch_b + write_b
        out["fetch_bytes_corrected"] = fetch_b
        out["write_bytes"] = write_b
        if "TCC_HIT_sum" in m:
            out["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
            # rocprofv3's MfmaUtil expression: sum over SIMDs of MFMA-busy cycles / (GPU-active cycles * SIMDs);
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note) -> divide by 8
            simds = 256 * 4
            active = m["GRBM_GUI_ACTIVE"] / 8.0
            out["mfma_util_percent"] = 100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (active * simds)
            out["mfma_flops_per_launch"] = m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, os.path.dirname(here))
        import bench  # kernel_src_sha(): which sources these numbers belong to
        tpath = os.path.join(here, "measured.json")
        t = json.load(open(tpath)) if os.path.exists(tpath) else {}
        kern = [v for k, v in out["kernels"].items() if "hybrid" in k]
        step_ms = sum(v["avg_ns"] for k, v in out["kernels"].items() if "hcspmm::" in k) / 1e6
        t[key] = {"traffic_bytes": fetch_b + write_b, "fetch_bytes": fetch_b, "write_bytes": write_b,
                  "l2_hit_rate": out.get("l2_hit_rate"), "mfma_util_percent": out.get("mfma_util_percent"),
                  "mfma_flops_per_launch": out.get("mfma_flops_per_launch"),
                  "kernel_ms": (kern[0]["avg_ns"] / 1e6) if kern else None, "step_kernels_ms": step_ms,
                  "source": os.path.relpath(os.path.join(dst, "%s_summary.json" % key), os.path.dirname(here)),
                  "kernel_src_sha": bench.kernel_src_sha()}
        json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
    json.dump(out, open(os.path.join(dst, "%s_summary.json" % key), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
