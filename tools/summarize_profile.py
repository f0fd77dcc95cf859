"""Turns the rocprofv3 output of tools/profile_bench.sh (gpurun_out/prof_<tag>/) into the
committed summaries under profiles/:
   <tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats, per-kernel durations
   <tag>_pmc_summary.json    HBM traffic per launch of the dominant kernel

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and
WRITE_SIZE are collected in separate passes (TCC slot budget), both are in KiB, and on
gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read, so the read side is
doubled ("fetch_bytes_corrected"); WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
KERNEL = "naive_step_kernel"


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    assert hits, pattern
    # (gpurun merges every call's files into gpurun_out/, and rocprofv3 names them by process id: the NEWEST is this
    # collection's -- hits[0] once copied a stale summary whose kernel signatures no longer existed)
    return max(hits, key=os.path.getmtime)


stats = one("trace/**/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
avg_ns = calls = None
for row in csv.DictReader(open(stats)):
    if KERNEL in row["Name"]:
        avg_ns, calls = float(row["AverageNs"]), int(row["Calls"])
        kname = row["Name"].split("(HIP_vector")[0]


def counter(pattern, name):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(one(pattern)))
            if r["Counter_Name"] == name and KERNEL in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


fetch_kib, nf = counter("pmc_fetch/**/*_counter_collection.csv", "FETCH_SIZE")
write_kib, nw = counter("pmc_write/**/*_counter_collection.csv", "WRITE_SIZE")
fetch_b, write_b = fetch_kib * 1024.0, write_kib * 1024.0
algorithmic = 80.0 * n  # read 40 + write 40 bytes per body per step
out = {
    "tag": tag, "n": n, "kernel": kname, "calls": calls, "avg_duration_ns": avg_ns,
    "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
    "fetch_bytes_corrected": 2.0 * fetch_b, "write_bytes": write_b,
    "hbm_bytes_per_launch": 2.0 * fetch_b + write_b,
    "algorithmic_hbm_bytes_per_launch": algorithmic,
    "launches_counted": [nf, nw],
    "note": "FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); "
            "the all-pairs kernel is VALU-bound: HBM traffic is reported, it is not the bound",
}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
