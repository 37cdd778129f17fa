#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/{kt,fetch,write,tcc}) into the small,
tracked summaries under profiles/ and the per-launch HBM traffic bench.py reports.

  kt     rocprofv3 --kernel-trace --stats      -> profiles/<tag>_kernel_stats.csv
  fetch  rocprofv3 --pmc FETCH_SIZE            -> per-dispatch KB fetched by L2 from the fabric
  write  rocprofv3 --pmc WRITE_SIZE
  tcc    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts
128-byte requests as 64 bytes for wide coalesced streams -> doubled; WRITE_SIZE is exact.
Units: KB per dispatch.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.environ.get("MF_PROFILE_SRC") or os.path.join(ROOT, "gpurun_out", "prof")
fuse_kernel = os.environ.get("MF_PROFILE_KERNEL", "mf::fuse_tiles_kernel")   # the tile kernel that did the work
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
workload_key = sys.argv[2] if len(sys.argv) > 2 else "distA_sequential_b64"
out_dir = os.environ.get("MF_PROFILE_OUT") or os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)


def one(pattern):
    """Newest file matching <subdir>/**/<name> (rocprofv3 nests its output under host / pid directories)."""
    sub, name = pattern.split("/", 1)
    name = name.split("/")[-1]
    files = sorted(glob.glob(os.path.join(src, sub, "**", name), recursive=True), key=os.path.getmtime)
    if not files:
        raise SystemExit(f"missing {pattern} under {src}")
    return files[-1]


# ---- kernel stats -----------------------------------------------------------------------------
rows = list(csv.DictReader(open(one("kt/runc/*_kernel_stats.csv"))))
keep = [r for r in rows if "mf::" in r["Name"]]
others = sorted((r for r in rows if "mf::" not in r["Name"]), key=lambda r: -float(r["TotalDurationNs"]))[:5]
with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in keep + others:
        r = dict(r)
        r["Name"] = r["Name"][:120]
        w.writerow(r)


try:
    srows = list(csv.DictReader(open(one("kt_single/runc/*_kernel_stats.csv"))))
    with open(os.path.join(out_dir, f"{tag}_single_frame_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(srows[0].keys()))
        w.writeheader()
        for r in srows:
            if "mf::" in r["Name"]:
                r = dict(r); r["Name"] = r["Name"][:120]; w.writerow(r)
except SystemExit:
    pass


def counter_means(path, counters):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "mf::" in r["Kernel_Name"] and r["Counter_Name"] in counters:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


fetch = counter_means(one("fetch/runc/*_counter_collection.csv"), {"FETCH_SIZE"})
write = counter_means(one("write/runc/*_counter_collection.csv"), {"WRITE_SIZE"})
try:
    tcc = counter_means(one("tcc/runc/*_counter_collection.csv"), {"TCC_HIT_sum", "TCC_MISS_sum"})
except SystemExit:
    tcc = {}

summary = {}
total = 0.0
for k in sorted(set(fetch) | set(write)):
    f_kb = fetch.get(k, {}).get("FETCH_SIZE", 0.0)
    w_kb = write.get(k, {}).get("WRITE_SIZE", 0.0)
    hbm = (2.0 * f_kb + w_kb) * 1024.0
    e = dict(FETCH_SIZE_KB_raw=f_kb, WRITE_SIZE_KB=w_kb, hbm_bytes_per_launch=hbm,
             note="hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B requests as 64 B)")
    if k in tcc and tcc[k].get("TCC_HIT_sum") is not None:
        h, m = tcc[k].get("TCC_HIT_sum", 0.0), tcc[k].get("TCC_MISS_sum", 0.0)
        e["L2_hit_rate"] = h / (h + m) if h + m else None
    summary[k] = e
    # (the step's total: the tile kernel that did the work, not the other instantiations that ran in the untimed warm-up)
    if "unproject_bin" not in k and not (k.startswith("mf::fuse_") and not k.startswith(fuse_kernel)):
        total += hbm
summary["_pipeline_total_hbm_bytes_per_step"] = total
with open(os.path.join(out_dir, f"{tag}_hbm_counters.json"), "w") as f:
    json.dump(summary, f, indent=1)

sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel source hash: bench.py refuses traffic measured on other sources)

tfile = os.path.join(out_dir, "traffic.json")
traffic = json.load(open(tfile)) if os.path.exists(tfile) else {}
if traffic.get("kernel_sources_sha16") != bench.sources_sha():
    traffic = {}                       # measured on other kernel sources: start over
fuse = [v for k, v in summary.items() if k.startswith(fuse_kernel)]
traffic["kernel_sources_sha16"] = bench.sources_sha()
traffic[workload_key] = fuse[0]["hbm_bytes_per_launch"] if fuse else None
traffic[workload_key + "_pipeline"] = total
traffic[workload_key + "_source"] = traffic["_source"] = (f"profiles/{tag}_hbm_counters.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                      "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, the gfx950 FETCH_SIZE correction of "
                      "/opt/skills/guides/MI355X_MICROARCH.md)")
with open(tfile, "w") as f:
    json.dump(traffic, f, indent=1)
print(json.dumps(summary, indent=1)[:3000])
