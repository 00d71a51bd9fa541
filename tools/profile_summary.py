"""Condense the rocprofv3 output of tools/profile.sh into the files that are committed under profiles/:
pmc_summary.json (counter sums per gtx kernel), pmc_traffic.json (HBM bytes per launch, read by bench.py for
roofline.traffic), kernel_stats.csv (rocprofv3's own --stats table) and kernel_stats_gtx.csv (the same rows for this
library's kernels and the rocPRIM sorts, names shortened)."""
import collections
import csv
import glob
import json
import shutil
import sys

out, reads = sys.argv[1], int(sys.argv[2])


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    return name.split("(")[0].split("<")[0]


summary = collections.defaultdict(dict)
launches = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = short(row.get("Kernel_Name", "?"))
        per[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
    for k, d in per.items():
        if k.startswith("gtx::"):
            summary[k].update(d)
            for c in d:
                launches[k][c] = cnt[(k, c)]
old = out + "/pmc_summary.json"
try:  # passes whose CSVs were too large to bring back were summarised on the GPU box
    for k, d in json.load(open(old)).items():
        if isinstance(d, dict):
            for c, v in d.items():
                summary[k].setdefault(c, v)
except (OSError, ValueError):
    pass
summary["_note"] = ("sums over the launches of one bench.py run (--reads %d, 1 step, no warm-up) per PMC pass; tools/profile.sh" % reads)
json.dump(summary, open(old, "w"), indent=1, sort_keys=True)

ks = sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True))
if ks:
    shutil.copy(ks[0], out + "/kernel_stats.csv")
    rows = list(csv.DictReader(open(ks[0])))
    with open(out + "/kernel_stats_gtx.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for r in rows:
            n = short(r["Name"])
            if n.startswith("gtx::") or "rocprim" in n:
                w.writerow([n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])

kernels = {}
for k, d in summary.items():
    if isinstance(d, dict) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        n = max(1, launches.get(k, {}).get("FETCH_SIZE", 1))
        # FETCH_SIZE / WRITE_SIZE count kilobytes; MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half of the bytes
        # read, so it is doubled; WRITE_SIZE is taken as reported.  The uncorrected sum is kept beside it.
        b = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / n
        kernels[k.replace("gtx::", "")] = {"launches": n, "fetch_size_kb": d["FETCH_SIZE"] / n, "write_size_kb": d["WRITE_SIZE"] / n,
                                           "hbm_bytes_per_launch": b,
                                           "hbm_bytes_per_launch_uncorrected": (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / n,
                                           "bytes_per_read_of_the_batch": b / reads}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile.sh) over one bench.py step of %d reads (cfg2, "
                   "no warm-up); per launch" % reads, "reads_per_launch": reads, "kernels": kernels}, open(out + "/pmc_traffic.json", "w"), indent=1)
print(open(out + "/kernel_stats_gtx.csv").read() if ks else "no kernel stats")
for k, v in kernels.items():
    print("%-40s launches=%d bytes/launch=%.4g (%.1f B per read of the batch)" % (k, v["launches"], v["hbm_bytes_per_launch"], v["bytes_per_read_of_the_batch"]))
