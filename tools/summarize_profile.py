#!/usr/bin/env python3
"""Turn gpurun_out/prof_TAG (tools/profile_round.sh) into the committed evidence under profiles/:
   rNN_TAG_bench.json, rNN_TAG_kernel_stats.csv, rNN_TAG_pmc_fetch_write.csv and pmc_summary.json.
Usage: tools/summarize_profile.py TAG ROUND_PREFIX   (e.g. c r01_c)"""
import csv, glob, json, os, sys, shutil, collections

tag, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, prefix + "_bench.json"), "w"), indent=1)
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, prefix + "_kernel_stats.csv"))


def per_kernel(dirname, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(src, dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


fetch, write = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE")
names = sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1]))
with open(os.path.join(dst, prefix + "_pmc_fetch_write.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_raw_KB_sum", "WRITE_SIZE_KB_sum"])
    for k in names[:60]:
        w.writerow([k[:120], fetch.get(k, write.get(k))[0], round(fetch.get(k, [0, 0])[1]), round(write.get(k, [0, 0])[1])])
dom = bench["roofline"]["kernel"]
key = dom.split("<")[0]
want_u64 = "u64" in dom


def match(name):
    if "rs_scatter" in key:                      # both scatter kernels (direct and LDS-reordered) belong to the class
        return "rs_scatter" in name and ("<unsigned long" in name) == want_u64
    return key in name


fl = [(k, v) for k, v in fetch.items() if match(k)]
wl = [(k, v) for k, v in write.items() if match(k)]
if fl and wl:
    launches = sum(v[0] for _, v in fl)
    fraw = sum(v[1] for _, v in fl) * 1024           # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
    wbytes = sum(v[1] for _, v in wl) * 1024
    hbm = (2 * fraw + wbytes) / launches
    summ = {"round": prefix, "workload_bytes": bench["config"]["bytes_per_gpu"], "kernel": dom, "launches": launches,
            "FETCH_SIZE_bytes_raw": fraw,
            "FETCH_correction": "x2: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for coalesced streaming reads (MI355X_MICROARCH.md, HBM)",
            "WRITE_SIZE_bytes": wbytes, "hbm_bytes_per_launch": round(hbm),
            "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
            "amplification": round(hbm / bench["roofline"]["algorithmic_bytes_per_launch"], 3),
            "collection": "two separate passes: rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace, python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (tools/profile_round.sh)"}
    json.dump(summ, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    print(json.dumps(summ, indent=1))
print("value", bench["value"], "dominant", dom, bench["roofline"]["frac"])
