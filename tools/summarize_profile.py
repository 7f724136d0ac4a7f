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
# HBM bytes per launch for EVERY kernel class of bench.py's table, so that roofline.traffic stays filled whichever class is dominant
# in the driver's run.  A class is what one ProfScope / prof_begin in the library covers (it may launch several kernels); its launches
# per step come from the bench line, the number of steps of the PMC run from the window kernel (one launch per step).
import re
CLASS_PATTERNS = {
    "window_levels_kernel": [r"window_levels_kernel", r"window_eager_kernel"],
    "rs_scatter_kernel<u64>": [r"ws_scatter_kernel", r"rs_scatter\w*<unsigned long", r"ss_scatter\w*"],
    "rs_scatter_kernel<u32>": [r"fs_scatter_kernel", r"rs_scatter\w*<unsigned int", r"msd_scatter\w*"],
    "rs_count_kernel": [r"ws_count_kernel", r"fs_count_kernel", r"rs_count_kernel", r"ss_count\w*", r"msd_count\w*"],
    "ws_leaf_sort_kernel": [r"ws_leaf_sort_kernel"],
    "ws_leaf_count_kernel": [r"ws_leaf_count_kernel"],
    "ws_run_kernels": [r"ws_run_\w+"],
    "flatten_round_kernel": [r"flatten_round_kernel"],
    "fs_image_kernel": [r"fs_image_kernel"],
    "pack_kernel": [r"pack_(cls_)?kernel"],
    "tile_bits_kernel": [r"tile_bits_(cls_)?kernel"],
    "literal_hist_kernel": [r"literal_hist_(cls_)?kernel"],
    "gaps_kernel": [r"gaps_kernel"],
    "sa_groups_kernel": [r"sa_groups_kernel"],
    "sa_build_keys_kernel": [r"sa_round_keys_kernel", r"sa_build_keys_kernel", r"ws_gather64_kernel"],
    "extract_kernels": [r"owner_\w+_kernel"],
    "cand_kernels": [r"cand_\w+_kernel"],
    "scan_kernels": [r"scan_reduce_kernel", r"scan_apply_kernel", r"rs_col\w+", r"ss_blocksum_kernel", r"ss_apply_kernel", r"ss_segbase_kernel"],
    "small_level_kernel": [r"small_level_kernel", r"eager_levels_kernel"],
}


def class_bytes(cls):
    pats = [re.compile(x) for x in CLASS_PATTERNS.get(cls, [re.escape(cls.split("<")[0])])]
    hit = lambda name: any(q.search(name) for q in pats)
    fraw = sum(v[1] for k, v in fetch.items() if hit(k)) * 1024      # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
    wb = sum(v[1] for k, v in write.items() if hit(k)) * 1024
    return fraw, wb


dom = bench["roofline"]["kernel"]
steps = max([v[0] for k, v in fetch.items() if "window_levels_kernel" in k or "window_eager_kernel" in k] + [0])
if steps and fetch and write:
    kernels = {}
    for cls, info in bench.get("kernels", {}).items():
        fraw, wb = class_bytes(cls)
        lps = info.get("launches_per_step") or 0
        if not lps or (fraw == 0 and wb == 0):
            continue
        kernels[cls] = {"launches_per_step": lps, "FETCH_SIZE_bytes_raw_per_step": round(fraw / steps), "WRITE_SIZE_bytes_per_step": round(wb / steps),
                        "hbm_bytes_per_launch": round((2 * fraw + wb) / steps / lps)}
    d = kernels.get(dom, {})
    summ = {"round": prefix, "workload_bytes": bench["config"]["bytes_per_gpu"], "kernel": dom, "steps_in_pmc_run": steps,
            "FETCH_correction": "x2: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for coalesced streaming reads (MI355X_MICROARCH.md, HBM)",
            "hbm_bytes_per_launch": d.get("hbm_bytes_per_launch"),
            "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
            "amplification": round(d["hbm_bytes_per_launch"] / bench["roofline"]["algorithmic_bytes_per_launch"], 3) if d else None,
            "pipeline_hbm_bytes_per_input_byte": round((2 * sum(v[1] for v in fetch.values()) + sum(v[1] for v in write.values())) * 1024 / steps / bench["config"]["bytes_per_gpu"], 1),
            "pipeline_hbm_bytes_per_input_byte_raw_fetch": round((sum(v[1] for v in fetch.values()) + sum(v[1] for v in write.values())) * 1024 / steps / bench["config"]["bytes_per_gpu"], 1),
            "kernels": kernels,
            "source": "profiles/%s_pmc_fetch_write.csv" % prefix,
            "collection": "two separate passes: rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace, python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (tools/profile_round.sh)"}
    json.dump(summ, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in summ.items() if k != "kernels"}, indent=1))
print("value", bench["value"], "dominant", dom, bench["roofline"]["frac"])
