"""Per-kernel-class, per-unit-of-work figures of one bench workload from the rocprofv3 passes of scripts/profile_bench.sh:
copies the run's summaries from gpurun_out/ (scratch) into profiles/ (tracked) and refreshes profiles/traffic.json, which
bench.py scales by the units and device time of its own run (Run.rooflines).

    python scripts/pmc_per_unit.py <tag> <workload> [<workload> ...]

Every frame of the profiled command (bench.py --plain: warm-up + timed steps, nothing else) is one step of the workload, so
  figure per unit = SUM over all dispatches of the class of the counter / (frames x units per step),
units per step from the bench line of the same command.  HBM-side bytes = FETCH_SIZE x 2 (gfx950 tallies 128-B requests at
64 B: MI355X guide, HBM section) + WRITE_SIZE, both in KB."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
tpath = os.path.join(root, "profiles", "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}

CLASSES = (("k_trace", r"pt::k_trace<false, false"), ("k_shade_a", r"pt::k_shade_a<"), ("k_shade_b", r"pt::k_shade_b<"), ("k_tail", r"pt::k_tail<"))


def totals(src, last=None, skip_frames=0):
    """{class: {counter: (dispatches, sum)}} over every PMC pass directory of the run.
    skip_frames = W: only dispatches behind the (W+1)-th k_init_image of the pass, i.e. the timed steps (the first frame of a
    process has no queue-length history and runs every bounce as a wavefront: other units than the steps the bench line counts);
    last = K: only the K last dispatches of a class in each pass (the microbenchmark's launches come after the scene set-up's
    own traversal launches)"""
    acc = {c: defaultdict(lambda: [0, 0.0]) for c, _ in CLASSES}
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        rows = {c: defaultdict(dict) for c, _ in CLASSES}                  # class -> counter -> {dispatch id: value}
        inits = set()
        for r in csv.DictReader(open(f)):
            if "k_init_image" in r["Kernel_Name"]:
                inits.add(int(r["Dispatch_Id"]))
            for c, pat in CLASSES:
                if re.search(pat, r["Kernel_Name"]):
                    rows[c][r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
        inits = sorted(inits)
        cutoff = inits[skip_frames] if skip_frames and len(inits) > skip_frames else -1
        for c, _ in CLASSES:
            for ctr, byid in rows[c].items():
                ids = [i for i in sorted(byid) if i > cutoff]
                if last is not None:
                    ids = ids[-last:]
                a = acc[c][ctr]
                a[0] += len(ids); a[1] += sum(byid[i] for i in ids)
    return acc


# --on-box: run where the counter CSVs are (they are too large to travel back): leaves <src>/per_unit.json, which a later run
# of this script without the flag merges into profiles/traffic.json together with the text summaries
on_box = "--on-box" in sys.argv
args = [a for a in sys.argv[2:] if a != "--on-box"]
for wl in args:
    src = os.path.join(root, "gpurun_out", f"prof_{tag}_{wl}")
    if not on_box:
        for name in ("kernel_stats.txt", "pmc_summary.txt", "bench_line.json"):
            if os.path.exists(os.path.join(src, name)):
                shutil.copy(os.path.join(src, name), os.path.join(root, "profiles", f"{tag}_{wl}_{name}"))
        pu = os.path.join(src, "per_unit.json")
        if os.path.exists(pu) and not glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
            traffic[wl] = json.load(open(pu))
            continue
    line = json.load(open(os.path.join(src, "pmc_line.json")))          # the bench line of a PMC pass: its steps / warmup / units
    if "rooflines" in line:
        frames = line["steps"]                                            # the timed steps only
        units = {e["kernel"].split("<")[0]: e["units_per_step"] for e in line["rooflines"]}
        cmd = f"bench.py --plain --steps {line['steps']} --warmup {line['warmup']} --workload {wl} (the {line['steps']} timed steps)"
        acc = totals(src, skip_frames=line["warmup"])
    else:                                                                 # the microbenchmark: launches of rays_per_launch rays
        units = {"k_trace": line["rays_per_launch"]}
        frames = None
        cmd = "bench.py --microbench-only"
        acc = totals(src, last=line["launches_timed"] + 1)                # the warm-up launch and the timed ones (the counting launch is another kernel)
    per = {}
    for c, _ in CLASSES:
        t = acc[c]
        if not t or not units.get(c):
            continue
        if frames is None:                                                # per launch: every dispatch of the class is one launch of `units` rays
            n = t["SQ_INSTS_VALU"][0] if "SQ_INSTS_VALU" in t else next(iter(t.values()))[0]
            den = {k: v[0] * units[c] for k, v in t.items()}
        else:
            den = {k: frames * units[c] for k in t}
        g = lambda k: (t[k][1] / den[k]) if k in t and den[k] > 0 else None
        e = {"tag": tag, "command": cmd, "summary_file": f"profiles/{tag}_{wl}_pmc_summary.txt", "dispatches": max(v[0] for v in t.values()),
             "valu_insts_per_unit": g("SQ_INSTS_VALU"), "salu_insts_per_unit": g("SQ_INSTS_SALU"), "vmem_insts_per_unit": g("SQ_INSTS_VMEM"),
             "l1_accesses_per_unit": g("TCP_TOTAL_CACHE_ACCESSES_sum")}
        if "SQ_THREAD_CYCLES_VALU" in t and "SQ_ACTIVE_INST_VALU" in t and t["SQ_ACTIVE_INST_VALU"][1] > 0:
            e["lanes_per_valu_inst"] = t["SQ_THREAD_CYCLES_VALU"][1] / t["SQ_ACTIVE_INST_VALU"][1]
        if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
            e["hbm_bytes_per_unit"] = (2.0 * t["FETCH_SIZE"][1] / den["FETCH_SIZE"] + t["WRITE_SIZE"][1] / den["WRITE_SIZE"]) * 1024.0
            e["fetch_kb_raw_total"] = t["FETCH_SIZE"][1]; e["write_kb_total"] = t["WRITE_SIZE"][1]
        if "TCC_HIT_sum" in t and "TCC_MISS_sum" in t and t["TCC_HIT_sum"][1] + t["TCC_MISS_sum"][1] > 0:
            e["l2_hit_rate"] = t["TCC_HIT_sum"][1] / (t["TCC_HIT_sum"][1] + t["TCC_MISS_sum"][1])
        if "SQ_ACTIVE_INST_VALU" in t and "SQ_BUSY_CYCLES" in t:
            e["sq_active_inst_valu_total"] = t["SQ_ACTIVE_INST_VALU"][1]
        per[c] = e
    traffic[wl] = {"tag": tag, "kernels": per}
    if on_box:
        json.dump(traffic[wl], open(os.path.join(src, "per_unit.json"), "w"), indent=1, sort_keys=True)
if not on_box:
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
print(json.dumps({k: traffic[k] for k in args}, indent=1))
