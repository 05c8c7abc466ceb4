"""copy the rocprofv3 summaries of scripts/profile_bench.sh from gpurun_out/ (scratch) into profiles/ (tracked) and
refresh profiles/traffic.json, the PMC-measured HBM-side bytes per traversal launch that bench.py reports as
roofline.traffic.  usage: python scripts/collect_profiles.py <tag> <workload> [<workload> ...]"""
import json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
tpath = os.path.join(root, "profiles", "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
for wl in sys.argv[2:]:
    src = os.path.join(root, "gpurun_out", f"prof_{tag}_{wl}")
    for name in ("kernel_stats.txt", "pmc_summary.txt", "bench_line.json"):
        shutil.copy(os.path.join(src, name), os.path.join(root, "profiles", f"{tag}_{wl}_{name}"))
    # per kernel class: the group (grid) with the most launches among the full-size grids.  k_trace<closest>: the plain build and
    # the one with target triangles (TGT, the light-sampling rays on large trees) are one population; k_tail: either build
    def measure(pattern, label):
        groups = {}
        for line in open(os.path.join(src, "pmc_summary.txt")):
            m = re.match(pattern + r"\s+grid\s+(\d+)\s+(\S+)\s+launches\s+(\d+)\s+mean\s+([\d.]+)", line)
            if m:
                n0, s0 = groups.setdefault(int(m.group(1)), {}).get(m.group(2), (0, 0.0))
                groups[int(m.group(1))][m.group(2)] = (n0 + int(m.group(3)), s0 + int(m.group(3)) * float(m.group(4)))
        groups = {g: v for g, v in groups.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v and "TCC_HIT_sum" in v}
        if not groups: return None
        grid = max(groups, key=lambda g: (groups[g]["FETCH_SIZE"][0], g))
        vals = {k: v[1] / v[0] for k, v in groups[grid].items()}
        launches = groups[grid]["FETCH_SIZE"][0]
        # FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X guide, HBM section): x2
        return {"tag": tag, "kernel": label, "fetch_size_kb_raw": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
                "bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
                "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]),
                "command": f"bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-sub --workload {wl} (mean over the {launches} {label} launches of grid {grid})"}
    per = {}
    for pattern, label in ((r"void pt::k_trace<false, false(?:, (?:false|true))?>", "k_trace<closest>"), (r"void pt::k_tail<\d>", "k_tail")):
        e = measure(pattern, label)
        if e: per[label] = e
    # the workload's own entry: its batched traversal launches when it has any (the large batches), else the tail kernel (small
    # batches run in it whole); the counting frames at the end of bench.py add a few wavefront launches to a tail-only run,
    # so "has any" means more launches than those
    tr, tl = per.get("k_trace<closest>"), per.get("k_tail")
    main = tr if tr and (not tl or int(re.search(r"over the (\d+)", tr["command"]).group(1)) > 2 * int(re.search(r"over the (\d+)", tl["command"]).group(1))) else (tl or tr)
    traffic[wl] = dict(main, kernels=per)
json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1))
