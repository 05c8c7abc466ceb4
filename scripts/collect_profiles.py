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
    # the batched k_trace<closest> launches: the group (kernel, grid) with the most launches among the full-size grids
    groups = {}
    for line in open(os.path.join(src, "pmc_summary.txt")):
        # closest-hit builds: plain and with target triangles (TGT, the light-sampling rays on large trees) -- one population
        m = re.match(r"void pt::k_trace<false, false(?:, (?:false|true))?>\s+grid\s+(\d+)\s+(\S+)\s+launches\s+(\d+)\s+mean\s+([\d.]+)", line)
        if m:
            n0, s0 = groups.setdefault(int(m.group(1)), {}).get(m.group(2), (0, 0.0))
            groups[int(m.group(1))][m.group(2)] = (n0 + int(m.group(3)), s0 + int(m.group(3)) * float(m.group(4)))
    for g in groups.values():
        for k, (n0, s0) in list(g.items()): g[k] = (n0, s0 / n0)
    grid = max(groups, key=lambda g: (groups[g].get("FETCH_SIZE", (0, 0))[0], g))
    vals = {k: v[1] for k, v in groups[grid].items()}
    launches = groups[grid]["FETCH_SIZE"][0]
    # FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X guide, HBM section): x2
    traffic[wl] = {"tag": tag, "kernel": "k_trace<closest>", "fetch_size_kb_raw": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
                   "bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
                   "l2_hit_rate": vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]),
                   "command": f"bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-sub --workload {wl} (mean over the {launches} k_trace<closest> launches of grid {grid})"}
json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1))
