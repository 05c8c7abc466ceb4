#!/bin/bash
# A/B runs of bench.py workloads under environment knobs; usage: bash scripts/ab_r2.sh <outdir> "<ENV=.. ENV=..>|<label>" ...
out=$1; shift; mkdir -p $out
for spec in "$@"; do
  envs=${spec%%|*}; label=${spec##*|}
  for wl in synth1m_4k synth1m_1080p cornell_1080p; do
    env $envs timeout -k 10 120 python -u bench.py --workload $wl --no-sub --no-cpu-baseline --steps 16 --warmup 3 > $out/${label}_$wl.json 2> $out/${label}_$wl.err || exit 1
  done
done
python - "$out" <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no line"); continue
    r = {e["kernel"].split("<")[0]: e["ms_per_step"] for e in d["rooflines"]}
    t = d["rooflines"][0]
    print("%-40s ms/step %7.3f  dev %7.3f  trace %6.3f shadeA %6.3f shadeB %6.3f tail %6.3f  nodes/ray %.1f launches %.1f lum %.6f" % (
        os.path.basename(f)[:-5], d["ms_per_step"], d["device_ms_per_step"], r["k_trace"], r["k_shade_a"], r["k_shade_b"], r["k_tail"], t["nodes_per_ray"], t["launches_per_step"], d["image_mean_luminance"]))
PY
