#!/bin/bash
# A/B of runtime switches in ONE call: the incoherent-ray microbenchmark (4 M rays, counting diagnostics) and bench workloads under each
# environment.  usage: bash scripts/ab_env_r3.sh <outdir> "<workloads>" "<ENV=.. ENV=..>|<label>" ...
out=$1; shift; wls=$1; shift; mkdir -p $out
for spec in "$@"; do
  envs=${spec%%|*}; label=${spec##*|}
  env $envs PIMGPU_TRACE_DIAG=1 timeout -k 10 200 python -u scripts/raybench.py 1000000 4194304 > $out/${label}_raybench.log 2>&1 || { echo "raybench failed: $label"; tail -5 $out/${label}_raybench.log; exit 1; }
  for wl in $wls; do
    env $envs timeout -k 10 200 python -u bench.py --workload $wl --no-sub --no-cpu-baseline --steps 16 --warmup 3 > $out/${label}_$wl.json 2> $out/${label}_$wl.err || { echo "bench failed: $label $wl"; tail -5 $out/${label}_$wl.err; exit 1; }
  done
done
python - "$out" <<'PY'
import json, glob, sys, os, re
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*_raybench.log")):
    for l in open(f):
        if re.match(r"^(uniform|surface)\s+n=", l) or "checksum" in l:
            print("%-20s %s" % (os.path.basename(f)[:-13], l.strip()))
for f in sorted(glob.glob(out + "/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no line"); continue
    r = {e["kernel"].split("<")[0]: e["ms_per_step"] for e in d["rooflines"]}
    t = d["rooflines"][0]
    print("%-32s ms/step %7.3f  dev %7.3f  trace %6.3f shadeA %6.3f shadeB %6.3f tail %6.3f  nodes/ray %.1f tris/ray %.1f lum %.6f sd %.6f" % (
        os.path.basename(f)[:-5], d["ms_per_step"], d["device_ms_per_step"], r["k_trace"], r["k_shade_a"], r["k_shade_b"], r["k_tail"], t["nodes_per_ray"], t["tris_per_ray"], d["image_mean_luminance"], d["pt_stddev"]))
PY
