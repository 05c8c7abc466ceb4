out=gpurun_out/t20; mkdir -p $out
for spec in "X=1|base" "PIMGPU_TAIL_BOUNCE=1|tb1" "PIMGPU_TAIL_BOUNCE=2|tb2" "PIMGPU_TAIL_WAVES=12|w12" "PIMGPU_TAIL_WAVES=20|w20"; do
  envs=${spec%%|*}; label=${spec##*|}
  env $envs timeout -k 10 250 python -u bench.py --workload synth20m_1080p --no-sub --no-cpu-baseline --steps 8 --warmup 2 > $out/$label.json 2> $out/$label.err || { echo failed $label; tail -3 $out/$label.err; exit 1; }
  python - $out/$label.json $label <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "ms/step %.3f dev %.3f build %.3f trace-only Mrays/s %.1f" % (d["ms_per_step"], d["device_ms_per_step"], d.get("bvh_rebuild_ms_per_step",0), d.get("trace_only_Mrays_per_s",0)))
PY
done
