"""Do whole frames overlap usefully on ONE GPU?  pimgpu_trace_image_samples with the device listed n times (n contexts, streams and
work-buffer sets on one GPU, a frame each per round) against n = 1 (the frames one after the other).
    python scripts/measure_concurrent_samples.py <workload> <n> <samples>      (one process per n: the device list is per process)"""
import sys, time, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pim_amd import api, scenes
import bench

wl_name, n, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
wl = bench.WORKLOADS[wl_name]
api.init_devices([0] * n)
sc, cam, g = bench.build_scene(wl["scene"])
t = api.Trace(wl["width"], wl["height"])
dof = scenes.default_dof()
def run(first):
    p = scenes.trace_params(0, seed=1, max_bounces=wl["bounces"]); p.sampleIndex = first; p.sampleWeight = 1.0
    t.trace_samples(g, cam, dof, p, S)
    return t.stats()
run(0); run(S)                       # warm-up: buffers, tail history
t0 = time.perf_counter()
reps = 3
rays = 0
for r in range(reps):
    st = run((r + 2) * S)
    rays += st.extensionRays + st.shadowRays + st.otherRays
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"workload": wl_name, "contexts": n, "samples": S, "ms_per_call": dt * 1e3, "ms_per_sample": dt * 1e3 / S, "Mrays_per_s": rays / reps / dt / 1e6}), flush=True)
t.close(); g.close(); api.shutdown()
