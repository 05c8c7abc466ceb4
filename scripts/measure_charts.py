"""chartnodes_create + ChartMaskFn (rendering/lightmap.c:723-767, 527-561) at scale: the 1M-triangle scene's triangles as one
world-space mesh, charts = runs of 1..8 consecutive nodes; whole calls (host arrays in and out), oracle beside on a sample.
usage: python scripts/measure_charts.py [tris] [--cpu]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from pim_amd import api, scenes
from tests import prepare_util as pu
from tests.util import load_oracle

tris = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1_000_000
api.init(0)
sc = scenes.synthetic_scene(target_tris=tris, seed=1234, extent=100.0, emissive_attempts=4, diffuse_only=True)
m = api.PimMaterial(); m.flags = 0
M = np.zeros((1, 4, 4), np.float32); M[0, 0, 0] = M[0, 1, 1] = M[0, 2, 2] = M[0, 3, 3] = 1.0
U = np.zeros((sc.positions.shape[0], 4), np.float32)
dr = api.Drawables([(np.ascontiguousarray(sc.positions, np.float32), np.ascontiguousarray(sc.normals, np.float32), U)], np.zeros(1, np.int32), M, [m])
tpu = 4.0
api.lightmap_chartnodes(dr, tpu)
t0 = time.perf_counter(); nodes = api.lightmap_chartnodes(dr, tpu); t_nodes = time.perf_counter() - t0
keep = nodes["area"] < 4.0                                  # the room's walls would be 400-texel charts: leave them to the host's splitter
nodes = np.ascontiguousarray(nodes[keep])
first = pu.chart_partition(nodes.shape[0], seed=1, sizes=(1, 1, 1, 2))
work = nodes.copy(); api.lightmap_chart_masks(work, first)
work = nodes.copy()
t0 = time.perf_counter(); areas, sizes, offsets, masks = api.lightmap_chart_masks(work, first); t_masks = time.perf_counter() - t0
out = {"what": "lightmap chart nodes + chart masks", "triangles": int(keep.shape[0]), "nodes": int(nodes.shape[0]), "charts": int(first.shape[0] - 1),
       "mask_bytes": int(masks.shape[0]), "set_fraction": float((masks == 255).mean()),
       "chartnodes_call_s": t_nodes, "chartnodes_per_s": keep.shape[0] / t_nodes, "chart_masks_calls_s": t_masks, "mask_texels_per_s": masks.shape[0] / t_masks}
if "--cpu" in sys.argv:
    lib = load_oracle(True)
    n = min(nodes.shape[0], 20000)
    sub_first = first[first <= n]
    sub = nodes[:sub_first[-1]].copy()
    t0 = time.perf_counter(); ao, so, oo, mo = pu.oracle_chart_masks(lib, sub, sub_first); t_cpu = time.perf_counter() - t0
    nc = sub_first.shape[0] - 1
    same = bool((so == sizes[:nc]).all() and (mo == masks[:mo.shape[0]]).all())
    out.update({"oracle_sample_charts": nc, "oracle_s": t_cpu, "oracle_mask_texels_per_s": mo.shape[0] / t_cpu, "oracle_threads": 1, "bit_identical": same})
print(json.dumps(out))
