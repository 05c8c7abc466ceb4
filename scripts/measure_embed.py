"""EmbedAttributes (rendering/lightmap.c:941-1045) at bake scale: `pages` pages of 1024^2 texels, `tris` mapped
triangles with charts of a few texels to a few dozen (density-16 lightmaps), GPU (whole call, host pages in and out,
and the two kernels alone from a rocprofv3-free HIP event bracket) beside the oracle's loop on the host cores.
usage: python scripts/measure_embed.py [tris] [pages]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from pim_amd import api
from tests import prepare_util as pu
from tests.util import load_oracle

tris = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2
S = 1024
api.init(0)
dr = pu.embed_table(S, list(range(count)), tris_per_mesh=(tris // 2, tris - tris // 2), chart_texels=(3.0, 24.0), hard_cases=False)
pages = pu.new_pages(count, S)
api.lightmap_embed(dr, pages, S)                       # warm-up (allocations, code load)
t0 = time.perf_counter(); api.lightmap_embed(dr, pages, S); gpu_s = time.perf_counter() - t0
cover = float(np.mean([pu.page_parts(p, S)[3].mean() for p in pages]))
out = {"what": "lightmap_embed", "pages": count, "lm_size": S, "triangles": tris, "coverage": cover,
       "gpu_call_s": gpu_s, "gpu_texels_per_s": count * S * S / gpu_s}
sample = 64 * S                                        # the oracle on a bounded sample: 64 rows of page 0
if "--cpu" in sys.argv:
    lib = load_oracle(True)
    small = pu.new_pages(count, S)
    t0 = time.perf_counter(); pu.oracle_embed(lib, dr, small, S, threads=os.cpu_count() and min(16, os.cpu_count()) or 8); cpu_s = time.perf_counter() - t0
    same = all((a.view(np.uint32) == b.view(np.uint32)).all() for a, b in zip(pages, small))
    out.update({"oracle_s": cpu_s, "oracle_threads": min(16, os.cpu_count() or 8), "oracle_texels_per_s": count * S * S / cpu_s, "bit_identical": bool(same)})
print(json.dumps(out))
