"""-m gpu: the two other callers of Pt_TraceRay -- Cubemap_Bake (rendering/cubemap.c:130-190) and AmbCube_Bake
(math/ambcube.c:5-32) -- on the GPU against the oracle, BIT-EXACT on every face texel / cube value, over two
progressive passes (the second blends into the first's result)."""
import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import gpu_init, assert_bits_equal
from tests.util import OracleScene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", ["boxes", "spheres_sky"])
def test_cubemap_bake_bit_exact(variant):
    gpu_init()
    sc = scenes.cornell_box() if variant == "boxes" else scenes.cornell_box("spheres", sky_size=16, emissive_attempts=32)
    g = api.Scene(sc.desc); o = OracleScene(sc.desc, accel=1)
    size = 24
    fg = [np.zeros(size * size * 3, np.float32) for _ in range(6)]
    fo = [np.zeros(size * size * 3, np.float32) for _ in range(6)]
    origin = np.array([0.3, -0.5, 0.7, 0.0], np.float32)
    for s in range(2):
        weight = 1.0 / (1.0 + s)
        g.cubemap_bake(fg, size, origin, weight, sampleIndex=s, seed=9, maxBounces=5)
        o.cubemap_bake(fo, size, origin, weight, sampleIndex=s, seed=9, maxBounces=5)
        for f in range(6):
            assert_bits_equal(fg[f], fo[f], f"face {f} pass {s}")
    assert np.isfinite(np.concatenate(fg)).all() and np.concatenate(fg).mean() > 0.001
    g.close()


def test_cubemap_bake_arguments():
    gpu_init()
    sc = scenes.cornell_box()
    g = api.Scene(sc.desc)
    faces = [np.zeros(3, np.float32) for _ in range(6)]
    with pytest.raises(api.PimGpuError):
        g.cubemap_bake(faces, 1, np.zeros(4, np.float32), 1.0)           # size 1: LinearInvert divides by size - 1
    g.cubemap_bake([np.zeros(0, np.float32) for _ in range(6)], 0, np.zeros(4, np.float32), 1.0)   # size 0: no-op (cubemap.c:172)
    g.close()


def test_ambcube_bake_bit_exact():
    gpu_init()
    sc = scenes.cornell_box()
    g = api.Scene(sc.desc); o = OracleScene(sc.desc, accel=1)
    cg = np.zeros((6, 4), np.float32); co = np.zeros((6, 4), np.float32)
    origin = np.array([-1.0, 0.5, 1.5, 0.0], np.float32)
    ng = no = 0
    for s, samples in enumerate((1000, 333, 1)):
        ng = g.ambcube_bake(cg, origin, samples, ng, sampleIndex=s, seed=4, maxBounces=4)
        no = o.ambcube_bake(co, origin, samples, no, sampleIndex=s, seed=4, maxBounces=4)
        assert ng == no == s + 1
        assert_bits_equal(cg, co, f"cube after pass {s}")
    assert (cg[:, :3] > 0).all()
    assert g.ambcube_bake(cg, origin, 0, ng) == ng + 1                  # zero samples: only the count moves
    assert_bits_equal(cg, co, "cube unchanged by an empty pass")
    g.close()


def test_raygen_runs_update_dists_first():
    """Pt_RayGen starts with PtScene_Update (path_tracer.c:2625), which reaches UpdateDists (:1034): the light histogram the
    first batch left in live[] must be folded into pdf / cdf before the second batch samples them.  Pt_TraceRay batches
    (pimgpu_trace_rays) and Cubemap_Bake do not update."""
    gpu_init()
    sc = scenes.cornell_box()
    g = api.Scene(sc.desc); o = OracleScene(sc.desc, accel=1)
    origin = np.array([0.5, -1.0, 0.25, 0.0], np.float32)
    before = g.download_lights()
    for s in range(3):
        cg, dg = g.raygen(origin, 40000, sampleIndex=s, seed=11, maxBounces=5)
        co, do = o.raygen(origin, 40000, sampleIndex=s, seed=11, maxBounces=5)
        assert_bits_equal(dg, do, f"directions {s}"); assert_bits_equal(cg, co, f"colors {s}")
        gl, ol = g.download_lights(), o.download_lights()
        for k in ("live", "pdf", "cdf"):
            assert_bits_equal(gl[k], ol[k], f"{k} after batch {s}")
    assert (gl["pdf"].view(np.uint32) != before["pdf"].view(np.uint32)).any(), "no cell gathered the 30 hits Dist1D_Update needs"
    # a Pt_TraceRay batch leaves pdf / cdf alone (only live[] moves)
    ro, rd = scenes.random_rays(sc, 20000, seed=2)
    g.trace_rays(ro, rd, sampleIndex=0, seed=3, maxBounces=4); o.trace_rays(ro, rd, sampleIndex=0, seed=3, maxBounces=4)
    gl2, ol2 = g.download_lights(), o.download_lights()
    assert_bits_equal(gl2["pdf"], gl["pdf"], "pdf untouched by trace_rays"); assert_bits_equal(gl2["live"], ol2["live"], "live")
    # AmbCube_Bake goes through Pt_RayGen
    cgc = np.zeros((6, 4), np.float32); coc = np.zeros((6, 4), np.float32)
    g.ambcube_bake(cgc, origin, 5000, 0, sampleIndex=7, seed=4, maxBounces=4); o.ambcube_bake(coc, origin, 5000, 0, sampleIndex=7, seed=4, maxBounces=4)
    assert_bits_equal(cgc, coc, "cube")
    gl3, ol3 = g.download_lights(), o.download_lights()
    for k in ("live", "pdf", "cdf"):
        assert_bits_equal(gl3[k], ol3[k], f"{k} after ambcube")
    assert (gl3["pdf"].view(np.uint32) != gl2["pdf"].view(np.uint32)).any()
    g.close()
