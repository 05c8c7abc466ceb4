"""-m gpu: REAL textures on the GPU against the oracle, BIT-EXACT (portable-libm variant of the oracle, which is within
4 ulp of the vectors compiled from the reference's headers).

scenes.textured_cornell(): non-square multi-texel rgba8 albedo / ROME textures, xy16 normal maps (one steep enough for
FixShadingNormal's reflect branch), uvs scaled outside [0, 1] and below 0, ROME alpha that is non-zero on 1 % of the
texels (EmissionPdf around its threshold) and on a stripe of a wall.  Covers SampleAlbedo / SampleRome / SampleNormal
(path_tracer.c:1328-1375), UvBilinearWrap_c32 / _xy16 (sampler.h:375-392), GetEmission (:1293-1326), EmissionPdf /
SetupEmissives (:784-883) and everything downstream of them: light grid, image, Pt_TraceRay batches, lightmap bake."""
import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import gpu_init, assert_bits_equal
from tests.test_gpu_image import _render_both
from tests.util import OracleScene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tex():
    gpu_init()
    sc = scenes.textured_cornell()
    g = api.Scene(sc.desc)
    o = OracleScene(sc.desc, accel=1)
    yield sc, g, o
    g.close()


def test_textured_light_setup_bit_exact(tex):
    sc, g, o = tex
    gi, oi = g.info(), o.info()
    assert gi.emissiveCount == oi.emissiveCount and gi.activeCells == oi.activeCells
    gl, ol = g.download_lights(), o.download_lights()
    for k in ("vertToEmit", "emitToVert", "cellLength", "pdf", "cdf"):
        assert_bits_equal(gl[k], ol[k], k)
    # the 1 %-alpha ROME puts triangles on both sides of EmissionPdf's threshold: of the walls that carry it (ceiling:
    # vertices 36..71, near wall: 180..215) some triangles are emissive and some are not
    e = gl["vertToEmit"][::3] >= 0
    for first in (36, 180):
        tri = e[first // 3:first // 3 + 12]
        assert tri.any() and not tri.all(), (first, tri)
    # the stripe: every triangle of the left wall (108..143) that sees it
    assert e[108 // 3:108 // 3 + 12].sum() >= 4


def test_textured_image_bit_exact():
    color = _render_both(scenes.textured_cornell(), 160, 90, 4, 6)
    assert np.isfinite(color).all() and color.mean() > 0.01
    # roulette only + another camera (looks at the metal wall with the steep normal map)
    _render_both(scenes.textured_cornell(), 96, 54, 2, 0, cam=scenes.camera((-3.0, -1.0, 0.5), (1.0, -0.1, -0.2)))


def test_textured_surfaces_really_vary(tex):
    """the branches are exercised: the albedo AOV takes many values on one wall, shading normals leave the face normal"""
    sc, g, o = tex
    n = 4000
    rng = np.random.default_rng(5)
    ro = np.zeros((n, 4), np.float32); rd = np.zeros((n, 4), np.float32)
    ro[:, :3] = rng.uniform(-3.0, 3.0, (n, 3)); ro[:, 1] = 0.0
    rd[:, 1] = -1.0                                    # straight down at the floor (albedo + normal map) and the boxes
    rg, ro_ = g.trace_rays(ro, rd, sampleIndex=0, seed=2, maxBounces=1), o.trace_rays(ro, rd, sampleIndex=0, seed=2, maxBounces=1)
    for k in ("color", "albedo", "normal"):
        assert_bits_equal(rg[k], ro_[k], k)
    assert np.unique(rg["albedo"].round(3), axis=0).shape[0] > 200
    tilt = 1.0 - np.abs(rg["normal"][:, 1])
    assert (tilt > 0.01).mean() > 0.3


def test_textured_trace_rays_and_intersect_bit_exact(tex):
    sc, g, o = tex
    ro, rd = scenes.random_rays(sc, 20000, seed=31)
    hg, ho = g.intersect(ro, rd), o.intersect(ro, rd)
    assert hg.tobytes() == ho.tobytes()
    rg, ro_ = g.trace_rays(ro, rd, sampleIndex=2, seed=8, maxBounces=0), o.trace_rays(ro, rd, sampleIndex=2, seed=8, maxBounces=0)
    for k in ("color", "albedo", "normal"):
        assert_bits_equal(rg[k], ro_[k], k)
    so, sd = scenes.surface_rays(sc, 20000, seed=6, offset=1e-4)
    rg, ro_ = g.trace_rays(so, sd, sampleIndex=1, seed=9, maxBounces=5), o.trace_rays(so, sd, sampleIndex=1, seed=9, maxBounces=5)
    for k in ("color", "albedo", "normal"):
        assert_bits_equal(rg[k], ro_[k], "surface " + k)


def test_textured_lightmap_bake_bit_exact(tex):
    sc, g, o = tex
    S = 64
    pages = scenes.lightmap_pages_for(sc, S, 2.0, max_pages=2)
    pg = [p.copy() for p in pages]; po = [p.copy() for p in pages]
    for passIndex, (slice_, spp) in enumerate([(1.0, 2), (0.7, 2)]):
        g.lightmap_bake(pg, S, 2.0, slice_, spp, passIndex=passIndex, seed=5, maxBounces=5)
        o.lightmap_bake(po, S, 2.0, slice_, spp, passIndex=passIndex, seed=5, maxBounces=5)
        for a, b in zip(pg, po):
            assert_bits_equal(a, b, f"page after pass {passIndex}")
    assert (pg[0][:S * S * 20] != 0).any()
