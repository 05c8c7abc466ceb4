"""-m gpu: the two batched traversal kernels of pim_amd/csrc/trace.hip -- k_trace (a lane puts a reached leaf aside until the
wave votes for the triangle code) and k_traceq (the leaves' triangles go to a per-wave ring in LDS and are tested 64 at a
time, the closest hit kept per ray as an atomic 64-bit minimum) -- forced one at a time (PIMGPU_TRACE_QUEUE), each in its
closest-hit, any-hit and visibility-query (target triangle) builds: scene set-up (probe and shadow rays of the light grid),
PtRayHit records and wavefront-only images all equal the oracle's bit for bit.  The library itself picks by launch size."""
import os

import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import gpu_init, assert_bits_equal
from tests.test_gpu_image import _render_both
from tests.util import OracleScene

pytestmark = pytest.mark.gpu

KNOBS = ("PIMGPU_TRACE_QUEUE", "PIMGPU_TAIL_BOUNCE", "PIMGPU_NEE_TARGET", "PIMGPU_TRACE_CHUNK", "PIMGPU_STACK_CAP")


@pytest.fixture
def knobs():
    old = {k: os.environ.get(k) for k in KNOBS}
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("queue", ["0", "1"])
def test_each_kernel_setup_hits_and_wavefront_images_bit_exact(knobs, queue):
    os.environ["PIMGPU_TRACE_QUEUE"] = queue
    gpu_init()
    sc = scenes.cornell_box("spheres", sky_size=8, emissive_attempts=64)          # 33 k triangles, refractive row, sky
    g = api.Scene(sc.desc)                                                        # light grid: closest-hit probes + any-hit shadow rays
    o = OracleScene(sc.desc, accel=1)
    gl, ol = g.download_lights(), o.download_lights()
    for k in ("vertToEmit", "emitToVert", "cellLength", "pdf", "cdf"):
        assert_bits_equal(gl[k], ol[k], k)
    for ro, rd in (scenes.random_rays(sc, 200_000, seed=5), scenes.surface_rays(sc, 200_000, seed=6, offset=1e-5)):
        hg, ho = g.intersect(ro, rd), o.intersect(ro, rd)
        assert hg.tobytes() == ho.tobytes()
    # bounded rays, a ragged count, dynamic chunks forced on a small tree
    ro, rd = scenes.random_rays(sc, 7001, seed=7)
    rng = np.random.default_rng(8)
    tn = rng.uniform(0.0, 2.0, 7001).astype(np.float32); tf = (tn + rng.uniform(-0.5, 6.0, 7001)).astype(np.float32)
    os.environ["PIMGPU_TRACE_CHUNK"] = "128"
    assert g.intersect(ro, rd, tn, tf).tobytes() == o.intersect(ro, rd, tn, tf).tobytes()
    os.environ.pop("PIMGPU_TRACE_CHUNK")
    g.close(); o.close()
    # every bounce a wavefront bounce (no tail kernel): extension rays through the plain build, light-sampling rays through the
    # target build (forced on: the host only uses it on large trees) and through the plain one
    os.environ["PIMGPU_TAIL_BOUNCE"] = "-1"
    for target in ("1", "0"):
        os.environ["PIMGPU_NEE_TARGET"] = target
        color = _render_both(scenes.cornell_box(), 160, 90, 2, 6)
        assert np.isfinite(color).all() and color.mean() > 0.01
    _render_both(sc, 96, 54, 2, 5, cam=scenes.camera((0.0, -2.0, 4.5), (0.0, -0.4, -1.0)))


def test_queued_kernel_reports_a_full_stack(knobs):
    """the queued kernel keeps 8 stack entries of a lane in LDS and the rest in the spill area: with the cap lowered to the LDS part
    of the per-lane kernel (12) a 33 k-triangle tree still fits, and the hits do not change"""
    os.environ["PIMGPU_TRACE_QUEUE"] = "1"
    gpu_init()
    sc = scenes.cornell_box("spheres", emissive_attempts=16)
    g = api.Scene(sc.desc)
    ro, rd = scenes.surface_rays(sc, 50_000, seed=3, offset=1e-5)
    want = g.intersect(ro, rd).tobytes()
    os.environ["PIMGPU_STACK_CAP"] = "12"
    try:
        assert g.intersect(ro, rd).tobytes() == want
    except api.PimGpuError as e:                         # deeper than 12 entries somewhere: reported, not silently wrong
        assert "stack" in str(e)
    g.close()
