"""-m gpu: a traversal stack that fills up is REPORTED (PIMGPU_E_LIMIT) by every caller of the traversal kernels, not
silently answered with dropped subtrees.  The stack holds 64 entries (12 in LDS + 52 spilled); PIMGPU_STACK_CAP (a test
knob, read per launch) lowers that to the LDS part so that an ordinary 33 k-triangle tree overflows."""
import os

import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import gpu_init

pytestmark = pytest.mark.gpu


def test_stack_overflow_is_reported_by_every_caller_and_does_not_stick():
    gpu_init()
    sc = scenes.cornell_box("spheres", emissive_attempts=16)
    g = api.Scene(sc.desc)
    cam, dof = scenes.camera((0.0, -2.0, 4.5), (0.0, -0.4, -1.0)), scenes.default_dof()
    t = api.Trace(256, 144)
    ro, rd = scenes.surface_rays(sc, 200000, seed=3)
    os.environ["PIMGPU_TAIL_BOUNCE"] = "-1"           # wavefront bounces only: the overflow comes from k_trace launches
    try:
        os.environ["PIMGPU_STACK_CAP"] = "12"
        with pytest.raises(api.PimGpuError, match="stack overflow"):
            t.trace(g, cam, dof, scenes.trace_params(0, seed=1, max_bounces=3))
        with pytest.raises(api.PimGpuError, match="stack overflow"):
            g.trace_rays(ro[:50000], rd[:50000], maxBounces=2)
        with pytest.raises(api.PimGpuError, match="stack overflow"):
            g.intersect(ro, rd)
        os.environ["PIMGPU_TAIL_BOUNCE"] = "0"        # the tail kernel's inline traversals report through stats[4]
        with pytest.raises(api.PimGpuError, match="stack overflow"):
            t.trace(g, cam, dof, scenes.trace_params(0, seed=1, max_bounces=3))
    finally:
        os.environ.pop("PIMGPU_STACK_CAP", None)
        os.environ.pop("PIMGPU_TAIL_BOUNCE", None)
    # with the full stack the same calls succeed: no stale overflow bit is blamed on them
    h = g.intersect(ro, rd)
    assert (h["type"] != api.HIT_NOTHING).mean() > 0.9
    t.clear()
    t.trace(g, cam, dof, scenes.trace_params(0, seed=1, max_bounces=3))
    g.trace_rays(ro[:1000], rd[:1000], maxBounces=2)
    t.close(); g.close()
