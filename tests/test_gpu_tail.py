"""-m gpu: the tail kernel (pim_amd/csrc/integrate.hip k_tail), which finishes the short queues of deep bounces in one
launch.  Where the wavefront hands over to it only moves time around: for EVERY switch point -- including bounce 0, the
whole integrator as one launch -- colour / albedo / normal, ray counts, autofocus state, light histograms and
pt_stddev equal the oracle's scalar Pt_TraceRay bit for bit (the checks of tests/test_gpu_image.py)."""
import os

import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import gpu_init
from tests.test_gpu_image import _render_both

pytestmark = pytest.mark.gpu


KNOBS = ("PIMGPU_TAIL_BOUNCE", "PIMGPU_TAIL_DENSE", "PIMGPU_GROUPS", "PIMGPU_GROUP_STAGGER", "PIMGPU_GROUP_MIN_PATHS", "PIMGPU_NEE_TARGET", "PIMGPU_TAIL_MAX_PATHS")


@pytest.fixture
def tail_env():
    old = {k: os.environ.get(k) for k in KNOBS}
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("switch", ["-1", "0", "1", "3", "7"])
def test_cornell_every_switch_point_bit_exact(tail_env, switch):
    os.environ["PIMGPU_TAIL_BOUNCE"] = switch
    color = _render_both(scenes.cornell_box(), 128, 72, 3, 8)
    assert np.isfinite(color).all() and color.mean() > 0.01


@pytest.mark.parametrize("switch", ["0", "2"])
def test_tail_refractive_sky_media_and_roulette_bit_exact(tail_env, switch):
    """the heavy cases inside the tail: thickness rays of refractive spheres, sky misses, a dense medium with in-scatter
    light rays, and the reference's roulette-only termination (no bounce clamp)"""
    os.environ["PIMGPU_TAIL_BOUNCE"] = switch
    sc = scenes.cornell_box("spheres", sky_size=16, emissive_attempts=64)
    _render_both(sc, 80, 45, 2, 6, cam=scenes.camera((0.0, -2.0, 4.5), (0.0, -0.4, -1.0)))
    _render_both(scenes.cornell_box(media=scenes.foggy_media(6.0)), 64, 36, 2, 5)
    _render_both(scenes.cornell_box(), 80, 45, 2, 0)


def test_tail_policy_uses_the_previous_call(tail_env):
    """a batch below the threshold (PIMGPU_TAIL_MAX_PATHS) goes to the tail kernel whole, from the first call on; above it
    the host switches where the previous call's queue fell below the threshold: the first call runs every bounce as a
    wavefront, later calls hand the deep bounces to the tail kernel -- same bits every way"""
    os.environ.pop("PIMGPU_TAIL_BOUNCE", None)
    gpu_init()
    sc = scenes.cornell_box()
    W, H = 160, 90
    images = []
    for max_paths in (None, str(W * H // 3)):
        if max_paths is None: os.environ.pop("PIMGPU_TAIL_MAX_PATHS", None)
        else: os.environ["PIMGPU_TAIL_MAX_PATHS"] = max_paths
        g = api.Scene(sc.desc)                          # a fresh scene: the light histograms adapt from frame to frame
        t = api.Trace(W, H)
        dof = scenes.default_dof()
        seen = []
        for s in range(3):
            t.trace(g, scenes.pt_test_camera(), dof, scenes.trace_params(s, seed=3, max_bounces=12))
            st = t.stats()
            seen.append((st.bounces, st.tailRays))
        if max_paths is None:
            assert all(b == 0 and r > 0 for b, r in seen)      # the whole integrator as one launch
        else:
            assert seen[0][0] == 12 and seen[0][1] < 200        # no history: all wavefront (but for the few parked paths)
            assert 1 <= seen[1][0] < 12 and seen[1][1] > 0     # history: short queues go to the tail kernel
            assert seen[2][0] == seen[1][0]
        images.append(t.download()[0])
        t.close(); g.close()
    os.environ["PIMGPU_TAIL_BOUNCE"] = "-1"
    g2 = api.Scene(sc.desc)
    t2 = api.Trace(W, H)
    dof2 = scenes.default_dof()
    for s in range(3):
        t2.trace(g2, scenes.pt_test_camera(), dof2, scenes.trace_params(s, seed=3, max_bounces=12))
        assert t2.stats().tailRays == 0
    ref = t2.download()[0]
    for a in images:
        assert (ref.view(np.uint32) == a.view(np.uint32)).all()
    t2.close(); g2.close()


def test_lightmap_bake_through_the_tail_bit_exact(tail_env):
    from tests.test_gpu_lightmap import test_cornell_bake_bit_exact
    os.environ["PIMGPU_TAIL_BOUNCE"] = "1"
    test_cornell_bake_bit_exact()


@pytest.mark.parametrize("groups,stagger,switch", [("2", "2", "3"), ("3", "1", "-1"), ("4", "0", "1"), ("8", "-1", "0")])
def test_image_shares_on_concurrent_streams_bit_exact(tail_env, groups, stagger, switch):
    """an image split into shares that run on their own streams (integrate_image), staggered or not, each share with
    its own tail switch: the same bits as the oracle, ray counts included"""
    os.environ["PIMGPU_GROUPS"] = groups
    os.environ["PIMGPU_GROUP_STAGGER"] = stagger
    os.environ["PIMGPU_GROUP_MIN_PATHS"] = "64"
    os.environ["PIMGPU_TAIL_BOUNCE"] = switch
    _render_both(scenes.cornell_box(), 128, 72, 3, 8)
    sc = scenes.cornell_box("spheres", sky_size=16, emissive_attempts=64)
    _render_both(sc, 80, 45, 2, 6, cam=scenes.camera((0.0, -2.0, 4.5), (0.0, -0.4, -1.0)))


@pytest.mark.parametrize("dense", ["0", "1"])
def test_both_builds_of_the_tail_kernel_bit_exact(tail_env, dense):
    """k_tail exists twice (2 or 4 waves per SIMD, picked by the size of the tree): either build, the oracle's bits"""
    os.environ["PIMGPU_TAIL_DENSE"] = dense
    os.environ["PIMGPU_TAIL_BOUNCE"] = "1"
    _render_both(scenes.cornell_box(), 128, 72, 2, 8)
    sc = scenes.cornell_box("spheres", sky_size=16, emissive_attempts=64)
    _render_both(sc, 80, 45, 2, 6, cam=scenes.camera((0.0, -2.0, 4.5), (0.0, -0.4, -1.0)))


@pytest.mark.parametrize("targeted", ["0", "1"])
def test_light_rays_as_visibility_queries_bit_exact(tail_env, targeted):
    """the wavefront's light-sampling rays run as plain closest-hit queries or as visibility queries of their light
    (k_trace TGT, picked by the size of the tree): either way the oracle's bits and ray counts, with occluders in front of
    the lights (spheres), media in-scatter rays and a sky"""
    os.environ["PIMGPU_NEE_TARGET"] = targeted
    os.environ["PIMGPU_TAIL_BOUNCE"] = "-1"
    _render_both(scenes.cornell_box(), 128, 72, 3, 8)
    sc = scenes.cornell_box("spheres", sky_size=16, emissive_attempts=64)
    _render_both(sc, 80, 45, 2, 6, cam=scenes.camera((0.0, -2.0, 4.5), (0.0, -0.4, -1.0)))
    _render_both(scenes.cornell_box(media=scenes.foggy_media(6.0)), 64, 36, 2, 5)
    os.environ["PIMGPU_TAIL_BOUNCE"] = "2"
    sc = scenes.synthetic_scene(target_tris=20000, seed=5, extent=30.0, emissive_attempts=32)
    _render_both(sc, 96, 54, 2, 6, cam=scenes.synthetic_camera(30.0))
