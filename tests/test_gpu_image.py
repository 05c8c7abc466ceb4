"""-m gpu: the wavefront integrator against the oracle's scalar Pt_TraceRay.

Both sides use per-pixel Prng streams seeded from the pixel coordinates, the same portable sin/cos/log/exp
and IEEE fp32 without FMA contraction, and the light-grid feedback is integer atomics, so the bar is
BIT-EXACT: every pixel of color / albedo / normal after several accumulated frames (with UpdateDists between
frames), the ray counts, the autofocus state, the light histograms, and pt_stddev."""
import os

import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import gpu_init, assert_bits_equal
from tests.util import OracleScene, copy_dof

pytestmark = pytest.mark.gpu


def _render_both(sc, W, H, frames, max_bounces, cam=None, check_lights=True):
    """the oracle once; the GPU under the library's own tail policy (a batch this small goes to the tail kernel whole) and,
    unless the calling test has pinned the switch itself, once more with two wavefront bounces before the tail -- the shape
    of the 4K bench workload -- so that the batched kernels stay covered by every scene that comes through here"""
    gpu_init()
    cam = cam or scenes.pt_test_camera()
    o = OracleScene(sc.desc, accel=1)
    img = [np.zeros((W * H, 3), np.float32) for _ in range(3)]
    do = scenes.default_dof()
    ostats, ofocal = [], []
    for s in range(frames):
        so = o.trace_image(img, W, H, cam, do, scenes.trace_params(s, seed=17, max_bounces=max_bounces))
        ostats.append((so.extensionRays, so.shadowRays, so.otherRays, so.paths)); ofocal.append(do.focalLength)
    mo, sod = o.stddev(img[0], W, H)
    ol = o.download_lights() if check_lights else None
    pinned = os.environ.get("PIMGPU_TAIL_BOUNCE")
    color = None
    for switch in ([pinned] if pinned is not None else [None, "2"]):
        if switch is not None: os.environ["PIMGPU_TAIL_BOUNCE"] = switch
        try:
            g = api.Scene(sc.desc)
            t = api.Trace(W, H)
            dg = scenes.default_dof()
            for s in range(frames):
                t.trace(g, cam, dg, scenes.trace_params(s, seed=17, max_bounces=max_bounces))
                sg = t.stats()
                assert (sg.extensionRays, sg.shadowRays, sg.otherRays, sg.paths) == ostats[s], f"frame {s} switch {switch}"
                assert dg.focalLength == ofocal[s]
            color, albedo, normal = t.download()
            assert_bits_equal(color, img[0], f"color (switch {switch})")
            assert_bits_equal(albedo, img[1], f"albedo (switch {switch})")
            assert_bits_equal(normal, img[2], f"normal (switch {switch})")
            # bits, except that a NaN is a NaN: a 1x1 image's stddev is inf * 0, and the default NaN of an invalid operation is
            # 0xFFC00000 on x86 and 0x7FC00000 on gfx950
            for got_v, want_v in zip(t.stddev(), (mo, sod)):
                assert (np.isnan(got_v) and np.isnan(want_v)) or np.float32(got_v).tobytes() == np.float32(want_v).tobytes(), (got_v, want_v)
            if check_lights:
                gl = g.download_lights()
                for k in ("live", "pdf", "cdf"):
                    assert_bits_equal(gl[k], ol[k], f"{k} (switch {switch})")
            t.close(); g.close()
        finally:
            if pinned is None: os.environ.pop("PIMGPU_TAIL_BOUNCE", None)
    return color


def test_cornell_pt_test_image_bit_exact():
    """BASELINE config 1 shape at reduced size: Cornell boxes, pt_test camera, 4-bounce clamp, 6 frames"""
    color = _render_both(scenes.cornell_box(), 160, 90, 6, 4)
    assert np.isfinite(color).all() and color.mean() > 0.01


def test_cornell_roulette_only_bit_exact():
    """the reference's own termination: no bounce clamp, russian roulette only (path_tracer.c:2319-2331)"""
    _render_both(scenes.cornell_box(), 96, 54, 3, 0)


def test_cornell_spheres_refractive_sky_bit_exact():
    """spheres variant: metal / dielectric / refractive rows (thickness rays), open sky cubemap on misses"""
    sc = scenes.cornell_box("spheres", sky_size=16, emissive_attempts=64)
    cam = scenes.camera((0.0, -2.0, 4.5), (0.0, -0.4, -1.0))
    _render_both(sc, 96, 54, 3, 6, cam=cam)


def test_foggy_cornell_media_bit_exact():
    """dense participating medium: delta tracking, in-scatter light rays, phase-function rejection sampling"""
    sc = scenes.cornell_box(media=scenes.foggy_media(6.0))
    _render_both(sc, 64, 36, 2, 5)


def test_trace_rays_and_raygen_bit_exact():
    gpu_init()
    sc = scenes.cornell_box()
    g = api.Scene(sc.desc)
    o = OracleScene(sc.desc, accel=1)
    ro, rd = scenes.random_rays(sc, 5000, seed=21)
    rg = g.trace_rays(ro, rd, sampleIndex=3, seed=5, maxBounces=5)
    ro_ = o.trace_rays(ro, rd, sampleIndex=3, seed=5, maxBounces=5)
    for k in ("color", "albedo", "normal"):
        assert_bits_equal(rg[k], ro_[k], k)
    origin = np.array([0.5, -1.0, 0.25, 0.0], np.float32)
    cg, dg = g.raygen(origin, 4096, sampleIndex=1, seed=9, maxBounces=4)
    co, do = o.raygen(origin, 4096, sampleIndex=1, seed=9, maxBounces=4)
    assert_bits_equal(dg, do, "directions")
    assert_bits_equal(cg, co, "colors")
    g.close()


@pytest.mark.parametrize("shares", ["1", "3", "8"])
def test_trace_image_host_equals_trace_and_download(shares):
    """pimgpu_trace_image_host (Pt_Trace as its callers use it): the pixels traced in shares, each share's part of the images
    copied home under the next share's tracing -- the images, the statistics and the autofocus state of pimgpu_trace_image +
    pimgpu_trace_download, bit for bit, whatever the number of shares (and those are checked against the oracle above)"""
    gpu_init()
    W, H = 200, 120
    sc = scenes.cornell_box("spheres", sky_size=16, emissive_attempts=64)
    cam = scenes.camera((0.0, -2.0, 4.5), (0.0, -0.4, -1.0))
    ga, gb = api.Scene(sc.desc), api.Scene(sc.desc)
    ta, tb = api.Trace(W, H), api.Trace(W, H)
    da, db = scenes.default_dof(), scenes.default_dof()
    host = [np.full((W * H, 3), np.nan, np.float32) for _ in range(3)]
    old = {k: os.environ.get(k) for k in ("PIMGPU_HOST_SHARES", "PIMGPU_HOST_SHARE_MIN")}
    os.environ["PIMGPU_HOST_SHARES"] = shares; os.environ["PIMGPU_HOST_SHARE_MIN"] = "2048"
    try:
        for s in range(3):
            p = scenes.trace_params(s, seed=5, max_bounces=6)
            ta.trace(ga, cam, da, p)
            tb.trace_to_host(gb, cam, db, p, host[0], host[1], host[2])
            sa, sb = ta.stats(), tb.stats()
            assert (sa.extensionRays, sa.shadowRays, sa.otherRays, sa.paths) == (sb.extensionRays, sb.shadowRays, sb.otherRays, sb.paths)
            assert da.focalLength == db.focalLength
            ref = ta.download()
            for name, a, b in zip(("color", "albedo", "normal"), ref, host):
                assert_bits_equal(b, a, f"{name} frame {s}")
            for name, a, b in zip(("color", "albedo", "normal"), ref, tb.download()):
                assert_bits_equal(b, a, f"device {name} frame {s}")
        la, lb = ga.download_lights(), gb.download_lights()
        for k in ("live", "pdf", "cdf"):
            assert_bits_equal(lb[k], la[k], k)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    ta.close(); tb.close(); ga.close(); gb.close()


@pytest.mark.parametrize("n", [1, 3, 65, 700])
def test_trace_rays_small_batches_bit_exact(n):
    """batches that leave most lanes of a wave (or all but one) without a path: in the one-launch integrator those lanes only
    help with the other lanes' rays (traverse_coop) -- same results as the oracle's scalar Pt_TraceRay"""
    gpu_init()
    sc = scenes.cornell_box("spheres", sky_size=16, emissive_attempts=64)
    g = api.Scene(sc.desc)
    o = OracleScene(sc.desc, accel=1)
    ro, rd = scenes.random_rays(sc, n, seed=100 + n)
    for k in range(2):
        rg = g.trace_rays(ro, rd, sampleIndex=k, seed=9, maxBounces=7)
        ro_ = o.trace_rays(ro, rd, sampleIndex=k, seed=9, maxBounces=7)
        for name in ("color", "albedo", "normal"):
            assert_bits_equal(rg[name], ro_[name], f"{name} call {k}")
    g.close()


@pytest.mark.parametrize("ntri", [0, 1, 5])
def test_tiny_scenes_render_bit_exact(ntri):
    """no triangle at all, a single-leaf tree and the smallest real tree under an open sky, through the one-launch integrator
    and through two wavefront bounces: every ray misses or meets the same few triangles"""
    full = scenes.cornell_box()
    b = scenes.SceneBuilder(seed=1, sky_size=8, emissive_attempts=8)
    m = b.material((0.6, 0.5, 0.4, 1.0), (0.7, 1.0, 0.0, 0.0))
    if ntri:
        b.add_arrays(full.positions[:ntri * 3], full.normals[:ntri * 3], full.uvs[:ntri * 3], m)
    _render_both(b.build(), 64, 36, 2, 4)


@pytest.mark.parametrize("W,H", [(1, 1), (1, 9), (130, 1), (67, 35), (2, 2)])
def test_image_sizes_that_fill_no_wave_or_row_bit_exact(W, H):
    """a single pixel (pt_stddev divides by N - 1 = 0 there: whatever the reference's arithmetic gives), one column, one row
    wider than two waves, and sizes that are multiples of nothing, three frames each"""
    _render_both(scenes.cornell_box(), W, H, 3, 4)


@pytest.mark.parametrize("bounces", [0, 1, 2])
def test_bounce_limits_at_the_bottom_bit_exact(bounces):
    """no bounce at all (the loop body never runs: the reference's result is what Pt_TraceRay initialises), one, two"""
    _render_both(scenes.cornell_box(), 48, 27, 2, bounces)
