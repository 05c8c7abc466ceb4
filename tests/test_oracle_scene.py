"""CPU tests of the oracle on whole scenes (no GPU): its three intersectors agree, the scene set-up is
what the Cornell recipe predicts, the two libm variants give statistically equal images, and the
harness' float32 flatten matches the oracle's restatement of FlattenDrawables."""
import ctypes

import numpy as np
import pytest

from pim_amd import scenes
from tests.util import OracleScene, load_oracle, copy_dof, bits


@pytest.fixture(scope="module")
def cornell():
    return scenes.cornell_box()


def test_cornell_recipe_counts(cornell):
    # 9 boxes x 12 = 108 triangles, 12 emissive (the light box), 7^3 grid cells at 1.5 m
    assert cornell.tri_count == 108
    o = OracleScene(cornell.desc, accel=1)
    info = o.info()
    assert info.emissiveCount == 12
    assert list(info.gridSize) == [7, 7, 7] and info.gridLen == 343
    lights = o.download_lights()
    assert (lights["emitToVert"] == np.arange(12) * 3 + 2 * 36).all()       # third entity = Cornell_Light
    assert (lights["vertToEmit"] >= 0).sum() == 36


@pytest.mark.parametrize("accel", [0, 1])
def test_intersectors_agree(cornell, accel):
    """fp32 watertight brute force == fp32 BVH bit for bit; both agree with the fp64 ground truth on the
    primitive for >= 99.99% of rays and on t to 1e-5 relative (BASELINE.json's gate)"""
    o = OracleScene(cornell.desc, accel=accel)
    ro, rd = scenes.random_rays(cornell, 50000, seed=3)
    p_a, tuv_a = o.intersect_raw(ro, rd)
    p_b, tuv_b = o.intersect_raw(ro, rd, brute=True)
    assert (p_a == p_b).all() and (bits(tuv_a) == bits(tuv_b)).all()
    p64, t64 = o.intersect_f64(ro, rd)
    assert (p_a == p64).mean() >= 0.9999
    same = (p_a == p64) & (p64 >= 0)
    # 1e-5 relative, plus 4 ulp of the largest coordinate (5 m) for hits a few mm from the origin, where the
    # fp32 vertex data itself limits t to an ABSOLUTE accuracy
    assert (np.abs(tuv_a[same, 0] - t64[same]) <= 1e-5 * t64[same] + 2e-6).all()


def test_synthetic_scene_bvh_equals_brute_force():
    sc = scenes.synthetic_scene(target_tris=6000, seed=5, extent=20.0, emissive_attempts=16)
    o = OracleScene(sc.desc, accel=1)
    ro, rd = scenes.random_rays(sc, 4000, seed=9)
    p_a, tuv_a = o.intersect_raw(ro, rd)
    p_b, tuv_b = o.intersect_raw(ro, rd, brute=True)
    assert (p_a == p_b).all() and (bits(tuv_a) == bits(tuv_b)).all()


def test_bvh_never_culls_what_the_triangle_test_accepts(cornell):
    """secondary rays starting 1e-5 off a surface: the fp32 test accepts a few "hits" a hair BEHIND the origin
    whose t rounds to +5e-7 (the contract is the fp32 test over all triangles); the per-ray box padding must keep
    the BVH walk from culling them -- found on the GPU with 2-triangle leaves, hence the large ray count"""
    for sc in (cornell, scenes.synthetic_scene(target_tris=3000, seed=7, extent=20.0, emissive_attempts=8)):
        o = OracleScene(sc.desc, accel=1)
        ro, rd = scenes.surface_rays(sc, 200000 if sc is cornell else 20000, seed=12, offset=1e-5)
        p_a, tuv_a = o.intersect_raw(ro, rd)
        p_b, tuv_b = o.intersect_raw(ro, rd, brute=True)
        assert (p_a == p_b).all() and (bits(tuv_a) == bits(tuv_b)).all()


def test_flatten_matches_oracle_restatement():
    lib = load_oracle()
    rng = np.random.default_rng(2)
    for mesh in (scenes.box_mesh(), scenes.sphere_mesh(6)):
        t = rng.uniform(-3, 3, 3).astype(np.float32)
        q = scenes.quat_lookat(scenes.normalize3(rng.normal(size=3).astype(np.float32)), np.array([0, 1, 0], np.float32))
        s = rng.uniform(0.5, 2, 3).astype(np.float32)
        P, N, U = scenes.flatten_entity(t, q, s, mesh)
        n = mesh[0].shape[0]
        mp = np.zeros((n, 4), np.float32); mp[:, :3] = mesh[0]; mp[:, 3] = 1
        mn = np.zeros((n, 4), np.float32); mn[:, :3] = mesh[1]
        mu = np.zeros((n, 4), np.float32); mu[:, :2] = mesh[2]
        oP, oN, oU = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32), np.zeros((n, 2), np.float32)
        t4 = np.zeros(4, np.float32); t4[:3] = t
        s4 = np.zeros(4, np.float32); s4[:3] = s
        lib.orc_flatten_entity(t4.ctypes.data_as(ctypes.c_void_p), q.ctypes.data_as(ctypes.c_void_p), s4.ctypes.data_as(ctypes.c_void_p),
                               mp.ctypes.data_as(ctypes.c_void_p), mn.ctypes.data_as(ctypes.c_void_p), mu.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_int32(n), oP.ctypes.data_as(ctypes.c_void_p), oN.ctypes.data_as(ctypes.c_void_p), oU.ctypes.data_as(ctypes.c_void_p))
        assert np.allclose(P, oP, rtol=0, atol=2e-6) and np.allclose(N[:, :3], oN[:, :3], rtol=0, atol=2e-6) and (U == oU).all()


def test_libm_variants_agree_statistically(cornell):
    """liborc.so (glibc libm, bit-exact to the reference headers) and liborc_p.so (portable libm, what the
    GPU is compared with) render the same image up to the few paths a last-ulp difference sends elsewhere"""
    W, H = 96, 54
    res = []
    for portable in (False, True):
        o = OracleScene(cornell.desc, accel=1, portable=portable)
        img = [np.zeros((W * H, 3), np.float32) for _ in range(3)]
        dof = scenes.default_dof()
        for s in range(8):
            o.trace_image(img, W, H, scenes.pt_test_camera(), dof, scenes.trace_params(s, max_bounces=4))
        res.append((img[0].copy(), o.stddev(img[0], W, H)))
    (a, (ma, sa)), (b, (mb, sb)) = res
    close = np.isclose(a, b, rtol=1e-3, atol=1e-5).all(axis=1).mean()
    assert close > 0.97, close
    assert abs(ma - mb) <= 0.02 * ma and abs(sa - sb) <= 0.1 * sa


def test_stddev_is_the_reference_formula(cornell):
    o = OracleScene(cornell.desc, accel=0)
    rng = np.random.default_rng(1)
    c = rng.uniform(0, 2, (40 * 30, 3)).astype(np.float32)
    m, s = o.stddev(c, 40, 30)
    lum = c.astype(np.float64).sum(1) / 3
    assert abs(m - lum.mean()) < 1e-4 and abs(s - lum.std(ddof=1)) < 1e-4


def test_cubemap_and_ambcube_bake_on_the_oracle(cornell):
    """Cubemap_Bake / AmbCube_Bake restatements: deterministic, progressive, and the ambient cube of the closed
    Cornell box is positive on all six axes"""
    o = OracleScene(cornell.desc, accel=1)
    size = 8
    a = [np.zeros(size * size * 3, np.float32) for _ in range(6)]
    b = [np.zeros(size * size * 3, np.float32) for _ in range(6)]
    origin = np.array([0.0, 0.0, 0.5, 0.0], np.float32)
    o.cubemap_bake(a, size, origin, 1.0, sampleIndex=0, seed=2, maxBounces=3)
    o.cubemap_bake(b, size, origin, 1.0, sampleIndex=0, seed=2, maxBounces=3)
    assert all((bits(x) == bits(y)).all() for x, y in zip(a, b)) and np.concatenate(a).mean() > 0
    first = [x.copy() for x in a]
    o.cubemap_bake(a, size, origin, 0.5, sampleIndex=1, seed=2, maxBounces=3)
    assert any((x != y).any() for x, y in zip(a, first))
    cube = np.zeros((6, 4), np.float32)
    n = o.ambcube_bake(cube, origin, 256, 0, sampleIndex=0, seed=2, maxBounces=3)
    assert n == 1 and (cube[:, :3] > 0).all()


def test_cubemap_convolve_on_the_oracle():
    """Cubemap_Convolve's restatement (oracle/pim_oracle.c, rendering/cubemap.c:192-303) against an independent numpy
    evaluation of the same split sum for a handful of texels (same streams, float64 accumulation: agreement to float
    rounding), plus the invariants: a uniform environment stays uniform, the 1 x 1 mip is blended towards zero, threads
    do not change a bit, weight blends."""
    from tests.util import oracle_cubemap_convolve, oracle_cubemap_elem_count
    size, samples = 16, 24
    elems = oracle_cubemap_elem_count(size)
    assert elems == 256 + 64 + 16 + 4 + 1
    rng = np.random.default_rng(1)
    faces = [np.ascontiguousarray(rng.random(size * size * 3).astype(np.float32)) for _ in range(6)]
    a = [np.zeros(elems * 4, np.float32) for _ in range(6)]
    b = [np.zeros(elems * 4, np.float32) for _ in range(6)]
    oracle_cubemap_convolve(faces, a, size, samples, 1.0, sampleIndex=2, seed=7, threads=1)
    oracle_cubemap_convolve(faces, b, size, samples, 1.0, sampleIndex=2, seed=7, threads=8)
    for f in range(6):
        assert a[f].tobytes() == b[f].tobytes()
    A = np.stack([x.reshape(-1, 4) for x in a])
    assert np.isfinite(A).all() and (A[:, elems - 1] == 0).all()                   # the 1 x 1 mip: NaN direction, nothing counted
    assert np.abs(A[:, :elems - 1, 3] - 1.0).max() < 1e-6
    lo, hi = np.concatenate(faces).min(), np.concatenate(faces).max()
    assert (A[:, :elems - 1, :3] >= lo - 1e-6).all() and (A[:, :elems - 1, :3] <= hi + 1e-6).all()   # a weighted mean of bilinear reads
    # rougher mips average over wider lobes: the variance over a mip's texels falls
    var = [A[:, o:o + n, :3].var() for o, n in ((0, 256), (256, 64), (320, 16))]
    assert var[0] > var[1] > var[2]
    # second call with weight 0.25 blends: new = old + (conv - old) * 0.25, conv from other streams (sampleIndex 3)
    c = [x.copy() for x in a]
    fresh = [np.zeros(elems * 4, np.float32) for _ in range(6)]
    oracle_cubemap_convolve(faces, c, size, samples, 0.25, sampleIndex=3, seed=7)
    oracle_cubemap_convolve(faces, fresh, size, samples, 1.0, sampleIndex=3, seed=7)
    for f in range(6):
        want = a[f] + (fresh[f] - a[f]) * np.float32(0.25)
        assert want.tobytes() == c[f].tobytes()
    # uniform environment
    uni = [np.full(size * size * 3, 0.5, np.float32) for _ in range(6)]
    u = [np.zeros(elems * 4, np.float32) for _ in range(6)]
    oracle_cubemap_convolve(uni, u, size, samples, 1.0, seed=1)
    U = np.stack([x.reshape(-1, 4) for x in u])[:, :elems - 1, :3]
    assert np.abs(U - 0.5).max() < 1e-6
