"""-m gpu: the BVH builder (Morton sort -> clustering rounds reported to the host without draining the stream -> collapse ->
compressed nodes) on the inputs that sit on its thresholds or defeat its heuristics, each under the library's structural
validation (PIMGPU_VALIDATE_BVH: every triangle once, boxes contain their triangles, the compressed copy is the same tree)
and against the oracle's brute-force closest hit, bit for bit; rebuilt once, so the second build runs on warm scratch.

  sizes      2, 3 (the smallest real tree), 511 / 512 / 513 (the finishing workgroup takes at most 512 clusters), 1023 / 1025
             (one clustering block), 4097
  geometry   all triangles the same triangle (one Morton code, every box equal: ties everywhere); points (zero-area triangles);
             triangles along one line (two Morton axes empty); a compact cloud plus one triangle 10^4 cloud diameters away (the
             quantised codes of the cloud collapse into one cell)"""
import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import gpu_init, assert_bits_equal
from tests.util import OracleScene

pytestmark = pytest.mark.gpu


def _scene(P):
    """P: (n, 3, 3) triangle corners"""
    n = P.shape[0]
    pos = np.zeros((n * 3, 4), np.float32); pos[:, :3] = P.reshape(-1, 3); pos[:, 3] = 1.0
    e1, e2 = P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]
    nrm = np.cross(e1, e2); ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = np.where(ln > 0, nrm / np.maximum(ln, 1e-30), np.array([0.0, 1.0, 0.0]))
    nor = np.zeros((n * 3, 4), np.float32); nor[:, :3] = np.repeat(nrm, 3, axis=0)
    uv = np.zeros((n * 3, 2), np.float32)
    b = scenes.SceneBuilder(seed=1, emissive_attempts=4)
    m = b.material((0.5, 0.5, 0.5, 1.0), (0.9, 1.0, 0.0, 0.0))
    b.add_arrays(pos, nor, uv, m)
    return b.build()


def _rays(P, n, seed):
    """towards points of the triangles from outside their bounds, plus random ones"""
    rng = np.random.default_rng(seed)
    lo, hi = P.reshape(-1, 3).min(0), P.reshape(-1, 3).max(0)
    c, r = 0.5 * (lo + hi), max(1e-3, float(np.linalg.norm(hi - lo)))
    tri = rng.integers(0, P.shape[0], n)
    w = rng.dirichlet([1, 1, 1], n)
    target = (P[tri] * w[:, :, None]).sum(1)
    d0 = rng.normal(size=(n, 3)); d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    origin = c + d0 * r * 1.5
    d = target - origin; d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    d[::5] = rng.normal(size=d[::5].shape); d[::5] /= np.linalg.norm(d[::5], axis=1, keepdims=True)
    ro = np.zeros((n, 4), np.float32); rd = np.zeros((n, 4), np.float32)
    ro[:, :3] = origin; rd[:, :3] = d
    return ro, rd


def _check(P, rays, monkeypatch, seed=1):
    gpu_init()
    monkeypatch.setenv("PIMGPU_VALIDATE_BVH", "1")
    sc = _scene(P.astype(np.float32))
    g = api.Scene(sc.desc)
    o = OracleScene(sc.desc, accel=0)
    try:
        ro, rd = _rays(P.astype(np.float64), rays, seed)
        want = o.intersect(ro, rd)
        for attempt in range(2):
            got = g.intersect(ro, rd)
            for field in ("type", "iVert", "flags", "wuvt", "normal"):
                assert_bits_equal(got[field], want[field], f"{field} (build {attempt})")
            g.rebuild_bvh()
        return got
    finally:
        g.close(); o.close()


def _cloud(n, seed, extent=10.0, size=0.3):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n, 1, 3))
    return c + rng.normal(scale=size, size=(n, 3, 3))


@pytest.mark.parametrize("n", [2, 3, 511, 512, 513, 1023, 1025, 4097])
def test_sizes_on_the_builders_thresholds(n, monkeypatch):
    got = _check(_cloud(n, seed=n), 3000, monkeypatch, seed=n)
    assert (got["type"] != api.HIT_NOTHING).mean() > 0.5


@pytest.mark.parametrize("copies", [1500, 60000])
def test_every_triangle_the_same(copies, monkeypatch):
    """(with "the lower index wins" among equal areas one pair merged per round: 1500 rounds, a chain deeper than the collapse
    takes; ploc_better's symmetric rule halves such a run every round)"""
    tri = np.array([[[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]])
    got = _check(np.repeat(tri, copies, axis=0), 2000, monkeypatch)
    hit = got["type"] != api.HIT_NOTHING
    assert hit.any() and (got["iVert"][hit] == 0).all()          # equal t everywhere: the lowest primID wins


def test_zero_area_triangles(monkeypatch):
    rng = np.random.default_rng(2)
    pts = rng.uniform(-5, 5, (900, 1, 3))
    P = np.repeat(pts, 3, axis=1)                                  # points
    P[::3] = _cloud(300, seed=4, extent=5.0)                       # ... among real triangles
    _check(P, 3000, monkeypatch)


def test_triangles_along_one_line(monkeypatch):
    n = 2000
    x = np.linspace(-50, 50, n)[:, None, None]
    P = np.zeros((n, 3, 3)); P[:, :, 0] = x[:, :, 0]
    P[:, 1, 1] = 0.04; P[:, 2, 2] = 0.04; P[:, 1, 0] += 0.01
    _check(P, 3000, monkeypatch)


def test_one_triangle_far_from_a_compact_cloud(monkeypatch):
    P = _cloud(3000, seed=9, extent=0.0006, size=0.00003)
    P[1234] += 12.0                                                # 13-bit Morton cells of 0.0015: the cloud is a handful of them
    _check(P, 3000, monkeypatch)
