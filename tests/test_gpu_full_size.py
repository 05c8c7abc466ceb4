"""-m gpu: parity at BASELINE config 3's scene size -- the 1M-triangle synthetic scene (seed 1234, 100 m).

The oracle walks its own binned-SAH BVH here (brute force is O(rays x 1M)); that walk is pinned bit-for-bit
against the brute-force loop on the smaller scenes of tests/test_oracle_scene.py.  Checked BIT-EXACT:
  - scene set-up: emissive maps, the 49^3-cell light grid's Dist1D lengths / pdf / cdf (RtcPointQuery walk,
    16 probe rays and 16 x E occlusion rays per cell);
  - >= 1M primary (camera) and >= 1M secondary (surface + 1e-5 N, shuffled) closest hits, SURVEY 8(d) gate (i);
  - an 8-bounce image (rays, autofocus, light histograms included)."""
import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import gpu_init, assert_bits_equal
from tests.util import OracleScene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    gpu_init()
    sc = scenes.synthetic_scene(target_tris=1000000, seed=1234, extent=100.0, emissive_attempts=16)
    g = api.Scene(sc.desc)
    o = OracleScene(sc.desc, accel=1)
    yield sc, g, o
    g.close()


def test_1m_scene_setup_bit_exact(big):
    sc, g, o = big
    gi, oi = g.info(), o.info()
    assert gi.triCount == oi.triCount == sc.tri_count and abs(sc.tri_count - 1000000) <= 2208
    assert gi.emissiveCount == oi.emissiveCount and list(gi.gridSize) == list(oi.gridSize) and gi.activeCells == oi.activeCells
    gl, ol = g.download_lights(), o.download_lights()
    for k in ("vertToEmit", "emitToVert", "cellLength", "pdf", "cdf"):
        assert_bits_equal(gl[k], ol[k], k)


def _primary_rays(sc, n, seed):
    cam = scenes.synthetic_camera(100.0)
    W = 1536; H = (n + W - 1) // W
    rng = np.random.default_rng(seed)
    px = (np.arange(W * H) % W + rng.random(W * H)) / W * 2.0 - 1.0
    py = (np.arange(W * H) // W + rng.random(W * H)) / H * 2.0 - 1.0
    q = np.array(cam.rotation, np.float32)
    eye = np.array(cam.position, np.float32)[:3]
    def rot(v):
        u = q[:3]; s = q[3]
        return 2.0 * np.dot(u, v) * u + (s * s - np.dot(u, u)) * v + 2.0 * s * np.cross(u, v)
    fwd, right, up = rot(np.array([0, 0, -1.0], np.float32)), rot(np.array([1.0, 0, 0], np.float32)), rot(np.array([0, 1.0, 0], np.float32))
    d = fwd[None, :] + (px * (W / H))[:, None] * right[None, :] + py[:, None] * up[None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ro = np.zeros((W * H, 4), np.float32); rd = np.zeros((W * H, 4), np.float32)
    ro[:, :3] = eye; rd[:, :3] = d
    return ro, rd


def test_1m_scene_two_million_hits_bit_exact(big):
    sc, g, o = big
    for name, (ro, rd) in (("primary", _primary_rays(sc, 1 << 20, 3)), ("secondary", scenes.surface_rays(sc, 1 << 20, seed=4, offset=1e-5))):
        assert ro.shape[0] >= 1 << 20
        hg, ho = g.intersect(ro, rd), o.intersect(ro, rd)
        for field in ("type", "iVert", "flags", "wuvt", "normal"):
            assert_bits_equal(hg[field], ho[field], name + " " + field)
        assert (hg["type"] != api.HIT_NOTHING).mean() > 0.99


def test_1m_scene_image_bit_exact(big):
    sc, g, o = big
    W, H = 480, 270
    cam = scenes.synthetic_camera(100.0)
    t = api.Trace(W, H)
    img = [np.zeros((W * H, 3), np.float32) for _ in range(3)]
    dg, do = scenes.default_dof(), scenes.default_dof()
    for s in range(2):
        p = scenes.trace_params(s, seed=23, max_bounces=8)
        t.trace(g, cam, dg, p)
        so = o.trace_image(img, W, H, cam, do, p)
        sg = t.stats()
        assert (sg.extensionRays, sg.shadowRays, sg.otherRays, sg.paths) == (so.extensionRays, so.shadowRays, so.otherRays, so.paths)
        assert dg.focalLength == do.focalLength
    color, albedo, normal = t.download()
    assert_bits_equal(color, img[0], "color"); assert_bits_equal(albedo, img[1], "albedo"); assert_bits_equal(normal, img[2], "normal")
    gl, ol = g.download_lights(), o.download_lights()
    for k in ("live", "pdf", "cdf"):
        assert_bits_equal(gl[k], ol[k], k)
    t.close()


def _frames_bit_exact(g, o, cam, W, H, frames, max_bounces, seed):
    """`frames` accumulated samples of a W x H image on both sides: every pixel of the three images, the ray counts, the
    autofocus state and the light histograms, bit for bit"""
    t = api.Trace(W, H)
    img = [np.zeros((W * H, 3), np.float32) for _ in range(3)]
    dg, do = scenes.default_dof(), scenes.default_dof()
    stats = []
    for s in range(frames):
        p = scenes.trace_params(s, seed=seed, max_bounces=max_bounces)
        t.trace(g, cam, dg, p)
        so = o.trace_image(img, W, H, cam, do, p)
        sg = t.stats()
        assert (sg.extensionRays, sg.shadowRays, sg.otherRays, sg.paths) == (so.extensionRays, so.shadowRays, so.otherRays, so.paths), f"frame {s}"
        assert dg.focalLength == do.focalLength
        stats.append(sg)
    color, albedo, normal = t.download()
    assert_bits_equal(color, img[0], "color"); assert_bits_equal(albedo, img[1], "albedo"); assert_bits_equal(normal, img[2], "normal")
    assert t.stddev() == o.stddev(img[0], W, H)
    gl, ol = g.download_lights(), o.download_lights()
    for k in ("live", "pdf", "cdf"):
        assert_bits_equal(gl[k], ol[k], k)
    t.close()
    return stats


def test_bench_workload_synth1m_4k_default_policy_bit_exact(monkeypatch):
    """bench.py's headline workload at its own size and under the library's OWN tail policy (no switch forced): the 1 M-triangle
    scene exactly as bench.build_scene makes it, 3840 x 2160, 8-bounce clamp, two frames -- the first has no queue-length history
    (every bounce a wavefront bounce, parked paths finished by the tail kernel), the second takes tail_policy's history branch
    (wavefront bounces until fewer than 2.5 M paths are left, then k_tail).  8.3 M paths per frame, 5.4 M-ray launches in
    128-ray dynamic chunks; the oracle needs about 5 s per frame on 16 threads."""
    for k in ("PIMGPU_TAIL_BOUNCE", "PIMGPU_TAIL_FRACTION", "PIMGPU_TAIL_MAX_PATHS", "PIMGPU_PARK", "PIMGPU_NEE_TARGET", "PIMGPU_TRACE_CHUNK", "PIMGPU_GROUPS"):
        monkeypatch.delenv(k, raising=False)
    gpu_init()
    sc = scenes.synthetic_scene(target_tris=1_000_000, seed=1234, extent=100.0, emissive_attempts=64)       # bench.build_scene("synth1m")
    g = api.Scene(sc.desc)
    o = OracleScene(sc.desc, accel=1)
    try:
        st = _frames_bit_exact(g, o, scenes.synthetic_camera(100.0), 3840, 2160, 2, 8, seed=1)
        assert st[0].bounces == 8 and st[0].tailPaths == 0                   # no history: the wavefront ran to the clamp
        assert 0 < st[1].bounces < 8 and 0 < st[1].tailPaths < 2_500_000      # history: the tail kernel took the short queue over
    finally:
        g.close(); o.close()


def test_bench_workload_cornell_1080p_8_bounces_bit_exact(monkeypatch):
    """BASELINE config 2 at its own size: cornell_box, 1920 x 1080, the 8-bounce clamp, two accumulated frames under the default
    policy (2.07 M paths: the whole integrator runs as one k_tail launch)"""
    for k in ("PIMGPU_TAIL_BOUNCE", "PIMGPU_TAIL_FRACTION", "PIMGPU_TAIL_MAX_PATHS", "PIMGPU_PARK", "PIMGPU_NEE_TARGET", "PIMGPU_TAIL_DENSE"):
        monkeypatch.delenv(k, raising=False)
    gpu_init()
    sc = scenes.cornell_box()
    g = api.Scene(sc.desc)
    o = OracleScene(sc.desc, accel=1)
    try:
        st = _frames_bit_exact(g, o, scenes.pt_test_camera(), 1920, 1080, 2, 8, seed=1)
        assert st[1].bounces == 0 and st[1].tailPaths == 1920 * 1080
    finally:
        g.close(); o.close()


def test_embed_full_size_pages_spot_checked_and_idempotent():
    """EmbedAttributes at bake scale (2 pages of 1024^2, 120 k mapped triangles): the oracle's per-texel loop on 3000 texels
    picked at random and on every texel of two rows, bit for bit; sampleCounts is 0 or 1; a second call gives the same pages"""
    import ctypes as C
    from tests import prepare_util as pu
    from tests.util import load_oracle
    gpu_init()
    S, count = 1024, 2
    dr = pu.embed_table(S, [0, 1], tris_per_mesh=(50_000, 70_000), chart_texels=(3.0, 30.0), hard_cases=False, seed=21)
    pages = api.lightmap_embed(dr, pu.new_pages(count, S, fill=-3.0), S)
    again = api.lightmap_embed(dr, pu.new_pages(count, S, fill=-3.0), S)
    rng = np.random.default_rng(4)
    items = np.concatenate([rng.integers(0, count * S * S, 3000), np.arange(517 * S, 518 * S), S * S + np.arange(3 * S, 4 * S)]).astype(np.int64)
    want = pu.new_pages(count, S, fill=-3.0)
    lib = load_oracle(True)
    lib.orc_lightmap_embed_items.restype = None
    lib.orc_lightmap_embed_items(C.byref(dr.desc), None, api._page_array(want, S), C.c_int32(S), items.ctypes.data_as(C.c_void_p),
                                 C.c_int64(items.shape[0]), C.c_int(16))
    covered = 0
    for i in range(count):
        g, a, w = pu.page_parts(pages[i], S), pu.page_parts(again[i], S), pu.page_parts(want[i], S)
        sel = items[(items // (S * S)) == i] % (S * S)
        for k, what in ((1, "position"), (2, "normal"), (3, "sampleCounts")):
            assert_bits_equal(np.ascontiguousarray(g[k][sel]), np.ascontiguousarray(w[k][sel]), f"page {i} {what}")
            assert (g[k].view(np.uint32) == a[k].view(np.uint32)).all()
        assert (g[0] == -3.0).all() and set(np.unique(g[3])) <= {0.0, 1.0}
        nrm = np.linalg.norm(g[2][g[3] > 0], axis=1)
        assert np.abs(nrm - 1.0).max() < 1e-3
        covered += int(w[3][sel].sum())
    assert covered > 500


def test_config4_lightmap_bake_of_the_1m_scene_at_density_16(big):
    """BASELINE config 4 at its real scene size and density: LmPack_Bake (rendering/lightmap.c:1204-1223) of the 1M-triangle
    scene at lm_density 16 into 1024^2 pages.  Bake inputs as SURVEY 8(d) prescribes (one chart per triangle, the packer
    is outside rows a-e): the two inward-facing floor triangles (a page each: 1600 texels clamped to the page) and every
    37th triangle of the objects (a third page of small charts), 1.47 M mapped texels.  EVERY probe and sample count of
    EVERY page against the oracle's BakeFn, BIT-EXACT (portable-libm variant), over two progressive passes with
    UpdateDists between them (the light histograms of pass 0 steer pass 1)."""
    sc, g, o = big
    S, tpm, spp = 1024, 16.0, 2
    L = S * S
    P = sc.positions[:84 * 3, :3].reshape(-1, 3, 3).astype(np.float64)          # the room: 7 boxes
    n = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    inner = np.nonzero((0.5 * np.linalg.norm(n, axis=1) > 1000.0) & ((n * -P.mean(1)).sum(1) > 0))[0]
    assert inner.size == 12
    tris = list(inner[:2]) + list(range(84, sc.tri_count, 37))
    pages = scenes.lightmap_pages_for(sc, S, tpm, max_pages=3, tri_indices=tris)
    assert len(pages) == 3
    mapped0 = [p[L * 26:].copy() for p in pages]
    assert sum(int((m > 0).sum()) for m in mapped0) > 1_000_000
    pg = [p.copy() for p in pages]
    po = [p.copy() for p in pages]
    o.threads = 16
    for passIndex in range(2):
        kw = dict(passIndex=passIndex, seed=7, maxBounces=8)
        g.lightmap_bake(pg, S, tpm, 1.0, spp, **kw)
        o.lightmap_bake(po, S, tpm, 1.0, spp, **kw)
        for ip in range(len(pages)):
            assert_bits_equal(pg[ip], po[ip], f"pass {passIndex} page {ip}")
            want = np.where(mapped0[ip] > 0, mapped0[ip] + spp * (passIndex + 1), 0.0).astype(np.float32)
            assert_bits_equal(pg[ip][L * 26:], want, f"pass {passIndex} page {ip} sampleCounts")
            assert_bits_equal(pg[ip][L * 20:L * 26], pages[ip][L * 20:L * 26], f"page {ip} position/normal untouched")
    gl, ol = g.download_lights(), o.download_lights()
    for k in ("live", "pdf", "cdf"):
        assert_bits_equal(gl[k], ol[k], k)
    lit = pg[2][:L * 4].reshape(L, 4)[mapped0[2] > 0]
    assert np.isfinite(lit).all() and (lit[:, :3] > 0).mean() > 0.2
