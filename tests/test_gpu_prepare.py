"""-m gpu: FlattenDrawables (rendering/path_tracer.c:692-782) and BakeBrdfFn (math/lighting.c:31-84) on the GPU
against the oracle, BIT-EXACT; then a scene built from the GPU-flattened arrays renders."""
import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import gpu_init, assert_bits_equal
from tests.util import load_oracle
from tests.prepare_util import draw_table, oracle_flatten, expected_counts, oracle_brdf_lut

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("count,holes", [(23, True), (400, True), (1, False), (0, False)])
def test_flatten_bit_exact(count, holes):
    gpu_init()
    dr = draw_table(count=count, seed=11 + count, with_holes=holes)
    V, Mc = expected_counts(dr)
    assert api.flatten_counts(dr) == (V, Mc)
    P, N, U, ids, mats = api.flatten(dr)
    oP, oN, oU, oids, omats = oracle_flatten(load_oracle(portable=True), dr, V, Mc)
    assert_bits_equal(P, oP, "positions"); assert_bits_equal(N, oN, "normals"); assert_bits_equal(U, oU, "uvs")
    assert (ids == oids).all()
    assert bytes(mats)[:48 * Mc] == bytes(omats)[:48 * Mc]


def test_flatten_many_instances_of_one_mesh():
    """the 1M-triangle scene's shape: thousands of entities over two meshes (each mesh is uploaded once)"""
    gpu_init()
    rng = np.random.default_rng(3)
    from tests.prepare_util import mesh4
    meshes = [mesh4(scenes.box_mesh()), mesh4(scenes.sphere_mesh(24))]
    count = 300
    ids = rng.integers(0, 2, count).astype(np.int32)
    M = np.zeros((count, 4, 4), np.float32)
    for i in range(count):
        q = scenes.quat_lookat(scenes.normalize3(rng.normal(size=3).astype(np.float32)), np.array([0, 1, 0], np.float32))
        c = scenes.trs_matrix(rng.uniform(-50, 50, 3).astype(np.float32), q, rng.uniform(0.5, 2, 3).astype(np.float32))
        for k in range(4):
            M[i, k, :3] = c[k]
        M[i, 3, 3] = 1.0
    dr = api.Drawables(meshes, ids, M, [api.PimMaterial() for _ in range(count)])
    V, Mc = api.flatten_counts(dr)
    assert V == int((ids == 0).sum()) * 36 + int((ids == 1).sum()) * 2208 * 3 and V > 300000
    P, N, U, mids, _ = api.flatten(dr)
    oP, oN, oU, oids, _ = oracle_flatten(load_oracle(portable=True), dr, V, Mc)
    assert_bits_equal(P, oP, "positions"); assert_bits_equal(N, oN, "normals"); assert_bits_equal(U, oU, "uvs")
    assert (mids == oids).all()


def test_flatten_bad_arguments():
    gpu_init()
    dr = draw_table(count=4, with_holes=False)
    dr.mesh_ids[0] = 99
    with pytest.raises(api.PimGpuError):
        api.flatten_counts(dr)


@pytest.mark.parametrize("sx,sy,passes", [(32, 32, [(0, 64)]), (128, 128, [(0, 16), (16, 26), (26, 36)]), (5, 9, [(0, 3), (3, 3)])])
def test_brdf_lut_bake_bit_exact(sx, sy, passes):
    gpu_init()
    t = np.zeros((sy, sx, 2), np.float32)
    for prev, num in passes:
        api.brdf_lut_bake(t, prev, num, seed=21)
    o = oracle_brdf_lut(load_oracle(portable=True), sx, sy, passes, seed=21)
    assert_bits_equal(t.reshape(-1, 2), o.reshape(-1, 2), "BRDF table")
    assert np.isfinite(t).all() and t.max() > 0.5


def _desc_with(sc, P, N, U, ids, mats, count, lut):
    import copy
    import ctypes
    d = api.PimSceneDesc.from_buffer_copy(bytes(sc.desc))
    d.vertCount, d.matCount = P.shape[0], count
    d.positions, d.normals, d.uvs, d.matIds = P.ctypes.data, N.ctypes.data, U.ctypes.data, ids.ctypes.data
    d.materials = ctypes.cast(mats, ctypes.c_void_p)
    d.brdfLut, d.brdfLutSize = lut.ctypes.data, lut.shape[0]
    return d


def test_gpu_prepared_inputs_render_bit_exact():
    """FlattenDrawables -> PtScene_New -> Pt_Trace with both producers on the GPU: Cornell's draw table flattened and its
    BRDF table baked on the GPU give a scene whose image equals the oracle's on the same inputs, and whose vertices are
    the harness' own (numpy) flatten up to its last-bit differences"""
    gpu_init()
    sc = scenes.cornell_box()
    dr = scenes.drawables_of(sc)
    P, N, U, ids, mats = api.flatten(dr)
    assert P.shape == sc.positions.shape and np.allclose(P, sc.positions, rtol=0, atol=4e-6)
    assert np.allclose(N, sc.normals, rtol=0, atol=4e-6) and (U == sc.uvs).all()
    lut = np.zeros((32, 32, 2), np.float32)
    api.brdf_lut_bake(lut, 0, 64, seed=7)
    d = _desc_with(sc, P, N, U, ids, mats, len(sc.entities), lut)
    g = api.Scene(d)
    from tests.util import OracleScene
    o = OracleScene(d, accel=1)
    W, H = 96, 54
    tr = api.Trace(W, H)
    img = [np.zeros((W * H, 3), np.float32) for _ in range(3)]
    dof_g, dof_o = scenes.default_dof(), scenes.default_dof()
    for s in range(2):
        tr.trace(g, scenes.pt_test_camera(), dof_g, scenes.trace_params(s, max_bounces=4))
        o.trace_image(img, W, H, scenes.pt_test_camera(), dof_o, scenes.trace_params(s, max_bounces=4))
    assert_bits_equal(tr.download()[0], img[0], "colour of the GPU-prepared scene")
    assert img[0].mean() > 0.01
    g.close()


# ---------------------------------------------------------------------------------------------
# EmbedAttributes (rendering/lightmap.c:941-1045)

def _embed_both(dr, count, S, ids=None, fill=-3.0):
    gpu_init()
    from tests import prepare_util as pu
    got = api.lightmap_embed(dr, pu.new_pages(count, S, fill), S, ids)
    want = pu.oracle_embed(load_oracle(True), dr, pu.new_pages(count, S, fill), S, ids)
    for i in range(count):
        g, w = pu.page_parts(got[i], S), pu.page_parts(want[i], S)
        assert (g[0] == fill).all()                                           # the probes are not touched
        assert_bits_equal(g[3], w[3], f"page {i} sampleCounts")
        # NaN (a fragment of a zero-area triangle divides 0 by 0): only its sign/payload may differ
        for k, what in ((1, "position"), (2, "normal")):
            nan = np.isnan(w[k])
            assert (np.isnan(g[k]) == nan).all(), f"page {i} {what}: NaN pattern"
            assert_bits_equal(np.where(nan, 0.0, g[k]).astype(np.float32), np.where(nan, 0.0, w[k]).astype(np.float32), f"page {i} {what}")
    return got


@pytest.mark.parametrize("S,ids,tris,hard", [(64, [0, 1], (60, 200), True), (48, [9, 4, 7], (300, 500), True),
                                             (256, [0, 1], (3000, 2000), False), (16, [5], (12, 12), True)])
def test_lightmap_embed_bit_exact(S, ids, tris, hard):
    """nearest triangle within the padding, first in drawable/vertex order on ties, duplicates, zero-area triangles
    (which flood their page, as in the reference), page ids that name no page, skipped drawables"""
    from tests import prepare_util as pu
    dr = pu.embed_table(S, ids, tris_per_mesh=tris, hard_cases=hard)
    pages = _embed_both(dr, len(ids), S, None if ids == list(range(len(ids))) else ids)
    cover = [pu.page_parts(p, S)[3].mean() for p in pages]
    assert max(cover) > 0.05


def test_lightmap_embed_empty_and_bad_arguments():
    from tests import prepare_util as pu
    gpu_init()
    S = 32
    dr = pu.embed_table(S, [77], tris_per_mesh=(20, 20), hard_cases=False)      # ids of the triangles name no page
    pages = _embed_both(dr, 2, S)
    assert all(not p[S * S * 20:].any() for p in pages)                          # nothing mapped: zeros, as Lightmap_New's calloc
    api.lightmap_embed(dr, [], S)                                                 # no pages: nothing to do
    import ctypes as C
    lib = api.load_library()
    arr = api._page_array(pages, S)
    E_INVALID = -1                                                                # include/pimgpu.h
    assert lib.pimgpu_lightmap_embed(None, None, arr, C.c_int32(2), C.c_int32(S)) == E_INVALID
    assert lib.pimgpu_lightmap_embed(C.byref(dr.desc), None, arr, C.c_int32(2), C.c_int32(0)) == E_INVALID
    assert lib.pimgpu_lightmap_embed(C.byref(dr.desc), None, None, C.c_int32(2), C.c_int32(S)) == E_INVALID
    dr.mesh_ids[0] = 99
    with pytest.raises(api.PimGpuError):
        api.lightmap_embed(dr, pages, S)


def test_lightmap_embed_feeds_the_bake():
    """embed -> bake: the pages EmbedAttributes fills are LmPack_Bake's input (lightmap.c:1140-1152); the GPU chain
    equals the oracle chain bit for bit"""
    from tests import prepare_util as pu
    from tests.util import OracleScene
    sc = scenes.cornell_box()
    S, tpm = 64, 2.0
    dr, count = pu.world_mesh_with_charts(sc, S, tpm)
    pg = _embed_both(dr, count, S, fill=0.0)
    po = pu.oracle_embed(load_oracle(True), dr, pu.new_pages(count, S), S)
    mapped = sum(int(pu.page_parts(p, S)[3].sum()) for p in pg)
    assert mapped > 500
    g = api.Scene(sc.desc); o = OracleScene(sc.desc, accel=1)
    g.lightmap_bake(pg, S, tpm, 1.0, 2, passIndex=0, seed=5, maxBounces=3)
    o.lightmap_bake(po, S, tpm, 1.0, 2, passIndex=0, seed=5, maxBounces=3)
    for a, b in zip(pg, po):
        assert_bits_equal(a, b, "baked page")
    lit = sum(int((pu.page_parts(p, S)[0].reshape(5, S * S, 4)[:, :, :3].sum(axis=(0, 2)) > 0).sum()) for p in pg)
    assert lit > 100
    g.close()


# ---------------------------------------------------------------------------------------------
# chartnodes_create, ChartMaskFn (rendering/lightmap.c:723-767, 527-561)

@pytest.mark.parametrize("count,tpu,sizes", [(9, 3.0, (1, 2, 5, 40)), (40, 1.5, (1, 3, 17)), (3, 8.0, (1,))])
def test_chartnodes_and_masks_bit_exact(count, tpu, sizes):
    from tests import prepare_util as pu
    gpu_init()
    lib = load_oracle(True)
    dr = pu.draw_table(count=count, seed=count)
    want = pu.oracle_chartnodes(lib, dr, tpu)
    got = api.lightmap_chartnodes(dr, tpu)
    assert got.shape == want.shape and got.shape[0] > 0
    assert_bits_equal(got.view(np.uint32).reshape(got.shape[0], -1), want.view(np.uint32).reshape(want.shape[0], -1), "chart nodes")
    first = pu.chart_partition(got.shape[0], seed=count, sizes=sizes)
    nodes_o = want.copy()
    ao, so, oo, mo = pu.oracle_chart_masks(lib, nodes_o, first)
    ag, sg, og, mg = api.lightmap_chart_masks(got, first)
    assert_bits_equal(got.view(np.uint32).reshape(got.shape[0], -1), nodes_o.view(np.uint32).reshape(got.shape[0], -1), "moved nodes")
    assert_bits_equal(ag, ao, "chart areas")
    assert (sg == so).all() and (og == oo).all()
    assert mg.shape == mo.shape and (mg == mo).all() and (mg == 255).any()


def test_chart_masks_sliver_floods_and_bad_arguments():
    """a zero-area triangle: sdTriangle2D calls every texel inside, so mask_tri sets the whole mask (as the reference)"""
    from tests import prepare_util as pu
    gpu_init()
    lib = load_oracle(True)
    nodes = np.zeros(3, api.CHARTNODE_DTYPE)
    nodes["triCoord"][0] = [0, 0, 9, 3, 2, 8]
    nodes["triCoord"][1] = [20, 20, 26, 20, 32, 20]                              # collinear
    nodes["triCoord"][2] = [5, 5, 5, 5, 40, 30]                                  # zero-length edge
    first = np.asarray([0, 1, 2, 3], np.int32)
    want = pu.oracle_chart_masks(lib, nodes.copy(), first)
    got = api.lightmap_chart_masks(nodes, first)
    assert (got[1] == want[1]).all() and (got[3] == want[3]).all()
    a1 = got[3][got[2][1]:got[2][2]]
    assert (a1 == 255).all()                                                      # the collinear triangle's mask is flooded
    with pytest.raises(api.PimGpuError):
        api.lightmap_chart_masks(nodes, np.asarray([0, 2, 2, 3], np.int32))      # an empty chart
