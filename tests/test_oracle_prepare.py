"""CPU: the oracle's restatements of FlattenDrawables (rendering/path_tracer.c:692-782) over a whole draw table and of
BakeBrdfFn (math/lighting.c:31-84).

Pinned by the reference: the per-vertex arithmetic through the `trs_flatten` golden record (f4x4_trs, f4x4_mul_pt,
f3x3_IM, f4_normalize3 from the reference's headers) and one LUT sample through `brdf_lut_sample`; both in
tests/test_oracle_golden.py.  Checked here: table handling (skipped entities, offsets, uv tails, material copy) and
the statistics of the whole LUT bake (parity unpinned as a whole: BakeBrdfFn is a static function of a translation
unit that cannot be compiled here, and the reference's table is seeded from the clock)."""
import numpy as np

from pim_amd import scenes
from tests.util import load_oracle
from tests.prepare_util import draw_table, oracle_flatten, expected_counts, oracle_brdf_lut


def test_flatten_table_layout_and_values():
    lib = load_oracle()
    dr = draw_table()
    V, Mc = expected_counts(dr)
    P, N, U, ids, mats = oracle_flatten(lib, dr, V, Mc)
    back, k = 0, 0
    for i, m in enumerate(dr.mesh_ids):
        if m < 0:
            continue
        n = dr.table[m].length
        assert mats[k].albedo == i and mats[k].rome == i + 100 and abs(mats[k].ior - (1.0 + 0.01 * i)) < 1e-6
        assert (ids[back:back + n] == k).all()
        if n:
            mp, mn, mu = dr.keep[3 * m], dr.keep[3 * m + 1], dr.keep[3 * m + 2]
            c = dr.matrices[i].reshape(4, 4).astype(np.float64)
            want = mp[:, 0:1] * c[0] + mp[:, 1:2] * c[1] + mp[:, 2:3] * c[2] + c[3]
            assert np.allclose(P[back:back + n], want, rtol=0, atol=2e-5) and (P[back:back + n, 3] == 1.0).all()
            im = np.linalg.inv(c[:3, :3]).T                         # c[:3,:3] is transpose(M3x3) as a matrix; row vectors: n @ inv^T
            wn = mn[:, :3].astype(np.float64) @ im
            wn /= np.linalg.norm(wn, axis=1, keepdims=True)
            assert np.allclose(N[back:back + n, :3], wn, rtol=0, atol=1e-5)
            whole = (n // 3) * 3
            assert (U[back:back + whole] == mu[:whole, :2]).all()
            assert (U[back + whole:back + n] == 0.0).all()           # the tail of a ragged mesh keeps Perm_Calloc's zeroes
        back += n; k += 1
    assert back == V and k == Mc


def test_flatten_matches_the_single_entity_restatement():
    """orc_flatten over a table == the per-entity restatement the scenes are built with (scenes.flatten_entity)"""
    lib = load_oracle()
    dr = draw_table(count=6, seed=9, with_holes=False)
    V, Mc = expected_counts(dr)
    P, N, U, ids, _ = oracle_flatten(lib, dr, V, Mc)
    rng = np.random.default_rng(9)
    rng.integers(0, 4, 6)
    back = 0
    box, sph = scenes.box_mesh(), scenes.sphere_mesh(6)
    for i, m in enumerate(dr.mesh_ids):
        t = rng.uniform(-30, 30, 3).astype(np.float32)
        q = scenes.quat_lookat(scenes.normalize3(rng.normal(size=3).astype(np.float32)), np.array([0, 1, 0], np.float32))
        s = rng.uniform(0.2, 3.0, 3).astype(np.float32)
        n = dr.table[m].length
        if n:
            mesh = (box, sph, tuple(a[:34] for a in box))[m]
            eP, eN, _ = scenes.flatten_entity(t, q, s, mesh)
            assert np.allclose(P[back:back + n], eP, rtol=0, atol=4e-6) and np.allclose(N[back:back + n], eN, rtol=0, atol=4e-6)
        back += n


def test_flatten_empty_table():
    lib = load_oracle()
    dr = draw_table(count=0)
    assert expected_counts(dr) == (0, 0)
    oracle_flatten(lib, dr, 0, 0)


def test_brdf_lut_statistics_and_progressive_blend():
    lib = load_oracle()
    a = oracle_brdf_lut(lib, 32, 32, [(0, 256)], seed=7)
    assert np.isfinite(a).all() and (a >= 0).all() and a[..., 1].max() < 4.0       # grazing texels are noisy at 256 samples
    # Fc <= 1, so x <= y; a mirror-smooth surface seen head on reflects about everything back (DFV.y -> 1)
    assert (a[..., 0] <= a[..., 1] + 1e-6).all()
    assert abs(a[1, 31, 1] - 1.0) < 0.05
    # same estimator as the harness' float64 numpy bake of the table (scenes.brdf_lut), different stream
    ref = scenes.brdf_lut(32, 256, seed=3).reshape(32, 32, 2)
    assert abs(a[4:, 4:].mean() - ref[4:, 4:].mean()) < 0.01
    assert np.abs(a[4:, 4:] - ref[4:, 4:]).mean() < 0.03
    # BrdfLut_Update (lighting.c:107-118): two passes blend by dt = delta/num and stay an average of the samples
    b = oracle_brdf_lut(lib, 32, 32, [(0, 128), (128, 256)], seed=7)
    assert np.abs(a[4:, 4:] - b[4:, 4:]).mean() < 0.03 and not np.array_equal(a, b)
    # nothing to add: untouched
    c = b.copy()
    lib.orc_brdf_lut_bake(c.ctypes.data, 32, 32, 256, 256, 7)
    assert np.array_equal(b, c)


def test_embed_matches_the_pairwise_golden_function():
    """orc_lightmap_embed (the reference's texel x drawable x triangle loop, lightmap.c:960-1031) against a
    re-derivation from the per-pair function the golden vectors pin (lm_embed_fragment): nearest triangle below the
    padding, the FIRST in drawable/vertex order on equal distance, probes untouched, unmapped drawables skipped"""
    import ctypes as C
    from tests import prepare_util as pu
    lib = load_oracle()
    S, ids = 24, [7, 3]
    dr = pu.embed_table(S, ids, tris_per_mesh=(20, 30), chart_texels=(1.5, 6.0))
    pages = pu.oracle_embed(lib, dr, pu.new_pages(2, S, fill=-3.0), S, page_tex_ids=ids, threads=3)
    hook = lib.orc_hook_lm_embed_fragment
    hook.restype = None
    # the reference's candidate order: drawables in table order (sky flag, missing mesh, no texIndices skipped), triangles in vertex order
    cands = []
    for e, m in enumerate(dr.mesh_ids):
        if m < 0 or dr.materials[e].flags & (2 | 16) or not dr.table[m].texIndices:
            continue
        arrays = dr.mesh_arrays[m]
        for t in range(arrays[0].shape[0] // 3):
            cands.append((arrays, t))
    assert len(cands) == 20 + 30 + 20                                        # mesh 0 is drawn twice; entity 5 (mesh 1) is sky
    inp = np.zeros(27, np.float32); out = np.zeros(8, np.uint32)
    for page, tex_id in enumerate(ids):
        probes, pos, nor, cnt = pu.page_parts(pages[page], S)
        assert (probes == -3.0).all()
        for texel in range(0, S * S, 5):
            px = (texel % S + 0.5, texel // S + 0.5)
            best = None
            for (P, N, U, T), t in cands:
                if T[t * 3, 3] != tex_id:
                    continue
                inp[0:6] = U[t * 3:t * 3 + 3, 2:].reshape(6); inp[6] = S; inp[7:9] = px
                inp[9:18] = P[t * 3:t * 3 + 3, :3].reshape(9); inp[18:27] = N[t * 3:t * 3 + 3, :3].reshape(9)
                hook(inp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
                o = out.view(np.float32)
                if o[0] < 2.0 and (best is None or o[0] < best[0]):
                    best = o.copy()
            if best is None:
                assert cnt[texel] == 0.0 and not pos[texel].any() and not nor[texel].any()
            else:
                assert cnt[texel] == 1.0
                np.testing.assert_array_equal(pos[texel], best[2:5]); np.testing.assert_array_equal(nor[texel], best[5:8])
    # the zero-length edge of the hard cases has zero area: sdTriangle2D then calls every texel inside (negative
    # distance), so its page is flooded -- the reference's behaviour, kept
    assert max(pu.page_parts(p, S)[3].mean() for p in pages) == 1.0
    plain = pu.oracle_embed(lib, pu.embed_table(S, ids, tris_per_mesh=(20, 30), chart_texels=(1.5, 6.0), hard_cases=False),
                            pu.new_pages(2, S), S, page_tex_ids=ids)
    assert all(0.05 < pu.page_parts(p, S)[3].mean() < 0.95 for p in plain)
