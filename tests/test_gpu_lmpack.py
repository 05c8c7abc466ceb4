"""-m gpu: the rest of LmPack_Pack (rendering/lightmap.c:1047-1102): the atlas search on the GPU (pimgpu_lightmap_atlas_place)
against the oracle's one-worker restatement of AtlasFn / atlas_search / mask_find / mask_fits / mask_write, and the whole
packer through the host layer's LmPack_Pack (chartnodes -> groups -> masks -> sort -> placement -> pages -> uvs / ids in the
meshes -> EmbedAttributes) against the oracle's chain, bit for bit, then one bake pass on the pages it made."""
import ctypes as C
import os

import numpy as np
import pytest

from pim_amd import api, scenes
from tests import prepare_util as pu
from tests.gpu_util import gpu_init, assert_bits_equal
from tests.util import load_oracle, OracleScene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def oracle_place(masks, offsets, sizes, areas, order, atlas_size, atlas_count, atlas_cap=0):
    lib = load_oracle(True)
    lib.orc_lightmap_atlas_place.restype = C.c_int32
    nc = order.shape[0]
    tr, idx = np.full((nc, 2), -7, np.int32), np.full(nc, -7, np.int32)
    used = lib.orc_lightmap_atlas_place(p(masks), p(offsets), p(sizes), p(areas), p(order), C.c_int32(nc), C.c_int32(atlas_size),
                                        C.c_int32(atlas_count), C.c_int32(atlas_cap), p(tr), p(idx))
    return tr, idx, used


def oracle_sort(areas):
    lib = load_oracle(True)
    lib.orc_lightmap_chart_sort.restype = None
    order = np.zeros(max(1, areas.shape[0]), np.int32)
    lib.orc_lightmap_chart_sort(p(areas), C.c_int32(areas.shape[0]), p(order))
    return order[:areas.shape[0]]


def random_masks(n, lo, hi, seed, fill=0.8):
    """chart masks as ChartMaskFn leaves them in kind: a blob with a 2-texel empty margin, some with holes"""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(lo, hi + 1, (n, 2)).astype(np.int32)
    lens = sizes[:, 0].astype(np.int64) * sizes[:, 1]
    offsets = np.zeros(n, np.int64); offsets[1:] = np.cumsum(lens)[:-1]
    masks = np.zeros(int(lens.sum()), np.uint8)
    for c in range(n):
        w, h = int(sizes[c, 0]), int(sizes[c, 1])
        m = (rng.random((h, w)) < fill)
        m[:1, :] = False; m[-1:, :] = False; m[:, :1] = False; m[:, -1:] = False
        if c % 3 == 0:
            m = np.triu(m)                                    # triangles leave room for the next chart inside their bounds
        masks[offsets[c]:offsets[c] + w * h] = np.where(m, 255, 0).astype(np.uint8).reshape(-1)
    areas = ((sizes[:, 0] - 4).clip(1) * (sizes[:, 1] - 4).clip(1)).astype(np.float32) * rng.uniform(0.9, 1.0, n).astype(np.float32)
    return masks, offsets, sizes, areas


@pytest.mark.parametrize("form", ["bits", "bytes"])
@pytest.mark.parametrize("n,lo,hi,S,seed", [(300, 5, 14, 64, 1), (120, 8, 40, 128, 2), (2000, 5, 9, 96, 3), (40, 30, 90, 256, 4), (60, 20, 400, 1024, 5), (30, 3, 150, 200, 6)])
def test_atlas_place_matches_one_worker(n, lo, hi, S, seed, form, monkeypatch):
    """both forms of the search: the atlas as bits in LDS (atlases up to 1024^2; charts whose bit rows fit LDS and, at 400 x 400,
    one kept in HBM) and byte for byte in HBM (larger atlases)"""
    gpu_init()
    if form == "bytes":
        monkeypatch.setenv("PIMGPU_ATLAS_BYTES", "1")
    masks, offsets, sizes, areas = random_masks(n, lo, hi, seed)
    order = oracle_sort(areas)
    lib = load_oracle(True)
    lib.orc_lightmap_atlas_estimate.restype = C.c_int32
    # the estimate counts bounds areas without the margins and runs short: further atlases are opened where the reference would spin
    count = lib.orc_lightmap_atlas_estimate(p(areas), C.c_int32(n), C.c_int32(S))
    want = oracle_place(masks, offsets, sizes, areas, order, S, count, count * 4 + 16)
    got = api.lightmap_atlas_place(masks, offsets, sizes, areas, order, S, count, count * 4 + 16)
    assert want[2] >= count and got[2] == want[2]
    assert (got[1] == want[1]).all() and (got[0] == want[0]).all()
    assert got[1].max() == got[2] - 1 and got[1].min() == 0          # atlases fill from the first one up
    # nothing overlaps: rebuild the atlases from the answer
    atlases = np.zeros((got[2], S, S), np.uint8)
    for c in range(n):
        w, h = int(sizes[c, 0]), int(sizes[c, 1])
        m = masks[offsets[c]:offsets[c] + w * h].reshape(h, w)
        x, y = int(got[0][c, 0]), int(got[0][c, 1])
        view = atlases[got[1][c], y:y + h, x:x + w]
        assert view.shape == (h, w) and not (view & m).any()
        view |= m


@pytest.mark.parametrize("n,lo,hi,S,seed,fill", [(600, 60, 68, 512, 7, 0.8), (400, 62, 66, 512, 10, 1.0), (800, 14, 18, 256, 8, 0.9), (500, 30, 34, 320, 9, 1.0),
                                                 (3000, 5, 20, 1024, 11, 1.0), (1500, 3, 64, 1000, 12, 0.7)])
def test_atlas_place_on_the_word_forms_thresholds(n, lo, hi, S, seed, fill):
    """the bit form tests 64 candidate columns per word for masks of up to 64 x 64 texels, with 16, 32 or 64 lanes per candidate
    row: masks around 16 / 32 / 64 rows and 64 columns (a run of set bits that ends at bit 63; one column more takes the form for
    wide masks), solid masks (one run per row) and ragged ones (several), an atlas row of 15.6 words, thousands of charts in one
    atlas (rounds of several candidate rows, resets of the hints as the areas fall)"""
    gpu_init()
    masks, offsets, sizes, areas = random_masks(n, lo, hi, seed, fill=fill)
    order = oracle_sort(areas)
    lib = load_oracle(True)
    lib.orc_lightmap_atlas_estimate.restype = C.c_int32
    count = lib.orc_lightmap_atlas_estimate(p(areas), C.c_int32(n), C.c_int32(S))
    want = oracle_place(masks, offsets, sizes, areas, order, S, count, count * 4 + 16)
    got = api.lightmap_atlas_place(masks, offsets, sizes, areas, order, S, count, count * 4 + 16)
    assert got[2] == want[2]
    assert (got[1] == want[1]).all() and (got[0] == want[0]).all()


def test_atlas_place_reports_what_the_reference_spins_on():
    gpu_init()
    masks, offsets, sizes, areas = random_masks(12, 6, 12, 5)
    sizes2 = sizes.copy()
    order = oracle_sort(areas)
    # one atlas that is too small for all of them and no more allowed: some chart fits nowhere
    want = oracle_place(masks, offsets, sizes, areas, order, 20, 1)
    assert want[2] == -1
    with pytest.raises(api.PimGpuError, match="fits in none"):
        api.lightmap_atlas_place(masks, offsets, sizes, areas, order, 20, 1)
    # a mask as large as the atlas can never be placed (mask_find's ranges are exclusive, lightmap.c:263-271)
    big = np.full(32 * 32, 255, np.uint8)
    with pytest.raises(api.PimGpuError, match="fits in none"):
        api.lightmap_atlas_place(big, np.zeros(1, np.int64), np.asarray([[32, 32]], np.int32), np.asarray([784.0], np.float32), np.zeros(1, np.int32), 32, 4)
    with pytest.raises(api.PimGpuError):
        api.lightmap_atlas_place(masks, offsets, sizes2, areas, np.zeros_like(order), 64, 2)       # not a permutation
    odd = masks.copy(); odd[offsets[3] + 20] = 7
    with pytest.raises(api.PimGpuError, match="neither 0 nor 0xff"):
        api.lightmap_atlas_place(odd, offsets, sizes, areas, order, 64, 2)
    assert api.lightmap_atlas_place(masks[:0], offsets[:0], sizes[:0], areas[:0], order[:0], 64, 1)[2] == 0


class Lightmap(C.Structure):
    _fields_ = [("probes", C.c_void_p * 5), ("position", C.c_void_p), ("normal", C.c_void_p), ("sampleCounts", C.c_void_p),
                ("size", C.c_int32), ("slot", C.c_uint32)]


class LmPack(C.Structure):
    _fields_ = [("axii", C.c_float * 20), ("lightmaps", C.POINTER(Lightmap)), ("lmCount", C.c_int32), ("lmSize", C.c_int32),
                ("texelsPerMeter", C.c_float)]


def lm_table(sc):
    """the scene's draw table with the mesh columns LmPack_Pack writes -- float4 uvs (zw = lightmap uv) and int4 texIndices --
    and one mesh PER ENTITY: chartnodes_assign writes the lightmap coordinates into the mesh (lightmap.c:919-930), so entities
    that share a mesh would overwrite each other's (in the reference as well; its scenes give every drawable its own mesh)"""
    dr0 = scenes.drawables_of(sc)
    meshes = []
    for m in dr0.mesh_ids:
        P, N, U, T = dr0.mesh_arrays[int(m)]
        meshes.append((P.copy(), N.copy(), U.copy(), np.full(P.shape, -1, np.int32)))
    mats = [dr0.materials[i] for i in range(dr0.desc.count)]
    return api.Drawables(meshes, np.arange(len(meshes), dtype=np.int32), dr0.matrices.reshape(-1, 4, 4).copy(), mats)


def oracle_pack(dr, S, tpu, dist, deg, seed, tex_ids_of):
    """the oracle's chain of LmPack_Pack: returns (pages, atlas count)"""
    lib = load_oracle(True)
    nodes = pu.oracle_chartnodes(lib, dr, tpu)
    n = nodes.shape[0]
    lib.orc_lightmap_chart_group.restype = C.c_int32
    grouped, first = np.zeros(n, api.CHARTNODE_DTYPE), np.zeros(n + 1, np.int32)
    nc = lib.orc_lightmap_chart_group(p(nodes), C.c_int32(n), C.c_float(dist), C.c_float(deg), C.c_float(S / np.float32(3.0)), C.c_uint32(seed), p(grouped), p(first))
    first = first[:nc + 1].copy()
    areas, sizes, offsets, masks = pu.oracle_chart_masks(lib, grouped, first)
    order = oracle_sort(areas)
    lib.orc_lightmap_atlas_estimate.restype = C.c_int32
    count = lib.orc_lightmap_atlas_estimate(p(areas), C.c_int32(nc), C.c_int32(S))
    tr, idx, used = oracle_place(masks, offsets, sizes, areas, order, S, count, count * 4 + 16)
    assert used >= 1
    ids = np.asarray(tex_ids_of(used), np.int32)
    lib.orc_lightmap_chart_assign.restype = None
    lib.orc_lightmap_chart_assign(C.byref(dr.desc), p(grouped), p(first), C.c_int32(nc), p(tr), p(idx), p(ids), C.c_int32(S))
    pages = pu.oracle_embed(lib, dr, pu.new_pages(used, S), S, ids)
    return pages, used, nc


@pytest.mark.parametrize("S,tpu,pages_ahead", [(128, 4.0, "1"), (64, 2.0, "1"), (64, 2.0, "0")])
def test_lmpack_pack_end_to_end(S, tpu, pages_ahead, monkeypatch):
    """LmPack_Pack under its own name: the meshes' lightmap uvs and texture ids, the number of pages and every texel of
    position | normal | sampleCounts equal the oracle's chain; then LmPack_Bake on those pages equals the oracle's bake.
    (The pages are page-locked on a second thread while the atlas search runs, or after it on the caller's: both.)"""
    monkeypatch.setenv("PIM_LMPACK_PAGES_AHEAD", pages_ahead)
    host = C.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))
    host.PtSys_Ok.restype = C.c_bool
    seed, mb = 21, 3
    cfg = (C.c_int32 * 5)(seed, mb, 0, 1, -1)
    host.PtSys_Configure(cfg); host.PtSys_Init()
    assert host.PtSys_Ok()
    sc = scenes.cornell_box()
    dr_g, dr_o = lm_table(sc), lm_table(sc)
    host.LmPack_PackEx.restype = LmPack
    host.LmPack_PackEx.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_uint32]
    host.LmPack_LastError.restype = C.c_char_p
    if S == 128:
        pack = host.LmPack_PackEx(C.byref(dr_g.desc), S, tpu, 0.1, 15.0, seed)
    else:
        # the reference's own prototype (lightmap.h:62-66): the draw table comes from the registered provider (Entities_Get /
        # Mesh_Get in the reference, lightmap.c:725-741), the seed of chart_split's streams from PtSys_Configure
        PROVIDER = C.CFUNCTYPE(C.c_bool, C.POINTER(api.PimDrawables), C.c_void_p)

        @PROVIDER
        def provide(out, user):
            C.memmove(out, C.byref(dr_g.desc), C.sizeof(api.PimDrawables))
            return True
        host.PtSys_SetDrawablesProvider(provide, None)
        host.LmPack_Pack.restype = LmPack
        host.LmPack_Pack.argtypes = [C.c_int32, C.c_float, C.c_float, C.c_float]
        pack = host.LmPack_Pack(S, tpu, 0.1, 15.0)
        host.PtSys_SetDrawablesProvider(None, None)
    assert pack.lmCount >= 1, host.LmPack_LastError()
    assert (pack.lmSize, pack.texelsPerMeter) == (S, tpu) and abs(pack.axii[4] - 0.57735) < 1e-6
    slots = [pack.lightmaps[i].slot >> 11 for i in range(pack.lmCount)]            # vkrTextureId: version : 8, type : 3, index : 21
    assert len(set(slots)) == pack.lmCount
    pages_o, used, nc = oracle_pack(dr_o, S, tpu, 0.1, 15.0, seed, lambda k: slots[:k] if k <= len(slots) else list(range(k)))
    assert used == pack.lmCount and nc > 20
    for (Pg, Ng, Ug, Tg), (Po, No, Uo, To) in zip(dr_g.mesh_arrays, dr_o.mesh_arrays):
        assert_bits_equal(Ug, Uo, "mesh uvs"); assert (Tg == To).all()
        assert (Ug[:, 2:] >= 0).all() and (Ug[:, 2:] <= 1).all() and (Tg[:, 3] > 0).all()
    pages_g = [np.ctypeslib.as_array(C.cast(pack.lightmaps[i].probes[0], C.POINTER(C.c_float)), shape=(S * S * 27,)) for i in range(pack.lmCount)]
    mapped = 0
    for a, b in zip(pages_g, pages_o):
        assert_bits_equal(a, b, "page after EmbedAttributes")
        mapped += int(a[S * S * 26:].sum())
    assert mapped > 0.2 * S * S                                                    # the room's walls fill a good part of the atlas
    # bake one pass on what the packer made, through LmPack_Bake's own name
    host.LmPack_Get.restype = C.POINTER(LmPack)
    host.LmPack_Del(host.LmPack_Get())
    C.memmove(host.LmPack_Get(), C.byref(pack), C.sizeof(LmPack))                  # *LmPack_Get() = pack (render_system.c:177)
    host.PtScene_NewFromSnapshot.restype = C.c_void_p
    host.PtScene_NewFromSnapshot.argtypes = [C.c_void_p]
    host.LmPack_Bake.argtypes = [C.c_void_p, C.c_float, C.c_int32]
    scene = C.c_void_p(host.PtScene_NewFromSnapshot(C.byref(sc.desc)))
    host.LmPack_Bake(scene, C.c_float(1.0), C.c_int32(2))
    o = OracleScene(sc.desc, accel=1)
    o.lightmap_bake(pages_o, S, tpu, 1.0, 2, passIndex=0, seed=seed, maxBounces=mb)
    for a, b in zip(pages_g, pages_o):
        assert_bits_equal(a, b, "baked page")
    assert any((a[:S * S * 20] != 0).any() for a in pages_g)
    host.PtScene_Del(scene)
    host.LmPack_Del(host.LmPack_Get())
