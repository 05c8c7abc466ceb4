"""-m gpu: the plain-C host layer (pim_amd/host/pim_host.c: Pim's own entry-point names over the C-ABI)
gives the same bits as the oracle: Pt_Trace's host images after several frames, Pt_Intersect, Pt_TraceRay."""
import ctypes as C
import os

import numpy as np
import pytest

from pim_amd import api, scenes
from tests.util import OracleScene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Int2(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32)]


class PtTrace(C.Structure):
    _fields_ = [("color", C.c_void_p), ("albedo", C.c_void_p), ("normal", C.c_void_p), ("denoised", C.c_void_p),
                ("imageSize", Int2), ("sampleWeight", C.c_float), ("gpu", C.c_void_p), ("sampleIndex", C.c_uint32),
                ("seed", C.c_uint32), ("maxBounces", C.c_int32), ("frameDt", C.c_float)]


class PtRayHit(C.Structure):
    _fields_ = [("wuvt", Float4), ("normal", Float4), ("type", C.c_int32), ("iVert", C.c_int32), ("flags", C.c_uint32)]


class PtResult(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("albedo", C.c_float * 3), ("normal", C.c_float * 3)]


def test_pim_named_entry_points_match_oracle():
    host = C.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))
    host.PtSys_Init.restype = C.c_bool
    host.PtScene_New.restype = C.c_void_p
    host.Pt_Intersect.restype = PtRayHit
    host.Pt_Intersect.argtypes = [C.c_void_p, Float4, Float4, C.c_float, C.c_float]
    host.Pt_TraceRay.restype = PtResult
    host.Pt_TraceRay.argtypes = [C.c_void_p, Float4, Float4, C.c_uint32, C.c_uint32, C.c_int32]
    assert host.PtSys_Init()
    sc = scenes.cornell_box()
    scene = C.c_void_p(host.PtScene_New(C.byref(sc.desc)))
    assert scene
    o = OracleScene(sc.desc, accel=1)
    W, H = 80, 45
    trace = PtTrace()
    host.PtTrace_New(C.byref(trace), Int2(W, H))
    trace.seed, trace.maxBounces = 17, 4
    dof, dof_o = api.PimDofInfo(), scenes.default_dof()
    host.DofInfo_New(C.byref(dof))
    assert bytes(dof) == bytes(dof_o)
    cam = scenes.pt_test_camera()
    img = [np.zeros((W * H, 3), np.float32) for _ in range(3)]
    for s in range(3):
        trace.sampleWeight = 1.0 / (s + 1)
        host.Pt_Trace(C.byref(trace), C.byref(dof), scene, C.byref(cam))
        o.trace_image(img, W, H, cam, dof_o, scenes.trace_params(s, seed=17, max_bounces=4))
    color = np.ctypeslib.as_array(C.cast(trace.color, C.POINTER(C.c_float)), shape=(W * H, 3))
    assert (color.view(np.uint32) == img[0].view(np.uint32)).all()
    assert trace.sampleIndex == 3

    ro = Float4(0.5, -1.0, 0.25, 0.0); rd = Float4(0.0, 0.6, 0.8, 0.0)
    h = host.Pt_Intersect(scene, ro, rd, 0.0, api.RCP_EPSILON)
    ho = o.intersect(np.array([[0.5, -1.0, 0.25, 0.0]], np.float32), np.array([[0.0, 0.6, 0.8, 0.0]], np.float32))[0]
    assert h.type == ho["type"] and h.iVert == ho["iVert"] and np.float32(h.wuvt.w) == ho["wuvt"][3]
    r = host.Pt_TraceRay(scene, ro, rd, 2, 5, 4)
    r_o = o.trace_rays(np.array([[0.5, -1.0, 0.25, 0.0]], np.float32), np.array([[0.0, 0.6, 0.8, 0.0]], np.float32), 2, 5, 4)[0]
    assert tuple(np.float32(v) for v in r.color) == tuple(r_o["color"])

    # Cubemap_Bake / AmbCube_Bake under their own names (rendering/cubemap.h:209, math/ambcube.h:69)
    class Cubemap(C.Structure):
        _fields_ = [("size", C.c_int32), ("color", C.c_void_p * 6)]
    size = 8
    fg = [np.zeros(size * size * 3, np.float32) for _ in range(6)]
    fo = [np.zeros(size * size * 3, np.float32) for _ in range(6)]
    cm = Cubemap(); cm.size = size
    for f in range(6):
        cm.color[f] = fg[f].ctypes.data
    origin = Float4(0.25, -0.5, 0.5, 0.0)
    host.Cubemap_Bake.argtypes = [C.c_void_p, C.c_void_p, Float4, C.c_float, C.c_uint32, C.c_uint32, C.c_int32]
    host.Cubemap_Bake(C.byref(cm), scene, origin, 1.0, 0, 3, 4)
    o.cubemap_bake(fo, size, np.array([0.25, -0.5, 0.5, 0.0], np.float32), 1.0, sampleIndex=0, seed=3, maxBounces=4)
    assert all((a.view(np.uint32) == b.view(np.uint32)).all() for a, b in zip(fg, fo))
    cube_g = np.zeros((6, 4), np.float32); cube_o = np.zeros((6, 4), np.float32)
    host.AmbCube_Bake.restype = C.c_int32
    host.AmbCube_Bake.argtypes = [C.c_void_p, C.c_void_p, Float4, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_int32]
    n = host.AmbCube_Bake(scene, cube_g.ctypes.data_as(C.c_void_p), origin, 200, 0, 1, 3, 4)
    assert n == 1 == o.ambcube_bake(cube_o, np.array([0.25, -0.5, 0.5, 0.0], np.float32), 200, 0, sampleIndex=1, seed=3, maxBounces=4)
    assert (cube_g.view(np.uint32) == cube_o.view(np.uint32)).all()
    host.PtTrace_Del(C.byref(trace))
    host.PtScene_Del(scene)


def test_scene_producers_under_pim_names():
    """FlattenDrawables + PtScene_New (PtScene_NewFromDrawables) and BrdfLut_New / BrdfLut_Update (math/lighting.c:86-118)
    through the plain-C host layer: same bits as the C-ABI calls they wrap"""
    host = C.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))
    host.PtSys_Init.restype = C.c_bool
    assert host.PtSys_Init()

    class BrdfLut(C.Structure):
        _fields_ = [("size", Int2), ("texels", C.POINTER(C.c_float))]
    host.BrdfLut_New.restype = BrdfLut
    host.BrdfLut_New.argtypes = [Int2, C.c_int32, C.c_uint32]
    lut = host.BrdfLut_New(Int2(32, 32), 48, 7)
    assert lut.texels and lut.size.x == 32
    host.BrdfLut_Update(C.byref(lut), 48, 64, C.c_uint32(7))
    table = np.ctypeslib.as_array(lut.texels, shape=(32, 32, 2)).copy()
    want = np.zeros((32, 32, 2), np.float32)
    api.brdf_lut_bake(want, 0, 48, seed=7); api.brdf_lut_bake(want, 48, 64, seed=7)
    assert (table.view(np.uint32) == want.view(np.uint32)).all() and table.max() > 0.5

    sc = scenes.cornell_box()
    dr = scenes.drawables_of(sc)
    rest = api.PimSceneDesc.from_buffer_copy(bytes(sc.desc))
    rest.brdfLut, rest.brdfLutSize = C.cast(lut.texels, C.c_void_p), 32
    host.PtScene_NewFromDrawables.restype = C.c_void_p
    scene = C.c_void_p(host.PtScene_NewFromDrawables(C.byref(dr.desc), C.byref(rest)))
    assert scene
    info = api.PimSceneInfo()
    assert api.load_library().pimgpu_scene_info(scene, C.byref(info)) == 0
    assert info.triCount == 108 and info.matCount == 9 and info.emissiveCount == 12
    host.Pt_Intersect.restype = PtRayHit
    host.Pt_Intersect.argtypes = [C.c_void_p, Float4, Float4, C.c_float, C.c_float]
    h = host.Pt_Intersect(scene, Float4(0.5, -1.0, 0.25, 0.0), Float4(0.0, 0.6, 0.8, 0.0), 0.0, api.RCP_EPSILON)
    assert h.type == api.HIT_TRIANGLE and h.wuvt.w > 0.0
    host.PtScene_Del(scene)
    host.BrdfLut_Del(C.byref(lut))
    assert not lut.texels


def test_lightmap_pages_under_pim_names():
    """Lightmap_New (lightmap.c:81-110: one zeroed allocation, probes[0] its base) + EmbedAttributes (:1034-1045) through
    the plain-C host layer: the same bits as the C-ABI call it wraps, probes left zero"""
    from tests import prepare_util as pu
    host = C.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))
    host.PtSys_Init.restype = C.c_bool
    assert host.PtSys_Init()

    class Lightmap(C.Structure):
        _fields_ = [("probes", C.POINTER(C.c_float) * 5), ("position", C.POINTER(C.c_float)), ("normal", C.POINTER(C.c_float)),
                    ("sampleCounts", C.POINTER(C.c_float)), ("size", C.c_int32)]
    S, ids = 48, [9, 4]
    lms = (Lightmap * 2)()
    for lm in lms:
        host.Lightmap_New(C.byref(lm), C.c_int32(S))
        assert lm.size == S and C.addressof(lm.position.contents) - C.addressof(lm.probes[0].contents) == S * S * 80
        assert C.addressof(lm.sampleCounts.contents) - C.addressof(lm.probes[0].contents) == S * S * 104
    dr = pu.embed_table(S, ids, tris_per_mesh=(80, 120), hard_cases=False)
    tex = np.asarray(ids, np.int32)
    host.LmPack_EmbedAttributes(C.byref(dr.desc), lms, tex.ctypes.data_as(C.c_void_p), C.c_int32(2))
    want = api.lightmap_embed(dr, pu.new_pages(2, S), S, ids)
    for lm, w in zip(lms, want):
        got = np.ctypeslib.as_array(lm.probes[0], shape=(S * S * 27,))
        assert (got.view(np.uint32) == w.view(np.uint32)).all() and got[S * S * 26:].sum() > 50
        host.Lightmap_Del(C.byref(lm))
        assert not lm.sampleCounts and lm.size == 0
