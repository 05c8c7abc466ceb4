"""-m gpu: the plain-C host layer (pim_amd/host/pim_host.c: Pim's own entry-point names AND signatures over the C-ABI).

test_reference_caller_*: tests/host/ref_caller.c, a C program that declares the reference's prototypes and structs itself
(rendering/path_tracer.h:13-103, lightmap.h:27-69, cubemap.h, ambcube.h) and never sees pim_host.h, is linked against
libpimhost.so and run; everything it computes -- Pt_Trace's host images after three progressive frames, the autofocus
state, Pt_Intersect, Pt_TraceRay, Pt_RayGen, Cubemap_Bake, AmbCube_Bake, two LmPack_Bake passes, the rebuild on a changed
modtime -- equals the oracle's bits (portable-libm variant), replayed here in the same order."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import assert_bits_equal, gpu_init
from tests.util import OracleScene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Int2(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32)]


class PtTrace(C.Structure):
    _fields_ = [("color", C.c_void_p), ("albedo", C.c_void_p), ("normal", C.c_void_p), ("denoised", C.c_void_p),
                ("imageSize", Int2), ("sampleWeight", C.c_float)]


class PtRayHit(C.Structure):
    _fields_ = [("wuvt", Float4), ("normal", Float4), ("type", C.c_int32), ("iVert", C.c_int32), ("flags", C.c_uint32)]


class PtResult(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("albedo", C.c_float * 3), ("normal", C.c_float * 3)]


def write_scene_file(path, sc, cam):
    """the scene snapshot + camera as tests/host/ref_caller.c reads them"""
    d = sc.desc
    with open(path, "wb") as f:
        f.write(struct.pack("<8i", 0x53434E45, d.vertCount, d.matCount, d.texCount, d.skySize, d.brdfLutSize, d.emissiveAttempts, d.seed))
        f.write(struct.pack("<f", d.metersPerCell)); f.write(bytes(d.media)); f.write(bytes(cam))
        f.write(sc.positions.tobytes()); f.write(sc.normals.tobytes()); f.write(sc.uvs.tobytes()); f.write(sc.matIds.tobytes())
        f.write(bytes(sc.materials)[:48 * d.matCount])
        for i in range(d.texCount):
            t = sc.textures[i]
            f.write(struct.pack("<3i", t.width, t.height, t.format)); f.write(sc.tex_arrays[i].tobytes())
        if d.skySize > 0:
            for k in range(6):
                f.write(sc.sky[k].tobytes())
        f.write(sc.lut.tobytes())


@pytest.mark.parametrize("devices", [None, "0,0"])
def test_reference_caller_gets_the_oracles_bits(tmp_path, devices):
    exe = os.path.join(ROOT, "pim_amd", "host", "build", "ref_caller")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "pim_amd", "host")])
    sc = scenes.cornell_box()
    cam = scenes.pt_test_camera()
    scene_path, out_path = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    write_scene_file(scene_path, sc, cam)
    env = dict(os.environ)
    if devices:
        env["PIM_GPU_DEVICES"] = devices            # the same caller fanned out over two contexts: same bits
    run = subprocess.run([exe, scene_path, out_path], capture_output=True, text=True, timeout=120, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "ref_caller ok" in run.stdout and "Vertex Count: 324" in run.stdout and "Emissive Count: 12" in run.stdout      # PtScene_Gui
    assert "Blade Count: 5" in run.stdout                                                                                # DofInfo_Gui
    data = open(out_path, "rb").read()
    off = 0

    def take(count, dtype=np.float32):
        nonlocal off
        a = np.frombuffer(data, dtype, count, off); off += a.nbytes
        return a

    # ---- the same calls, in the same order, on the oracle (PtSys_Configure: seed 17, 4 bounces, dt 1/60) ----
    o = OracleScene(sc.desc, accel=1)
    W, H, seed, mb = 80, 45, 17, 4
    img = [np.zeros((W * H, 3), np.float32) for _ in range(3)]
    dof = scenes.default_dof()
    for s in range(3):
        o.trace_image(img, W, H, cam, dof, scenes.trace_params(s, seed=seed, max_bounces=mb))      # Pt_Trace call k = sample k
    for k, name in enumerate(("color", "albedo", "normal")):
        assert_bits_equal(take(W * H * 3).reshape(-1, 3), img[k], name)
    got_dof = take(7)
    assert got_dof[1] == np.float32(dof.focalLength) and got_dof.view(np.uint8)[24] == 1
    ro = np.array([[0.5, -1.0, 0.25, 0.0]], np.float32); rd = np.array([[0.0, 0.6, 0.8, 0.0]], np.float32)
    ho = o.intersect(ro, rd)
    hit = np.frombuffer(data, api.RAYHIT_DTYPE, 1, off); off += 48
    for fld in ("wuvt", "normal", "type", "iVert", "flags"):
        assert_bits_equal(hit[fld], ho[fld], "Pt_Intersect " + fld)
    for k in range(3):                                                                           # Pt_TraceRay call k = stream k
        r = np.frombuffer(data, api.RESULT_DTYPE, 1, off); off += 36
        want = o.trace_rays(ro, rd, sampleIndex=k, seed=seed, maxBounces=mb)
        for fld in ("color", "albedo", "normal"):
            assert_bits_equal(r[fld], want[fld], f"Pt_TraceRay {k} {fld}")
    origin = np.array([0.25, -0.5, 0.5, 0.0], np.float32)
    co, do = o.raygen(origin, 256, sampleIndex=0, seed=seed, maxBounces=mb)
    assert_bits_equal(take(256 * 4).reshape(-1, 4), co, "Pt_RayGen colors"); assert_bits_equal(take(256 * 4).reshape(-1, 4), do, "Pt_RayGen directions")
    fo = [np.zeros(64 * 3, np.float32) for _ in range(6)]
    o.cubemap_bake(fo, 8, origin, 1.0, sampleIndex=0, seed=seed, maxBounces=mb)
    for f in range(6):
        assert_bits_equal(take(64 * 3), fo[f], f"Cubemap_Bake face {f}")
    from tests.util import oracle_cubemap_convolve
    conv = [np.zeros(85 * 4, np.float32) for _ in range(6)]
    oracle_cubemap_convolve(fo, conv, 8, 64, 1.0, sampleIndex=0, seed=seed)
    oracle_cubemap_convolve(fo, conv, 8, 64, 0.5, sampleIndex=1, seed=seed)
    assert take(1, np.int32)[0] == 4                                                             # Cubemap_New: 1 + log2(8) mips
    for f in range(6):
        assert_bits_equal(take(85 * 4), conv[f], f"Cubemap_Convolve face {f}")
    cube = np.zeros((6, 4), np.float32)
    n = o.ambcube_bake(cube, origin, 200, 0, sampleIndex=1, seed=seed, maxBounces=mb)               # the second Pt_RayGen batch
    assert_bits_equal(take(24).reshape(6, 4), cube, "AmbCube_Bake"); assert take(1, np.int32)[0] == n == 1
    S = 32
    page = np.zeros(S * S * 27, np.float32)
    pos = page[S * S * 20:S * S * 23].reshape(S, S, 3); nor = page[S * S * 23:S * S * 26].reshape(S, S, 3); cnt = page[S * S * 26:].reshape(S, S)
    ys, xs = np.meshgrid(np.arange(4, 28), np.arange(4, 28), indexing="ij")
    pos[4:28, 4:28, 0] = np.float32(-3.0) + np.float32(0.25) * xs.astype(np.float32); pos[4:28, 4:28, 1] = -4.95
    pos[4:28, 4:28, 2] = np.float32(-3.0) + np.float32(0.25) * ys.astype(np.float32)
    nor[4:28, 4:28, 1] = 1.0; cnt[4:28, 4:28] = 1.0
    o.lightmap_bake([page], S, 4.0, 1.0, 3, passIndex=0, seed=seed, maxBounces=mb)
    o.lightmap_bake([page], S, 4.0, 0.5, 2, passIndex=1, seed=seed, maxBounces=mb)
    got_page = take(S * S * 27)
    assert_bits_equal(got_page, page, "LmPack_Bake page")
    assert (got_page[:S * S * 4] != 0).any() and got_page[S * S * 26:].max() >= 4.0
    hit2 = np.frombuffer(data, api.RAYHIT_DTYPE, 1, off); off += 48
    assert hit2.tobytes() == hit.tobytes()                          # the rebuilt scene answers alike
    same, rebuilt, sz_trace, sz_lm = take(4, np.int32)
    assert (same, rebuilt) == (0, 1) and sz_trace == 48 and sz_lm == 72
    assert off == len(data)


def test_scene_producers_under_pim_names():
    """FlattenDrawables + PtScene_New (PtScene_NewFromDrawables) and BrdfLut_New / BrdfLut_Update (math/lighting.c:86-118)
    through the plain-C host layer: same bits as the C-ABI calls they wrap"""
    host = C.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))
    host.PtSys_Ok.restype = C.c_bool
    cfg = (C.c_int32 * 5)(7, 0, 0, 1, -1)                 # PtSysConfig: seed 7
    host.PtSys_Configure(cfg)
    host.PtSys_Init()
    assert host.PtSys_Ok()

    class BrdfLut(C.Structure):
        _fields_ = [("size", Int2), ("texels", C.POINTER(C.c_float))]
    host.BrdfLut_New.restype = BrdfLut
    host.BrdfLut_New.argtypes = [Int2, C.c_int32]
    lut = host.BrdfLut_New(Int2(32, 32), 48)
    assert lut.texels and lut.size.x == 32
    host.BrdfLut_Update(C.byref(lut), 48, 64)
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
    host.PtScene_Gpu.restype = C.c_void_p
    assert api.load_library().pimgpu_scene_info(C.c_void_p(host.PtScene_Gpu(scene)), C.byref(info)) == 0
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
    host.PtSys_Ok.restype = C.c_bool
    host.PtSys_Init()
    assert host.PtSys_Ok()

    class Lightmap(C.Structure):
        _fields_ = [("probes", C.POINTER(C.c_float) * 5), ("position", C.POINTER(C.c_float)), ("normal", C.POINTER(C.c_float)),
                    ("sampleCounts", C.POINTER(C.c_float)), ("size", C.c_int32), ("slot", C.c_uint32)]
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


def test_lightmap_checkpoint_resumes_bit_exact(tmp_path):
    """LmPack_Save / LmPack_Load (lightmap.c:1225-1308) around LmPack_Bake: a bake that is saved after its first pass,
    dropped, loaded back (Lightmap_New + Crate_Get + Lightmap_Upload per page) and continued equals the uninterrupted bake
    and the oracle bit for bit -- probes with their running lobe weights in .w, sampleCounts, position, normal; the
    uploader sees the kGiDirections RGBA32F layers of every page (Lightmap_Upload, lightmap.c:135-144)"""
    from tests.test_host_crate import Crate, Guid, Lightmap, LmPack
    host = C.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))
    libc = C.CDLL(None)
    libc.calloc.restype = C.c_void_p
    libc.calloc.argtypes = [C.c_size_t, C.c_size_t]
    host.PtSys_Ok.restype = C.c_bool
    host.LmPack_Get.restype = C.POINTER(LmPack)
    host.PtScene_NewFromSnapshot.restype = C.c_void_p
    host.PtScene_NewFromSnapshot.argtypes = [C.c_void_p]
    host.LmPack_Bake.argtypes = [C.c_void_p, C.c_float, C.c_int32]
    for f in ("Crate_Open", "Crate_Close", "LmPack_Save", "LmPack_Load"):
        getattr(host, f).restype = C.c_bool
    host.Crate_Open.argtypes = [C.POINTER(Crate), C.c_char_p]
    seed, mb, S = 11, 3, 32
    cfg = (C.c_int32 * 5)(seed, mb, 0, 1, -1)                    # PtSysConfig: seed, maxBounces, frameDt (default), downloadAovs, device
    sc = scenes.cornell_box()
    uploads = []
    UPLOAD = C.CFUNCTYPE(None, C.c_uint32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p)

    @UPLOAD
    def on_upload(slot, layer, data, nbytes, user):
        uploads.append((layer, nbytes, np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_float)), shape=(nbytes // 4,)).copy()))
    host.PtSys_SetLightmapUploader(on_upload, None)

    def fresh_pack():
        pack = host.LmPack_Get()
        host.LmPack_Del(pack)
        pack.contents.lightmaps = C.cast(libc.calloc(1, C.sizeof(Lightmap)), C.POINTER(Lightmap))
        pack.contents.lmCount, pack.contents.lmSize, pack.contents.texelsPerMeter = 1, S, 4.0
        host.Lightmap_New(C.byref(pack.contents.lightmaps[0]), C.c_int32(S))
        page = np.ctypeslib.as_array(C.cast(pack.contents.lightmaps[0].probes[0], C.POINTER(C.c_float)), shape=(S * S * 27,))
        fill_page(page)
        return pack, page

    def fill_page(page):
        pos = page[S * S * 20:S * S * 23].reshape(S, S, 3); nor = page[S * S * 23:S * S * 26].reshape(S, S, 3); cnt = page[S * S * 26:].reshape(S, S)
        ys, xs = np.meshgrid(np.arange(4, 28), np.arange(4, 28), indexing="ij")
        pos[4:28, 4:28, 0] = np.float32(-3.0) + np.float32(0.25) * xs.astype(np.float32); pos[4:28, 4:28, 1] = -4.95
        pos[4:28, 4:28, 2] = np.float32(-3.0) + np.float32(0.25) * ys.astype(np.float32)
        nor[4:28, 4:28, 1] = 1.0; cnt[4:28, 4:28] = 1.0

    # the oracle's two passes
    o = OracleScene(sc.desc, accel=1)
    want = np.zeros(S * S * 27, np.float32); fill_page(want)
    o.lightmap_bake([want], S, 4.0, 1.0, 3, passIndex=0, seed=seed, maxBounces=mb)
    o.lightmap_bake([want], S, 4.0, 1.0, 2, passIndex=1, seed=seed, maxBounces=mb)

    # uninterrupted
    host.PtSys_Configure(cfg); host.PtSys_Init()
    assert host.PtSys_Ok()
    scene = C.c_void_p(host.PtScene_NewFromSnapshot(C.byref(sc.desc)))
    assert scene
    pack, page = fresh_pack()
    assert len(uploads) == 5 and all(n == S * S * 16 for _, n, _ in uploads)         # Lightmap_New ends with Lightmap_Upload (lightmap.c:122)
    host.LmPack_Bake(scene, C.c_float(1.0), C.c_int32(3)); host.LmPack_Bake(scene, C.c_float(1.0), C.c_int32(2))
    straight = page.copy()
    assert_bits_equal(straight, want, "uninterrupted bake")
    host.PtScene_Del(scene)

    # the same with the pages resident in HBM: nothing comes back before LmPack_Sync (mode 2) / after every call (mode 1)
    for mode in (2, 1):
        host.PtSys_Configure(cfg)
        scene = C.c_void_p(host.PtScene_NewFromSnapshot(C.byref(sc.desc)))
        pack, page = fresh_pack()
        host.LmPack_SetResident(C.c_int32(mode))
        host.LmPack_Bake(scene, C.c_float(1.0), C.c_int32(3))
        if mode == 2:
            assert not page[:S * S * 20].any()                   # still the zeroed probes on the host
        host.LmPack_Bake(scene, C.c_float(1.0), C.c_int32(2))
        if mode == 2:
            assert not page[:S * S * 20].any()
            host.LmPack_Sync()
        assert_bits_equal(page, straight, f"resident bake, mode {mode}")
        host.LmPack_SetResident(C.c_int32(0))
        host.PtScene_Del(scene)

    # saved after pass 0, dropped, loaded, continued
    host.PtSys_Configure(cfg)
    scene = C.c_void_p(host.PtScene_NewFromSnapshot(C.byref(sc.desc)))
    pack, page = fresh_pack()
    host.LmPack_Bake(scene, C.c_float(1.0), C.c_int32(3))
    after_first = page.copy()
    crate = Crate()
    path = str(tmp_path / "bake.crate").encode()
    assert host.Crate_Open(C.byref(crate), path)
    assert host.LmPack_Save(C.byref(crate), pack)
    assert host.Crate_Close(C.byref(crate))
    assert os.path.getsize(path) == 24576 + 24 + S * S * 108
    host.LmPack_Del(pack)
    assert pack.contents.lmCount == 0 and not pack.contents.lightmaps
    del uploads[:]
    assert host.Crate_Open(C.byref(crate), path)
    assert host.LmPack_Load(C.byref(crate), pack)
    assert host.Crate_Close(C.byref(crate))
    assert (pack.contents.lmCount, pack.contents.lmSize, pack.contents.texelsPerMeter) == (1, S, 4.0)
    page = np.ctypeslib.as_array(C.cast(pack.contents.lightmaps[0].probes[0], C.POINTER(C.c_float)), shape=(S * S * 27,))
    assert_bits_equal(page, after_first, "loaded page")
    # Lightmap_New's upload of the zeroed page, then LmPack_Load's upload of the loaded one: layer i = probes[i]
    assert [l for l, _, _ in uploads] == [0, 1, 2, 3, 4] * 2
    for layer, nbytes, data in uploads[5:]:
        assert nbytes == S * S * 16
        assert_bits_equal(data, after_first[S * S * 4 * layer:S * S * 4 * (layer + 1)], f"uploaded layer {layer}")
    host.LmPack_Bake(scene, C.c_float(1.0), C.c_int32(2))
    assert_bits_equal(page, straight, "resumed bake")
    host.LmPack_Del(pack)
    host.PtScene_Del(scene)
    host.PtSys_SetLightmapUploader(None, None)


def test_page_locked_blocks_come_zeroed():
    """pimgpu_host_alloc stands for Tex_Calloc (lightmap.c:90-110): blocks above 32 MB are cleared by several threads, in parts
    rounded to pages; sizes on and off those boundaries, dirtied and freed in between so that a reused block shows"""
    gpu_init()
    for count in (1, 4095, (32 << 20) // 4 - 1, (48 << 20) // 4 + 3, 27 * 1024 * 1024 + 5):
        a = api.host_array(count, np.float32)
        assert a.shape == (count,) and not a.any()
        a[:] = 1.0
        del a
        b = api.host_array(count, np.uint32)
        assert not b.any()
        del b
