"""The lightmap checkpoint (LmPack_Save / LmPack_Load, Crate, Guid_HashStr; pim_amd/host/pim_crate.c) against
tests/golden/ref_lmpack.crate -- a file written by the REFERENCE's own assets/crate.c, common/guid.c, common/fnv1a.c and
io/fstr.c, compiled where they lie by `make -C oracle ref` and driven by oracle/ref_crate.c (which restates LmPack_Save's
sequence of Crate_Set calls, rendering/lightmap.c:1225-1262, because lightmap.c itself needs the Vulkan headers).
Host C, no GPU: these tests run everywhere.  LmPack_Load and the resumed bake need page-locked pages and run with the
GPU tests (tests/test_gpu_host_api.py)."""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_lmpack.crate")
K_DIRS = 5


class Guid(ctypes.Structure):
    _fields_ = [("a", ctypes.c_uint64), ("b", ctypes.c_uint64)]
    _pack_ = 16


class Crate(ctypes.Structure):
    _fields_ = [("file", ctypes.c_void_p), ("_pad", ctypes.c_uint64), ("ids", Guid * 1024), ("offsets", ctypes.c_int32 * 1024), ("sizes", ctypes.c_int32 * 1024)]


class Lightmap(ctypes.Structure):
    _fields_ = [("probes", ctypes.c_void_p * K_DIRS), ("position", ctypes.c_void_p), ("normal", ctypes.c_void_p),
                ("sampleCounts", ctypes.c_void_p), ("size", ctypes.c_int32), ("slot", ctypes.c_uint32)]


class LmPack(ctypes.Structure):
    _fields_ = [("axii", ctypes.c_float * (4 * K_DIRS)), ("lightmaps", ctypes.POINTER(Lightmap)), ("lmCount", ctypes.c_int32),
                ("lmSize", ctypes.c_int32), ("texelsPerMeter", ctypes.c_float)]


@pytest.fixture(scope="module")
def host():
    lib = ctypes.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))
    lib.Guid_HashStr.restype = Guid
    lib.Guid_HashStr.argtypes = [ctypes.c_char_p]
    for f in ("Crate_Open", "Crate_Close", "Crate_Get", "Crate_Set", "Crate_Rm", "Crate_Stat", "LmPack_Save"):
        getattr(lib, f).restype = ctypes.c_bool
    lib.Crate_Open.argtypes = [ctypes.POINTER(Crate), ctypes.c_char_p]
    lib.Crate_Close.argtypes = [ctypes.POINTER(Crate)]
    lib.Crate_Get.argtypes = [ctypes.POINTER(Crate), Guid, ctypes.c_void_p, ctypes.c_int32]
    lib.Crate_Set.argtypes = [ctypes.POINTER(Crate), Guid, ctypes.c_void_p, ctypes.c_int32]
    lib.Crate_Rm.argtypes = [ctypes.POINTER(Crate), Guid]
    lib.Crate_Stat.argtypes = [ctypes.POINTER(Crate), Guid, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    lib.LmPack_Save.argtypes = [ctypes.POINTER(Crate), ctypes.POINTER(LmPack)]
    return lib


def page_values(page, floats):
    """oracle/ref_crate.c::page_value"""
    k = np.arange(floats, dtype=np.float32)
    return ((np.float32(page * 1000) + k) * np.float32(0.25) - np.float32(7.0)).astype(np.float32)


def make_pack(lm_size, pages):
    """an LmPack over numpy pages of the reference's block layout [5] float4 | float3 | float3 | float (lightmap.c:81-110)"""
    n = lm_size * lm_size
    lms = (Lightmap * len(pages))()
    for i, p in enumerate(pages):
        base = p.ctypes.data
        for d in range(K_DIRS):
            lms[i].probes[d] = base + 16 * n * d
        lms[i].position = base + 16 * n * K_DIRS
        lms[i].normal = base + 16 * n * K_DIRS + 12 * n
        lms[i].sampleCounts = base + 16 * n * K_DIRS + 24 * n
        lms[i].size = lm_size
    pack = LmPack()
    pack.lightmaps = lms
    pack.lmCount = len(pages)
    pack.lmSize = lm_size
    pack.texelsPerMeter = 16.0
    return pack, lms


def test_struct_layouts():
    assert ctypes.sizeof(Guid) == 16 and ctypes.sizeof(Crate) == 16 + 1024 * 24
    assert ctypes.sizeof(Lightmap) == 72 and ctypes.sizeof(LmPack) == 104


def test_guid_hash_matches_the_reference_file(host):
    raw = open(GOLDEN, "rb").read()
    ids = np.frombuffer(raw[:16384], dtype=np.uint64).reshape(1024, 2)
    for slot, name in enumerate([b"lmpack", b"lightmap_0", b"lightmap_1", b"extra", b"small"]):
        g = host.Guid_HashStr(name)
        assert (g.a, g.b) == (int(ids[slot, 0]), int(ids[slot, 1])), name
    # the hash folds ASCII case (common/fnv1a.c:3-11) and the empty string is the null id
    g1, g2 = host.Guid_HashStr(b"LightMap_0"), host.Guid_HashStr(b"lightmap_0")
    assert (g1.a, g1.b) == (g2.a, g2.b)
    g0 = host.Guid_HashStr(b"")
    assert (g0.a, g0.b) == (0, 0)


def test_save_sequence_is_byte_identical_to_the_reference(host, tmp_path):
    """the whole file: header tables (ids, offsets, sizes incl. the split-off free extent) and every blob"""
    path = str(tmp_path / "pack.crate").encode()
    lm_size, floats = 4, 16 * 27
    pages = [page_values(i, floats) for i in range(2)]
    pack, keep = make_pack(lm_size, pages)
    crate = Crate()
    assert host.Crate_Open(ctypes.byref(crate), path)
    assert host.LmPack_Save(ctypes.byref(crate), ctypes.byref(pack))
    extra = ((np.arange(2000) * 7 + 3) & 255).astype(np.uint8)
    assert host.Crate_Set(ctypes.byref(crate), host.Guid_HashStr(b"extra"), extra.ctypes.data, 100)
    assert host.Crate_Set(ctypes.byref(crate), host.Guid_HashStr(b"small"), extra.ctypes.data + 100, 50)
    assert host.Crate_Set(ctypes.byref(crate), host.Guid_HashStr(b"extra"), extra.ctypes.data + 500, 80)      # in place, keeps 100
    pages[1][:] = page_values(1, floats) + np.float32(0.5)
    assert host.LmPack_Save(ctypes.byref(crate), ctypes.byref(pack))                                            # over the first save
    off, size = ctypes.c_int32(), ctypes.c_int32()
    assert host.Crate_Stat(ctypes.byref(crate), host.Guid_HashStr(b"small"), ctypes.byref(off), ctypes.byref(size))
    assert (off.value, size.value) == (28156, 50)
    assert host.Crate_Close(ctypes.byref(crate))
    mine, ref = open(path, "rb").read(), open(GOLDEN, "rb").read()
    # the second save rewrote page 0 too (same bytes) -- the generator only rewrote page 1; the files must still be equal
    assert len(mine) == len(ref) == 28206
    assert mine == ref


def test_reading_the_reference_file(host, tmp_path):
    path = str(tmp_path / "ref.crate").encode()
    open(path, "wb").write(open(GOLDEN, "rb").read())
    crate = Crate()
    assert host.Crate_Open(ctypes.byref(crate), path)
    hdr = np.zeros(6, np.int32)
    assert host.Crate_Get(ctypes.byref(crate), host.Guid_HashStr(b"lmpack"), hdr.ctypes.data, 24)
    assert list(hdr[:5]) == [2, 5, 2, 4, 1728] and hdr[5:6].view(np.float32)[0] == 16.0     # DiskLmPack, lightmap.h:46-54
    page = np.zeros(16 * 27, np.float32)
    assert host.Crate_Get(ctypes.byref(crate), host.Guid_HashStr(b"lightmap_0"), page.ctypes.data, 1728)
    assert (page.view(np.uint32) == page_values(0, 16 * 27).view(np.uint32)).all()
    assert host.Crate_Get(ctypes.byref(crate), host.Guid_HashStr(b"lightmap_1"), page.ctypes.data, 1728)
    assert (page.view(np.uint32) == (page_values(1, 16 * 27) + np.float32(0.5)).view(np.uint32)).all()
    # asking for more than the blob holds, or for an unknown id, fails (crate.c:206-218)
    assert not host.Crate_Get(ctypes.byref(crate), host.Guid_HashStr(b"small"), page.ctypes.data, 51)
    assert not host.Crate_Get(ctypes.byref(crate), host.Guid_HashStr(b"nothing"), page.ctypes.data, 4)
    assert host.Crate_Close(ctypes.byref(crate))
    assert not host.Crate_Close(ctypes.byref(crate))           # already closed (crate.c:177-194)


def test_free_and_reuse_keep_the_table_consistent(host, tmp_path):
    """Crate_Rm / a grown blob: where the reference's crate_free loses the extent's offset (see pim_crate.c), this one must
    keep a consistent extent list -- extents in file order, contiguous, blobs readable after any sequence of calls"""
    path = str(tmp_path / "churn.crate").encode()
    crate = Crate()
    assert host.Crate_Open(ctypes.byref(crate), path)
    rng = np.random.default_rng(5)
    live = {}
    for step in range(300):
        name = b"blob_%d" % rng.integers(0, 24)
        if name in live and rng.random() < 0.3:
            assert host.Crate_Rm(ctypes.byref(crate), host.Guid_HashStr(name))
            del live[name]
        else:
            data = rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8)
            assert host.Crate_Set(ctypes.byref(crate), host.Guid_HashStr(name), data.ctypes.data, len(data))
            live[name] = data
        offs, sizes = np.array(crate.offsets[:]), np.array(crate.sizes[:])
        used = int(np.argmax(sizes == 0)) if (sizes == 0).any() else 1024
        assert offs[0] == 24576 and (offs[1:used] == offs[:used - 1] + sizes[:used - 1]).all()
        assert int(sizes[:used].astype(np.int64).sum()) == 1 << 30
    for name, data in live.items():
        out = np.zeros(len(data), np.uint8)
        assert host.Crate_Get(ctypes.byref(crate), host.Guid_HashStr(name), out.ctypes.data, len(data))
        assert (out == data).all()
    assert host.Crate_Close(ctypes.byref(crate))
    # and after a reopen
    assert host.Crate_Open(ctypes.byref(crate), path)
    for name, data in live.items():
        out = np.zeros(len(data), np.uint8)
        assert host.Crate_Get(ctypes.byref(crate), host.Guid_HashStr(name), out.ctypes.data, len(data))
        assert (out == data).all()
    assert host.Crate_Close(ctypes.byref(crate))
