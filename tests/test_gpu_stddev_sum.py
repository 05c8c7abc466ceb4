"""-m gpu: CalcStdDev's two serial fp32 sums (render_system.c:1374-1394) formed in parallel (k_fs_maps / k_fs_walk,
integrate.hip) against the one-wave chain that adds the terms one after the other (PIMGPU_STDDEV_SERIAL=1), bit for bit, on
colour buffers made to exercise what the parallel form rests on: ties (terms a few binades below the sum), zeros and long
black stretches, single bright pixels, wide ranges, overflow, and the terms it hands back to the chain (negative, NaN,
infinite, denormal).  Small sizes are checked against the oracle as well."""
import ctypes as C
import os
import struct
import time

import numpy as np
import pytest

from pim_amd import api
from tests.gpu_util import gpu_init
from tests.util import load_oracle

pytestmark = pytest.mark.gpu


def _bits(pair):
    return tuple(struct.unpack("<I", struct.pack("<f", v))[0] for v in pair)


@pytest.fixture(scope="module")
def hip():
    gpu_init()
    h = C.CDLL("libamdhip64.so")
    h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    h.hipMemcpy.restype = C.c_int
    return h


def _both(hip, t, color):
    """(parallel, chain) results of pimgpu_trace_stddev on the given colour buffer"""
    color = np.ascontiguousarray(color, np.float32)
    assert color.size == t.width * t.height * 3
    dst = t.buffers()[0]
    assert hip.hipMemcpy(dst, color.ctypes.data, color.nbytes, 1) == 0
    os.environ.pop("PIMGPU_STDDEV_SERIAL", None)
    fast = t.stddev()
    fast_na = t.stddev_noalloc() if t.width * t.height <= 3840 * 2160 else fast
    os.environ["PIMGPU_STDDEV_SERIAL"] = "1"
    try:
        chain = t.stddev()
    finally:
        os.environ.pop("PIMGPU_STDDEV_SERIAL", None)
    assert _bits(fast_na) == _bits(fast)
    return fast, chain


def _images(n, rng):
    """name -> colour buffer of n pixels"""
    out = {}
    out["uniform"] = rng.random((n, 3), dtype=np.float32)
    out["zeros"] = np.zeros((n, 3), np.float32)
    eq = np.full((n, 3), 0.75, np.float32); out["all equal"] = eq
    wide = (10.0 ** rng.uniform(-20, 5, (n, 3))).astype(np.float32); out["wide range"] = wide
    for frac in (1e-4, 1e-2):
        sp = np.zeros((n, 3), np.float32)
        k = max(1, int(n * frac))
        idx = rng.choice(n, k, replace=False)
        sp[idx] = rng.random((k, 3), dtype=np.float32) * 4
        out[f"sparse {frac}"] = sp
    lead = rng.random((n, 3), dtype=np.float32); lead[: n // 2] = 0; out["black first half"] = lead
    tail = rng.random((n, 3), dtype=np.float32); tail[n // 3:] = 0; out["black after a third"] = tail
    for where, name in ((0, "bright first"), (n // 2, "bright middle"), (n - 1, "bright last")):
        b = rng.random((n, 3), dtype=np.float32) * 0.01; b[where] = 3.0e6; out[name] = b
    ov = rng.random((n, 3), dtype=np.float32); ov[n // 2] = 3.0e38; out["overflow"] = ov
    pw = np.zeros((n, 3), np.float32); pw[:, 0] = np.float32(2.0) ** rng.integers(-12, 3, n).astype(np.float32); out["powers of two"] = pw
    return out


@pytest.mark.parametrize("w,h", [(1, 1), (2, 1), (7, 3), (64, 64), (333, 211), (4096, 1), (4097, 1), (1920, 1080)])
def test_parallel_sum_equals_the_chain(hip, w, h):
    rng = np.random.default_rng(w * 10007 + h)
    t = api.Trace(w, h)
    o = load_oracle(True) if w * h <= 333 * 211 else None
    for name, img in _images(w * h, rng).items():
        fast, chain = _both(hip, t, img)
        assert _bits(fast) == _bits(chain), (name, fast, chain)
        if w * h == 1920 * 1080 and name != "overflow":         # (3e38 in every channel: the pixel's own term is infinite -> the chain's)
            # the parallel rounds finish these themselves: a chain over a noticeable part of the frame would show (23 ms for all of it)
            t0 = time.perf_counter(); t.stddev(); dt = 1e3 * (time.perf_counter() - t0)
            print(f"\n  {name:22s} {dt:6.2f} ms", end="")
            assert dt < 6.0, (name, dt)
        if o is not None and name != "overflow" and w * h > 1:
            m, sd = C.c_float(), C.c_float()
            o.orc_stddev(img.ctypes.data_as(C.c_void_p), C.c_int32(w), C.c_int32(h), C.byref(m), C.byref(sd))
            assert _bits(fast) == _bits((m.value, sd.value)), name


def test_terms_the_parallel_sum_does_not_take_go_to_the_chain(hip):
    w, h = 640, 360
    rng = np.random.default_rng(5)
    t = api.Trace(w, h)
    base = rng.random((w * h, 3), dtype=np.float32)
    for name, v in (("negative", -5.0), ("nan", np.nan), ("inf", np.inf), ("denormal", 1e-42)):
        img = base.copy()
        if name == "denormal":
            img[:] = 0
        img[w * h // 3] = v
        fast, chain = _both(hip, t, img)
        assert _bits(fast) == _bits(chain), (name, fast, chain)


def test_4k_frame_and_its_time(hip):
    w, h = 3840, 2160
    rng = np.random.default_rng(11)
    t = api.Trace(w, h)
    img = rng.random((w * h, 3), dtype=np.float32)
    img[: w * 200] = 0                      # black rows on top, as a sky-less frame has
    fast, chain = _both(hip, t, img)
    assert _bits(fast) == _bits(chain)
    t.stddev()
    t0 = time.perf_counter(); t.stddev(); t1 = time.perf_counter()
    os.environ["PIMGPU_STDDEV_SERIAL"] = "1"
    try:
        t2 = time.perf_counter(); t.stddev(); t3 = time.perf_counter()
    finally:
        os.environ.pop("PIMGPU_STDDEV_SERIAL", None)
    print(f"\npimgpu_trace_stddev at 3840x2160: parallel {1e3 * (t1 - t0):.2f} ms, chain {1e3 * (t3 - t2):.2f} ms")
    assert (t1 - t0) < (t3 - t2) / 4
