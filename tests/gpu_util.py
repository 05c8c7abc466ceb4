"""shared fixtures of the -m gpu tests: one pimgpu_init per process"""
import numpy as np

from pim_amd import api

_READY = False


def gpu_init():
    global _READY
    if not _READY:
        api.init(0)
        _READY = True


def assert_bits_equal(a, b, what):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shapes {a.shape} / {b.shape}"
    if a.size == 0:
        return
    av = a.view(np.uint32) if a.dtype == np.float32 else a
    bv = b.view(np.uint32) if b.dtype == np.float32 else b
    bad = np.nonzero((av != bv).reshape(av.shape[0], -1).any(axis=1))[0] if av.ndim > 1 else np.nonzero(av != bv)[0]
    assert bad.size == 0, f"{what}: {bad.size} of {av.shape[0]} differ, first at {bad[0]}: gpu {a[bad[0]]} oracle {b[bad[0]]}"
