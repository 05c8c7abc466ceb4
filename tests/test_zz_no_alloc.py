"""-m gpu: pimgpu_trace_stddev_noalloc (the one entry point that allocates nothing: its per-pixel terms live in a
file-scope device array whose address pimgpu_init took with hipGetSymbolAddress) against pimgpu_trace_stddev, through
the C-ABI, bit for bit, on two calls in a row -- the second on a smaller image, so stale terms of the first call sit
behind the second call's range -- and the size limit."""
import ctypes as C

import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import gpu_init

pytestmark = pytest.mark.gpu


def _rendered(g, W, H, frames):
    t = api.Trace(W, H)
    cam, dof = scenes.pt_test_camera(), scenes.default_dof()
    for s in range(frames):
        t.trace(g, cam, dof, scenes.trace_params(s, seed=5, max_bounces=4))
    return t


def test_stddev_noalloc_matches_stddev_twice_second_smaller():
    gpu_init()
    lib = api.load_library()
    sc = scenes.cornell_box()
    g = api.Scene(sc.desc)
    big, small = _rendered(g, 200, 120, 3), _rendered(g, 64, 36, 2)
    for t in (big, small):
        want = t.stddev()
        m, s = C.c_float(-1.0), C.c_float(-1.0)
        rc = lib.pimgpu_trace_stddev_noalloc(t.handle, C.byref(m), C.byref(s))
        assert rc == 0, lib.pimgpu_last_error()
        got = (m.value, s.value)
        assert np.float32(got[0]).view(np.uint32) == np.float32(want[0]).view(np.uint32), (got, want)
        assert np.float32(got[1]).view(np.uint32) == np.float32(want[1]).view(np.uint32), (got, want)
        assert got[0] > 0.0 and got[1] > 0.0
    # once more on the big image after the small one: same bits as before
    assert big.stddev_noalloc() == big.stddev()
    # beyond the static array: a clear error, not a fault
    huge = api.Trace(3840, 2161)
    m, s = C.c_float(), C.c_float()
    assert lib.pimgpu_trace_stddev_noalloc(huge.handle, C.byref(m), C.byref(s)) == -5
    assert b"3840x2160" in lib.pimgpu_last_error()
    edge = api.Trace(3840, 2160)          # exactly the limit: a zero image -> mean 0, stddev 0
    assert edge.stddev_noalloc() == edge.stddev() == (0.0, 0.0)
    for t in (big, small, huge, edge):
        t.close()
    g.close()
