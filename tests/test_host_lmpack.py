"""The sequential steps of LmPack_Pack in the host layer (pim_amd/host/pim_lmpack.c: chart_group with its bucketed plane_find,
chart_split, chart_sort's quicksort) against the oracle's one-for-one restatement of rendering/lightmap.c:333-357, 451-518,
563-660 and common/sort.c:14-75 (oracle/orc_lmpack.c, linear plane_find as in the reference).  Host C on both sides: no GPU.
The placement and the whole LmPack_Pack run with the GPU tests (tests/test_gpu_lmpack.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from pim_amd import api, scenes
from tests import prepare_util as pu
from tests.util import load_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    lib = C.CDLL(os.path.join(ROOT, "pim_amd", "libpimhost.so"))
    lib.LmPack_GroupCharts.restype = C.c_int32
    lib.LmPack_GroupCharts.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_void_p]
    lib.LmPack_SortCharts.restype = None
    lib.LmPack_SortCharts.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    return lib


def group_both(host, nodes, dist, deg, max_width, seed):
    lib = load_oracle(True)
    lib.orc_lightmap_chart_group.restype = C.c_int32
    n = nodes.shape[0]
    want_nodes, want_first = np.zeros(n, api.CHARTNODE_DTYPE), np.zeros(n + 1, np.int32)
    nc = lib.orc_lightmap_chart_group(nodes.ctypes.data_as(C.c_void_p), C.c_int32(n), C.c_float(dist), C.c_float(deg), C.c_float(max_width),
                                      C.c_uint32(seed), want_nodes.ctypes.data_as(C.c_void_p), want_first.ctypes.data_as(C.c_void_p))
    got_nodes, got_first = nodes.copy(), np.zeros(n + 1, np.int32)
    gc = host.LmPack_GroupCharts(got_nodes.ctypes.data, n, dist, deg, max_width, seed, got_first.ctypes.data)
    assert gc == nc
    assert (got_first[:nc + 1] == want_first[:nc + 1]).all()
    assert got_nodes.tobytes() == want_nodes.tobytes()
    return got_nodes, got_first[:nc + 1]


@pytest.mark.parametrize("count,tpu,max_width", [(30, 4.0, 64 / 3.0), (60, 2.0, 1024 / 3.0), (12, 16.0, 32 / 3.0)])
def test_chart_group_matches_the_reference_scan(host, count, tpu, max_width):
    """boxes share planes pairwise, spheres do not; small atlases make the wide faces split (chart_split, several levels deep)"""
    dr = pu.draw_table(count=count, seed=count + 1)
    nodes = pu.oracle_chartnodes(load_oracle(True), dr, tpu)
    assert nodes.shape[0] > 30
    got, first = group_both(host, nodes, 0.1, 15.0, max_width, seed=7)
    sizes = np.diff(first)
    assert sizes.min() >= 1 and sizes.sum() == nodes.shape[0] and ((sizes >= 2).any() or max_width < 20)
    # every node is in exactly one chart
    key = lambda a: sorted(zip(a["drawableIndex"].tolist(), a["vertIndex"].tolist()))
    assert key(got) == key(nodes)


def test_chart_group_cornell_walls_split(host):
    sc = scenes.cornell_box()
    dr = scenes.drawables_of(sc)
    nodes = pu.oracle_chartnodes(load_oracle(True), dr, 8.0)
    assert nodes.shape[0] == 108
    whole, first_whole = group_both(host, nodes, 0.1, 15.0, 1e9, seed=3)         # nothing is too wide: plane groups only
    split, first_split = group_both(host, nodes, 0.1, 15.0, 40.0, seed=3)        # 10 m walls at 8 texels/m are 80 wide
    assert first_split.shape[0] > first_whole.shape[0]
    # another seed: other means, (in general) another split of the same walls
    group_both(host, nodes, 0.1, 15.0, 40.0, seed=4)


def test_chart_group_edge_cases(host):
    nodes = np.zeros(9, api.CHARTNODE_DTYPE)
    rng = np.random.default_rng(2)
    nodes["triCoord"] = rng.uniform(0, 30, (9, 6)).astype(np.float32)
    nodes["plane"][:, 2] = 1.0
    nodes["plane"][:, 3] = [0.0, 0.05, 0.099, 0.11, 0.2, 0.0, np.nan, 5e30, -0.05]
    nodes["plane"][4, :3] = [0.0, 0.3, 0.95]                                      # tilted by more than 15 degrees
    nodes["plane"][5, :3] = 0.0                                                   # a degenerate triangle's plane: matches nothing, itself included
    nodes["vertIndex"] = np.arange(9) * 3
    group_both(host, nodes, 0.1, 15.0, 1e9, seed=1)
    group_both(host, nodes, 0.0, 15.0, 1e9, seed=1)                                # nothing is closer than 0: one chart per node
    got, first = group_both(host, nodes, 1e-3, 15.0, 1e9, seed=1)
    # identical triangles cannot be divided by 2-means: the chart is kept whole after a bounded number of tries
    same = np.zeros(6, api.CHARTNODE_DTYPE)
    same["triCoord"] = [0, 0, 500, 0, 0, 500]; same["plane"][:, 2] = 1.0; same["vertIndex"] = np.arange(6) * 3
    got, first = group_both(host, same, 0.1, 15.0, 10.0, seed=9)
    assert first.tolist() == [0, 6]
    group_both(host, nodes[:1], 0.1, 15.0, 1.0, seed=1)
    group_both(host, nodes[:0], 0.1, 15.0, 1.0, seed=1)
    # distance terms whose quotient by the threshold does not fit an int64 (a tiny threshold, huge terms): those planes leave the
    # buckets and are compared one by one -- equal huge terms still share a chart, as in the reference's scan
    big = np.zeros(8, api.CHARTNODE_DTYPE)
    big["triCoord"] = rng.uniform(0, 30, (8, 6)).astype(np.float32)
    big["plane"][:, 2] = 1.0
    big["plane"][:, 3] = [5e30, 1.0, 5e30, 1e-3, 1.0, 3e35, 3e35, np.inf]
    big["vertIndex"] = np.arange(8) * 3
    got, first = group_both(host, big, 1e-20, 15.0, 1e9, seed=1)
    assert np.diff(first).tolist().count(2) == 3                               # the two 5e30, the two 1.0, the two 3e35
    group_both(host, big, 0.1, 15.0, 1e9, seed=1)


def test_chart_sort_is_the_references_quicksort(host):
    lib = load_oracle(True)
    lib.orc_lightmap_chart_sort.restype = None
    rng = np.random.default_rng(11)
    for n in (0, 1, 3, 4, 5, 9, 64, 1000, 20000):
        for ties in (False, True):
            areas = rng.uniform(1, 500, n).astype(np.float32)
            if ties:
                areas = np.round(areas / 25).astype(np.float32) * 25           # many equal areas: their order is the algorithm's
            want, got = np.zeros(max(1, n), np.int32), np.zeros(max(1, n), np.int32)
            lib.orc_lightmap_chart_sort(areas.ctypes.data_as(C.c_void_p), C.c_int32(n), want.ctypes.data_as(C.c_void_p))
            host.LmPack_SortCharts(areas.ctypes.data, n, got.ctypes.data)
            assert (got[:n] == want[:n]).all()
            assert sorted(got[:n].tolist()) == list(range(n)) and (np.diff(areas[got[:n]]) >= 0).all()
