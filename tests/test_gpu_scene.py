"""-m gpu: scene set-up (SetupEmissives, SetupLightGrid, UpdateDists) against the oracle, BIT-EXACT:
emissive maps, per-cell Dist1D lengths, pdf and cdf tables."""
import numpy as np
import pytest

from pim_amd import api, scenes
from tests.gpu_util import gpu_init, assert_bits_equal
from tests.util import OracleScene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["boxes", "spheres"])
def test_cornell_light_setup_bit_exact(name):
    gpu_init()
    sc = scenes.cornell_box(name, emissive_attempts=64 if name == "spheres" else 1000)
    g = api.Scene(sc.desc)
    o = OracleScene(sc.desc, accel=1)
    gi, oi = g.info(), o.info()
    assert gi.emissiveCount == oi.emissiveCount and list(gi.gridSize) == list(oi.gridSize) and gi.activeCells == oi.activeCells
    gl, ol = g.download_lights(), o.download_lights()
    for k in ("vertToEmit", "emitToVert", "cellLength", "pdf", "cdf", "live"):
        assert_bits_equal(gl[k], ol[k], k)
    g.close()


def test_synthetic_light_setup_bit_exact():
    gpu_init()
    sc = scenes.synthetic_scene(target_tris=4000, seed=8, extent=12.0, emissive_attempts=32, meters_per_cell=1.5)
    g = api.Scene(sc.desc)
    o = OracleScene(sc.desc, accel=1)
    gl, ol = g.download_lights(), o.download_lights()
    for k in ("vertToEmit", "emitToVert", "cellLength", "pdf", "cdf"):
        assert_bits_equal(gl[k], ol[k], k)
    assert 0 < g.info().activeCells <= g.info().gridLen
    g.close()
