"""-m gpu: the light-histogram exchange entry points of the one-process-per-GPU form (pimgpu_scene_live_get_dev / _set_dev,
include/pimgpu.h) through the C-ABI: TWO replicas of a scene in one process stand for two ranks -- each traces its own samples,
and before every round the increments both made since the last UpdateDists are summed (pim_amd/dist.py's formula, the sum done on
the device with torch tensors as bench.py does), written back into both and UpdateDists run on each.  Checked bit for bit against
ONE oracle scene that traces both ranks' samples of a round after one UpdateDists: every replica's images, its distributions,
and the histograms (live_A + live_B - base = the oracle's)."""
import numpy as np
import pytest

from pim_amd import api, scenes
from pim_amd import dist as pdist
from tests.gpu_util import gpu_init, assert_bits_equal
from tests.util import OracleScene

pytestmark = pytest.mark.gpu

W, H, ROUNDS = 96, 54, 4        # enough paths that cells pass Dist1D_Update's 30-hit threshold (math/dist1d.c:140-144)


class _TwoRanks:
    """stands in for torch.distributed in allreduce_light_histograms: `other` is the peer's increment"""
    class ReduceOp:
        SUM = "sum"

    def __init__(self):
        self.peer = None

    def all_reduce(self, t, op=None):
        t.add_(self.peer)

    def get_backend(self):
        return "nccl"


def _params(sample, update, weight):
    p = scenes.trace_params(0, seed=3, max_bounces=4, update_dists=update)
    p.sampleIndex = sample
    p.sampleWeight = weight            # a rank's image is the running mean of ITS samples
    return p


@pytest.mark.parametrize("prims", ["boxes", "spheres"])
def test_two_replicas_exchange_histograms_like_one_process(prims):
    import torch
    gpu_init()
    sc = scenes.cornell_box(prims=prims, emissive_attempts=64)
    reps = [api.Scene(sc.desc), api.Scene(sc.desc)]
    traces = [api.Trace(W, H), api.Trace(W, H)]
    o = OracleScene(sc.desc, accel=1, threads=4)
    cam = scenes.pt_test_camera()
    try:
        n = reps[0].live_count()
        assert n == reps[0].info().gridLen * reps[0].info().emissiveCount > 0
        live = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in reps]
        base = torch.zeros(n, dtype=torch.int32, device="cuda")
        reps[0].live_get_dev(base.data_ptr())
        assert_bits_equal(base.cpu().numpy().view(np.uint32), o.download_lights()["live"], "initial histograms")
        want = [[np.zeros((W * H, 3), np.float32) for _ in range(3)] for _ in reps]
        pdf0 = o.download_lights()["pdf"].copy()
        for i in range(ROUNDS):
            # the exchange, as bench.py's step() does it on every rank
            for r, g in enumerate(reps):
                g.live_get_dev(live[r].data_ptr())
            summed = []
            for r in range(2):
                d = _TwoRanks(); d.peer = live[1 - r] - base
                summed.append(pdist.allreduce_light_histograms(live[r], base, 2, d))
            torch.cuda.synchronize()
            assert torch.equal(summed[0], summed[1])
            for r, g in enumerate(reps):
                g.live_set_dev(summed[r].data_ptr())
                g.update_dists()
            reps[0].live_get_dev(base.data_ptr())
            # one process: UpdateDists once, then both ranks' samples of the round
            for r, g in enumerate(reps):
                s = pdist.sample_indices(r, 2, ROUNDS)[i]
                dg, do = scenes.default_dof(), scenes.default_dof()
                dg.autoFocus = do.autoFocus = 0
                traces[r].trace(g, cam, dg, _params(s, 0, 1.0 / (i + 1)))
                o.trace_image(want[r], W, H, cam, do, _params(s, 1 if r == 0 else 0, 1.0 / (i + 1)))
        one = o.download_lights()
        assert not np.array_equal(one["pdf"], pdf0), "the feedback never passed the 30-hit threshold: the test would prove nothing"
        for r in range(2):
            color, albedo, normal = traces[r].download()
            assert_bits_equal(color, want[r][0], f"replica {r} color"); assert_bits_equal(albedo, want[r][1], f"replica {r} albedo")
            assert_bits_equal(normal, want[r][2], f"replica {r} normal")
            gl = reps[r].download_lights()
            assert_bits_equal(gl["pdf"], one["pdf"], f"replica {r} pdf"); assert_bits_equal(gl["cdf"], one["cdf"], f"replica {r} cdf")
        la, lb = reps[0].download_lights()["live"], reps[1].download_lights()["live"]
        b = base.cpu().numpy().view(np.uint32)
        assert_bits_equal(la + lb - b, one["live"], "histograms: the common base plus both replicas' last increments")
        assert (one["live"] > b).any()
    finally:
        for t in traces:
            t.close()
        for g in reps:
            g.close()
        o.close()


def test_live_exchange_argument_errors():
    gpu_init()
    sc = scenes.cornell_box()
    g = api.Scene(sc.desc)
    try:
        with pytest.raises(api.PimGpuError):
            g.live_get_dev(0)
        with pytest.raises(api.PimGpuError):
            g.live_set_dev(0)
    finally:
        g.close()
