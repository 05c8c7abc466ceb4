"""CPU, world_size 2, gloo: the N > 1 path of bench.py -- sample sharding by index plus ONE reduce of the
per-rank mean images -- gives the image a single process renders with all the samples (up to fp32 reassociation
of the final mean), using the CPU oracle as the renderer on each rank; and the lightmap texel tiling is a
partition."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pim_amd import dist as pdist  # noqa: E402

W, H, STEPS = 48, 27, 3


def _render(indices):
    from pim_amd import scenes
    from tests.util import OracleScene
    sc = scenes.cornell_box(emissive_attempts=64)
    o = OracleScene(sc.desc, accel=1, threads=2)
    img = [np.zeros((W * H, 3), np.float32) for _ in range(3)]
    dof = scenes.default_dof()
    dof.autoFocus = 0
    for k, s in enumerate(indices):
        p = scenes.trace_params(k, seed=3, max_bounces=3, update_dists=0)
        p.sampleIndex = s
        o.trace_image(img, W, H, scenes.pt_test_camera(), dof, p)
    return img[0]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = _render(pdist.sample_indices(rank, world, STEPS))
    t = torch.from_numpy(mine.copy())
    pdist.reduce_mean_image(t, world, dist, dst=0)
    if rank == 0:
        np.save(out, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sample_sharding_and_reduce_world2(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    # single process, same 6 samples: mean of the two shards' means == overall mean (equal counts)
    a = _render(pdist.sample_indices(0, 2, STEPS))
    b = _render(pdist.sample_indices(1, 2, STEPS))
    assert np.array_equal(got, (a + b) * np.float32(0.5))
    assert sorted(pdist.sample_indices(0, 2, STEPS) + pdist.sample_indices(1, 2, STEPS)) == list(range(2 * STEPS))
    assert got.mean() > 0.01


def test_texel_tiles_partition():
    for total in (0, 1, 4095, 4096, 4097, 64 * 64 * 37 + 5):
        for world in (1, 2, 3, 8):
            seen = 0
            prev_end = 0
            for r in range(world):
                b, e = pdist.texel_range(r, world, total)
                assert b == prev_end or b == e == total
                assert b % (64 * 64) == 0 or b == total
                seen += e - b; prev_end = e
            assert seen == total


def test_sample_ranges_and_tiles_partition():
    for total in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                b, e = pdist.sample_range(r, world, total)
                got += list(range(b, e))
                assert abs((e - b) - total / world) < 1.0
            assert got == list(range(total))
    for lm_count, lm_size in ((1, 64), (3, 1024), (2, 100)):
        per_row = (lm_size + 63) // 64
        for world in (1, 2, 8):
            tiles = sorted(t for r in range(world) for t in pdist.tiles_of(r, world, lm_count, lm_size))
            assert tiles == list(range(lm_count * per_row * per_row))
            counts = [len(pdist.tiles_of(r, world, lm_count, lm_size)) for r in range(world)]
            assert max(counts) - min(counts) <= 1


def _worker_unequal(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 5
    b, e = pdist.sample_range(rank, world, total)
    t = torch.from_numpy(_render(list(range(b, e))).copy())
    pdist.reduce_mean_image(t, world, dist, dst=0, weight=e - b, total=total)
    if rank == 0:
        np.save(out, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_with_a_remainder_world2(tmp_path):
    """--total-spp 5 on 2 ranks: 3 + 2 samples, weighted sum == the mean of all five samples (fp32 reassociation aside)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker_unequal, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    a, b = _render([0, 1, 2]), _render([3, 4])
    want = a * np.float32(3.0 / 5.0) + b * np.float32(2.0 / 5.0)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)
    assert got.mean() > 0.01


# ---- the light histograms across ranks: the N-process render equals the single-process one BIT FOR BIT ----

ROUNDS = 4


def _scene():
    from pim_amd import scenes
    from tests.util import OracleScene
    sc = scenes.cornell_box(emissive_attempts=64)
    return OracleScene(sc.desc, accel=1, threads=2)


def _trace(o, img, sample, update):
    from pim_amd import scenes
    dof = scenes.default_dof()
    dof.autoFocus = 0
    p = scenes.trace_params(0, seed=3, max_bounces=4, update_dists=update)
    p.sampleIndex = sample
    return o.trace_image(img, W2, H2, scenes.pt_test_camera(), dof, p)


W2, H2 = 96, 54        # enough paths that cells pass Dist1D_Update's 30-hit threshold (math/dist1d.c:140-144): the pdfs really move


def _worker_live(rank, world, port, out):
    """what bench.py --gpus N does each step: sum the ranks' histogram increments, UpdateDists, remember the result as the
    next base, then trace this rank's sample with no further update"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = _scene()
    img = [np.zeros((W2 * H2, 3), np.float32) for _ in range(3)]
    base = torch.from_numpy(o.download_lights()["live"].view(np.int32).copy())
    for i, s in enumerate(pdist.sample_indices(rank, world, ROUNDS)):
        live = torch.from_numpy(o.download_lights()["live"].view(np.int32).copy())
        live = pdist.allreduce_light_histograms(live, base, world, dist)
        o.set_live(live.numpy().view(np.uint32))
        o.update_dists()
        base = torch.from_numpy(o.download_lights()["live"].view(np.int32).copy())
        img_s = [np.zeros((W2 * H2, 3), np.float32) for _ in range(3)]
        _trace(o, img_s, s, 0)
        w = np.float32(1.0 / (i + 1))
        for k in range(3):
            img[k] += (img_s[k] - img[k]) * w
    lights = o.download_lights()
    t = torch.from_numpy(img[0].copy())
    pdist.reduce_mean_image(t, world, dist, dst=0)
    np.savez(out + f".{rank}.npz", pdf=lights["pdf"], cdf=lights["cdf"], live=lights["live"], base=base.numpy().view(np.uint32), mine=img[0])
    if rank == 0:
        np.save(out, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_light_histograms_allreduced_world2_equals_one_process_bit_for_bit(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker_live, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    # ONE process, ONE scene, the same rounds: UpdateDists at the start of a round, then both ranks' samples of that round
    o = _scene()
    a = np.zeros((W2 * H2, 3), np.float32); b = np.zeros((W2 * H2, 3), np.float32)
    pdf0 = o.download_lights()["pdf"].copy()
    for i in range(ROUNDS):
        for rank, acc in ((0, a), (1, b)):
            img_s = [np.zeros((W2 * H2, 3), np.float32) for _ in range(3)]
            _trace(o, img_s, pdist.sample_indices(rank, 2, ROUNDS)[i], 1 if rank == 0 else 0)
            acc += (img_s[0] - acc) * np.float32(1.0 / (i + 1))
    one = o.download_lights()
    assert not np.array_equal(one["pdf"], pdf0), "the feedback never passed the 30-hit threshold: the test would prove nothing"
    # every rank's own image, its sampling distributions ...
    assert np.array_equal(r0["mine"].view(np.uint32), a.view(np.uint32)) and np.array_equal(r1["mine"].view(np.uint32), b.view(np.uint32))
    for r in (r0, r1):
        assert np.array_equal(r["pdf"].view(np.uint32), one["pdf"].view(np.uint32)) and np.array_equal(r["cdf"].view(np.uint32), one["cdf"].view(np.uint32))
    # ... the histograms (each rank holds the common base plus its own last-round increments) and the reduced image
    assert np.array_equal(r0["base"], r1["base"])
    assert np.array_equal(r0["live"] + r1["live"] - r0["base"], one["live"]) and (one["live"] > r0["base"]).any()
    assert np.array_equal(got.view(np.uint32), ((a + b) * np.float32(0.5)).view(np.uint32))
