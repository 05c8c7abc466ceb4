"""Run by tests/test_gpu_multi_device.py in a process of its own (the library keeps ONE device list per process):
pimgpu_init_devices with a list of N entries -- all device 0 on the one-GPU test box: N contexts, streams, scene replicas and
image copies on one GPU, which exercises every multi-device path except the physical peer copies -- against the oracle, bit for bit.

    python -m tests.multi_device_worker <N>"""
import ctypes as C
import sys

import numpy as np

from pim_amd import api, scenes
from tests.gpu_util import assert_bits_equal
from tests.util import OracleScene


def stage(name):
    """progress on stderr: a process that dies between two of these has said where"""
    print(f"[worker] {name}", file=sys.stderr, flush=True)


def main(n):
    stage("init")
    api.init_devices([0] * n)
    lib = api.load_library()
    assert lib.pimgpu_device_count() == n
    sc = scenes.textured_cornell()
    g = api.Scene(sc.desc)
    o = OracleScene(sc.desc, accel=1)
    gl, ol = g.download_lights(), o.download_lights()
    for k in ("vertToEmit", "emitToVert", "cellLength", "pdf", "cdf", "live"):
        assert_bits_equal(gl[k], ol[k], k)

    stage("image")
    # image: bands of rows, one per context; 50 rows do not divide by 3
    W, H = 96, 50
    t = api.Trace(W, H)
    img = [np.zeros((W * H, 3), np.float32) for _ in range(3)]
    cam = scenes.pt_test_camera()
    dg, do = scenes.default_dof(), scenes.default_dof()
    for s in range(4):
        p = scenes.trace_params(s, seed=17, max_bounces=5)
        t.trace(g, cam, dg, p)
        so = o.trace_image(img, W, H, cam, do, p)
        sg = t.stats()
        assert (sg.extensionRays, sg.shadowRays, sg.otherRays, sg.paths) == (so.extensionRays, so.shadowRays, so.otherRays, so.paths), f"frame {s}"
        assert dg.focalLength == do.focalLength
    color, albedo, normal = t.download()
    assert_bits_equal(color, img[0], "color"); assert_bits_equal(albedo, img[1], "albedo"); assert_bits_equal(normal, img[2], "normal")
    assert t.stddev() == o.stddev(img[0], W, H)                    # gathers the bands on context 0
    assert t.stddev_noalloc() == t.stddev()
    gl, ol = g.download_lights(), o.download_lights()
    for k in ("live", "pdf", "cdf"):
        assert_bits_equal(gl[k], ol[k], k + " after 4 frames")      # the replicas' histograms were summed before every UpdateDists
    assert (gl["live"] > 0).any()

    stage("raygen")
    # the entry points that run on the first device see (and feed) the same histograms
    origin = np.array([0.5, -1.0, 0.25, 0.0], np.float32)
    cg, dgn = g.raygen(origin, 20000, sampleIndex=1, seed=3, maxBounces=4)
    co, don = o.raygen(origin, 20000, sampleIndex=1, seed=3, maxBounces=4)
    assert_bits_equal(cg, co, "raygen colors")
    t.trace(g, cam, dg, scenes.trace_params(4, seed=17, max_bounces=5)); o.trace_image(img, W, H, cam, do, scenes.trace_params(4, seed=17, max_bounces=5))
    assert_bits_equal(t.download()[0], img[0], "color after raygen")

    stage("sample-index sharding")
    # BASELINE config 3's sharding inside one process (pimgpu_trace_image_samples): 5 samples over n contexts, round by round;
    # ONE oracle scene plays every context (its UpdateDists at the start of a round sees all the contexts' increments, as the
    # summed replicas do), each context with its own running mean and its own autofocus state
    t2 = api.Trace(W, H)
    img2 = [np.zeros((W * H, 3), np.float32) for _ in range(3)]
    dg2, do2 = scenes.default_dof(), scenes.default_dof()
    first = scenes.trace_params(0, seed=29, max_bounces=5)
    t2.trace(g, cam, dg2, first); o.trace_image(img2, W, H, cam, do2, first)          # something to blend into
    rays_seen = []

    def samples_case(S, weight, firstIndex):
        """one pimgpu_trace_image_samples call and the same schedule on the oracle; img2 / dg2 / do2 carry over"""
        ps = scenes.trace_params(0, seed=29, max_bounces=5); ps.sampleIndex = firstIndex; ps.sampleWeight = weight
        t2.trace_samples(g, cam, dg2, ps, S)
        acc = [[np.zeros((W * H, 3), np.float32) for _ in range(3)] for _ in range(n)]
        dofs = [api.PimDofInfo.from_buffer_copy(do2) for _ in range(n)]
        rays = 0
        for r in range((S + n - 1) // n):
            for i in range(n):
                k = r * n + i
                if k < S:
                    q = scenes.trace_params(0, seed=29, max_bounces=5, update_dists=1 if i == 0 else 0)
                    q.sampleIndex = firstIndex + k; q.sampleWeight = 1.0 / (r + 1)
                    so = o.trace_image(acc[i], W, H, cam, dofs[i], q)
                    rays += so.extensionRays + so.shadowRays + so.otherRays
        counts = [(S - i + n - 1) // n if i < S else 0 for i in range(n)]
        assert sum(counts) == S
        for c in range(3):
            mean = acc[0][c] * (np.float32(counts[0]) / np.float32(S))
            for i in range(1, n):
                if counts[i] > 0:
                    mean = mean + acc[i][c] * (np.float32(counts[i]) / np.float32(S))
            img2[c] = img2[c] + (mean - img2[c]) * np.float32(weight)
        got = t2.download()
        for c, name in enumerate(("color", "albedo", "normal")):
            assert_bits_equal(got[c], img2[c], f"{S} samples: " + name)
        sg = t2.stats()
        assert sg.extensionRays + sg.shadowRays + sg.otherRays == rays and sg.paths == S * W * H
        assert dg2.focalLength == dofs[0].focalLength
        C.memmove(C.byref(do2), C.byref(dofs[0]), C.sizeof(do2))
        rays_seen.append(rays)

    samples_case(5, 0.25, 10)
    samples_case(max(1, n - 1), 0.5, 40)                 # fewer samples than contexts: the last context sits the call out
    samples_case(2 * n, 1.0, 50)                         # weight 1: the image becomes old + (mean - old), as Pt_Trace's lerp gives it
    assert all(r > 0 for r in rays_seen)
    try:
        t2.trace_samples(g, cam, dg2, scenes.trace_params(0, seed=29, max_bounces=5), 0)
        raise AssertionError("a call for 0 samples must be refused")
    except api.PimGpuError:
        pass
    # the contexts' bands were refreshed: a row-sharded frame on top of the result
    nxt = scenes.trace_params(3, seed=29, max_bounces=5)
    t2.trace(g, cam, dg2, nxt); o.trace_image(img2, W, H, cam, do2, nxt)
    assert_bits_equal(t2.download()[0], img2[0], "a row-sharded frame after the sample-sharded call")
    gl, ol = g.download_lights(), o.download_lights()
    for k in ("live", "pdf", "cdf"):
        assert_bits_equal(gl[k], ol[k], k + " after the sample-sharded call")
    t2.close()

    stage("lightmap bake")
    # lightmap bake: 64x64 tiles round-robin over the contexts; 96^2 pages = 2x2 tiles with a ragged edge
    S = 96
    pages = scenes.lightmap_pages_for(sc, S, 3.0, max_pages=2)
    pg = [p.copy() for p in pages]; po = [p.copy() for p in pages]
    for passIndex, (slice_, spp) in enumerate([(1.0, 2), (0.6, 2)]):
        g.lightmap_bake(pg, S, 3.0, slice_, spp, passIndex=passIndex, seed=5, maxBounces=4)
        o.lightmap_bake(po, S, 3.0, slice_, spp, passIndex=passIndex, seed=5, maxBounces=4)
        for a, b in zip(pg, po):
            assert_bits_equal(a, b, f"page after pass {passIndex}")
    begin, end = S * S // 3, S * S + 1000                           # a sub-range across the page boundary
    g.lightmap_bake(pg, S, 3.0, 1.0, 1, passIndex=9, seed=5, maxBounces=4, texelBegin=begin, texelEnd=end)
    o.lightmap_bake(po, S, 3.0, 1.0, 1, passIndex=9, seed=5, maxBounces=4, texelBegin=begin, texelEnd=end)
    for a, b in zip(pg, po):
        assert_bits_equal(a, b, "sub-range pages")

    stage("resident lightmaps")
    # the same with pages resident in HBM: every context keeps a replica and bakes its tiles of it, the download merges them
    hs = [p.copy() for p in pages]; po2 = [p.copy() for p in pages]
    lm = api.LightmapSet(len(pages), S)
    lm.upload(hs)
    for passIndex, (slice_, spp) in enumerate([(1.0, 2), (0.6, 2)]):
        lm.bake(g, 3.0, slice_, spp, passIndex=passIndex + 20, seed=5, maxBounces=4)
        o.lightmap_bake(po2, S, 3.0, slice_, spp, passIndex=passIndex + 20, seed=5, maxBounces=4)
    lm.download(hs)
    for a, b in zip(hs, po2):
        assert_bits_equal(a, b, "resident pages")
    lm.close()

    stage("rebuild, shutdown, second list")
    g.rebuild_bvh()
    ro, rd = scenes.random_rays(sc, 5000, seed=2)
    assert g.intersect(ro, rd).tobytes() == o.intersect(ro, rd).tobytes()
    t.close(); g.close()
    api.shutdown()
    # a second list in the same process after shutdown
    api.init_devices([0])
    sc2 = scenes.cornell_box()
    g = api.Scene(sc2.desc); g.close()
    api.shutdown()
    stage("done")
    print(f"MULTI-DEVICE OK {n}", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]))
