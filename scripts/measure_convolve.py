"""Cubemap_Convolve on the GPU (pimgpu_cubemap_convolve, host faces in and out) against the oracle on the host cores.
    python scripts/measure_convolve.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pim_amd import api
from tests.util import oracle_cubemap_convolve

api.init(0)
rng = np.random.default_rng(1)
for size, samples in ((64, 64), (256, 64), (256, 1024), (1024, 64)):
    elems = api.cubemap_elem_count(size)
    faces = [np.ascontiguousarray(rng.random(size * size * 3).astype(np.float32)) for _ in range(6)]
    conv = [np.zeros(elems * 4, np.float32) for _ in range(6)]
    api.cubemap_convolve(faces, conv, size, samples, 1.0)
    t0 = time.perf_counter(); reps = 5
    for r in range(reps):
        api.cubemap_convolve(faces, conv, size, samples, 0.5, sampleIndex=r + 1)
    gpu = (time.perf_counter() - t0) / reps
    texels = 6 * sum((size >> m) ** 2 for m in range(size.bit_length()))
    rec = {"size": size, "samples": samples, "texels": texels, "gpu_ms_per_call_host_to_host": gpu * 1e3, "gpu_Msamples_per_s": texels * samples / gpu / 1e6}
    if texels * samples <= 6e8:
        co = [np.zeros(elems * 4, np.float32) for _ in range(6)]
        t0 = time.perf_counter()
        oracle_cubemap_convolve(faces, co, size, samples, 1.0, threads=16)
        cpu = time.perf_counter() - t0
        rec.update({"oracle_16_threads_ms": cpu * 1e3, "oracle_Msamples_per_s": texels * samples / cpu / 1e6})
    print(json.dumps(rec), flush=True)
api.shutdown()
