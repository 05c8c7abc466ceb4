"""-m gpu: the device-list form of the C-ABI (pimgpu_init_devices): scene replicas, an image rendered as one band of rows
per context, lightmap tiles round-robin, the replicas' light histograms summed before every UpdateDists -- every result
equal to the oracle's (= the one-device result) bit for bit.  The test box has one GPU, so the list names device 0 two and
three times: as many contexts, streams, replicas and host threads as entries, on one GPU (tests/multi_device_worker.py).
Each list runs in a process of its own: the library keeps one device list per process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [1, 2, 3])
def test_device_list_matches_oracle(n):
    # a child that dies names the HIP call it died in: native frames on a fatal signal, one stderr line per call of the library's init
    env = dict(os.environ, PIMGPU_SEGV_BACKTRACE="1", PIMGPU_INIT_TRACE="1")
    run = subprocess.run([sys.executable, "-u", "-X", "faulthandler", "-m", "tests.multi_device_worker", str(n)], cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    ok = run.returncode == 0 and f"MULTI-DEVICE OK {n}" in run.stdout
    assert ok, run.stdout[-3000:] + run.stderr[-3000:]
