import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def second_gpu_process(gpu):
    """Tests that start a CHILD process on the box's one GPU while this process holds it: some boxes of the pool do not let a
    second process onto the device (round 6: such children hung at their first device work).  A 60-second probe decides; a box
    that fails it SKIPS those tests instead of hanging in them."""
    import subprocess

    code = "import torch; t = torch.ones(1024, device='cuda'); torch.cuda.synchronize(); assert float(t.sum()) == 1024.0"
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=60)
        ok = r.returncode == 0
    except subprocess.TimeoutExpired:
        ok = False
    if not ok:
        pytest.skip("a second process cannot use this box's GPU while the test process holds it (probe hung or failed)")
    return True
