import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "image2video-synthesis-using-cinns_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Returns (arrays: dict[str, np.ndarray], meta: dict) of tests/golden/<name>.npz."""
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    arrays = {k: f[k] for k in f.files if k != "meta"}
    meta = json.loads(bytes(f["meta"]).decode())
    return arrays, meta


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2 in float64 (the parity metric of SURVEY §8d)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="session")
def torch_sd():
    import torch

    def conv(sd):
        return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    return conv
