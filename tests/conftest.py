import os
import sys

import numpy as np
import pytest

os.environ.setdefault("AC_TESTING", "1")   # enables the library's test hook ac_set_force_generic (read when first called)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a `-m gpu` run on a box without a GPU is a configuration error, not a silent skip
    pass


@pytest.fixture(autouse=True)
def _deterministic_draws():
    """Every test starts from the same generator state (CPU and device): draws without an explicit generator are the same
    in every run, so a test cannot pass or fail by the luck of its data."""
    import torch
    torch.manual_seed(20260)
    yield


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


def rel_peak(a, b):
    """max_frame(||a-b||_inf / ||b||_inf) over frames (axis 2 = filters) -- SURVEY 8(c) MDCT metric."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    num = np.max(np.abs(a - b), axis=2)
    den = np.maximum(np.max(np.abs(b), axis=2), 1e-30)
    return float(np.max(num / den))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def rel_elem(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.abs(b)))


def tonality_err(t, t_ref):
    """Worst |t - t_ref| in units of the bar of SURVEY 8(c): element-wise 1e-4 relative with atol 1e-6 near 0
    (<= 1.0 passes).  Works on numpy arrays and on torch tensors."""
    try:
        import torch
        if isinstance(t, torch.Tensor) or isinstance(t_ref, torch.Tensor):
            t = t.detach().double().cpu().numpy() if isinstance(t, torch.Tensor) else t
            t_ref = t_ref.detach().double().cpu().numpy() if isinstance(t_ref, torch.Tensor) else t_ref
    except ImportError:
        pass
    t = np.asarray(t, dtype=np.float64)
    t_ref = np.asarray(t_ref, dtype=np.float64)
    return float(np.max(np.abs(t - t_ref) / (1e-4 * np.abs(t_ref) + 1e-6)))
