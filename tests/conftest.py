"""pytest configuration: the `gpu` marker + shared fixtures (synthetic subject, models, golden files)."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def psnr(a, b):
    """The reference's own formula (im2mesh/utils/eval.py:6-9)."""
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else -10.0 * np.log10(mse)


@pytest.fixture(scope="session")
def scene():
    from arah_release_amd import synthetic
    return synthetic.SyntheticScene(seed=0)


_MODELS = {}


def get_model(name, device="cpu"):
    """Synthetic-weight model per builtin config (cached: the hypernetwork has 87 M parameters)."""
    from arah_release_amd import config
    key = (name, str(device))
    if key not in _MODELS:
        _MODELS[key] = config.build_synthetic_model(name, device=device)
    return _MODELS[key]


@pytest.fixture(scope="session")
def model_factory():
    return get_model
