"""pytest configuration: the ``gpu`` marker and shared helpers."""

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Float tolerance of the parity claim: 1e-5 relative (fp32), with a small absolute floor for values
# that are differences of O(1) quantities.  Decisions (bool / integer outputs) are compared exactly.
RTOL = 1e-5
ATOL = 1e-6


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a MI355X (runs the HIP kernels through the C ABI)")


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def assert_close(got, ref, rtol=RTOL, atol=ATOL, what=""):
    got = got.detach().cpu() if isinstance(got, torch.Tensor) else torch.as_tensor(got)
    ref = ref.detach().cpu() if isinstance(ref, torch.Tensor) else torch.as_tensor(ref)
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} != {tuple(ref.shape)}"
    if ref.dtype in (torch.bool, torch.int32, torch.int64, torch.uint8):
        bad = (got != ref).sum().item()
        assert bad == 0, f"{what}: {bad} of {ref.numel()} exact-compare elements differ"
        return
    torch.testing.assert_close(got, ref.to(got.dtype), rtol=rtol, atol=atol, equal_nan=True, msg=lambda m: f"{what}: {m}")


def assert_exact(got, ref, what=""):
    got = got.detach().cpu()
    ref = ref.detach().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} != {tuple(ref.shape)}"
    same = (got == ref) | (torch.isnan(got.float()) & torch.isnan(ref.float()))
    bad = (~same).sum().item()
    assert bad == 0, f"{what}: {bad} of {ref.numel()} elements are not bit-identical"


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
