"""The repository's entry module: its small end-to-end check of the hot path against the oracle step runs as part of the
GPU suite too."""
import importlib
import os
import sys

import pytest


@pytest.mark.gpu
def test_smoke_step_matches_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    entry = importlib.import_module("__graft_entry__")
    entry.smoke()
