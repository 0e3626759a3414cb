"""The C-ABI library loads on a machine without a GPU and exports every symbol the header declares."""

import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    import glob

    syms = set()
    for path in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms.update(re.findall(r"\b(ilx_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


def test_header_declares_something():
    syms = declared_symbols()
    assert "ilx_version" in syms and "ilx_last_error" in syms and "ilx_mdp_step" in syms and len(syms) >= 20


def test_library_exports_every_declared_symbol():
    from isaaclab_amd import _abi

    missing = [s for s in declared_symbols() if not hasattr(_abi.lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    assert _abi.lib.ilx_version() == 1


def test_argument_validation_needs_no_gpu():
    from isaaclab_amd import _abi

    # unknown op and negative sizes are rejected before any HIP call
    assert _abi.math_op(999, None, None, None, None, None, None, 4, None) == -1
    assert b"unknown op" in _abi.lib.ilx_last_error()
    assert _abi.math_op(0, None, None, None, None, None, None, -1, None) == -1


def test_cpu_tensors_are_refused():
    import torch

    from isaaclab_amd.utils import math as m

    with pytest.raises(Exception, match="GPU"):
        m.quat_mul(torch.zeros(2, 4), torch.zeros(2, 4))


def test_command_list_bookkeeping_needs_no_gpu():
    """Recording state, the error cases and an empty list's replay: no HIP call is made."""
    from isaaclab_amd import _abi
    from isaaclab_amd.utils.command_list import CommandList

    assert _abi.commands_end() == -1 and b"not recording" in _abi.lib.ilx_last_error()
    assert _abi.commands_begin(None) == -1
    assert _abi.commands_replay(None, None) == -1
    commands = CommandList()
    with pytest.raises(Exception, match="nothing recorded"):
        commands.replay(stream=0)
    with commands.record():
        handle = ctypes.c_void_p()
        assert _abi.commands_begin(ctypes.byref(handle)) == -1 and b"recording already" in _abi.lib.ilx_last_error()
        assert _abi.commands_replay(commands._handle, None) == -1 and b"still being recorded" in _abi.lib.ilx_last_error()
        # argument checks run as usual while recording
        assert _abi.math_op(999, None, None, None, None, None, None, 4, None) == -1
    assert len(commands) == 0
    commands.replay(stream=0)          # nothing to issue
    with pytest.raises(RuntimeError, match="boom"):
        with commands.record():
            raise RuntimeError("boom")
    assert commands._handle is None and _abi.commands_end() == -1   # the failed block left no recording behind
    commands.destroy()
