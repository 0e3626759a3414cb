"""Command lists (``ilx_commands_*``): the launches of a step recorded once and re-issued by a C loop.

No counterpart in the reference.  A HIP graph replay costs the host 14 us however few nodes the graph has; a launch
issued from C costs 2.9 us, so a step of two to four short launches (the Cartpole, Reach and in-hand configurations)
is issued faster as a list.  The rules are a captured graph's: nothing runs while recording, the list holds the
device pointers it was recorded with, and what changes from step to step lives in device memory.

    commands = CommandList()
    with commands.record():
        env.step(actions)          # kept, not issued
    for _ in range(steps):
        commands.replay()          # on the current stream
"""
from __future__ import annotations

import contextlib
import ctypes

from .. import _abi


class CommandList:
    def __init__(self):
        self._handle = None

    @contextlib.contextmanager
    def record(self):
        """Record the library's launches made by this thread inside the block (replacing what the list held).  Work that
        is not this library's -- torch operations, copies -- is not recorded: it runs as usual, once."""
        self.destroy()
        handle = ctypes.c_void_p()
        _abi.check(_abi.commands_begin(ctypes.byref(handle)), "ilx_commands_begin")
        self._handle = handle
        try:
            yield self
        except BaseException:
            _abi.commands_end()
            self.destroy()
            raise
        _abi.check(_abi.commands_end(), "ilx_commands_end")

    def __len__(self) -> int:
        if self._handle is None:
            return 0
        n = ctypes.c_int()
        _abi.check(_abi.commands_count(self._handle, ctypes.byref(n)), "ilx_commands_count")
        return n.value

    def replay(self, stream: int | None = None):
        if self._handle is None:
            raise _abi.IlxError("CommandList.replay: nothing recorded")
        _abi.check(_abi.commands_replay(self._handle, _abi.stream() if stream is None else stream), "ilx_commands_replay")

    def destroy(self):
        if self._handle is not None:
            _abi.commands_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
