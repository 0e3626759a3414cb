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


class SegmentedCommandList:
    """A step with a simulator in it: one command list per stretch of launches between two physics steps.

    The step drivers call ``env.on_physics(substep)`` where the simulator's step belongs.  ``record`` cuts the recording
    there, ``replay(physics)`` issues the stretches in order and calls ``physics(substep)`` between them -- the
    simulator's own stepping call, which is not this library's and is therefore not recorded.  The simulator must
    write its results where the recorded launches read them (the lists hold those addresses).

        step = SegmentedCommandList()
        step.record(env, lambda: env.step(actions))
        step.replay(lambda k: sim.step())
    """

    def __init__(self):
        self.segments: list[CommandList] = []

    def record(self, env, one_step):
        for seg in self.segments:
            seg.destroy()
        self.segments = []
        stack = contextlib.ExitStack()

        def open_segment():
            commands = CommandList()
            stack.enter_context(commands.record())
            self.segments.append(commands)

        def cut(substep):
            stack.close()          # ends the stretch before the physics step ...
            open_segment()         # ... and starts the one behind it

        before = env.on_physics
        env.on_physics = cut
        try:
            open_segment()
            one_step()
        finally:
            env.on_physics = before
            stack.close()
        return self

    def __len__(self) -> int:
        return sum(len(seg) for seg in self.segments)

    def replay(self, physics=None, stream: int | None = None):
        for k, seg in enumerate(self.segments):
            if k > 0 and physics is not None:
                physics(k - 1)
            seg.replay(stream)
