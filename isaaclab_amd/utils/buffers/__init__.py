"""Buffers with the reference's module layout (``isaaclab/utils/buffers``)."""

from .circular_buffer import CircularBuffer  # noqa: F401
from .delay_buffer import DelayBuffer  # noqa: F401
from .timestamped_buffer import TimestampedBuffer  # noqa: F401
