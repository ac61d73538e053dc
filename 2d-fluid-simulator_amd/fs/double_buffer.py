"""DoubleBuffer: two device fields and a reference swap (reference: fs/double_buffer.py:4-18).

Which PHYSICAL buffer a kernel writes matters: the step kernels only write fluid / not-wall cells, so
the other cells keep whatever that buffer held one or two swaps ago and later kernels read them
(SURVEY.md hazard H5).  swap() therefore only exchanges the two references - nothing is copied or cleared.
"""
from . import runtime


class DoubleBuffer:
    def __init__(self, resolution, n_channel, device=None):
        dev = device if device is not None else runtime.current_device(resolution)
        self.current = dev.alloc(n_channel)
        self.next = dev.alloc(n_channel)

    def swap(self):
        self.current, self.next = self.next, self.current

    def reset(self):
        self.current.fill(0)
        self.next.fill(0)
