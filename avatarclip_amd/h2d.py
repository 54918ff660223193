"""Host -> device uploads of the small per-iteration values (camera pose, light direction, ...) that do not stall the launch
queue.  A `tensor.to(device)` from pageable memory is a blocking copy ordered behind everything already enqueued on the
stream: one of them after the render kernels makes the host wait for the render and then launch the ~1000 small CLIP kernels
with the GPU idling in between.  Here the values go through a ring of pinned staging rows (non-blocking copies; a row is
reused only after the event recorded behind its copy has completed) and constants are uploaded once per device."""
import threading

import numpy as np
import torch

_consts = {}
_rings = {}
_lock = threading.Lock()     # the view of the next iteration is prepared on a helper thread (Runner.prefetch_view)


def const(values, device, dtype=torch.float32):
    """a constant tensor (cached per device): CLIP mean / std, the camera up vector, ..."""
    a = np.asarray(values, np.float64)
    key = (a.tobytes(), a.shape, str(device), dtype)
    t = _consts.get(key)
    if t is None:
        t = torch.tensor(a.tolist(), dtype=dtype, device=device)
        _consts[key] = t
    return t


class _Ring:
    SLOTS, WIDTH = 64, 64   # rows of 64 doubles (512 B): poses, eyes, directions

    def __init__(self, device):
        self.device = device
        self.buf = torch.empty(self.SLOTS, self.WIDTH, dtype=torch.float64).pin_memory()
        self.buf32 = torch.empty(self.SLOTS, self.WIDTH, dtype=torch.float32).pin_memory()   # float32 values travel as float32: no conversion launch
        self.events = [None] * self.SLOTS
        self.slot_locks = [threading.Lock() for _ in range(self.SLOTS)]
        self.i = 0

    def put(self, a, dtype):
        n = a.size
        with _lock:
            slot = self.i % self.SLOTS
            self.i += 1
        # the whole wait -> fill -> copy -> record sequence of a slot is one critical section: with the helper-thread prefetch two
        # threads can wrap the ring onto the same slot (SLOTS uploads by one during the other's fill), and the second must neither
        # overwrite the pinned row before its copy has been issued nor miss the event that guards it
        with self.slot_locks[slot]:
            ev = self.events[slot]
            if ev is not None:
                ev.synchronize()
            if dtype == torch.float32:
                row = self.buf32[slot, :n]
                row.copy_(torch.from_numpy(a.reshape(-1).astype(np.float32)))
            else:
                row = self.buf[slot, :n]
                row.copy_(torch.from_numpy(a.reshape(-1)))
            with torch.cuda.device(self.device):     # the copy AND its guard event go to this device's current stream
                out = row.to(self.device, non_blocking=True).to(dtype).reshape(a.shape)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
            self.events[slot] = ev
        return out


def upload(values, device, dtype=torch.float32):
    """values (array-like of <= 64 numbers) -> device tensor of `dtype` without blocking the host on the stream."""
    a = np.ascontiguousarray(np.asarray(values, np.float64))
    device = torch.device(device)
    if device.type != "cuda" or a.size > _Ring.WIDTH:
        return torch.from_numpy(a).to(device=device, dtype=dtype)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    with _lock:
        ring = _rings.get(key)
        if ring is None:
            ring = _rings[key] = _Ring(device)
    return ring.put(a, dtype)
