"""CTU-row bands across devices: the host-side partition and the hand-off pipeline.

A picture is cut into bands of whole CTU rows, one band per rank (one rank = one GPU).  Under WaveFrontSynchro=1 the only data
that crosses a band boundary downwards is what `hm355_export_boundary` packs from the band's last CTU row (bottom sample lines,
decision arrays, CABAC states: the WPP hand-off points of TEncSlice.cpp:740-755,855-858); nothing flows upwards during the search.
So the ranks form a pipeline over groups of pictures: rank r receives the boundary rows of a group from rank r-1, searches its
band of those pictures, and sends its own last rows to rank r+1 while it goes on with the next group.

The transport is injected (`send(array, dst)` returning a handle with .wait(), `recv(nbytes, src)`), so that the same schedule
runs over RCCL between GPUs (bench.py --shard rows), over gloo in the CPU tests, and through a plain dict when one process
plays all ranks (tests/test_gpu_parity.py).  The engine is anything with run_rows / export_boundary / import_boundary
(hm355.Encoder; a recording fake in tests/test_multi_gpu_cpu.py).
"""
import numpy as np


def band_rows(h_ctu, world, rank):
    """CTU rows [first, last] of `rank`: as even as possible, the first h_ctu % world ranks one row more; (0, -1) = no rows
    (more ranks than CTU rows)."""
    base, extra = divmod(h_ctu, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, first + count - 1


def picture_groups(num_pictures, group):
    """[(first_slot, n), ...] in pipeline order"""
    return [(s, min(group, num_pictures - s)) for s in range(0, num_pictures, group)]


def run_banded(engine, num_pictures, group, h_ctu, rank, world, send, recv, qp, buffer=None):
    """Search this rank's band of `num_pictures` pictures resident in the engine's slots, `group` pictures per launch.
    Returns the kernel milliseconds this rank spent.  Ranks without rows only forward nothing and return 0.
    `buffer(nbytes)` (optional) allocates a device-resident byte buffer (an object with data_ptr()): the boundary rows then go from the
    slots into it and from it into the slots on the device (hm355_export_boundary / hm355_import_boundary accept device pointers), and
    the transport is handed the device buffer -- no host hop.  Without it the rows travel as numpy arrays."""
    first, last = band_rows(h_ctu, world, rank)
    if last < first:
        return 0.0
    # the next rank that owns rows (ranks beyond h_ctu own none)
    has_below = rank + 1 < world and band_rows(h_ctu, world, rank + 1)[1] >= band_rows(h_ctu, world, rank + 1)[0]
    has_above = first > 0
    nbytes = engine.boundary_bytes()
    pending, kernel_ms = [], 0.0
    for (slot0, n) in picture_groups(num_pictures, group):
        if has_above:
            data = recv(n * nbytes, rank - 1)
            for i in range(n):
                if hasattr(data, "data_ptr"):
                    engine.import_boundary_ptr(slot0 + i, first - 1, data.data_ptr() + i * nbytes)
                else:
                    engine.import_boundary(slot0 + i, first - 1, data[i * nbytes:(i + 1) * nbytes])
        ms, _ = engine.run_rows(slot0, n, qp, first, last)
        kernel_ms += ms
        if has_below:
            if buffer is not None:
                out = buffer(n * nbytes)
                for i in range(n):
                    engine.export_boundary_ptr(slot0 + i, last, out.data_ptr() + i * nbytes)
            else:
                out = np.concatenate([engine.export_boundary(slot0 + i, last) for i in range(n)])
            pending.append(send(out, rank + 1))
    for h in pending:
        if h is not None:
            h.wait()
    return kernel_ms


class _Sent:
    """an asynchronous send in flight: owns the tensor until the send has completed"""

    def __init__(self, work, tensor):
        self.work, self.tensor = work, tensor

    def wait(self):
        self.work.wait()
        self.tensor = None


class TorchTransport:
    """send / recv of byte buffers over torch.distributed point-to-point.  Backend "nccl" = RCCL between GPUs: the buffers are device
    tensors from `buffer()`, filled and consumed on the device (no host hop); "gloo" (device None): numpy arrays through host tensors."""

    def __init__(self, dist, torch, device=None):
        self.dist, self.torch, self.device = dist, torch, device

    def buffer(self, nbytes):
        return self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device)

    def send(self, data, dst):
        t = data if hasattr(data, "data_ptr") else self.torch.from_numpy(np.ascontiguousarray(data, np.uint8))
        if self.device is not None and t.device != self.device:
            t = t.to(self.device)
        return _Sent(self.dist.isend(t, dst), t)      # the tensor lives until wait()

    def recv(self, nbytes, src):
        t = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.device if self.device is not None else "cpu")
        self.dist.recv(t, src)
        return t if self.device is not None else t.numpy()
