"""Host -> HBM prefetch: the part of upstream's DataLoader loop that feeds the network,
`for sample in loader: bev = sample.to(device); model(bev)`, with the copy taken off the step's critical path.

The C ABI takes device pointers (include/v2x_amd.h); a caller that holds host buffers pays the host link.  One bench
step's sweeps are 671 MB = 11.7 ms at the measured 57 GB/s -- 46 % of the 25 ms step if it is issued in line.  Here a
background thread pulls batches from any iterable, stages them in PINNED host memory and issues the copies on a
dedicated HIP stream `depth` batches ahead; the consumer only makes its compute stream wait on the copy's event, so the
link runs under the previous step's kernels.

    for batch in DevicePrefetcher(loader, device, depth=2):      # batch: same structure, tensors on `device`
        model(batch["points"], ...)

Structure: dict / list / tuple of numpy arrays or torch tensors (anything else is passed through untouched).  The device
tensors of a batch are handed back to the caching allocator only after the consumer asked for the NEXT batch
(record_stream on the compute stream), so a step may keep using its input while the following copy is in flight.
"""
import queue
import threading

import numpy as np
import torch


def _map(obj, fn):
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(v, fn) for v in obj)
    return fn(obj)


class DevicePrefetcher:
    _END = object()

    def __init__(self, iterable, device, depth=2):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher feeds the MI355X: device must be a cuda device")
        self.iterable, self.depth = iterable, depth
        self.stream = torch.cuda.Stream(device=self.device)
        self.bytes_copied = 0

    def _stage(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x)
        if not torch.is_tensor(x):
            return x
        if x.device.type != "cpu":
            return x
        if not x.is_pinned():
            x = x.contiguous().pin_memory()
        self.bytes_copied += x.numel() * x.element_size()
        return x.to(self.device, non_blocking=True)

    def _worker(self, q):
        try:
            torch.cuda.set_device(self.device)
            for batch in self.iterable:
                with torch.cuda.stream(self.stream):
                    dev = _map(batch, self._stage)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                q.put((dev, ev))
            q.put((self._END, None))
        except BaseException as e:   # surfaced in the consumer thread
            q.put((e, None))

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth)
        t = threading.Thread(target=self._worker, args=(q,), daemon=True)
        t.start()
        while True:
            dev, ev = q.get()
            if dev is self._END:
                break
            if isinstance(dev, BaseException):
                raise dev
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            _map(dev, lambda x: x.record_stream(cur) if torch.is_tensor(x) and x.device.type == "cuda" else None)
            yield dev
        t.join()
