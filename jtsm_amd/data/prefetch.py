"""DevicePrefetcher — the host -> HBM leg of the input contract (the reference leaves it to
`x["image"].to(self.device)` inside the model, mcnn.py:308, one blocking copy per tensor).

Every tensor of a batch (list[dict], values: tensors / Instances / plain Python) is packed by a worker thread into ONE
pinned staging arena and crosses PCIe as ONE asynchronous copy on a side HIP stream; the device tensors handed out
are views into a device arena.  `depth` arenas rotate, so batch i+1 is packed and in flight while the model computes
on batch i; an arena is overwritten only after its copy has left host memory (host side) and after the compute
stream has passed the point where the consumer asked for the next batch (device side).

Byte diet on the link: `oh_labels` (0/1) and `sem_seg` (<= 255) travel as uint8 and are widened on the device.
"""
import queue
import threading

import numpy as np
import torch

from ..structures import Boxes, Instances

_ALIGN = 256
_NARROW = {"oh_labels": torch.uint8, "sem_seg": torch.uint8}


def _walk(obj, path, out):
    """Collect (path, tensor) for every CPU tensor reachable in a batch element."""
    if isinstance(obj, torch.Tensor):
        out.append((path, obj))
    elif isinstance(obj, Boxes):
        out.append((path + ("tensor",), obj.tensor))
    elif isinstance(obj, Instances):
        for k, v in obj.get_fields().items():
            _walk(v, path + (k,), out)
    elif isinstance(obj, dict):
        for k, v in obj.items():
            _walk(v, path + (k,), out)


def _rebuild(obj, path, table):
    if isinstance(obj, torch.Tensor):
        return table[path]
    if isinstance(obj, Boxes):
        return Boxes(table[path + ("tensor",)])
    if isinstance(obj, Instances):
        return Instances(obj.image_size, **{k: _rebuild(v, path + (k,), table) for k, v in obj.get_fields().items()})
    if isinstance(obj, dict):
        return {k: _rebuild(v, path + (k,), table) for k, v in obj.items()}
    return obj


class _Slot:
    def __init__(self):
        self.pinned = self.device = self.host = None
        self.free = None            # event on the compute stream: the consumer is done with this slot's tensors
        self.copied = None          # event on the copy stream: the arena has left host memory
        self.launched = threading.Event()   # main thread has queued this slot's copy (worker may look at `copied`)
        self.launched.set()

    def reserve(self, nbytes, device):
        if self.pinned is None or self.pinned.numel() < nbytes:
            cap = int(nbytes * 1.25) + _ALIGN
            self.pinned = torch.empty(cap, dtype=torch.uint8).pin_memory()
            self.host = self.pinned.numpy()          # same memory, for GIL-free single-thread copies
            self.device = torch.empty(cap, dtype=torch.uint8, device=device)


_NP = {torch.uint8: np.uint8, torch.int16: np.int16, torch.int32: np.int32, torch.int64: np.int64,
       torch.float32: np.float32, torch.float64: np.float64, torch.bool: np.bool_, torch.float16: np.float16}


class DevicePrefetcher:
    def __init__(self, batches, device, depth=2):
        if torch.device(device).type != "cuda":
            raise RuntimeError("DevicePrefetcher stages batches into HBM: it needs the HIP device")
        self.batches, self.device, self.depth = batches, torch.device(device), max(2, int(depth))
        self.stream = torch.cuda.Stream(self.device)
        self.slots = [_Slot() for _ in range(self.depth)]
        self.bytes_last = 0

    # ---- worker thread: host tensors -> the slot's pinned arena (numpy copies: no GIL, no intra-op thread pool)
    def _pack(self, batch, slot):
        found = []
        for i, elem in enumerate(batch):
            _walk(elem, (i,), found)
        plan, off = [], 0
        for path, t in found:
            wire = _NARROW.get(path[-1], t.dtype) if t.dtype in (torch.int32, torch.int64) else t.dtype
            nbytes = t.numel() * torch.empty((), dtype=wire).element_size()
            plan.append((path, t.dtype, tuple(t.shape), wire, off, nbytes))
            off = (off + nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
        slot.launched.wait()
        if slot.copied is not None:
            slot.copied.synchronize()                 # the previous content of this arena has crossed the link
        slot.reserve(off, self.device)
        for (path, dtype, shape, wire, o, nbytes), (_, t) in zip(plan, found):
            if not nbytes:
                continue
            src = t.detach().contiguous().numpy()
            if wire != dtype and (src.min() < 0 or src.max() > 255):
                raise ValueError("%s holds values outside 0..255: cannot travel as uint8" % (path,))
            dst = slot.host[o:o + nbytes].view(_NP[wire]).reshape(shape)
            np.copyto(dst, src, casting="unsafe")
        slot.launched.clear()
        return plan, off

    def _worker(self, q):
        try:
            for k, batch in enumerate(self.batches):
                slot = self.slots[k % self.depth]
                q.put((batch, slot) + self._pack(batch, slot))
            q.put(None)
        except BaseException as e:   # surfaces in the consumer
            q.put(e)

    # ---- consumer thread: one async copy per batch on the side stream, device views, widening
    def _launch(self, batch, slot, plan, off):
        self.bytes_last = off
        # the stream the training step runs on — looked up BEFORE entering the side stream's context (inside it,
        # current_stream() is the side stream itself and record_stream() on it would be a no-op)
        consumer = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.stream):
            if slot.free is not None:
                self.stream.wait_event(slot.free)                                 # do not overwrite tensors still in use
            slot.device[:off].copy_(slot.pinned[:off], non_blocking=True)         # the batch's one H2D copy
            slot.copied = torch.cuda.Event()
            slot.copied.record(self.stream)
            slot.launched.set()
            table = {}
            for path, dtype, shape, wire, o, nbytes in plan:
                d = slot.device[o:o + nbytes].view(wire).view(shape)
                if wire != dtype:
                    d = d.to(dtype)                                                # widen on the device
                    d.record_stream(consumer)   # allocated on the side stream, read on the compute stream: the caching
                    # allocator must not hand this block to a later widening while a step still reads it
                table[path] = d
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return [_rebuild(elem, (i,), table) for i, elem in enumerate(batch)], ready

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth - 1)
        threading.Thread(target=self._worker, args=(q,), daemon=True).start()

        def take():
            item = q.get()
            if isinstance(item, BaseException):
                raise item
            return None if item is None else (item[1],) + self._launch(*item)

        pending = take()
        while pending is not None:
            cur, out, ready = pending
            pending = take()                     # the next batch's copy is queued before this step's kernels
            torch.cuda.current_stream(self.device).wait_event(ready)
            yield out
            cur.free = torch.cuda.Event()
            cur.free.record(torch.cuda.current_stream(self.device))              # consumer came back: slot reusable
