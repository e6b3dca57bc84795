"""Data-parallel plumbing of the training step: one process per GPU, gradients averaged over ranks with
a bucketed all-reduce overlapped with the backward (the reference wraps its model the same way,
detectron2/engine/defaults.py:288-291, minus `find_unused_parameters`: every trainable parameter of the
JTSM composite receives a gradient each step).  Backend "nccl" is RCCL on ROCm; "gloo" is used
by the CPU tests."""
import os

import torch
import torch.distributed as dist


def env_ranks():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend, device=None):
    rank, world, _ = env_ranks()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def wrap_data_parallel(model, device=None, bucket_cap_mb=64):
    """DDP with buffers left alone (FrozenBN statistics never change) and no unused-parameter search."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    ids = [device.index] if (device is not None and device.type == "cuda") else None
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, broadcast_buffers=False,
                                                     find_unused_parameters=False, bucket_cap_mb=bucket_cap_mb,
                                                     gradient_as_bucket_view=True)


def max_over_ranks(seconds, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def fence(device=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize()
