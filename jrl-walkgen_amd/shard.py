"""Multi-GPU plumbing of the MPC-tick path: one process per GPU, gaits sharded by contiguous index range,
ONE collective (broadcast of the constant model block from rank 0), no data-path exchange.

The reference has no distribution at all (single-threaded library); independent gait instances simply do not
interact, so the only thing ranks must agree on is the robot/algorithm constant block (wg_model_t), which
rank 0 owns and broadcasts once at start-up -- over RCCL/xGMI on GPUs ("nccl" backend), over gloo in the CPU
tests.  Everything here is backend-agnostic torch.distributed.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched plainly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend):
    rank, local_rank, world = env_world()
    if backend == "nccl" and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)          # before the group exists: RCCL binds its communicator to the current device
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def broadcast_struct(obj, device, src=0):
    """Broadcast a ctypes.Structure's bytes from `src` to every rank, in place."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return obj
    n = C.sizeof(obj)
    buf = torch.frombuffer(bytearray(bytes(memoryview(obj).cast("B"))), dtype=torch.uint8).to(device)
    dist.broadcast(buf, src=src)
    raw = bytes(buf.cpu().numpy().tobytes())
    C.memmove(C.byref(obj), raw, n)
    return obj


def shard_range(total, rank, world):
    """Contiguous [lo, hi) slice of `total` gaits owned by `rank` (remainder spread over the first ranks): the C ABI's
    wg_shard_range (host arithmetic, include/wg_mpc.h) -- one rule for the Python harness and for host/fleet_bench.cpp."""
    from . import wgmpc
    lo, hi = C.c_longlong(), C.c_longlong()
    rc = wgmpc.lib().wg_shard_range(C.c_longlong(total), int(rank), int(world), C.byref(lo), C.byref(hi))
    if rc != 0:
        raise ValueError(f"wg_shard_range({total}, {rank}, {world}) -> {rc}")
    return lo.value, hi.value


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
