"""Multi-GPU sharding of the OFDM path: independent IQ streams, one process per GPU.

The path has no exchange step: stream ``s`` is modulated, passed through
its channel and demodulated entirely on one GPU.  The only collective is the
end-of-run reduction of the packet counters (sum) and of the elapsed time (max), a
few int64 over RCCL (backend "nccl" on ROCm) or gloo on CPU.
"""
import os

COUNTER_KEYS = ("symbols", "samples", "packets", "crc_ok", "frames", "peaks")


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process if absent)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def streams_of_rank(nstreams, rank, world_size):
    """Block-cyclic assignment of stream ids to ranks: stream s lives on rank s % world_size."""
    return [s for s in range(nstreams) if s % world_size == rank]


def init_process_group(backend=None, force=False):
    """Initialise torch.distributed when launched under torchrun -- also with ONE rank (torchrun --nproc-per-node 1
    sets RANK / MASTER_ADDR: the RCCL communicator and every collective below then really run, on one GPU) -- or when
    ``force`` is set; returns (rank, local_rank, world).  A plain ``python bench.py`` has no RANK: nothing to set up."""
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if (world > 1 or launched or force) and not dist.is_initialized():
        if backend is None:
            import torch
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            import torch
            kw["device_id"] = torch.device("cuda", local_rank)   # binds the communicator (and barrier()) to this rank's GPU
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_counters(stats, device=None):
    """Sum the per-rank packet counters over all ranks (all_reduce); returns a dict."""
    import torch
    import torch.distributed as dist
    vals = torch.tensor([int(stats.get(k, 0)) for k in COUNTER_KEYS], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
    return {k: int(v) for k, v in zip(COUNTER_KEYS, vals.tolist())}


def reduce_max(value, device=None):
    """Max of a float over all ranks (the slowest rank defines the step time)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class _DeviceF32(object):
    """Zero-copy view of a raw device buffer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def allreduce_sensed(msgs, device=None):
    """Cooperative sensing (BASELINE config 5): element-wise MAX of the sensed power
    spectra over all ranks, in place -- every GPU then decides on what ANY antenna heard.
    ``msgs`` is a torch tensor (CPU with gloo, device with RCCL) or an (engine device pointer,
    nmsgs, fft_size) triple from Engine.sense_device_msgs().  A few KB to a few MB per dwell: one
    all_reduce, latency-bound over xGMI."""
    import torch
    import torch.distributed as dist
    if isinstance(msgs, tuple):
        ptr, nm, S = msgs
        if not ptr or not nm:
            return None
        msgs = torch.as_tensor(_DeviceF32(ptr, (nm, S)), device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(msgs, op=dist.ReduceOp.MAX)
    return msgs
