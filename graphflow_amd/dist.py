"""Data-parallel helpers (one process per GPU, torch.distributed: backend "nccl" is RCCL on ROCm, "gloo" on CPU).

The SMP path has exactly one exchange per step: the sum of the flat parameter-gradient buffer over ranks, the
analogue of the reference's serial add_gradient loop in Threaded_BatchLearn (GraphFlow/SMP_omega.h:730-740, 784-786).
Everything else (the molecule batch) is sharded with no communication."""
import os


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def shard(n_total, rank, world):
    """Contiguous slice [lo, hi) of n_total items for this rank (sizes differ by at most one)."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init(backend=None, device=None):
    import torch.distributed as dist
    world, rank, local = env_world()
    if world == 1 and not os.environ.get("GF_FORCE_DIST"):  # GF_FORCE_DIST=1: exercise the collective path with one rank
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        import torch
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def allreduce_sum_(flat, dist):
    """In-place sum of one flat gradient buffer over all ranks (a single collective per step)."""
    if dist is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def max_over_ranks(value, dist, device="cpu"):
    import torch
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
