"""Batch-axis sharding across GPUs: independent MPC instances, one process per GPU, no data-path
collective.  torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests) is used only around the timed
region: barrier, max-over-ranks of the elapsed time / residuals, gather of per-rank summaries."""
import os

import numpy as np


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_seed(base_seed, rank):
    """Each rank draws its own instances: seed = base + rank (rank 0 of any world size equals the single-GPU run)."""
    return base_seed + rank


def shard_range(global_batch, world, rank):
    """Contiguous block of ceil(B / world) instances per rank (strong-scaling split of a fixed global batch)."""
    per = -(-global_batch // world)
    lo = min(rank * per, global_batch)
    return lo, min(lo + per, global_batch)


class Group:
    """Thin wrapper so that bench.py and the gloo tests share the reduction code."""

    def __init__(self, world, backend=None, device=None):
        self.world = world
        self.device = device
        self.dist = None
        if world > 1:
            import torch.distributed as dist
            self.dist = dist
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    kw["device_id"] = device
                dist.init_process_group(backend=backend or "gloo", **kw)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, values):
        """Element-wise max over ranks of a list of floats."""
        if self.dist is None:
            return [float(v) for v in values]
        import torch
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t.cpu()]

    def gather(self, values):
        """[world][len(values)] array of every rank's floats (all_gather)."""
        if self.dist is None:
            return np.asarray([list(values)], dtype=float)
        import torch
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.cpu().numpy() for o in out])

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def aggregate_throughput(instances_per_rank, steps, elapsed_max):
    """Whole-job SQP iterations per second: all ranks' instance-iterations over the slowest rank's time."""
    return float(np.sum(instances_per_rank)) * steps / elapsed_max
