"""Batch-axis sharding across GPUs (BASELINE.json north_star): independent MPC instances, one process per GPU.

Two ways to run N GPUs, both over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests):
  * weak scaling — every rank draws its own shard (shard_seed), no data-path collective; RCCL only for the barrier and the
    max-over-ranks timing around the timed region;
  * the north star's data path (strong scaling of a fixed global batch, BASELINE config 4: 256 instances -> 32 per GPU) —
    rank 0 owns the global problem: `broadcast_image` sends the shared problem image (the POD model description), `BatchShards.scatter`
    the contiguous ceil(B / world) instance blocks of x_init / x / u / node parameters, every rank solves its block, `BatchShards.gather`
    returns the solutions (x, u, performance indices, KKT residuals) to rank 0.  The tensors live on the group's device (HBM under
    nccl), so a shard goes GPU -> xGMI -> GPU -> hsqp_upload_device without host staging."""
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
        # gloo with tensors in HBM (HSQP_DIST_BACKEND=gloo: the data path of N ranks dry-run on ONE GPU, where RCCL refuses two ranks on
        # a device): its scatter / gather / all_gather take CPU tensors only, so those collectives are staged through the host
        self.stage = backend == "gloo" and device is not None and getattr(device, "type", "cpu") != "cpu"
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
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device if self.device is not None and not self.stage else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t.cpu()]

    def gather(self, values):
        """[world][len(values)] array of every rank's floats (all_gather)."""
        if self.dist is None:
            return np.asarray([list(values)], dtype=float)
        import torch
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device if self.device is not None and not self.stage else "cpu")
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.cpu().numpy() for o in out])

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def broadcast_image(group, payload):
    """Rank 0's bytes (the shared problem image: hsqp_model_desc) to every rank — one broadcast of O(100 KiB)."""
    if group.dist is None:
        return bytes(payload)
    import torch
    dev = group.device if group.device is not None and not group.stage else "cpu"
    n = torch.tensor([len(payload) if payload is not None else 0], dtype=torch.int64, device=dev)
    group.dist.broadcast(n, src=0)
    buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev) if env_rank()[0] == 0 else torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    group.dist.broadcast(buf, src=0)
    return bytes(buf.cpu().numpy().tobytes())


class BatchShards:
    """Contiguous blocks of ceil(B / world) instances per rank (shard_range).  dist.scatter / dist.gather need equal-sized pieces, so
    the last blocks are padded with copies of instance 0 (solved and discarded; B = 256 over 1/2/4/8 ranks needs no padding)."""

    def __init__(self, group, global_batch):
        self.group, self.B, self.world = group, int(global_batch), group.world
        self.per = -(-self.B // self.world)
        self.rank = env_rank()[0] if group.dist is not None else 0
        self.lo, self.hi = shard_range(self.B, self.world, self.rank)
        self.count = self.hi - self.lo

    def _dev(self):
        return self.group.device if self.group.device is not None else "cpu"

    def scatter(self, global_tensor, row_shape, dtype=None):
        """Rank 0 passes the global [B, *row_shape] tensor (others None); every rank gets its padded block [per, *row_shape]."""
        import torch
        dtype = dtype or torch.float64
        local = torch.empty((self.per, *row_shape), dtype=dtype, device=self._dev())
        if self.group.dist is None:
            local.copy_(global_tensor[: self.per])
            return local
        pieces = None
        if self.rank == 0:
            pieces = []
            for r in range(self.world):
                lo, hi = shard_range(self.B, self.world, r)
                blk = global_tensor[lo:hi]
                if hi - lo < self.per:
                    blk = torch.cat([blk, global_tensor[:1].expand(self.per - (hi - lo), *row_shape)], dim=0)
                pieces.append(blk.contiguous())
        if self.group.stage:
            host = torch.empty((self.per, *row_shape), dtype=dtype)
            self.group.dist.scatter(host, None if pieces is None else [p.cpu() for p in pieces], src=0)
            local.copy_(host)
        else:
            self.group.dist.scatter(local, pieces, src=0)
        return local

    def gather(self, local):
        """Every rank passes its padded block; rank 0 gets the global [B, ...] tensor (others None)."""
        import torch
        if self.group.dist is None:
            return local[: self.B].clone()
        if self.group.stage:
            host = local.contiguous().cpu()
            outh = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            self.group.dist.gather(host, outh, dst=0)
            out = None if outh is None else [o.to(local.device) for o in outh]
        else:
            out = [torch.empty_like(local) for _ in range(self.world)] if self.rank == 0 else None
            self.group.dist.gather(local.contiguous(), out, dst=0)
        if self.rank != 0:
            return None
        return torch.cat([out[r][: shard_range(self.B, self.world, r)[1] - shard_range(self.B, self.world, r)[0]] for r in range(self.world)], dim=0)


def aggregate_throughput(instances_per_rank, steps, elapsed_max):
    """Whole-job SQP iterations per second: all ranks' instance-iterations over the slowest rank's time."""
    return float(np.sum(instances_per_rank)) * steps / elapsed_max
