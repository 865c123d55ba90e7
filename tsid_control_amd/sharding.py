"""Env-batch sharding across the GPUs of one node: contiguous blocks of envs per rank, no data-path
collective (envs never interact: one WalkController + one MjData per env in the reference,
main.py:46-51), plus the one collective the north star names - an all-gather of the per-env
observations (RCCL over xGMI on ROCm via torch.distributed's "nccl" backend; gloo on CPU in tests)."""
import os

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int):
    """Rank r owns envs [lo, hi): contiguous blocks, remainder spread over the first ranks."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_distributed(backend=None):
    """One process per GPU; reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        be = backend or os.environ.get("TSIDB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if be == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(be, rank=rank, world_size=world)
    return rank, local, world


class ObsGather:
    """all-gather of obs[N_local, D] into a preallocated [N_total, D].  Equal shards use the
    single-buffer all_gather_into_tensor; ragged shards are padded to the largest shard, gathered the
    same way, and compacted."""

    def __init__(self, n_local: int, dim: int, world: int, dtype, device, sizes=None):
        self.world = world
        self.sizes = sizes or [n_local] * world
        self.equal = all(s == self.sizes[0] for s in self.sizes)
        self.out = torch.empty(sum(self.sizes), dim, dtype=dtype, device=device)
        if not self.equal:
            self.pad = max(self.sizes)
            self.stage = torch.zeros(self.pad, dim, dtype=dtype, device=device)
            self.wide = torch.empty(world * self.pad, dim, dtype=dtype, device=device)

    def __call__(self, obs: torch.Tensor, async_op=False):
        if self.world == 1:
            self.out.copy_(obs)
            return None
        if self.out.is_cuda and dist.get_backend() == "gloo":
            # rehearsal on a box without RCCL peers: stage through the host
            wide = torch.empty(self.out.shape, dtype=self.out.dtype)
            if self.equal:
                dist.all_gather_into_tensor(wide, obs.cpu().contiguous())
                self.out.copy_(wide)
                return None
            raise NotImplementedError("ragged gloo gather of device tensors")
        if self.equal:
            return dist.all_gather_into_tensor(self.out, obs.contiguous(), async_op=async_op)
        self.stage[:obs.shape[0]].copy_(obs)
        dist.all_gather_into_tensor(self.wide, self.stage)
        off = 0
        for r, s in enumerate(self.sizes):
            self.out[off:off + s].copy_(self.wide[r * self.pad:r * self.pad + s])
            off += s
        return None
