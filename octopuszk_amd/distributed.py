"""Multi-GPU variable-base MSM: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI), index-range sharding, one tiny exchange.

Mirrors VariableBaseMSM.distributedMSM (VariableBaseMSM.java:775-786): Spark's
`mapPartitions(serialMSMPartition) -> reduce(GroupT::add)` becomes: every rank runs the
single-GPU pipeline on its contiguous slice of (scalar, base) pairs, the ranks all-gather
their 192-byte (G1) / 384-byte (G2) affine partials, and every rank adds the world_size
partials with the HIP point-sum kernel.  RCCL has no elliptic-curve reduction operator, so
the "all-reduce" is all-gather + local sum; the message is < 4 KiB, i.e. latency-bound, and
there is no other data-path collective.
"""
import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of n pairs owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_partials(partial: torch.Tensor, group=None) -> torch.Tensor:
    """partial: uint8 [192|384] on this rank's device -> uint8 [world * len] (rank order)."""
    world = dist.get_world_size(group)
    out = torch.empty(world * partial.numel(), dtype=torch.uint8, device=partial.device)
    dist.all_gather_into_tensor(out, partial.contiguous(), group=group)
    return out


def distributed_var_msm(local_partial_fn, sum_fn, type_: int = 1, group=None, always_collective=False) -> torch.Tensor:
    """local_partial_fn() -> this rank's partial (wire-out bytes tensor);
    sum_fn(gathered, world, type_) -> the normalised sum.  The defaults used on GPUs are
    device.VarMsmWorkspace.run and device.points_sum; tests inject CPU stand-ins over gloo.
    always_collective: run the all-gather and the point sum even in a world of one (how the RCCL code path is
    exercised on a one-GPU box: the sum of one affine point is that point)."""
    partial = local_partial_fn()
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_collective):
        return partial
    gathered = all_gather_partials(partial, group)
    return sum_fn(gathered, dist.get_world_size(group), type_)


def gpu_var_msm(ws, d_bases, d_scalars, group=None) -> torch.Tensor:
    """The production composition on one rank: HIP MSM on the local slice + RCCL all-gather +
    HIP point sum.  `ws` is a device.VarMsmWorkspace for the local slice size."""
    from . import device as dev
    return distributed_var_msm(lambda: ws.run(d_bases, d_scalars), dev.points_sum, ws.type, group)


def distributed_var_double_msm(local_partial_fn, sum_fn, group=None, always_collective=False) -> torch.Tensor:
    """VariableBaseMSM.distributedDoubleMSM (VariableBaseMSM.java:805-818: per-partition doubleMSM, then reduce(add) of
    the G1 and of the G2 component): local_partial_fn() -> this rank's 576-byte partial, G1 (192) || G2 (384) as
    ozk_var_double_msm_host lays them out; ONE all-gather of the 576-byte records, then the two point sums."""
    partial = local_partial_fn()
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always_collective):
        return partial
    world = dist.get_world_size(group)
    rec = all_gather_partials(partial, group).view(world, 576)
    g1 = sum_fn(rec[:, :192].contiguous().view(-1), world, 1)
    g2 = sum_fn(rec[:, 192:].contiguous().view(-1), world, 2)
    return torch.cat([g1.view(-1), g2.view(-1)])


def gpu_var_double_msm(ws_g1, ws_g2, d_bases_g1, d_bases_g2, d_scalars, group=None) -> torch.Tensor:
    """The production composition of the double MSM on one rank: the two HIP MSMs over the local slice (the same
    scalars), one RCCL all-gather of the 576-byte partial, two HIP point sums."""
    from . import device as dev

    def local():
        return torch.cat([ws_g1.run(d_bases_g1, d_scalars).view(-1), ws_g2.run(d_bases_g2, d_scalars).view(-1)])

    return distributed_var_double_msm(local, dev.points_sum, group)
