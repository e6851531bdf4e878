"""Point sharding over ranks: points are independent, so a shard is a
contiguous block of global point ids and nothing crosses ranks on the data path.
The only collectives are the ones the benchmark contract needs: a barrier and a
MAX over ranks of the elapsed time."""
from __future__ import annotations


def weak_shard(points_per_rank: int, rank: int) -> tuple[int, int]:
    """(global offset, count) of this rank when every rank owns `points_per_rank` points."""
    return rank * points_per_rank, points_per_rank


def strong_shard(total_points: int, world: int, rank: int) -> tuple[int, int]:
    """(global offset, count): contiguous block partition of `total_points`, remainders to the
    first ranks."""
    base, rem = divmod(total_points, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """MAX all-reduce of a scalar (the bench's elapsed time)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
