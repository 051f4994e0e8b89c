"""Sample-count sharding of one render over the ranks of a torch.distributed job (one process per GPU).

Every rank renders its own contiguous block of sample indices of every pixel; random streams are keyed by
(pixel, sample index), so the sum of the rank films is the same image whatever the number of ranks.  The only
exchange step is one sum all-reduce of the film (RCCL over xGMI on GPUs, gloo in the CPU tests)."""


def sample_range(rank, world, ns_total):
    """[first, first + count) of the ns_total samples that `rank` renders; the blocks tile 0..ns_total."""
    base, rem = divmod(ns_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def reduce_film(film):
    """sums the per-rank films in place on every rank; a no-op outside a process group"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(film, op=dist.ReduceOp.SUM)
    return film
