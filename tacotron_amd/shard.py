"""Utterance sharding across the GPUs of one node: utterances are independent,
weights are replicated, so a job splits into contiguous shards - no collective on the data path.
Ranks only meet for the timing barrier / max-over-ranks in bench.py."""
import numpy as np


def shard_bounds(n_items, world):
    """Contiguous, balanced [lo, hi) per rank; the first n_items % world ranks get one more."""
    base, rem = divmod(n_items, world)
    lo, out = 0, []
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def shard_by_length(lengths, world):
    """Length-sort (stable, descending) then deal contiguous shards, so each rank pads to a similar T_in.
    Returns per-rank index arrays into the original order."""
    order = np.argsort(-np.asarray(lengths), kind='stable')
    return [order[lo:hi] for lo, hi in shard_bounds(len(order), world)]


def gather_order(shards):
    """Inverse permutation: concatenating per-rank results in rank order and indexing with this
    restores the original utterance order."""
    cat = np.concatenate(shards) if shards else np.zeros(0, np.int64)
    inv = np.empty_like(cat)
    inv[cat] = np.arange(len(cat))
    return inv


def plan_batches(lengths, world, batch_size):
    """The launch plan of a sharded job: (shards, tasks) with `shards` from shard_by_length and `tasks` the batches
    [(rank, index array of <= batch_size utterances)] in issue order - round-robin over the ranks, so that every
    device receives its k-th batch before any receives its (k+1)-th."""
    shards = shard_by_length(lengths, world)
    per_rank = [[idx[lo:lo + batch_size] for lo in range(0, len(idx), batch_size)] for idx in shards]
    tasks = []
    for k in range(max((len(b) for b in per_rank), default=0)):
        for r, batches in enumerate(per_rank):
            if k < len(batches):
                tasks.append((r, batches[k]))
    return shards, tasks
