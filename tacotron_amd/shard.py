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
