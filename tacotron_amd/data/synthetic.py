"""Synthetic character-id batches (BASELINE.json configs: "synthetic 150-char inputs")."""
import numpy as np


def synth_ids(rng, B, T_in, ragged=True, min_len=None):
    """Synthetic character ids over the ids basic_cleaners can produce:
    {28-53 a-z, 54-64 punctuation/space, 75-84 Icelandic lower, 85-94 digits}, eos=1 last,
    right-padded with id 0 (reference data/batch.py:121-134)."""
    reach = np.array(list(range(28, 65)) + list(range(75, 95)))
    min_len = min_len or max(2, T_in // 2)
    lens = rng.integers(min_len, T_in + 1, size=B) if ragged else np.full(B, T_in)
    lens[0] = T_in
    ids = np.zeros((B, T_in), np.int32)
    for b in range(B):
        n = int(lens[b])
        ids[b, :n - 1] = rng.choice(reach, size=n - 1)
        ids[b, n - 1] = 1
    return ids, lens.astype(np.int32)
