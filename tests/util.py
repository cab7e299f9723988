import numpy as np


def rel_l1(a, b):
    """sum|a-b| / sum|b| - the tolerance BASELINE.json's north_star states (1e-3) is on this."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-300))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def synth_ids(rng, B, T_in, ragged=True, min_len=None):
    """Synthetic character ids over the ids basic_cleaners can produce:
    {28-53 a-z, 54-64 punctuation/space, 75-84 Icelandic lower, 85-94 digits}, eos=1 last,
    right-padded with id 0 (batch.py:121-134)."""
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
