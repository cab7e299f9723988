import numpy as np


def rel_l1(a, b):
    """sum|a-b| / sum|b| - the tolerance BASELINE.json's north_star states (1e-3) is on this."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-300))


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


from tacotron_amd.data.synthetic import synth_ids  # noqa: E402,F401  (kept here for the tests' imports)
