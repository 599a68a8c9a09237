"""Formula-generated weights and inputs for the golden fixtures.

Nothing large is committed: every weight tensor is a closed-form function of its
state_dict key and shape, reproducible here (numpy) and anywhere else.  Shared by
tests/golden/make_golden.py (which loads them into the *reference* modules) and by
the tests (which load the same tensors into the oracle / the HIP modules).
"""
import zlib

import numpy as np


def _phase(name):
    h = zlib.crc32(name.encode())
    a = 0.37 + (h % 1009) / 1009.0 * 0.9
    b = ((h >> 11) % 997) / 997.0 * 6.283
    return a, b


def formula_tensor(name, shape):
    """float32 tensor for state_dict key `name`."""
    n = int(np.prod(shape)) if len(shape) else 1
    a, b = _phase(name)
    base = np.sin(a * np.arange(n, dtype=np.float64) + b)
    leaf = name.split(".")[-1]
    parent = name.split(".")[-2] if "." in name else ""
    if parent.startswith("norm"):
        t = 1.0 + 0.1 * base if leaf == "weight" else 0.05 * base
    elif leaf in ("bias", "in_proj_bias"):
        t = 0.05 * base
    else:
        fan_in = shape[-1]
        t = base * (1.2 / np.sqrt(fan_in))
    return t.reshape(shape).astype(np.float32)


def formula_state_dict(template_sd, skip=("position_encoding.pe",)):
    """template_sd: mapping key -> object with .shape.  Returns key -> float32 ndarray."""
    out = {}
    for k, v in template_sd.items():
        if k in skip:
            continue
        out[k] = formula_tensor(k, tuple(v.shape))
    return out


def formula_input(tag, S, B, D, pad_from=None):
    """uniform-[0,1)-like input (S, B, D), zero from `pad_from` on for odd b (ragged padding)."""
    a, b = _phase("input." + tag)
    i = np.arange(S * B * D, dtype=np.float64)
    x = 0.5 + 0.5 * np.sin(a * 1.7 * i + b)
    x = x.reshape(S, B, D)
    if pad_from is not None:
        x[pad_from:, 1::2, :] = 0.0
    return x.astype(np.float32)


def sample_indices(n, k=1024):
    """deterministic strided sample of a flat tensor"""
    if n <= k:
        return np.arange(n)
    return (np.arange(k, dtype=np.int64) * (n // k)) + (np.arange(k) % 7)


def summarize(t, full_max=4096):
    """tensor -> dict of fixture arrays: full if small, else sample + sums."""
    t = np.asarray(t, dtype=np.float32)
    flat = t.reshape(-1)
    if flat.size <= full_max:
        return {"full": t}
    idx = np.minimum(sample_indices(flat.size), flat.size - 1)
    return {"sample": flat[idx], "sum": np.float64(flat.astype(np.float64).sum()),
            "l2": np.float64(np.sqrt((flat.astype(np.float64) ** 2).sum()))}
