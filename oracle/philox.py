"""Philox4x32-10 dropout-mask generator (TEST INFRASTRUCTURE — oracle side).

The reference draws dropout masks from torch's CPU generator (`nn.Dropout`,
/root/reference/model.py:1181,1197,1218 ...); a GPU can never reproduce that
stream, so the build defines its own counter-based stream and the oracle
restates it here in numpy so that train-mode (dropout on) parity is bit-exact
in the *mask* and 1e-4 in the values.

Contract (shared with gan_ffn_amd/csrc/philox.h):

  For a 2-D tensor [R x C] at dropout site `site` with rng state (seed, offset):
      ctr  = (c0, c1, c2, c3) = ((r >> 2) * C + c, site, offset_lo, offset_hi)
      key  = (seed_lo, seed_hi)
      word = philox4x32_10(ctr, key)[r & 3]
      keep = word >= floor(p * 2**32)
      y    = keep ? x / (1 - p) : 0
  i.e. one Philox call covers the 4 consecutive ROWS r..r+3 of one column, which
  is exactly what one lane of an MFMA 32x32 / 16x16 accumulator holds.
  Attention probabilities use R = (b*H + h) * 112 + i, C = 128, c = j.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All args are array-likes of uint32 values
    (broadcastable); returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64)
    c3 = np.asarray(c3, dtype=np.uint64)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def threshold(p):
    """u32 drop threshold: an element is DROPPED iff word < threshold."""
    return int(np.floor(float(p) * 4294967296.0)) & 0xFFFFFFFF if p > 0 else 0


def keep_mask(R, C, p, site, seed, offset):
    """bool [R, C] keep-mask for the contract above."""
    if p <= 0.0:
        return np.ones((R, C), dtype=bool)
    G = (R + 3) // 4
    c0 = np.arange(G * C, dtype=np.uint64)
    w = philox4x32_10(c0, np.uint64(site & 0xFFFFFFFF),
                      np.uint64(offset & 0xFFFFFFFF), np.uint64((offset >> 32) & 0xFFFFFFFF),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(w, axis=0)            # [4, G*C]; word index = r & 3
    words = words.reshape(4, G, C).transpose(1, 0, 2).reshape(G * 4, C)[:R]
    return words >= np.uint32(threshold(p))


def attn_keep_mask(B, H, S, p, site, seed, offset):
    """bool [B*H, S, S] keep-mask for attention probabilities
    (R = bh*112 + i, C = 128, c = j)."""
    if p <= 0.0:
        return np.ones((B * H, S, S), dtype=bool)
    assert S <= 112
    full = keep_mask(B * H * 112, 128, p, site, seed, offset)
    return full.reshape(B * H, 112, 128)[:, :S, :S].copy()
