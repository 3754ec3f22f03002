"""ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

numpy restatement of the device-side negative sampler's stream (graph_recsys_benchmark_amd/csrc/sampler.hip):
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
reference constants), key = 64-bit seed, counter = (row lo, row hi, attempt, offset), first output word w -> item
floor(w * num_items / 2^32), rejection against the user's training positives for the 'unseen' strategy.
The sampler is an ADDITION to the reference, so there is no reference fixture for it: this file
pins the published algorithm (known-answer vector below) and the kernel is checked against this file bit for bit.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over the counter words (uint64 arrays holding 32-bit values); returns the 4 output words."""
    c0, c1, c2, c3 = (np.asarray(a, dtype=np.uint64) & MASK for a in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0), p1 & MASK,
                          (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1), p0 & MASK)
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def known_answer():
    """Random123 kat_vectors: philox4x32-10, counter = key = all ones -> 408f276d 41c83b0e a20bc7c6 6d5451fd;
    counter = key = 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8."""
    ones = 0xFFFFFFFF
    a = [int(x[0]) for x in philox4x32_10([ones], [ones], [ones], [ones], ones, ones)]
    b = [int(x[0]) for x in philox4x32_10([0], [0], [0], [0], 0, 0)]
    return a, b


def sample_negatives(pos_u, pos_i, k, item_lo, num_items, seen_keys=None, seed=0, offset=0, max_attempts=64):
    """[n_pos*k, 3] int64 triples (u, i+, i-) and the number of rows whose attempts ran out."""
    pos_u, pos_i = np.asarray(pos_u, dtype=np.int64), np.asarray(pos_i, dtype=np.int64)
    rows = np.arange(pos_u.shape[0] * k, dtype=np.uint64)
    u = np.repeat(pos_u, k)
    neg = np.full(rows.shape, item_lo, dtype=np.int64)
    todo = np.ones(rows.shape, dtype=bool)
    seen = None if seen_keys is None else np.asarray(seen_keys, dtype=np.int64)
    for attempt in range(max_attempts):
        idx = np.nonzero(todo)[0]
        if idx.size == 0:
            break
        r = rows[idx]
        w, _, _, _ = philox4x32_10(r & MASK, r >> np.uint64(32), np.full(r.shape, attempt, dtype=np.uint64),
                                   np.full(r.shape, offset, dtype=np.uint64), seed & 0xFFFFFFFF, seed >> 32)
        j = ((w * np.uint64(num_items)) >> np.uint64(32)).astype(np.int64)
        neg[idx] = item_lo + j
        if seen is None:
            todo[idx] = False
        else:
            key = u[idx] * num_items + j
            pos = np.searchsorted(seen, key)
            hit = (pos < seen.size) & (seen[np.minimum(pos, seen.size - 1)] == key)
            todo[idx] = hit
    out = np.stack([u, np.repeat(pos_i, k), neg], axis=1)
    return out, int(todo.sum())
