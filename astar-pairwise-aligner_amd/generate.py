"""Seeded synthetic DNA pairs.

The reference draws its inputs from the un-vendored `pa-generate` crate, so this is our own
generator: counter-based splitmix64 (deterministic, vectorised, easy to re-implement natively).
`a` is uniform over ACGT; `b` is `a` with floor(e*n) edits, each uniformly one of
{substitution to a different base, insertion of a random base, deletion} at a uniform position of `a`.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser, vectorised over uint64 (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def rand_u64(seed: int, stream: int, count: int) -> np.ndarray:
    """count pseudo-random uint64: splitmix64(splitmix64(seed, stream) + i)."""
    with np.errstate(over="ignore"):
        base = splitmix64(np.array([np.uint64(seed & 0xFFFFFFFFFFFFFFFF)]) * np.uint64(0xD1342543DE82EF95)
                          + np.uint64(stream))[0]
        idx = np.arange(count, dtype=np.uint64)
        return splitmix64(base + idx * np.uint64(0x9E3779B97F4A7C15))


def random_sequence(n: int, seed: int, stream: int = 0) -> bytes:
    return _ACGT[(rand_u64(seed, stream, n) >> np.uint64(62)).astype(np.intp)].tobytes()


def mutate(a: bytes, e: float, seed: int) -> bytes:
    """Apply floor(e*len(a)) uniform edits to `a` (one pass, O(n))."""
    n = len(a)
    k = int(e * n)
    if n == 0 or k == 0:
        return a
    av = np.frombuffer(a, dtype=np.uint8)
    pos = (rand_u64(seed, 1, k) % np.uint64(n)).astype(np.int64)
    kind = (rand_u64(seed, 2, k) % np.uint64(3)).astype(np.int64)  # 0 sub, 1 ins, 2 del
    base = (rand_u64(seed, 3, k) >> np.uint64(62)).astype(np.int64)
    code = np.searchsorted(_ACGT, av)  # 0..3 for ACGT
    # per-position effects (later edits at the same position overwrite earlier substitutions)
    deleted = np.zeros(n, bool)
    deleted[pos[kind == 2]] = True
    sub_shift = np.zeros(n, np.int64)
    sp = pos[kind == 0]
    sub_shift[sp] = 1 + base[kind == 0] % 3  # shift by 1..3 => always a different base
    new_code = (code + sub_shift) % 4
    ins_cnt = np.bincount(pos[kind == 1], minlength=n)
    # emit: for each i, ins_cnt[i] random bases, then (unless deleted) the (substituted) base
    keep = (~deleted).astype(np.int64)
    out_len = int(ins_cnt.sum() + keep.sum())
    starts = np.cumsum(ins_cnt + keep) - (ins_cnt + keep)
    out = np.empty(out_len, np.uint8)
    # kept bases
    kept_idx = np.nonzero(keep)[0]
    out[starts[kept_idx] + ins_cnt[kept_idx]] = _ACGT[new_code[kept_idx]]
    # inserted bases
    tot_ins = int(ins_cnt.sum())
    if tot_ins:
        owner = np.repeat(np.arange(n), ins_cnt)
        within = np.arange(tot_ins) - np.repeat(np.cumsum(ins_cnt) - ins_cnt, ins_cnt)
        ins_bases = (rand_u64(seed, 4, tot_ins) >> np.uint64(62)).astype(np.intp)
        out[starts[owner] + within] = _ACGT[ins_bases]
    return out.tobytes()


def generate_pair(n: int, e: float, seed: int) -> tuple[bytes, bytes]:
    a = random_sequence(n, seed)
    return a, mutate(a, e, seed)
