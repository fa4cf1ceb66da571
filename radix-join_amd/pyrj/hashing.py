"""The library's key hashes, restated in numpy (host logic; lets tests check device output).

INT32 keys use murmur3's fmix32, INT64/FP64 keys fmix64 (the reference hashes with fmix64,
src/execute.cpp:21-27; the choice of hash is not observable in results).  Both are bijections,
which is why partitions can store hashed keys and un-hash on emit (csrc/rj_kernels.hip).
Radix digits come from the LOW bits, LDS slot bits from the bits above them, and the sharding
digit (which rank owns a key) from the TOP bits.
"""
import numpy as np

M32 = np.uint32(0xFFFFFFFF)


def fmix32(k):
    h = np.asarray(k).astype(np.uint32, copy=True)
    with np.errstate(over="ignore"):
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    return h


def unfmix32(h):
    h = np.asarray(h).astype(np.uint32, copy=True)
    with np.errstate(over="ignore"):
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x7ED1B41D)
        h ^= (h >> np.uint32(13)) ^ (h >> np.uint32(26))
        h *= np.uint32(0xA5CB9243)
        h ^= h >> np.uint32(16)
    return h


def owner_rank(keys_int32, n_ranks):
    """Which rank owns a key in the sharded join: the top log2(n_ranks) hash bits."""
    rb = (n_ranks - 1).bit_length()
    if rb == 0:
        return np.zeros(np.asarray(keys_int32).shape[0], dtype=np.int64)
    return (fmix32(np.asarray(keys_int32).view(np.uint32)) >> np.uint32(32 - rb)).astype(np.int64)
