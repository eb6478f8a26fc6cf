#!/usr/bin/env python3
"""Occupancy quality of the table hash (kmer_bits.h kh_hash_n) and of cheaper variants, CPU only.

For several structured key sets (canonical k-mers of a random genome, of low-complexity sequence,
sequential integers, strided integers) it reports the chi-square / dof of the region histogram
(top RB bits) and of the in-region start histogram (next 12 bits), and the fullest region relative
to the mean -- next to splitmix64 as the yardstick.  ~1.0 is ideal for chi2/dof."""
import sys
import numpy as np

FC = [0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D, 0x27D4EB2F]
M32 = np.uint64(0xFFFFFFFF)


def feistel(keys, k, rounds=4, xorshift=False, mul24=None, mulhi_even=False):
    """mulhi_even=False: the hash of rounds 1-5 (every round keeps the top k bits of the product's LOW word).
    mulhi_even=True: the shipped hash since round 6 -- rounds 2 and 4 keep bits k .. 2k-1 of the full product (v_mul_hi_u32 of
    the left-aligned half): two instructions per round in the written-out window instead of three (window.hip.h)."""
    keys = keys.astype(np.uint64)
    mask = np.uint64((1 << k) - 1)
    if mul24 is None:
        mul24 = 16 <= k <= 24
    L = (keys >> np.uint64(k)) & mask
    R = keys & mask
    for i, c in enumerate(FC[:rounds]):
        if mulhi_even and i % 2 == 1:
            t = ((R * np.uint64(c)) >> np.uint64(k)) & mask     # R < 2^32, c < 2^32: the product fits 64 bits
        else:
            cc = np.uint64((c & 0xFFFFFF) | 1) if mul24 else np.uint64(c)
            t = (R * cc) & M32
            if xorshift:
                t ^= t >> np.uint64(15)
            t >>= np.uint64(32 - k)
        t = (L ^ t) & mask
        L, R = R, t
    return (L << np.uint64(k)) | R


def splitmix(keys, k):
    z = keys.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z >> np.uint64(64 - 2 * k)


def canon_kmers(seq, k):
    codes = np.frombuffer(seq, dtype=np.uint8)
    codes = ((codes >> 1) ^ (codes >> 2)) & 3
    n = len(codes) - k + 1
    f = np.zeros(n, dtype=np.uint64)
    r = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        f = (f << np.uint64(2)) | codes[j:j + n].astype(np.uint64)
        r |= (np.uint64(3) - codes[j:j + n].astype(np.uint64)) << np.uint64(2 * j)
    return np.unique(np.minimum(f, r))


def stats(h, k, rb):
    top = (h >> np.uint64(2 * k - rb)).astype(np.int64)
    start = ((h >> np.uint64(2 * k - rb - 12)) & np.uint64(4095)).astype(np.int64)
    out = []
    for v, bins in ((top, 1 << rb), (start, 4096)):
        c = np.bincount(v, minlength=bins).astype(np.float64)
        e = len(v) / bins
        out.append(((c - e) ** 2 / e).sum() / (bins - 1))
    c = np.bincount(top, minlength=1 << rb)
    out.append(c.max() / (len(top) / (1 << rb)))
    return out


def probe_stats(h, k):
    """Linear probing inside 4096-slot regions at load ~0.5 (region count chosen from the key count):
    mean and maximum displacement from the start slot (wrap ignored: regions are processed as open runs)."""
    rb = max(1, int(np.floor(np.log2(len(h) / 2048))))
    slot = (h >> np.uint64(2 * k - rb - 12)).astype(np.int64)   # region * 4096 + start
    slot.sort()
    # position = max(slot, previous position + 1): pos_i = i + max_{j<=i}(slot_j - j)
    idx = np.arange(len(slot))
    pos = idx + np.maximum.accumulate(slot - idx)
    disp = pos - slot
    return rb, disp.mean(), disp.max(), len(h) / (4096 << rb)


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 21
    rb = 10
    rng = np.random.default_rng(1)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=3_000_000)].tobytes()
    lowc = (b"ACACACACAT" * 40 + b"GGGGGGGGGGGGGGGGGGGGC" * 20) * 400
    lowc = bytes(np.where(rng.random(len(lowc)) < 0.02, rng.integers(65, 69, len(lowc)), np.frombuffer(lowc, dtype=np.uint8)).astype(np.uint8))
    lowc = bytes(b if b in b"ACGT" else 65 for b in lowc)
    sets = {
        "genome": canon_kmers(genome, k),
        "low-complexity": canon_kmers(lowc, k),
        "sequential": np.arange(2_000_000, dtype=np.uint64),
        "stride 2^k": (np.arange(2_000_000, dtype=np.uint64) << np.uint64(k)) & np.uint64((1 << (2 * k)) - 1),
        "stride 4097": (np.arange(2_000_000, dtype=np.uint64) * np.uint64(4097)) & np.uint64((1 << (2 * k)) - 1),
    }
    variants = {
        "feistel 4r, mulhi in rounds 2 and 4 (shipped)": lambda x: feistel(x, k, mulhi_even=True),
        "feistel 4r (rounds 1-5)": lambda x: feistel(x, k),
        "feistel 4r + xorshift (earlier)": lambda x: feistel(x, k, xorshift=True),
        "feistel 3r": lambda x: feistel(x, k, rounds=3),
        "feistel 3r + xorshift": lambda x: feistel(x, k, rounds=3, xorshift=True),
        "splitmix64": lambda x: splitmix(x, k),
    }
    print(f"k={k}, regions 2^{rb}; columns: chi2/dof region, chi2/dof start, max/mean region")
    for sname, keys in sets.items():
        keys = np.unique(keys)
        print(f"-- {sname}: {len(keys)} keys")
        for vname, fn in variants.items():
            hh = fn(keys)
            a, b, c = stats(hh, k, rb)
            prb, dm, dx, load = probe_stats(hh, k)
            print(f"   {vname:46s} {a:8.3f} {b:8.3f} {c:7.3f}   probing at load {load:.2f}: mean disp {dm:6.3f}, max {dx}")


if __name__ == "__main__":
    main()
