"""GPU parity tests: the HIP path, called through the C ABI (include/kmerhip.h via
krust_amd.native), against the CPU oracle on the same inputs -- bit-exact maps.

Run with `pytest -m gpu` on an MI355X.  Nothing here reads /root/reference."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "krust_kats.json")) as f:
    KATS = json.load(f)
with open(os.path.join(HERE, "golden", "derived_fixture_tables.json")) as f:
    DERIVED = json.load(f)

SEED = 20260130
NCPU = max(1, min(os.cpu_count() or 1, 16))


@pytest.fixture(scope="module")
def K():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU (run through gpurun)"
    import krust_amd
    krust_amd.lib()  # ImportError if the HIP extension is missing: no silent fallback
    return krust_amd


@pytest.fixture(params=["direct", "partition"])
def path(request):
    """Both insert strategies must give the same table: device-scope atomics straight into HBM,
    and the two-level partition + LDS region rebuild."""
    return request.param


def flat(records, quals=None):
    """records -> one flat buffer with '\\n' separators (the kh_push layout)."""
    b = b"\n".join(records) + b"\n"
    q = None
    if quals is not None:
        q = b"\n".join(quals) + b"\n"
        assert len(q) == len(b)
    return b, q


def gpu_count(K, records, k, quals=None, min_quality=None, path=None, **kw):
    kw["path"] = path
    b, q = flat(records, quals)
    with K.DeviceCounter(k, min_quality=min_quality, **kw) as dc:
        dc.push(b, q)
        st = dc.finish()
        d = dc.as_dict()
    assert st["kmers"] == sum(d.values())
    assert st["distinct"] == len(d)
    return d


def oracle_dict(records, k, quals=None, min_quality=None):
    return O.count_records(records, k, quals=quals, min_quality=min_quality).as_dict()


# ---------------------------------------------------------------------------
# the reference's own known-answer tests, through the device path
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("kat", KATS["count_kats"], ids=lambda k: k["name"])
def test_count_kats(K, kat, path):
    recs = [r.encode() for r in kat["records"]]
    got = {K.unpack(key, kat["k"]): c for key, c in gpu_count(K, recs, kat["k"], path=path).items()}
    if kat["exact"]:
        assert got == kat["counts"]
    else:
        for key, c in kat["counts"].items():
            assert got.get(key) == c
    for key in kat.get("absent", []):
        assert key not in got
    assert got == O.count_records(recs, kat["k"]).as_str_dict(kat["k"])


@pytest.mark.parametrize("kat", KATS["quality_kats"], ids=lambda k: k["name"])
def test_quality_kats(K, kat, path):
    seq = kat["seq"].encode()
    qual = kat["qual"].encode() if kat["qual"] is not None else None
    got = gpu_count(K, [seq], kat["k"], quals=None if qual is None else [qual], min_quality=kat["min_quality"], path=path)
    if "distinct" in kat:
        assert len(got) == kat["distinct"] and list(got.values()) == [kat["only_count"]]
    if kat.get("nonempty"):
        assert got == oracle_dict([seq], kat["k"])


def test_equal_map_and_histogram_kats(K):
    for kat in KATS["equal_map_kats"]:
        a = gpu_count(K, [r.encode() for r in kat["a"]], kat["k"])
        b = gpu_count(K, [r.encode() for r in kat["b"]], kat["k"])
        assert a == b and a
    for kat in KATS["histogram_kats"]:
        b, _ = flat([r.encode() for r in kat["records"]])
        with K.DeviceCounter(kat["k"]) as dc:
            dc.push(b)
            dc.finish()
            assert tuple(kat["contains_line"]) in dc.histogram(kat["min_count"])


def test_bad_k(K):
    for k in KATS["kmer_length"]["err"]:
        with pytest.raises(K.KmerLengthError):
            K.DeviceCounter(k)


def _parse_fixture(path):
    recs, quals = [], []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    if lines and lines[0].startswith(b">"):
        return [lines[i + 1] for i in range(0, len(lines) - 1, 2)], None
    for i in range(0, len(lines) - 1, 4):
        recs.append(lines[i + 1])
        quals.append(lines[i + 3])
    return recs, quals


@pytest.mark.parametrize("row", DERIVED["tables"], ids=lambda r: f'{r["fixture"]}-k{r["k"]}-q{r["min_quality"]}')
def test_fixture_tables(K, row, fixtures_dir, path):
    recs, quals = _parse_fixture(os.path.join(fixtures_dir, row["fixture"]))
    b, q = flat(recs, quals)
    with K.DeviceCounter(row["k"], min_quality=row["min_quality"], path=path) as dc:
        dc.push(b, q)
        st = dc.finish()
        assert dc.as_str_dict() == row["counts"]
        assert [list(x) for x in dc.histogram(1)] == row["histogram"]
        assert st["kmers"] == row["total"] and st["distinct"] == row["distinct"]


# ---------------------------------------------------------------------------
# random / ragged / edge inputs vs the oracle
# ---------------------------------------------------------------------------

def _dirty(rng, n, p_bad=0.02, lower=True):
    alpha = np.frombuffer(b"ACGTacgt" if lower else b"ACGT", dtype=np.uint8)
    s = alpha[rng.integers(0, alpha.size, size=n)]
    bad = rng.random(n) < p_bad
    s = np.where(bad, np.frombuffer(b"NnRY*-", dtype=np.uint8)[rng.integers(0, 6, size=n)], s)
    return s.astype(np.uint8).tobytes()


@pytest.mark.parametrize("k", [1, 2, 3, 5, 11, 15, 16, 17, 21, 31, 32])
def test_random_ragged_records(K, k, path):
    rng = np.random.default_rng(1000 + k)
    recs = [_dirty(rng, int(n)) for n in rng.integers(0, 400, size=300)]
    recs += [b"", b"A", b"N" * 50, b"A" * 200, b"ACGT" * 50, _dirty(rng, 5000, p_bad=0.0)]
    assert gpu_count(K, recs, k, path=path) == oracle_dict(recs, k)


@pytest.mark.parametrize("k,minq", [(4, 20), (21, 20), (31, 20), (32, 0), (21, 255), (5, 40), (21, 41)])
def test_random_quality_masking(K, k, minq, path):
    rng = np.random.default_rng(77 + k + minq)
    recs, quals = [], []
    for n in rng.integers(0, 500, size=200):
        recs.append(_dirty(rng, int(n)))
        quals.append(rng.choice(np.frombuffer(b"!#+5?IJ~\xff", dtype=np.uint8), size=int(n)).astype(np.uint8).tobytes())
    assert gpu_count(K, recs, k, quals=quals, min_quality=minq, path=path) == oracle_dict(recs, k, quals=quals, min_quality=minq)
    # qual present but no threshold, and threshold but no qual: nothing is filtered (run.rs:543)
    assert gpu_count(K, recs, k, quals=quals, min_quality=None, path=path) == oracle_dict(recs, k)
    assert gpu_count(K, recs, k, quals=None, min_quality=minq, path=path) == oracle_dict(recs, k)


def test_empty_and_tiny_inputs(K, path):
    for k in (1, 21, 32):
        assert gpu_count(K, [], k, path=path) == {}
        assert gpu_count(K, [b""], k, path=path) == {}
        assert gpu_count(K, [b"ACGT"[: k - 1] if k <= 4 else b"A" * (k - 1)], k, path=path) == {}
    with K.DeviceCounter(21, path=path) as dc:
        dc.push(b"")
        assert dc.finish()["kmers"] == 0 and dc.result_size() == 0 and dc.histogram() == []


def test_one_long_record_and_tile_boundaries(K, path):
    """A single 300 kb record (hg38-style long record): windows cross the 4096-position tile
    boundaries and the workgroup range boundaries; N runs and soft-masked blocks inside."""
    rng = np.random.default_rng(5)
    s = bytearray(_dirty(rng, 300_000, p_bad=0.0))
    s[50_000:50_700] = b"N" * 700
    s[4090:4100] = b"acgtacgtac"
    s[8191] = ord("N")
    for k in (21, 32, 7):
        assert gpu_count(K, [bytes(s)], k, path=path) == oracle_dict([bytes(s)], k)


def test_device_pointer_alignment(K, path):
    """kh_push_device with base/qual pointers at every 16-byte phase (aligned fast path and the
    byte-wise quality fallback)."""
    import torch
    rng = np.random.default_rng(11)
    recs, quals = [], []
    for n in rng.integers(1, 300, size=120):
        recs.append(_dirty(rng, int(n)))
        quals.append(rng.choice(np.frombuffer(b"#5I", dtype=np.uint8), size=int(n)).astype(np.uint8).tobytes())
    b, q = flat(recs, quals)
    want = oracle_dict(recs, 21, quals=quals, min_quality=20)
    n = len(b)
    for boff, qoff in [(0, 0), (1, 1), (5, 5), (15, 15), (3, 0), (0, 7), (9, 12)]:
        tb = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        tq = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        tb[boff:boff + n] = torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
        tq[qoff:qoff + n] = torch.frombuffer(bytearray(q), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        with K.DeviceCounter(21, min_quality=20, path=path) as dc:
            dc.push_device(tb.data_ptr() + boff, tq.data_ptr() + qoff, n)
            dc.finish()
            assert dc.as_dict() == want, (boff, qoff)


def test_multiple_pushes_and_reset(K, path):
    rng = np.random.default_rng(3)
    recs = [_dirty(rng, int(n)) for n in rng.integers(0, 300, size=400)]
    want = oracle_dict(recs, 13)
    with K.DeviceCounter(13, path=path) as dc:
        for i in range(0, len(recs), 37):  # k-mers never span pushes: each push holds whole records
            dc.push(flat(recs[i:i + 37])[0])
        dc.finish()
        assert dc.as_dict() == want
        dc.reset()
        assert dc.finish()["kmers"] == 0 and dc.result_size() == 0
        dc.push(flat(recs)[0])
        dc.finish()
        assert dc.as_dict() == want


@pytest.mark.parametrize("k,minq", [(21, None), (31, 20), (25, None), (15, None), (17, 20)],
                         ids=["k21", "k31q20", "k25", "k15", "k17q20"])
def test_many_batches_into_one_table_match_the_direct_path(K, k, minq):
    """2000 partitioned batches at shifting offsets against the same pushes through the direct path.
    Regression for a level-1 write-out bug: the destination of a partition's run is biased by the run's
    tile-local offset and wraps below zero for the pool's first chunks; the value -1 (about one batch
    in 300, in the workgroup that takes the pool's first chunks) was also the "drop" marker, so that run
    was never written and level 2 read whatever the previous batch had left there -- totals conserved,
    a handful of wrong keys.  Each batch has exactly one such workgroup, hence many small batches."""
    import torch
    reads, rl = 110_000, 150
    tb = torch.empty(reads * (rl + 1), dtype=torch.uint8, device="cuda")
    tq = torch.empty_like(tb) if minq is not None else None
    qp = tq.data_ptr() if tq is not None else None
    K.synth_reads_device(tb.data_ptr(), qp, SEED, 1 << 20, rl, 0, reads)
    torch.cuda.synchronize()
    # 8 tiles per batch; offsets are arbitrary (records may be cut: same cut both ways).  2000 batches at k = 21 (the bug's own
    # configuration: one batch in ~300 met it), 1000 for the other payload widths / masks
    span, step, nb = 131072, 7963, (2000 if k == 21 else 1000)
    assert (nb - 1) * step + span <= tb.numel()
    tables = {}
    for path in ("direct", "partition"):
        with K.DeviceCounter(k, min_quality=minq, path=path, capacity_hint=8_000_000) as dc:
            for b in range(nb):
                dc.push_device(tb.data_ptr() + b * step, None if qp is None else qp + b * step, span)
            st = dc.finish()
            assert st["grows"] == 0
            tables[path] = dc.result(sort=True) + (st["kmers"],)
    (dk, dcnt, dn), (pk, pcnt, pn) = tables["direct"], tables["partition"]
    assert dn == pn == int(dcnt.sum()) == int(pcnt.sum())
    assert dk.size == pk.size and np.array_equal(dk, pk) and np.array_equal(dcnt, pcnt)
    # third leg: the CPU oracle on the same 2000 spans (k-mers never span pushes: one separator between spans),
    # whole map by total, distinct count and the order-independent digest of all (key, count) pairs
    hb = tb.cpu().numpy()
    hq = tq.cpu().numpy() if tq is not None else None
    idx = (np.arange(nb, dtype=np.int64)[:, None] * step + np.arange(span + 1, dtype=np.int64)[None, :]).ravel()
    sep = np.tile(np.arange(span + 1) == span, nb)
    cat = hb[np.minimum(idx, hb.size - 1)]
    cat[sep] = ord("\n")
    catq = None
    if hq is not None:
        catq = hq[np.minimum(idx, hq.size - 1)]
        catq[sep] = ord("\n")
    total, distinct, digest = O.count_flat_radix(cat, k, qual=catq, min_quality=minq, nthreads=NCPU)
    with np.errstate(over="ignore"):
        got = int(_np_mix64(pk ^ _np_mix64(pcnt)).sum(dtype=np.uint64))
    assert (pn, pk.size, got) == (total, distinct, digest)


# restatement of the table hash (krust_amd/csrc/kmer_bits.h kh_hash_n / kh_unhash_n), to BUILD keys with chosen hash bits
_FC = (0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D, 0x27D4EB2F)


def _feistel_f(r, c, k, i):
    if i % 2:  # rounds 2 and 4 (kh_feistel_g): bits k .. 2k-1 of the full product
        return ((r * c) >> k) & ((1 << k) - 1)
    t = (r * ((c & 0xFFFFFF) | 1)) & 0xFFFFFFFF if 16 <= k <= 24 else (r * c) & 0xFFFFFFFF
    return t >> (32 - k) if k < 32 else t


def _table_hash(key, k):
    mask = (1 << k) - 1
    L, R = (key >> k) & mask, key & mask
    for i, c in enumerate(_FC):
        L, R = R, (L ^ _feistel_f(R, c, k, i)) & mask
    return (L << k) | R


def _table_unhash(h, k):
    mask = (1 << k) - 1
    L, R = (h >> k) & mask, h & mask
    for i, c in reversed(list(enumerate(_FC))):
        L, R = (R ^ _feistel_f(L, c, k, i)) & mask, L
    return (L << k) | R


def test_payload_equal_to_the_lds_free_marker(K):
    """At k = 21 with 1024 level-1 partitions the 32-bit payload is ALL of the hash below the partition
    digit, and the LDS image of region_count_kernel32 marks free slots with 0xFFFFFFFF: a key whose
    payload is 0xFFFFFFFF (one per partition; ~1 in 4 full-size runs meets one) takes the "special"
    route -- counted on the side, placed at write-back.  Random data never gets there, so build those
    keys by inverting the hash, plus their neighbours 0xFFFFFFFE / 0xFFFFFFFD that probe into the same
    slots, and count them among ordinary reads: fresh table, then a second push into the filled one."""
    k = 21
    rng = np.random.default_rng(11)
    for key in [int(x) for x in rng.integers(0, 1 << 42, size=200)]:  # pin the restatement on the library's own hash
        assert _table_unhash(_table_hash(key, k), k) == key
        assert K.owner(key, k, 1 << 20) == _table_hash(key, k) >> (2 * k - 20)
    recs, n_special = [], 0
    for p1 in range(1024):
        for low, reps in ((0xFFFFFFFF, 1 + p1 % 5), (0xFFFFFFFE, 2), (0xFFFFFFFD, 1)):
            key = _table_unhash((p1 << 32) | low, k)
            if K.canonical(key, k)[0] != key:
                continue  # its reverse complement is the canonical one, with some other hash
            n_special += low == 0xFFFFFFFF
            recs += [K.unpack(key, k).encode()] * reps
    assert n_special > 300
    genome = _dirty(rng, 1 << 16, p_bad=0.0, lower=False)
    recs += [genome[o:o + 150] for o in rng.integers(0, (1 << 16) - 150, size=20_000)]
    order = rng.permutation(len(recs))
    recs = [recs[i] for i in order]
    want = oracle_dict(recs, k)
    b, _ = flat(recs)
    for hint in (8_000_000, 3_000_000):  # 4 / 1 level-2 buckets per partition
        with K.DeviceCounter(k, path="partition", capacity_hint=hint) as dc:
            dc.push(b)
            st = dc.finish()
            assert st["grows"] == 0 and dc.as_dict() == want
            dc.push(b)  # the special keys are now OLD keys of their regions
            half = len(recs) // 2
            dc.push(flat(recs[:half])[0])
            dc.finish()
            want3 = oracle_dict(recs + recs + recs[:half], k)
            assert dc.as_dict() == want3
            # NEW keys whose probe sequence runs over the slot of an OLD key with the free-marker payload: that
            # slot is marked taken in the LDS image (it used to look free -- a newcomer claimed it, its count went
            # to the old key and the newcomer was lost: distinct short by one, totals conserved; seen as
            # 2,102,811,924 instead of ...926 distinct on the hg38-sized input of tests/test_gpu_scale.py)
            late = []
            for p1 in range(1024):
                for low in (0xFFFFFFFC, 0xFFFFFFFB, 0xFFFFFFFA, 0xFFFFFFF9):
                    key = _table_unhash((p1 << 32) | low, k)
                    if K.canonical(key, k)[0] == key:
                        late.append(K.unpack(key, k).encode())
            assert len(late) > 1000
            dc.push(flat(late)[0])
            st = dc.finish()
            want4 = dict(want3)
            for key, c in oracle_dict(late, k).items():
                want4[key] = want4.get(key, 0) + c
            assert st["distinct"] == len(want4) and dc.as_dict() == want4


@pytest.mark.parametrize("narrow", ["1", "0"], ids=["image", "wide-only"])
def test_shard_image_takes_the_free_marker_payload(K, monkeypatch, narrow):
    """Round 6: kh_merge_across builds the shard as the 8-byte image (shard.hip.h shard_merge_narrow_kernel), whose LDS form
    marks free slots with the payload 0xFFFFFFFF.  In a world of one rank at k = 21 the payload is ALL 32 hash bits behind the
    level-1 digit, so a key can carry that very value: summed apart, placed by one lane at the end.  The keys built by
    inverting the hash (as above), with the neighbours that probe over their slots, through RCCL's world of one."""
    monkeypatch.setenv("KMERHIP_NARROW", narrow)
    k = 21
    rng = np.random.default_rng(12)
    recs, n_special = [], 0
    for p1 in range(1024):
        for low, reps in ((0xFFFFFFFF, 1 + p1 % 5), (0xFFFFFFFE, 2), (0xFFFFFFFD, 1), (0xFFFFFFFC, 1)):
            key = _table_unhash((p1 << 32) | low, k)
            if K.canonical(key, k)[0] != key:
                continue
            n_special += low == 0xFFFFFFFF
            recs += [K.unpack(key, k).encode()] * reps
    assert n_special > 300
    genome = _dirty(rng, 1 << 16, p_bad=0.0, lower=False)
    recs += [genome[o:o + 150] for o in rng.integers(0, (1 << 16) - 150, size=20_000)]
    recs = [recs[i] for i in rng.permutation(len(recs))]
    want = oracle_dict(recs, k)
    b, _ = flat(recs)
    for pieces in ("1", "4"):
        monkeypatch.setenv("KMERHIP_MERGE_PIECES", pieces)
        with K.DeviceCounter(k, capacity_hint=3_000_000) as dc:   # 2^11 regions: packed pairs (31 bits below the region index)
            dc.comm_init(1, 0, K.comm_unique_id())
            dc.push(b)
            info = dc.merge_across()
            assert info["path"].startswith("regions-packed") and info["conserved"] == 1, info
            st = dc.finish()
            assert st["slot_bytes"] == (8 if narrow == "1" else 16)
            assert st["distinct"] == len(want) and dc.as_dict() == want
            probe = np.array(list(want)[:3000], dtype=np.uint64)
            assert dc.lookup(probe).tolist() == [want[int(x)] for x in probe]


def test_shard_image_counts_beyond_32_bits_fall_back(K):
    """The image's counts are 32-bit: a target region that takes in 2^32 occurrences or more fails with code 2, the host widens the
    shard, and that region's units go in again through the direct path -- u64 counts are the reference's range (src/run.rs:569).
    One rank's packed export (a homopolymer with 1.66 G copies among ordinary reads) merged as THREE senders."""
    import torch
    k = 21
    n_a = 1_600_000
    poly = np.full((n_a, 151), ord("A"), dtype=np.uint8)
    poly[:, 150] = 10
    other, _ = O.synth_reads(SEED, 1 << 18, 150, 0, 50_000, with_qual=False)
    m = O.OracleMap()
    m.scan_flat(other, k, nthreads=NCPU)
    ta = torch.from_numpy(poly.reshape(-1)).cuda()
    to = torch.from_numpy(other.copy()).cuda()
    torch.cuda.synchronize()
    pushes = 8
    with K.DeviceCounter(k, capacity_hint=6_000_000, path="partition") as dc:
        dc.push_device(to.data_ptr(), None, to.numel())
        for _ in range(pushes):
            dc.push_device(ta.data_ptr(), None, ta.numel())
        st = dc.finish()
        R = st["table_slots"] // 4096
        dk = torch.empty(st["distinct"], dtype=torch.int64, device="cuda")
        rc = torch.empty(R, dtype=torch.int32, device="cuda")
        parts, R2 = dc.export_regions_packed_device(1, dk.data_ptr(), st["distinct"], rc.data_ptr(), R)
        assert R2 == R and int(parts.sum()) == st["distinct"]
    one = dict(m.as_dict())
    one[0] = one.get(0, 0) + pushes * n_a * 130
    assert 3 * one[0] > 1 << 32 > one[0]
    with K.DeviceCounter(k, capacity_hint=6_000_000) as dc:
        dc.set_shard(0, 1)
        dc.merge_regions_packed_device(R, [dk.data_ptr()] * 3, [rc.data_ptr()] * 3)
        st = dc.finish()
        assert st["slot_bytes"] == 16 and st["distinct"] == len(one) and st["kmers"] == 3 * sum(one.values())
        assert dc.as_dict() == {key: 3 * c for key, c in one.items()}
        assert int(dc.lookup(np.array([0], dtype=np.uint64))[0]) == 3 * one[0]


def test_lazy_reset_never_leaks_old_entries(K, path):
    """kh_reset does not clear the table (the next FRESH partitioned pass rewrites every region, any
    other use clears first).  Fill the table, reset, then go through each way of using a reset table:
    a second, DIFFERENT data set whose buckets leave most regions empty; immediate output calls; a
    lookup of the old keys; a merge."""
    rng = np.random.default_rng(17)
    big = [_dirty(rng, 400, p_bad=0.0, lower=False) for _ in range(3000)]       # ~1.1 M distinct 15-mers
    small = [_dirty(rng, 60, p_bad=0.0, lower=False) for _ in range(40)]       # a few thousand: most regions get no bucket
    want_big, want_small = oracle_dict(big, 15), oracle_dict(small, 15)
    with K.DeviceCounter(15, path=path, capacity_hint=2_000_000) as dc:
        dc.push(flat(big)[0])
        dc.finish()
        assert dc.as_dict() == want_big
        old_keys = np.fromiter(want_big.keys(), dtype=np.uint64)[:5000]
        dc.reset()                                                               # -> output straight away
        assert dc.result_size() == 0 and dc.histogram() == []
        dc.reset()
        assert not dc.lookup(old_keys).any()                                     # -> lookup straight away
        dc.push(flat(big)[0]); dc.finish()
        dc.reset()
        dc.push(flat(small)[0])                                                  # -> a different, small data set
        st = dc.finish()
        assert st["distinct"] == len(want_small)
        assert dc.as_dict() == want_small
        assert not dc.lookup(np.array([k for k in old_keys.tolist() if k not in want_small][:2000], dtype=np.uint64)).any()
        dc.reset()
        keys = np.fromiter(want_small.keys(), dtype=np.uint64)
        dc.merge_pairs(keys, np.ones_like(keys))                                 # -> a merge
        dc.finish()
        assert dc.as_dict() == {int(k): 1 for k in keys.tolist()}
        dc.reset()
        dc.reset()                                                               # twice in a row stays empty
        dc.push(flat(big)[0]); dc.finish()
        assert dc.as_dict() == want_big


def test_table_growth_from_tiny_capacity(K, path):
    """No capacity hint: the table starts small and must grow (rehash) without losing counts."""
    bases, _ = O.synth_reads(SEED, 1 << 22, 150, 0, 60_000, with_qual=False)
    m = O.OracleMap()
    total = m.scan_flat(bases, 21, nthreads=NCPU)
    with K.DeviceCounter(21, path=path) as dc:
        dc.push(bases)
        st = dc.finish()
        assert st["grows"] >= 1 or path == "partition"
        assert st["kmers"] == total and st["distinct"] == len(m)
        keys, cnts = dc.result()
    okeys, ocnts = m.arrays()
    assert np.array_equal(keys, okeys) and np.array_equal(cnts, ocnts)


def test_host_staging_chunk_boundary(K, path):
    """kh_push stages the host buffer in 64 MiB chunks; windows crossing a chunk boundary must
    be counted exactly once (the k-1 halo is re-sent)."""
    n_reads = 460_000  # 69.5 MB > one staging chunk
    bases, qual = O.synth_reads(SEED, 1 << 20, 150, 0, n_reads)
    # make the chunk boundary fall inside a long N-free stretch: replace separators near it
    bases = bases.copy()
    bases[(64 << 20) - 500:(64 << 20) + 500] = np.frombuffer(b"ACGT" * 250, dtype=np.uint8)
    for k, minq in ((31, None), (21, 20)):
        m = O.OracleMap()
        total = m.scan_flat(bases, k, qual=qual, min_quality=minq, nthreads=NCPU)
        with K.DeviceCounter(k, min_quality=minq, capacity_hint=len(m), path=path) as dc:
            dc.push(bases, qual)
            st = dc.finish()
            assert st["kmers"] == total and st["distinct"] == len(m)
            keys, cnts = dc.result()
        okeys, ocnts = m.arrays()
        assert np.array_equal(keys, okeys) and np.array_equal(cnts, ocnts)


@pytest.mark.parametrize("with_qual", [False, True])
def test_accumulation_buffer_smaller_than_a_staging_chunk(K, monkeypatch, path, with_qual):
    """KMERHIP_ACC_MAX_MB=1 caps the device accumulation buffers at 1 MiB -- far below the 64 MiB staging chunk of a
    pageable kh_push.  A chunk must then be cut to the room that is left (round 3 copied the whole chunk: a write past
    the buffer's end); every seam needs its k-1 look-back."""
    monkeypatch.setenv("KMERHIP_ACC_MAX_MB", "1")
    bases, qual = O.synth_reads(SEED + 5, 1 << 18, 150, 0, 60_000)  # 9 MB: nine buffers' worth, pageable numpy memory
    k, minq = (21, 20) if with_qual else (25, None)
    cut = 4_000_000  # (inside a read: k-mers never span pushes, so the oracle sees the same two pushes)
    parts = [(bases[:cut], qual[:cut] if with_qual else None), (bases[cut:], qual[cut:] if with_qual else None)]
    m2 = O.OracleMap()
    t2 = sum(m2.scan_flat(b, k, qual=q, min_quality=minq, nthreads=NCPU) for b, q in parts)
    with K.DeviceCounter(k, min_quality=minq, capacity_hint=len(m2), path=path) as dc:
        for b, q in parts:
            dc.push(b, q)
        st = dc.finish()
        keys, cnts = dc.result()
    assert st["kmers"] == t2 and st["distinct"] == len(m2)
    okeys, ocnts = m2.arrays()
    assert np.array_equal(keys, okeys) and np.array_equal(cnts, ocnts)


def test_min_count_histogram_lookup(K, path):
    bases, _ = O.synth_reads(SEED, 1 << 14, 150, 0, 20_000, with_qual=False)  # high coverage: big counts
    m = O.OracleMap()
    m.scan_flat(bases, 15, nthreads=NCPU)
    od = m.as_dict()
    with K.DeviceCounter(15, path=path) as dc:
        dc.push(bases)
        dc.finish()
        for mc in (1, 2, 50, 10**9):
            want = {k: c for k, c in od.items() if c >= mc}
            assert dc.result_size(mc) == len(want)
            assert dc.as_dict(mc) == want
            assert dc.histogram(mc) == m.histogram(mc)
        probe = np.array(list(od.keys())[:1000] + [0xDEADBEEF, 5, 2**40 + 12345], dtype=np.uint64)
        got = dc.lookup(probe)
        assert got.tolist() == [od.get(int(k), 0) for k in probe]
    # counts far beyond the dense histogram range (k=1: two keys with huge counts)
    m1 = O.OracleMap()
    m1.scan_flat(bases, 1, nthreads=NCPU)
    with K.DeviceCounter(1, path=path) as dc:
        dc.push(bases)
        dc.finish()
        assert dc.histogram(1) == m1.histogram(1)
        assert dc.as_dict() == m1.as_dict()


def test_synth_generator_matches_oracle(K):
    import torch
    for (glen, rl, first, n) in [(1 << 20, 150, 0, 3000), (1 << 27, 150, 99_999_000, 1001), (5000, 36, 7, 513)]:
        tb = torch.empty(n * (rl + 1), dtype=torch.uint8, device="cuda")
        tq = torch.empty(n * (rl + 1), dtype=torch.uint8, device="cuda")
        K.synth_reads_device(tb.data_ptr(), tq.data_ptr(), SEED, glen, rl, first, n)
        torch.cuda.synchronize()
        ob, oq = O.synth_reads(SEED, glen, rl, first, n)
        assert np.array_equal(tb.cpu().numpy(), ob)
        assert np.array_equal(tq.cpu().numpy(), oq)


def test_determinism_digest(K, path):
    """Same input twice -> identical multiset (atomics race, results must not)."""
    bases, qual = O.synth_reads(SEED, 1 << 18, 150, 0, 50_000)
    digests = []
    for _ in range(3):
        with K.DeviceCounter(21, min_quality=20, path=path) as dc:
            dc.push(bases, qual)
            dc.finish()
            k, c = dc.result()
            digests.append((int(np.bitwise_xor.reduce(k * np.uint64(0x9E3779B97F4A7C15) + c)), int(c.sum()), k.size))
    assert digests[0] == digests[1] == digests[2]


# ---------------------------------------------------------------------------
# multi-GPU merge, exercised as logical shards on one device
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_owner_partitioned_merge_logical_shards(K, nshards):
    import torch
    n_reads = 24_000
    bases, _ = O.synth_reads(SEED, 1 << 16, 150, 0, n_reads, with_qual=False)
    m = O.OracleMap()
    m.scan_flat(bases, 21, nthreads=NCPU)
    want = m.as_dict()
    per = n_reads // nshards
    exports = []
    for s in range(nshards):  # each "rank" counts its contiguous read range
        lo, hi = s * per, (n_reads if s == nshards - 1 else (s + 1) * per)
        with K.DeviceCounter(21) as dc:
            dc.push(bases[lo * 151: hi * 151])
            st = dc.finish()
            cap = st["distinct"]
            dk = torch.empty(cap, dtype=torch.int64, device="cuda")
            dcnt = torch.empty(cap, dtype=torch.int64, device="cuda")
            parts = dc.export_by_owner_device(nshards, dk.data_ptr(), dcnt.data_ptr(), cap)
            assert int(parts.sum()) == cap
            offs = np.concatenate([[0], np.cumsum(parts)]).astype(np.int64)
            hk = dk.cpu().numpy().view(np.uint64)
            for p in range(nshards):
                seg = hk[offs[p]:offs[p + 1]]
                assert all(K.owner(int(x), 21, nshards) == p for x in seg[:50])
            exports.append((dk, dcnt, offs))
    merged = {}
    for p in range(nshards):  # owner p merges the p-th segment of every shard's export
        with K.DeviceCounter(21) as dc:
            for dk, dcnt, offs in exports:
                n = int(offs[p + 1] - offs[p])
                dc.merge_pairs_device(dk.data_ptr() + 8 * int(offs[p]), dcnt.data_ptr() + 8 * int(offs[p]), n)
            dc.finish()
            d = dc.as_dict()
        assert not (set(d) & set(merged))
        assert all(K.owner(k, 21, nshards) == p for k in list(d)[:200])
        merged.update(d)
    assert merged == want


@pytest.mark.parametrize("packed", [0, 1, 2], ids=["wide", "packed64", "heads32"])
@pytest.mark.parametrize("nshards,k,minq,recv_hint", [(2, 21, None, 3_000_000), (4, 21, None, 3_000_000), (8, 21, None, 3_000_000),
                                                      (4, 31, 20, 3_000_000), (8, 9, None, 3_000_000),
                                                      (4, 21, None, 40_000_000), (4, 21, None, 20_000), (4, 19, None, 3_000_000),
                                                      (2, 17, 20, 20_000)])
def test_region_ordered_merge_logical_shards(K, nshards, k, minq, recv_hint, packed):
    """The power-of-two fast path of the multi-GPU merge, as logical shards on one device:
    region-ordered export from every 'rank', then each owner rebuilds its hash-range shard in LDS
    from the senders' region segments.  Union of the shards == single-table result."""
    import torch
    n_reads = 40_000
    bases, qual = O.synth_reads(SEED, 1 << 17, 150, 0, n_reads)
    m = O.OracleMap()
    m.scan_flat(bases, k, qual=qual, min_quality=minq, nthreads=NCPU)
    want = m.as_dict()
    per = n_reads // nshards
    exports = []
    nreg = None
    for s in range(nshards):
        lo, hi = s * per, (n_reads if s == nshards - 1 else (s + 1) * per)
        with K.DeviceCounter(k, min_quality=minq, capacity_hint=3_000_000) as dc:  # same hint -> same table size
            dc.push(bases[lo * 151: hi * 151], qual[lo * 151: hi * 151])
            st = dc.finish()
            R = st["table_slots"] // 4096
            dk = torch.empty(max(st["distinct"], 1), dtype=torch.int64, device="cuda")
            dcnt = torch.empty(max(st["distinct"], 1), dtype=torch.int64, device="cuda")
            rc = torch.empty(R, dtype=torch.int32, device="cuda")
            hb = 2 * k - (R.bit_length() - 1)  # hash bits below the region index
            if packed == 1:  # one u64 per pair: count << 32 | 32 hash bits below the region index
                res = dc.export_regions_packed_device(nshards, dk.data_ptr(), st["distinct"], rc.data_ptr(), R)
                if hb > 32:
                    assert res is None  # not representable: the caller takes the wide route
                    pytest.skip("packed form not representable for this k / table size")
                parts, R2 = res
            elif packed == 2:  # u32 heads: hb hash bits | addend - 1; large counts split into several heads
                res = dc.export_regions_heads_device(nshards, dk.data_ptr(), 2 * st["distinct"], rc.data_ptr(), R)
                if not (1 <= hb <= 28):
                    assert res is None
                    pytest.skip("heads not representable for this k / table size")
                parts, R2 = res
                assert int(parts.sum()) >= st["distinct"]
            else:
                parts, R2 = dc.export_regions_device(nshards, dk.data_ptr(), dcnt.data_ptr(), st["distinct"], rc.data_ptr(), R)
            assert R2 == R and int(rc.sum().item()) == int(parts.sum()) and (packed == 2 or int(parts.sum()) == st["distinct"])
            nreg = R if nreg is None else nreg
            assert R == nreg
            offs = np.concatenate([[0], np.cumsum(parts)]).astype(np.int64)
            exports.append((dk, dcnt, rc, offs))
    merged = {}
    per_r = nreg // nshards
    for o in range(nshards):
        with K.DeviceCounter(k, capacity_hint=recv_hint) as dc:  # receiver tables larger / smaller than the senders' too
            dc.set_shard(o, nshards)
            if packed == 1:
                dc.merge_regions_packed_device(nreg, [e[0].data_ptr() + 8 * int(e[3][o]) for e in exports],
                                               [e[2].data_ptr() + 4 * per_r * o for e in exports])
            elif packed == 2:
                dc.merge_regions_heads_device(nreg, [e[0].data_ptr() + 4 * int(e[3][o]) for e in exports],
                                              [e[2].data_ptr() + 4 * per_r * o for e in exports])
            else:
                dc.merge_regions_device(nreg,
                                        [e[0].data_ptr() + 8 * int(e[3][o]) for e in exports],
                                        [e[1].data_ptr() + 8 * int(e[3][o]) for e in exports],
                                        [e[2].data_ptr() + 4 * per_r * o for e in exports])
            st = dc.finish()
            d = dc.as_dict()
            assert st["distinct"] == len(d)
            probe = np.array(list(d)[:500] + [12345], dtype=np.uint64)  # lookups use the sharded placement
            assert dc.lookup(probe).tolist() == [d.get(int(x), 0) for x in probe]
            # Round 6: a fresh merge of packed pairs / heads leaves the shard as the 8-byte image (count << 32 | the 32 hash bits
            # behind the shard table's level-1 digit) wherever those 32 bits hold what the region index does not -- unless the table
            # had to grow (overflowing regions go through the 16-byte table).  The image names a key by the hash bits BELOW the
            # owner's: the keys of the OTHER shards must not alias one of this shard's in a lookup.
            regions = st["table_slots"] // 4096
            p1_bits = 10 if regions > 1024 else regions.bit_length() - 1
            sh = nshards.bit_length() - 1
            xbits = 2 * k - sh - p1_bits
            # (... and the geometry the image kernel's 32-bit arithmetic covers, merge.hip geo_fast: targets no coarser than the senders'
            #  regions, the shard's level-1 digit ending inside or at the senders' x, no hash bits behind that x)
            s_p1 = 10 if nreg > 1024 else nreg.bit_length() - 1
            fast = regions >= nreg // nshards and 0 <= sh + p1_bits - s_p1 < 32 and 2 * k - s_p1 <= 32
            if st["grows"] == 0:
                assert st["slot_bytes"] == (8 if packed in (1, 2) and 1 <= xbits <= 32 and fast else 16), (st, xbits, fast)
            everyone = np.array(list(want)[:: max(1, len(want) // 20_000)], dtype=np.uint64)
            assert dc.lookup(everyone).tolist() == [d.get(int(x), 0) for x in everyone]
            assert dict(dc.histogram()) == dict(zip(*[a.tolist() for a in np.unique(np.array(list(d.values()), dtype=np.uint64), return_counts=True)]))
        assert not (set(d) & set(merged))
        assert all(K.owner(key, k, nshards) == o for key in list(d)[:300])
        merged.update(d)
    assert merged == want


@pytest.mark.parametrize("fmt", [0, 1, 2], ids=["wide", "packed64", "heads32"])
@pytest.mark.parametrize("nshards,npieces", [(2, 2), (4, 4), (2, 8)])
@pytest.mark.parametrize("scenario", ["fresh", "dirty-reversed", "interrupted", "tiny-table"])
def test_region_merge_in_pieces_logical_shards(K, fmt, nshards, npieces, scenario):
    """kh_set_region_window: the region-ordered exchange cut into pieces of every owner's region range (so
    that export / all-to-all / merge of successive pieces can overlap).  A windowed export looks like the
    export of a table that is empty outside the piece; the merge rebuilds the matching share of the
    shard.  Receivers: a fresh table; a lazily reset (dirty) one with the pieces in reverse order; one
    that is queried between two pieces (the rest then merges into a non-empty table); one that has to
    grow.  Union of the shards == single-table result, whatever the route."""
    import torch
    k, n_reads = 19, 30_000
    bases, _ = O.synth_reads(SEED, 1 << 17, 150, 0, n_reads)
    m = O.OracleMap()
    m.scan_flat(bases, k, nthreads=NCPU)
    want = m.as_dict()
    per = n_reads // nshards
    unit = 4 if fmt == 2 else 8
    exports, nreg = [], None
    for s_ in range(nshards):
        lo, hi = s_ * per, (n_reads if s_ == nshards - 1 else (s_ + 1) * per)
        with K.DeviceCounter(k, capacity_hint=3_000_000) as dc:
            dc.push(bases[lo * 151: hi * 151])
            st = dc.finish()
            R = st["table_slots"] // 4096
            nreg = R if nreg is None else nreg
            assert R == nreg
            pieces, total = [], 0
            for piece in range(npieces):
                dc.set_region_window(piece, npieces)
                dk = torch.empty(2 * st["distinct"], dtype=torch.int64, device="cuda")
                dcnt = torch.empty(st["distinct"], dtype=torch.int64, device="cuda")
                rc = torch.empty(R, dtype=torch.int32, device="cuda")
                if fmt == 1:
                    parts, _ = dc.export_regions_packed_device(nshards, dk.data_ptr(), st["distinct"], rc.data_ptr(), R)
                elif fmt == 2:
                    parts, _ = dc.export_regions_heads_device(nshards, dk.data_ptr(), 2 * st["distinct"], rc.data_ptr(), R)
                else:
                    parts, _ = dc.export_regions_device(nshards, dk.data_ptr(), dcnt.data_ptr(), st["distinct"], rc.data_ptr(), R)
                rch = rc.cpu().numpy().reshape(nshards, npieces, -1)
                assert int(rch.sum()) == int(parts.sum()) and int(rch[:, piece].sum()) == int(parts.sum())  # zero outside the piece
                total += int(parts.sum())
                pieces.append((dk, dcnt, rc, np.concatenate([[0], np.cumsum(parts)]).astype(np.int64)))
            dc.set_region_window(0, 1)
            assert total >= st["distinct"] and (fmt == 2 or total == st["distinct"])
            exports.append(pieces)
    merged = {}
    per_r = nreg // nshards
    order = list(range(npieces))
    if scenario == "dirty-reversed":
        order.reverse()
    for o in range(nshards):
        with K.DeviceCounter(k, capacity_hint=1000 if scenario == "tiny-table" else 3_000_000) as dc:
            if scenario == "dirty-reversed":  # leave other keys in the table, then the lazy reset
                dc.push(bases[:151 * 3000][::-1].copy())
                dc.finish()
                dc.reset()
            dc.set_shard(o, nshards)
            for i, piece in enumerate(order):
                dc.set_region_window(piece, npieces)
                ptrs = [e[piece][0].data_ptr() + unit * int(e[piece][3][o]) for e in exports]
                rcs = [e[piece][2].data_ptr() + 4 * per_r * o for e in exports]
                if fmt == 1:
                    dc.merge_regions_packed_device(nreg, ptrs, rcs)
                elif fmt == 2:
                    dc.merge_regions_heads_device(nreg, ptrs, rcs)
                else:
                    dc.merge_regions_device(nreg, ptrs, [e[piece][1].data_ptr() + 8 * int(e[piece][3][o]) for e in exports], rcs)
                if scenario == "interrupted" and i == 0:
                    mid = dc.finish()["distinct"]  # touches the table: the pieces still missing count as empty
                    assert 0 < mid == dc.result_size()
            dc.set_region_window(0, 1)
            st = dc.finish()
            d = dc.as_dict()
            assert st["distinct"] == len(d)
        assert not (set(d) & set(merged))
        merged.update(d)
    assert merged == want


@pytest.mark.parametrize("fmt", [0, 1, 2], ids=["wide", "packed64", "heads32"])
def test_region_merge_in_pieces_growth_while_pieces_are_missing(K, fmt):
    """The first piece sizes the shard table for npieces pieces like itself.  Here it comes from a small
    table and the later ones from a large one, so the table has to grow while pieces are still
    unwritten: they are emptied first (growing rehashes every region), the rest merges the ordinary way."""
    import torch
    k, nshards, npieces = 19, 2, 4
    bases, _ = O.synth_reads(SEED, 1 << 17, 150, 0, 30_000)
    unit = 4 if fmt == 2 else 8
    tables, exports = [], []
    for n_reads in (1_500, 30_000):  # the small sender, the large sender
        m = O.OracleMap()
        m.scan_flat(bases[:151 * n_reads], k, nthreads=NCPU)
        tables.append(m.as_dict())
        with K.DeviceCounter(k, capacity_hint=3_000_000) as dc:
            dc.push(bases[:151 * n_reads])
            st = dc.finish()
            R = st["table_slots"] // 4096
            pieces = []
            for piece in range(npieces):
                dc.set_region_window(piece, npieces)
                dk = torch.empty(2 * st["distinct"], dtype=torch.int64, device="cuda")
                dcnt = torch.empty(st["distinct"], dtype=torch.int64, device="cuda")
                rc = torch.empty(R, dtype=torch.int32, device="cuda")
                if fmt == 1:
                    parts, _ = dc.export_regions_packed_device(nshards, dk.data_ptr(), st["distinct"], rc.data_ptr(), R)
                elif fmt == 2:
                    parts, _ = dc.export_regions_heads_device(nshards, dk.data_ptr(), 2 * st["distinct"], rc.data_ptr(), R)
                else:
                    parts, _ = dc.export_regions_device(nshards, dk.data_ptr(), dcnt.data_ptr(), st["distinct"], rc.data_ptr(), R)
                pieces.append((dk, dcnt, rc, np.concatenate([[0], np.cumsum(parts)]).astype(np.int64)))
            exports.append(pieces)
    per_r = R // nshards
    for o in range(nshards):
        with K.DeviceCounter(k, capacity_hint=1000) as dc:
            dc.push(bases[:151 * 2000][::-1].copy())  # something else first, then the lazy reset
            dc.finish()
            dc.reset()
            dc.set_shard(o, nshards)
            for piece in range(npieces):
                e = exports[0 if piece == 0 else 1][piece]
                dc.set_region_window(piece, npieces)
                ptr, rcp = [e[0].data_ptr() + unit * int(e[3][o])], [e[2].data_ptr() + 4 * per_r * o]
                if fmt == 1:
                    dc.merge_regions_packed_device(R, ptr, rcp)
                elif fmt == 2:
                    dc.merge_regions_heads_device(R, ptr, rcp)
                else:
                    dc.merge_regions_device(R, ptr, [e[1].data_ptr() + 8 * int(e[3][o])], rcp)
            dc.set_region_window(0, 1)
            st = dc.finish()
            assert st["grows"] >= 1
            got = dc.as_dict()
        want = {}
        for src, keep in ((tables[0], lambda p: p == 0), (tables[1], lambda p: p != 0)):
            for key, cnt in src.items():
                fine = K.owner(key, k, nshards * npieces)  # top bits of the hash: owner, then the piece inside it
                if fine // npieces == o and keep(fine % npieces):
                    want[key] = cnt
        assert got == want


def test_heads_export_splits_large_counts_and_refuses_huge_ones(K):
    """32-bit heads carry addend - 1 in the bits the hash leaves free (k = 19, 2^11 regions: 5 bits).  Counts
    above 32 travel as several heads of the same key; a count above 64 x 32 makes the table not
    representable (the caller then takes a wider unit)."""
    import torch
    k = 19
    for n_reads, representable in ((12_000, True), (80_000, False)):
        bases, _ = O.synth_reads(SEED, 1 << 12, 150, 0, n_reads, with_qual=False)   # 4 kbp genome: counts ~ n_reads / 31
        m = O.OracleMap()
        m.scan_flat(bases, k, nthreads=NCPU)
        want = m.as_dict()
        assert (max(want.values()) > 64 * 32) == (not representable) and max(want.values()) > 32
        with K.DeviceCounter(k, capacity_hint=3_000_000) as dc:
            dc.push(bases)
            st = dc.finish()
            R = st["table_slots"] // 4096
            buf = torch.empty(8 * st["distinct"] + 1024, dtype=torch.int32, device="cuda")
            rc = torch.empty(R, dtype=torch.int32, device="cuda")
            res = dc.export_regions_heads_device(2, buf.data_ptr(), buf.numel(), rc.data_ptr(), R)
            if not representable:
                assert res is None
                continue
            parts, _ = res
            assert int(parts.sum()) > st["distinct"]  # some keys needed more than one head
            offs = np.concatenate([[0], np.cumsum(parts)]).astype(np.int64)
        merged = {}
        for o in range(2):
            with K.DeviceCounter(k, capacity_hint=3_000_000) as sh:
                sh.set_shard(o, 2)
                sh.merge_regions_heads_device(R, [buf.data_ptr() + 4 * int(offs[o])], [rc.data_ptr() + 4 * (R // 2) * o])
                sh.finish()
                merged.update(sh.as_dict())
        assert merged == want


def test_heads_export_uses_region_pass_counts(K):
    """Right after a FRESH partitioned pass the heads export takes its per-region counts from that pass
    (no counting pass over the table).  They must equal what the counting kernel finds on the same table
    built through the direct path, with and without split heads; and be dropped once the table changes."""
    import torch
    k = 19
    bases, _ = O.synth_reads(SEED, 1 << 13, 150, 0, 30_000, with_qual=False)   # counts ~ 480: several heads per key
    more, _ = O.synth_reads(SEED + 1, 1 << 13, 150, 0, 2_000, with_qual=False)
    got = {}
    for path in ("partition", "direct"):
        with K.DeviceCounter(k, capacity_hint=3_000_000, path=path) as dc:
            dc.push(bases)
            st = dc.finish()
            R = st["table_slots"] // 4096
            buf = torch.empty(64 * st["distinct"] + 1024, dtype=torch.int32, device="cuda")
            rc = torch.empty(R, dtype=torch.int32, device="cuda")
            parts, _ = dc.export_regions_heads_device(4, buf.data_ptr(), buf.numel(), rc.data_ptr(), R)
            heads = buf[: int(parts.sum())].cpu().numpy().copy()
            got[path] = (parts.copy(), rc.cpu().numpy().copy(), np.sort(heads))
            if path == "partition":  # a later push invalidates the shortcut: counts must follow the table
                dc.push(more)
                dc.finish()
                parts2, _ = dc.export_regions_heads_device(4, buf.data_ptr(), buf.numel(), rc.data_ptr(), R)
                assert int(parts2.sum()) == int(rc.sum().item()) > int(parts.sum())
    assert np.array_equal(got["partition"][0], got["direct"][0])
    assert np.array_equal(got["partition"][1], got["direct"][1])
    assert np.array_equal(got["partition"][2], got["direct"][2])
    assert int(got["partition"][0].sum()) > st["distinct"]


@pytest.mark.parametrize("k,agg", [(19, "1"), (21, "0")], ids=["k19-summed-in-lds", "k21-entry-by-entry"])
def test_heads_counts_survive_a_short_overflow_list(K, monkeypatch, k, agg):
    """Round 5: the arena level 2 leaves what does not fit a bucket's arena or bin in an overflow list that is inserted AFTER the
    region pass -- a few entries on nearly every real batch -- and until now any entry dropped the region pass's per-region head
    counts for the whole table (a 4 ms counting pass at the next export, configs[3]'s size).  Now the regions the list touched
    are counted again and the others' counts stand.  Here: reads with bursts of one repeated read (bins that overflow between two flushes: a list
    of thousands of entries, new keys and added counts alike), a table large enough for the arena path; the heads export of the
    partitioned table -- counts from the pass + the recount -- must equal, region by region and head by head, the export of the
    same table built through the direct path (whose export counts every region)."""
    import torch
    monkeypatch.setenv("KMERHIP_OVF_AGG", agg)   # both insert kernels: the entry-by-entry one marks what it has applied as consumed
    monkeypatch.setenv("KMERHIP_HEADS_ALWAYS", "1")  # (a context without a communicator leaves no head counts behind since round 5)
    n_reads = 200_000
    bases, _ = O.synth_reads(SEED + 5, 1 << 22, 150, 0, n_reads, with_qual=False)
    v = bases.reshape(n_reads, 151)
    rng = np.random.default_rng(55)
    # bursts: 150 consecutive copies of one read, sixty times over -- a key's copies arrive together, more than its level-2 bin
    # holds between two flushes, and go to the overflow list; counts stay far below what heads can carry (64 << 9)
    for b in range(60):
        rep = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=150)
        i0 = int(rng.integers(0, n_reads - 150))
        v[i0:i0 + 150, :150] = rep
    monkeypatch.setenv("KMERHIP_TABLE_REGIONS", str(1024 * 512))   # 512 buckets per partition: bins of 64 payloads
    got = {}
    for path in ("partition", "direct"):
        with K.DeviceCounter(k, capacity_hint=40_000_000, path=path, trace=(path == "partition")) as dc:
            dc.push(bases)
            st = dc.finish()
            R = st["table_slots"] // 4096
            buf = torch.empty(4 * st["distinct"] + (1 << 20), dtype=torch.int32, device="cuda")
            rc = torch.empty(R, dtype=torch.int32, device="cuda")
            res = dc.export_regions_heads_device(4, buf.data_ptr(), buf.numel(), rc.data_ptr(), R)
            assert res is not None
            parts, _ = res
            heads = buf[: int(parts.sum())].cpu().numpy().copy()
            got[path] = (parts.copy(), rc.cpu().numpy().copy(), np.sort(heads), st["stage_ms"])
    assert got["partition"][3]["level2"] > 0 and got["partition"][3]["level2_count"] == 0      # the arena path took the batch
    assert np.array_equal(got["partition"][1], got["direct"][1]), "per-region head counts differ"
    assert np.array_equal(got["partition"][0], got["direct"][0]) and np.array_equal(got["partition"][2], got["direct"][2])


@pytest.mark.parametrize("k,nshards", [(5, 3), (11, 4), (13, 2)])
def test_dense_export_and_merge_small_k(K, k, nshards):
    """k <= 13: the table as a dense array of 4^k counts (what an all-reduce(sum) merges), and back into
    per-owner tables.  The 'all-reduce' is a sum over logical ranks here."""
    import torch
    n_reads = 6000
    bases, _ = O.synth_reads(SEED, 1 << 16, 150, 0, n_reads, with_qual=False)
    m = O.OracleMap()
    m.scan_flat(bases, k, nthreads=NCPU)
    want = m.as_dict()
    n = 1 << (2 * k)
    total = torch.zeros(n, dtype=torch.int64, device="cuda")
    per = n_reads // nshards
    for s_ in range(nshards):
        lo, hi = s_ * per, (n_reads if s_ == nshards - 1 else (s_ + 1) * per)
        with K.DeviceCounter(k) as dc:
            dc.push(bases[lo * 151: hi * 151])
            dc.finish()
            arr = torch.full((n,), -1, dtype=torch.int64, device="cuda")   # the export must zero what it does not set
            torch.cuda.synchronize()  # (the context has its own stream: torch's fill must be done first)
            dc.export_dense_device(arr.data_ptr(), n)
            d = dc.as_dict()
        host = arr.cpu().numpy()
        nz = np.flatnonzero(host)
        assert len(nz) == len(d) and all(d.get(int(i)) == int(host[i]) for i in nz)
        total += arr
    torch.cuda.synchronize()
    merged = {}
    for o in range(nshards):
        with K.DeviceCounter(k) as dc:
            dc.merge_dense_device(total.data_ptr(), n, o, nshards)
            dc.finish()
            d = dc.as_dict()
        assert all(K.owner(key, k, nshards) == o for key in d)
        assert not (set(d) & set(merged))
        merged.update(d)
    assert merged == want
    with K.DeviceCounter(14) as dc:
        with pytest.raises(K.native.KmerHipError) as e:
            dc.export_dense_device(total.data_ptr(), 1 << 28)
        assert e.value.status == K.native.KH_ERR_RANGE


def test_shard_table_rejects_reads_until_reset(K):
    """A shard table holds only keys of its hash range: reads cannot be pushed into it (state
    error, nothing counted); kh_reset turns it back into a full table."""
    bases, _ = O.synth_reads(SEED, 1 << 16, 150, 0, 5_000, with_qual=False)
    m = O.OracleMap()
    m.scan_flat(bases, 21, nthreads=NCPU)
    with K.DeviceCounter(21) as dc:
        dc.set_shard(3, 4)
        with pytest.raises(K.KmerHipError) as e:
            dc.push(bases)
        assert e.value.status == -7
        dc.reset()
        dc.push(bases)
        dc.finish()
        assert dc.as_dict() == m.as_dict()
        with pytest.raises(K.KmerHipError):   # not empty any more
            dc.set_shard(0, 2)
        for bad in ((0, 3), (4, 4), (0, 0)):
            dc.reset()
            with pytest.raises(K.KmerHipError):
                dc.set_shard(*bad)


def test_merge_pairs_host(K):
    with K.DeviceCounter(9) as dc:
        dc.merge_pairs(np.array([1, 2, 3, 2], dtype=np.uint64), np.array([10, 20, 30, 5], dtype=np.uint64))
        dc.push(b"AAAAAAAAA\n")  # key 0
        dc.finish()
        assert dc.as_dict() == {0: 1, 1: 10, 2: 25, 3: 30}


# ---------------------------------------------------------------------------
# BASELINE.json configs[1] at full size: size-independent properties + sampled parity
# ---------------------------------------------------------------------------

@pytest.fixture(scope="module")
def big_reads(K):
    """The metric's 100 M reads -- bases AND qualities -- generated ONCE for the module (round 5: every full-size test made
    its own 15-30 GB, copied them to the host and had the oracle scan them; the GPU suite took 783 s of the driver's 1200).
    The 10 M-read configurations are the first 10 M reads of the same set (the generator is counter based: reads [0, n) are a
    prefix), the bases are the same with and without qualities.  `sample(n, k, minq)`: the oracle's k-mer total and exact
    counts of the 1/1024 key sample of the first n reads, computed once per (n, k, minq)."""
    import types
    import torch
    n, rl = 100_000_000, 150
    nbytes = n * (rl + 1)
    tb = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    tq = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    K.synth_reads_device(tb.data_ptr(), tq.data_ptr(), SEED, 1 << 27, rl, 0, n)
    torch.cuda.synchronize()
    ob, oq = O.synth_reads(SEED, 1 << 27, rl, n - 500, 500)
    assert np.array_equal(tb[-500 * (rl + 1):].cpu().numpy(), ob) and np.array_equal(tq[-500 * (rl + 1):].cpu().numpy(), oq)  # the device generator = the oracle's
    host = {"b": None, "q": None}
    cache = {}

    def hb():
        if host["b"] is None:
            host["b"] = tb.cpu().numpy()
        return host["b"]

    def hq():
        if host["q"] is None:
            host["q"] = tq.cpu().numpy()
        return host["q"]

    def sample(n_reads, k, minq):
        key = (n_reads, k, minq)
        if key not in cache:
            m = O.OracleMap()
            nb = n_reads * (rl + 1)
            total = m.scan_flat(hb()[:nb], k, qual=hq()[:nb] if minq is not None else None, min_quality=minq, sample_mask=1023, nthreads=NCPU)
            cache[key] = (total,) + m.arrays()
        return cache[key]

    yield types.SimpleNamespace(tb=tb, tq=tq, hb=hb, hq=hq, sample=sample, rl=rl, n=n)
    del tb, tq
    host.clear()
    cache.clear()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n_reads", [10_000_000, 100_000_000], ids=["10M", "100M"])
def test_full_size_reads_k21(K, path, big_reads, n_reads):
    """BASELINE configs[1] (10 M x 150 bp) and the size the headline metric is quoted on (100 M x 150 bp),
    through size-independent properties plus exact counts on a 1/1024 key sample."""
    if n_reads > 10_000_000 and path == "direct":
        pytest.skip("full size through the partitioned path only (the direct path is covered at 10 M)")
    rl, k = 150, 21
    nbytes = n_reads * (rl + 1)
    tb = big_reads.tb
    total, skeys, scnts = big_reads.sample(n_reads, k, None)  # exact counts on 1/1024 of the keys
    with K.DeviceCounter(k, capacity_hint=int(1.4e8 + 12.0 * n_reads), path=path) as dc:
        dc.push_device(tb.data_ptr(), None, nbytes)
        st = dc.finish()
        assert st["kmers"] == total                                  # every valid window counted once
        assert dc.result_size() == st["distinct"]
        hist = dc.histogram()
        assert sum(f for _, f in hist) == st["distinct"]             # checksum of checksums
        assert sum(c * f for c, f in hist) == st["kmers"]
        assert [c for c, _ in hist] == sorted(c for c, _ in hist)    # ascending (BTreeMap order)
        assert np.array_equal(dc.lookup(skeys), scnts)               # sampled keys: exact counts
        keys, cnts = dc.result(sort=False)
        assert int(cnts.sum()) == total and keys.size == st["distinct"]
        sel = (np.array([O.mix64(int(x)) for x in keys[:200000]], dtype=np.uint64) & np.uint64(1023)) == 0
        assert set(keys[:200000][sel].tolist()) <= set(skeys.tolist())
        del keys, cnts
        # idempotence: counting the same input again doubles every count, adds no key
        dc.push_device(tb.data_ptr(), None, nbytes)
        st2 = dc.finish()
        assert st2["distinct"] == st["distinct"] and st2["kmers"] == 2 * total
        assert np.array_equal(dc.lookup(skeys), 2 * scnts)


@pytest.mark.parametrize("n_reads", [10_000_000, 100_000_000], ids=["10M", "100M"])
def test_full_size_reads_k31_q20(K, path, big_reads, n_reads):
    """BASELINE configs[2] (k = 31, N bases + --min-quality 20 masking; 64-bit payload path) at 10 M and
    at the full 100 M x 150 bp: total, histogram checksums, exact counts on a 1/1024 key sample."""
    if n_reads > 10_000_000 and path == "direct":
        pytest.skip("full size through the partitioned path only (the direct path is covered at 10 M)")
    rl, k, minq = 150, 31, 20
    nbytes = n_reads * (rl + 1)
    tb, tq = big_reads.tb, big_reads.tq
    total, skeys, scnts = big_reads.sample(n_reads, k, minq)
    assert 0 < total < n_reads * (rl - k + 1)                        # the masks did remove windows
    hint = int(1.4e8 + 53.0 * n_reads)  # genome k-mers + ~53 error / boundary k-mers per read at k = 31
    with K.DeviceCounter(k, min_quality=minq, capacity_hint=hint, path=path) as dc:
        dc.push_device(tb.data_ptr(), tq.data_ptr(), nbytes)
        st = dc.finish()
        assert st["kmers"] == total
        hist = dc.histogram()
        assert sum(f for _, f in hist) == st["distinct"] == dc.result_size()
        assert sum(c * f for c, f in hist) == total
        assert np.array_equal(dc.lookup(skeys), scnts)
    if n_reads > 10_000_000:
        return
    # without the quality buffer the same context counts more windows (run.rs:543: both must be Some)
    with K.DeviceCounter(k, min_quality=minq, capacity_hint=hint, path=path) as dc:
        dc.push_device(tb.data_ptr(), None, nbytes)
        assert dc.finish()["kmers"] > total


def _np_mix64(z):
    """oracle ko_mix64 (splitmix64 finaliser), vectorised with wrap-around."""
    z = z.astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


@pytest.mark.parametrize("k,minq", [(21, None), (31, 20)], ids=["k21", "k31-q20"])
def test_full_map_digest_10M_reads(K, big_reads, k, minq):
    """BASELINE configs[1] with the WHOLE map compared, not a key sample: the CPU counts every k-mer of
    the 10 M reads (oracle radix formulation) and the order-independent digest sum(mix(key ^ mix(count)))
    of all ~235 M (key, count) pairs must equal the digest of the pairs copied back from the device,
    together with the distinct count and the total."""
    n_reads, rl = 10_000_000, 150
    nbytes = n_reads * (rl + 1)
    tb, tq = big_reads.tb, (big_reads.tq if minq is not None else None)
    # (the oracle's count of the same 10 M reads -- from its own generator, ko_count_flat_radix_mt -- has been running in a child
    #  process since the session started: tests/bg_oracle.py; a lone run of this test computes it here)
    import bg_oracle
    res = bg_oracle.collect("digest_k21" if minq is None else "digest_k31q20", NCPU)
    total, distinct, digest = res["total"], res["distinct"], res["digest"]
    assert _np_mix64(np.array([12345], dtype=np.uint64))[0] == O.mix64(12345)  # the vectorised mix is the oracle's
    with K.DeviceCounter(k, min_quality=minq, capacity_hint=int(1.4e8 + 12.0 * n_reads), path="partition") as dc:
        dc.push_device(tb.data_ptr(), tq.data_ptr() if tq is not None else None, nbytes)
        st = dc.finish()
        keys, cnts = dc.result(sort=False)
    assert st["kmers"] == total and st["distinct"] == distinct == keys.size
    with np.errstate(over="ignore"):
        got = int(_np_mix64(keys ^ _np_mix64(cnts)).sum(dtype=np.uint64))
    assert got == digest


@pytest.mark.parametrize("mode", ["units", "stand-down", "unaligned"])
@pytest.mark.parametrize("k,hint", [(21, 3_000_000), (21, 40_000_000), (19, 8_000_000), (15, 2_000_000), (21, 0)],
                         ids=["k21-2buckets", "k21-64buckets", "k19", "k15", "k21-unhinted"])
def test_level2_unit_writer_and_its_stand_down(K, monkeypatch, mode, k, hint):
    """The EXACT level 2 (count -> scan -> scatter; what the arena path falls back to) with 32-bit payloads writes whole aligned 64-byte units (segments padded with sentinels that the
    region pass skips, tails carried in LDS); a partition too large for its 32-bit offsets makes it stand down for
    the batch in favour of the unaligned kernel (forced here by KMERHIP_P2_FORCE_WIDE: the real condition needs
    > 4 G k-mers with one level-1 digit); KMERHIP_P2_LINES=0 is the unaligned kernel alone.  All three must give
    the oracle's map, over several batches into one table (non-fresh region passes see the sentinels too)."""
    monkeypatch.setenv("KMERHIP_L2_ARENA", "0")  # (the exact level 2: the arena path has its own test below)
    if mode == "stand-down":
        monkeypatch.setenv("KMERHIP_P2_FORCE_WIDE", "1")
    elif mode == "unaligned":
        monkeypatch.setenv("KMERHIP_P2_LINES", "0")
    n_reads = 70_000
    bases, _ = O.synth_reads(SEED, 1 << 19, 150, 0, n_reads, with_qual=False)
    m = O.OracleMap()
    m.scan_flat(bases, k, nthreads=NCPU)
    want_k, want_c = m.arrays()
    import torch
    tb = torch.from_numpy(bases).cuda()
    torch.cuda.synchronize()
    with K.DeviceCounter(k, capacity_hint=hint, path="partition") as dc:
        cut = [0, 20_000 * 151, 20_001 * 151, 45_000 * 151, n_reads * 151]  # batches of very different sizes
        for a, b in zip(cut, cut[1:]):
            dc.push_device(tb.data_ptr() + a, None, b - a)  # (every push_device is a batch of its own)
        st = dc.finish()
        assert st["kmers"] == m.total() and st["part_batches"] >= 3
        keys, cnts = dc.result()
        assert np.array_equal(keys, want_k) and np.array_equal(cnts, want_c)


@pytest.mark.parametrize("minq", [None, 20], ids=["noqual", "q20"])
@pytest.mark.parametrize("k,hint,generic", [(21, 6_000_000, False), (21, 6_000_000, True), (21, 40_000_000, False),
                                            (20, 6_000_000, False), (17, 3_000_000, False),
                                            (22, 6_000_000, False), (25, 6_000_000, False), (31, 6_000_000, False), (31, 6_000_000, True),
                                            (32, 6_000_000, False), (31, 40_000_000, False), (19, 6_000_000, "pay64")],
                         ids=["k21-written-out", "k21-c++window", "k21-64buckets", "k20", "k17",
                              "k22-bins64", "k25-bins64", "k31-bins64", "k31-bins64-generic", "k32-bins64", "k31-bins64-32buckets", "k19-forced-64bit-payloads"])
def test_level1_bins_kernel_overflow_and_masks(K, monkeypatch, k, hint, generic, minq):
    """Level 1: per-partition bins in LDS, flushed in whole 64-byte segments -- with 32-bit payloads twice per tile
    (part1_bins_kernel: the hand-written k = 21 window and the C++ one), with 64-bit payloads (k >= 22,
    part1_bins64_kernel: 16-payload bins) every two windows, with -Q every four.  The input is made to hit everything the
    kernel treats specially: 12 % of the reads are homopolymers / dinucleotide repeats (thousands of payloads of one
    tile for ONE bin: the overflow path, second ranks, descriptors in the emptied bin), N runs and lower case
    (windows without a key: the waste counters), reads of every length mod 16 (tile seams), and quality masking.
    Several batches of very different sizes into one table; against the oracle."""
    if generic == "pay64":
        monkeypatch.setenv("KMERHIP_PAYLOAD", "64")   # (read at kh_create: the 24-bit-multiplier instance of the 64-bit kernel)
    elif generic:
        monkeypatch.setenv("KMERHIP_GENERIC_K", "1")
    rng = np.random.default_rng(4242 + k)
    n_reads = 60_000
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=1 << 18)
    recs, quals = [], []
    for i in range(n_reads):
        n = int(rng.integers(30, 260))
        u = rng.random()
        if u < 0.05:
            s = np.full(n, ord("A") if i % 2 else ord("T"), dtype=np.uint8)
        elif u < 0.09:
            s = np.resize(np.frombuffer(b"AC" if i % 2 else b"GT", dtype=np.uint8), n).copy()
        elif u < 0.12:
            s = np.resize(np.frombuffer(b"ACGTTGCA", dtype=np.uint8), n).copy()
        else:
            o = int(rng.integers(0, genome.size - n))
            s = genome[o:o + n].copy()
            if u > 0.9:
                a = int(rng.integers(0, n))
                s[a:a + int(rng.integers(1, 6))] = ord("N")
            if u > 0.97:
                s = np.frombuffer(s.tobytes().lower(), dtype=np.uint8).copy()
        recs.append(s.tobytes())
        quals.append(rng.choice(np.frombuffer(b"#+5?I", dtype=np.uint8), size=n).astype(np.uint8).tobytes())
    flat = b"\n".join(recs) + b"\n"
    qflat = b"\n".join(quals) + b"\n"
    bases = np.frombuffer(flat, dtype=np.uint8)
    qual = np.frombuffer(qflat, dtype=np.uint8)
    m = O.OracleMap()
    m.scan_flat(bases, k, qual=qual if minq is not None else None, min_quality=minq, nthreads=NCPU)
    want_k, want_c = m.arrays()
    import torch
    tb = torch.from_numpy(bases.copy()).cuda()
    tq = torch.from_numpy(qual.copy()).cuda() if minq is not None else None
    torch.cuda.synchronize()
    ends = np.flatnonzero(bases == 10) + 1  # record ends: batches are cut at record boundaries
    cut = [0, int(ends[999]), int(ends[1000]), int(ends[30_000]), bases.size]
    with K.DeviceCounter(k, min_quality=minq, capacity_hint=hint, path="partition") as dc:
        for a, b in zip(cut, cut[1:]):
            dc.push_device(tb.data_ptr() + a, tq.data_ptr() + a if tq is not None else None, b - a)
        st = dc.finish()
        assert st["kmers"] == m.total() and st["part_batches"] >= 3
        keys, cnts = dc.result()
        assert np.array_equal(keys, want_k) and np.array_equal(cnts, want_c)


@pytest.mark.parametrize("mode", ["arena", "exact", "list-full", "no-skew-limit", "unaligned-exact"])
@pytest.mark.parametrize("k,skewed", [(21, False), (21, True), (31, True), (19, False), (21, "many"), (21, "light"), (31, "light")],
                         ids=["k21", "k21-skewed", "k31-skewed", "k19", "k21-many-heavy", "k21-few-heavy-partitions", "k31-few-heavy-partitions"])
def test_level2_arena_path_and_its_fallbacks(K, monkeypatch, mode, k, skewed):
    """Level 2 without a counting pass (part2_arena_kernel: per-bucket arenas sized from the level-1 partition totals,
    one workgroup per partition, what does not fit goes to an overflow list inserted through the direct path after the
    region pass) against the oracle, and every way it can step aside: switched off (`exact`), the overflow list
    declared full after a few entries (`list-full`: the batch is redone through the exact path), no limit on how uneven
    the partitions may be (`no-skew-limit`: on skewed input the heavy buckets then really go through the list), and the
    exact path with the unaligned scatter.  Skewed input: a fifth of the reads are copies of four short repeats;
    `many-heavy`: twenty reads 2000 times each, i.e. ~2600 k-mers each heavier than a bucket's arena slack, spread over
    nearly all of the 1024 partitions -- every workgroup takes a private 8192-entry segment of the overflow list, so the
    list's cursor ends far beyond its capacity without the list being full (the insert kernel must stop at the capacity).
    The table is sized (hint 50 M -> 2^15 regions) so that the geometry is the arena path's: 1024 partitions x 32 buckets;
    which path a batch took shows in the stage times (the arena path has no counting pass)."""
    if mode == "exact":
        monkeypatch.setenv("KMERHIP_L2_ARENA", "0")
    elif mode == "list-full":
        monkeypatch.setenv("KMERHIP_L2_OVF_CAP", "64")
        monkeypatch.setenv("KMERHIP_L2_SKEW_X", "0")
    elif mode == "no-skew-limit":
        monkeypatch.setenv("KMERHIP_L2_SKEW_X", "0")
    elif mode == "unaligned-exact":
        monkeypatch.setenv("KMERHIP_L2_ARENA", "0")
        monkeypatch.setenv("KMERHIP_P2_LINES", "0")
    rng = np.random.default_rng(900 + k)
    n_reads = 80_000
    bases, _ = O.synth_reads(SEED + k, 1 << 19, 150, 0, n_reads, with_qual=False)
    bases = bases.copy()
    if skewed == "many":
        v = bases.reshape(n_reads, 151)
        heavy = v[:20, :150].copy()                      # twenty reads, 2000 copies of each
        for j, i in enumerate(rng.choice(np.arange(20, n_reads), size=40_000, replace=False)):
            v[i, :150] = heavy[j % 20]
    elif skewed:
        # a fifth of the reads are the four repeats: their level-1 partitions hold more than the room behind the arenas, the
        # batch takes the exact path as a whole.  "light": a twentieth -- the few heavy partitions go through the exact
        # kernels, all the others through the arena kernel, in one batch (round 3).
        v = bases.reshape(n_reads, 151)
        reps = [np.resize(np.frombuffer(r, dtype=np.uint8), 150) for r in (b"A", b"AC", b"ACGTTGCA", b"GATTACA")]
        for i in rng.choice(n_reads, size=n_reads // (20 if skewed == "light" else 5), replace=False):
            v[i, :150] = reps[i % 4]
    m = O.OracleMap()
    m.scan_flat(bases, k, nthreads=NCPU)
    want_k, want_c = m.arrays()
    import torch
    tb = torch.from_numpy(bases).cuda()
    torch.cuda.synchronize()
    with K.DeviceCounter(k, capacity_hint=50_000_000, path="partition") as dc:
        cut = [0, 30_000 * 151, 30_001 * 151, n_reads * 151]
        for a, b in zip(cut, cut[1:]):
            dc.push_device(tb.data_ptr() + a, None, b - a)
        st = dc.finish()
        assert st["kmers"] == m.total() and st["table_slots"] == 1 << 27
        counted = st["stage_ms"]["level2_count"] > 0          # some batch went through count -> scan -> scatter
        if mode in ("exact", "unaligned-exact", "list-full"):
            assert counted or mode == "list-full" and not skewed
        elif not skewed or mode == "no-skew-limit" or skewed in ("many", "light"):
            # (many-heavy: nearly every workgroup reserves a segment of the overflow list -- the list is sized for that)
            assert not counted, "the arena path stepped aside where it should not have"
        keys, cnts = dc.result()
        assert np.array_equal(keys, want_k) and np.array_equal(cnts, want_c)


_WINDOW_READS = {}


def _window_reads():
    """24,000 reads with N, lower case and low qualities, lengths of every residue mod 16, both strands -- built once for the
    46 cases of the test below (a Python loop per record: it was most of each case's run time)."""
    if not _WINDOW_READS:
        rng = np.random.default_rng(7700)
        n_reads = 24_000
        genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=1 << 17)
        recs, quals = [], []
        for i in range(n_reads):
            n = int(rng.integers(20, 240))
            o = int(rng.integers(0, genome.size - n))
            s = genome[o:o + n].copy()
            u = rng.random()
            if u > 0.9:
                a = int(rng.integers(0, n))
                s[a:a + int(rng.integers(1, 4))] = ord("N")
            if u < 0.05:
                s = np.frombuffer(s.tobytes().lower(), dtype=np.uint8).copy()
            if i % 2:
                s = np.frombuffer(s.tobytes().translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))[::-1], dtype=np.uint8).copy()  # both strands
            recs.append(s.tobytes())
            quals.append(rng.choice(np.frombuffer(b"#5IIIIII", dtype=np.uint8), size=n).astype(np.uint8).tobytes())
        _WINDOW_READS["bases"] = np.frombuffer(b"\n".join(recs) + b"\n", dtype=np.uint8)
        _WINDOW_READS["qual"] = np.frombuffer(b"\n".join(quals) + b"\n", dtype=np.uint8)
    return _WINDOW_READS["bases"], _WINDOW_READS["qual"]


@pytest.mark.parametrize("minq", [None, 20], ids=["noqual", "q20"])
@pytest.mark.parametrize("k", list(range(10, 33)))
def test_written_out_window_every_k(K, k, minq):
    """The level-1 window is generated for every k = 11..32 (krust_amd/csrc/window.hip.h: one kernel per k; 4-byte
    payloads up to k = 21, the 8-byte key above; k = 10 takes the C++ window and rides along).  Reads with N, lower
    case and low qualities, lengths of every residue mod 16, three batches of different sizes through the partitioned
    path into a table of 2^12 regions (1024 level-1 partitions: the geometry the written-out window is for); the whole
    map against the oracle.  (The reference serves k = 1..32 uniformly, src/kmer.rs:100-110.)"""
    bases, qual = _window_reads()
    m = O.OracleMap()
    m.scan_flat(bases, k, qual=qual if minq is not None else None, min_quality=minq, nthreads=NCPU)
    want_k, want_c = m.arrays()
    import torch
    tb = torch.from_numpy(bases.copy()).cuda()
    tq = torch.from_numpy(qual.copy()).cuda() if minq is not None else None
    torch.cuda.synchronize()
    ends = np.flatnonzero(bases == 10) + 1
    cut = [0, int(ends[499]), int(ends[9_000]), bases.size]
    with K.DeviceCounter(k, min_quality=minq, capacity_hint=6_000_000, path="partition") as dc:
        for a, b in zip(cut, cut[1:]):
            dc.push_device(tb.data_ptr() + a, tq.data_ptr() + a if tq is not None else None, b - a)
        st = dc.finish()
        assert st["kmers"] == m.total() and st["part_batches"] == 3 and st["table_slots"] == 1 << 24
        keys, cnts = dc.result()
        assert np.array_equal(keys, want_k) and np.array_equal(cnts, want_c)


@pytest.mark.parametrize("cut", ["100000", "2500", "0"], ids=["heavy-keys-only", "ordinary-buckets-too", "off"])
@pytest.mark.parametrize("k", [21, 31])
def test_hot_buckets_are_counted_apart_from_the_region_pass(K, monkeypatch, k, cut):
    """A bucket dominated by one key (poly-A: 10 % of the S100M-shaped reads put 65 M copies into one bucket) would keep one
    workgroup of the region pass busy for twice as long as the whole pass: buckets above a threshold are skipped by the
    pass and counted by hot_buckets_kernel instead (slices over the whole grid, LDS sums, a few device atomics each).
    Here with the threshold lowered (KMERHIP_HOT_CUT) so that the repeats' buckets are hot -- and, with 2500, a good part
    of the ordinary buckets as well (mean 2900 payloads: the LDS table is applied and refilled many times) -- on a fresh
    table and on a filled one (the skipped region must stay as it is), 4- and 8-byte payloads, the map against the
    oracle after every batch, lookups and histogram from the widened table."""
    monkeypatch.setenv("KMERHIP_HOT_CUT", cut)
    rng = np.random.default_rng(5100 + k)
    n_reads = 80_000
    bases, _ = O.synth_reads(SEED + 3 * k, 1 << 19, 150, 0, n_reads, with_qual=False)
    bases = bases.copy()
    v = bases.reshape(n_reads, 151)
    reps = [np.resize(np.frombuffer(r, dtype=np.uint8), 150) for r in (b"A", b"AC", b"ACGTTGCA", b"GATTACA")]
    for i in rng.choice(n_reads, size=n_reads // 8, replace=False):
        v[i, :150] = reps[i % 4]
    import torch
    tb = torch.from_numpy(bases).cuda()
    torch.cuda.synchronize()
    half = (n_reads // 2) * 151
    m = O.OracleMap()
    with K.DeviceCounter(k, capacity_hint=6_000_000, path="partition") as dc:     # 2^12 regions
        for a, b in ((0, half), (half, bases.size)):
            m.scan_flat(bases[a:b], k, nthreads=NCPU)
            dc.push_device(tb.data_ptr() + a, None, b - a)
            st = dc.finish()
            assert st["kmers"] == m.total() and st["distinct"] == len(m), (a, st)
            want_k, want_c = m.arrays()
            keys, cnts = dc.result()
            assert np.array_equal(keys, want_k) and np.array_equal(cnts, want_c), a
        poly_a = 0                                                               # AAAA...A, the heaviest key
        assert int(dc.lookup(np.array([poly_a], dtype=np.uint64))[0]) == int(want_c[np.searchsorted(want_k, poly_a)])
        assert dc.histogram() == m.histogram()


@pytest.mark.parametrize("k", [31, 21])
def test_quality_masked_range_is_sized_from_its_survival_rate(K, monkeypatch, k):
    """A range pushed with qualities and --min-quality: the partition buffers are sized from a sample of the windows that
    survive masking (survival_sample_kernel), not from "every window", so BASELINE configs[2] (k = 31, -Q 20, ~0.4 of the
    windows left) is one batch instead of two.  Here: a budget that holds 0.7 of the worst case -- one batch with the
    sample, two with KMERHIP_SURVIVAL=1; an estimate that is far too small (0.02) makes level 1 run out of pool, which is
    noticed before anything but the pool was written, and the tiles run again at full size.  Same map every time."""
    rng = np.random.default_rng(4242 + k)
    n_reads, rl = 60_000, 150
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=1 << 20)
    offs = rng.integers(0, genome.size - rl, size=n_reads)
    bases = np.empty(n_reads * (rl + 1), dtype=np.uint8)
    bases.reshape(n_reads, rl + 1)[:, :rl] = genome[offs[:, None] + np.arange(rl)[None, :]]
    bases.reshape(n_reads, rl + 1)[:, rl] = 10
    # every read is good ('I') up to a cut point and bad ('#') behind it: a little under half of the windows survive -Q 20
    cutp = rng.integers(0, 200, size=n_reads)
    q = np.where(np.arange(rl + 1)[None, :] < cutp[:, None], ord("I"), ord("#")).astype(np.uint8)
    q[:, rl] = 10
    qual = q.reshape(-1)
    m = O.OracleMap()
    m.scan_flat(bases, k, qual=qual, min_quality=20, nthreads=NCPU)
    want_k, want_c = m.arrays()
    windows = bases.size
    share = m.total() / windows
    assert 0.1 < share < 0.6, share
    import torch
    tb, tq = torch.from_numpy(bases).cuda(), torch.from_numpy(qual.copy()).cuda()
    torch.cuda.synchronize()
    per_key = 20 if k > 21 else 11
    monkeypatch.setenv("KMERHIP_PART_BUDGET_GB", repr(0.7 * per_key * windows / 2**30))
    for survival, batches in ((None, 1), ("1", 2), ("0.02", None), ("0.9", 2)):
        if survival is None:
            monkeypatch.delenv("KMERHIP_SURVIVAL", raising=False)
        else:
            monkeypatch.setenv("KMERHIP_SURVIVAL", survival)
        with K.DeviceCounter(k, min_quality=20, capacity_hint=6_000_000, path="partition") as dc:   # 2^12 regions: 4-byte payloads at k = 21
            dc.push_device(tb.data_ptr(), tq.data_ptr(), bases.size)
            st = dc.finish()
            assert st["kmers"] == m.total()
            if batches is not None:
                assert st["part_batches"] == batches, (survival, st["part_batches"])
            keys, cnts = dc.result()
            assert np.array_equal(keys, want_k) and np.array_equal(cnts, want_c), survival
        # without qualities (or without a threshold) nothing is sampled and nothing changes
    monkeypatch.delenv("KMERHIP_SURVIVAL", raising=False)
    m2 = O.OracleMap()
    m2.scan_flat(bases, k, nthreads=NCPU)
    with K.DeviceCounter(k, capacity_hint=6_000_000, path="partition") as dc:
        dc.push_device(tb.data_ptr(), tq.data_ptr(), bases.size)
        st = dc.finish()
        assert st["kmers"] == m2.total() and st["part_batches"] == 2


@pytest.mark.parametrize("narrow", ["1", "0"], ids=["narrow-image", "wide-only"])
@pytest.mark.parametrize("k", [21, 17, 13])
def test_narrow_table_image_and_its_transitions(K, monkeypatch, k, narrow):
    """Round 3: a partitioned pass with 32-bit payloads keeps the table as an 8-byte image (count << 32 | payload) and only
    widens it to 16-byte {key, count} slots when something needs those.  One table through every transition: fresh narrow
    pass -> pass over the narrow image -> results / histogram / lookups / min-count straight from the image -> a small push
    (direct path: widens) -> another partitioned pass (stays wide) -> reset -> narrow again.  Against the oracle at every
    step, and with KMERHIP_NARROW=0 (never narrow) for the same answers."""
    monkeypatch.setenv("KMERHIP_NARROW", narrow)
    n_reads = 90_000
    bases, _ = O.synth_reads(SEED + 3 * k, 1 << 19, 150, 0, n_reads, with_qual=False)
    import torch
    tb = torch.from_numpy(bases.copy()).cuda()
    torch.cuda.synchronize()

    def oracle(upto):
        m = O.OracleMap()
        m.scan_flat(bases[: upto * 151], k, nthreads=NCPU)
        return m

    def check(dc, m):
        wk, wc = m.arrays()
        st = dc.finish()
        assert st["kmers"] == m.total() and st["distinct"] == len(m)
        keys, cnts = dc.result()
        assert np.array_equal(keys, wk) and np.array_equal(cnts, wc)
        sel = wc >= 3
        k3, c3 = dc.result(min_count=3)
        assert np.array_equal(k3, wk[sel]) and np.array_equal(c3, wc[sel]) and dc.result_size(3) == int(sel.sum())
        hist = dc.histogram()
        u, f = np.unique(wc, return_counts=True)
        assert hist == list(zip(u.tolist(), f.tolist()))
        probe = np.concatenate([wk[:: max(1, wk.size // 5000)], np.array([0, 1, (1 << (2 * k)) - 1], dtype=np.uint64)])
        want = np.array([m.get(int(x)) for x in probe[-3:]], dtype=np.uint64)   # (the oracle map's own lookup: as_dict() built millions of entries for three)
        got = dc.lookup(probe)
        assert np.array_equal(got[:-3], wc[:: max(1, wk.size // 5000)]) and np.array_equal(got[-3:], want)
        # a key with bits beyond 2k is no k-mer of this k: absent, not an alias of the k-mer in its low bits
        assert not dc.lookup(wk[:100] | np.uint64(1 << 63)).any() and not dc.lookup(wk[:100] | np.uint64(1 << (2 * k))).any()

    with K.DeviceCounter(k, capacity_hint=6_000_000, path="partition") as dc:
        dc.push_device(tb.data_ptr(), None, 40_000 * 151)                      # fresh
        check(dc, oracle(40_000))
        dc.push_device(tb.data_ptr() + 40_000 * 151, None, 30_000 * 151)       # over the image
        check(dc, oracle(70_000))
    with K.DeviceCounter(k, capacity_hint=6_000_000) as dc:                    # path chosen per push
        dc.push_device(tb.data_ptr(), None, 60_000 * 151)                      # partitioned (fresh)
        dc.push_device(tb.data_ptr() + 60_000 * 151, None, 2_000 * 151)        # small: the direct path -- the image is widened
        check(dc, oracle(62_000))
        dc.push_device(tb.data_ptr() + 62_000 * 151, None, 28_000 * 151)       # partitioned over the 16-byte table
        check(dc, oracle(90_000))
        dc.reset()
        dc.push_device(tb.data_ptr(), None, 40_000 * 151)                      # fresh again after the reset
        check(dc, oracle(40_000))


@pytest.mark.parametrize("hot", ["hot-buckets", "no-hot-buckets"])
def test_count_beyond_32_bits(K, monkeypatch, hot):
    """a count of 2^32 and more (poly-A pushed batch after batch).  The 8-byte table image keeps 32-bit
    counts: the region pass that would take a count past them fails that region (code 2), the host widens the table to
    16-byte slots, re-inserts the bucket and stays wide.  u64 counts are the reference's range (src/run.rs:569).
    Since the hot-bucket kernel a bucket of 208 M copies no longer goes through the region pass at all (the first such
    batch widens the table): `no-hot-buckets` switches that off to keep the image's own overflow path under test."""
    if hot == "no-hot-buckets":
        monkeypatch.setenv("KMERHIP_HOT_CUT", "0")
    k = 21
    rng = np.random.default_rng(5)
    n_a = 1_600_000
    poly = np.full((n_a, 151), ord("A"), dtype=np.uint8)
    poly[:, 150] = 10
    other, _ = O.synth_reads(SEED, 1 << 18, 150, 0, 50_000, with_qual=False)
    m = O.OracleMap()
    m.scan_flat(other, k, nthreads=NCPU)
    wk, wc = m.arrays()
    import torch
    ta = torch.from_numpy(poly.reshape(-1)).cuda()
    to = torch.from_numpy(other.copy()).cuda()
    torch.cuda.synchronize()
    per_push = n_a * 130
    pushes = (1 << 32) // per_push + 2
    with K.DeviceCounter(k, capacity_hint=6_000_000, path="partition") as dc:
        dc.push_device(to.data_ptr(), None, to.numel())
        for _ in range(pushes):
            dc.push_device(ta.data_ptr(), None, ta.numel())
        st = dc.finish()
        want0 = pushes * per_push + m.get(0)
        assert want0 > 1 << 32
        assert st["kmers"] == m.total() + pushes * per_push
        assert int(dc.lookup(np.array([0], dtype=np.uint64))[0]) == want0
        keys, cnts = dc.result()
        exp = dict(m.as_dict())
        exp[0] = want0
        assert dict(zip(keys.tolist(), cnts.tolist())) == exp
        assert dc.histogram()[-1] == (want0, 1)


# ---------------------------------------------------------------------------
# round 4: tables of any multiple of 1024 regions (kernels.hip.h TableGeom)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("k", [21, 31])
@pytest.mark.parametrize("b2", [800, 1024])
def test_more_than_768_buckets_per_partition_with_heavy_buckets(K, monkeypatch, b2, k):
    """Level 2's 1024-bucket instance (128-byte bins, 64-byte units, write positions in LDS): a power of two and not, both
    payload widths, with a burst of one read's copies whose buckets outgrow their bins and arenas (overflow list)."""
    monkeypatch.setenv("KMERHIP_TABLE_REGIONS", str(1024 * b2))
    n_reads = 200_000
    bases, _ = O.synth_reads(SEED + b2 + k, 1 << 21, 150, 0, n_reads, with_qual=False)
    burst = np.tile(bases[: 151], 3000)  # 3000 copies of one read: its buckets outgrow their arenas
    bases = np.concatenate([bases, burst])
    m = O.OracleMap()
    total = m.scan_flat(bases, k, nthreads=NCPU)
    want_k, want_c = m.arrays()
    import torch
    tb = torch.from_numpy(bases).cuda()
    torch.cuda.synchronize()
    with K.DeviceCounter(k, capacity_hint=len(m), path="partition") as dc:
        half = (n_reads // 2) * 151
        dc.push_device(tb.data_ptr(), None, half)
        dc.push_device(tb.data_ptr() + half, None, len(bases) - half)
        st = dc.finish()
        assert st["table_slots"] == 1024 * b2 * 4096, st
        assert st["kmers"] == total and st["distinct"] == len(m)
        keys, cnts = dc.result()
        assert np.array_equal(keys, want_k) and np.array_equal(cnts, want_c)
        assert dc.histogram() == m.histogram()


@pytest.mark.parametrize("narrow", ["1", "0"], ids=["narrows", "never"])
@pytest.mark.parametrize("k,rbits,minq", [(22, 15, None), (23, 15, 20), (24, 16, None), (25, 18, None), (26, 20, 20)])
def test_level2_narrows_8_byte_payloads_below_the_region_index(K, monkeypatch, k, rbits, minq, narrow):
    """Round 6: k >= 22 carries 8-byte payloads (the hash below the level-1 digit); in a power-of-two table of
    2^rbits regions with 2k - rbits <= 32, level 2 writes the 4 bytes below the REGION index instead and the rest of the batch is a
    32-bit batch over the virtual geometry (rbits, 1): 32-bit region kernel, 8-byte table image -- which k >= 22 never had.
    Fresh pass, a pass over the image, results / histogram / lookups / min-count from it; then a batch with a heavy level-1
    partition (10 % poly-A reads): its narrowed level 2 is run again wide and the image widened -- the same map.  With
    KMERHIP_L2_NARROW=0 the 8-byte flow of rounds 1-5, for the same answers."""
    import torch
    monkeypatch.setenv("KMERHIP_TABLE_REGIONS", str(1 << rbits))
    monkeypatch.setenv("KMERHIP_L2_NARROW", narrow)
    n_reads = 120_000
    bases, qual = O.synth_reads(SEED + k, 1 << 21, 150, 0, n_reads)
    tb, tq = torch.from_numpy(bases).cuda(), torch.from_numpy(qual).cuda()
    torch.cuda.synchronize()
    m = O.OracleMap()
    m.scan_flat(bases, k, qual=qual if minq is not None else None, min_quality=minq, nthreads=NCPU)
    half = (n_reads // 2) * 151
    with K.DeviceCounter(k, min_quality=minq, capacity_hint=len(m), path="partition") as dc:
        dc.push_device(tb.data_ptr(), tq.data_ptr() if minq is not None else None, half)                 # fresh
        dc.push_device(tb.data_ptr() + half, tq.data_ptr() + half if minq is not None else None, len(bases) - half)   # over the image
        st = dc.finish()
        assert st["table_slots"] == 4096 << rbits and st["grows"] == 0
        assert st["slot_bytes"] == (8 if narrow == "1" else 16), st
        want_k, want_c = m.arrays()
        keys, cnts = dc.result()
        assert st["kmers"] == m.total() and np.array_equal(keys, want_k) and np.array_equal(cnts, want_c)
        assert dc.histogram() == m.histogram()
        probe = np.concatenate([want_k[:: max(1, len(want_k) // 4000)], np.array([0, (1 << (2 * k)) - 1], dtype=np.uint64)])
        assert np.array_equal(dc.lookup(probe), np.array([m.get(int(x)) for x in probe], dtype=np.uint64))
        k2, c2 = dc.result(min_count=2)
        assert np.array_equal(k2, want_k[want_c >= 2]) and np.array_equal(c2, want_c[want_c >= 2])
        # a heavy level-1 partition: 12 k poly-A reads among 108 k ordinary ones -> the narrowed level 2 steps aside for this batch
        skew = bases.copy().reshape(n_reads, 151)
        skew[::10, :150] = ord("A")
        skew = skew.reshape(-1)
        ts = torch.from_numpy(skew).cuda()
        torch.cuda.synchronize()
        dc.push_device(ts.data_ptr(), tq.data_ptr() if minq is not None else None, len(skew))
        m.scan_flat(skew, k, qual=qual if minq is not None else None, min_quality=minq, nthreads=NCPU)
        st = dc.finish()
        want_k, want_c = m.arrays()
        keys, cnts = dc.result()
        assert st["kmers"] == m.total() and np.array_equal(keys, want_k) and np.array_equal(cnts, want_c)


def test_a_batch_sized_for_a_narrowing_level2_that_cannot_narrow_runs_again_in_smaller_batches(K, monkeypatch):
    """Batches of 8-byte payloads that level 2 is expected to narrow are sized at 15 bytes per window instead of 20.  Where the
    batch cannot narrow after all -- here: heavy level-1 partitions (a fifth of the reads poly-A) -- and the 8-byte level-2 output
    does not fit beside the pool (KMERHIP_L2_NO_ROOM_WIDE: as if), the same tiles run again in batches sized for it
    (batch.hip KH_RETRY_WIDE).  Same map as the oracle's, more batches than the narrowing run needs."""
    import torch
    k, n_reads = 24, 150_000
    monkeypatch.setenv("KMERHIP_TABLE_REGIONS", str(1 << 16))
    bases, _ = O.synth_reads(SEED + 5, 1 << 21, 150, 0, n_reads, with_qual=False)
    skew = bases.copy().reshape(n_reads, 151)
    skew[::5, :150] = ord("A")
    skew = skew.reshape(-1)
    m = O.OracleMap()
    m.scan_flat(skew, k, nthreads=NCPU)
    want_k, want_c = m.arrays()
    ts = torch.from_numpy(skew).cuda()
    torch.cuda.synchronize()
    got_batches = {}
    for tag, env in (("retry", "1"), ("room", None)):
        if env:
            monkeypatch.setenv("KMERHIP_L2_NO_ROOM_WIDE", env)
        else:
            monkeypatch.delenv("KMERHIP_L2_NO_ROOM_WIDE", raising=False)
        with K.DeviceCounter(k, capacity_hint=len(m), path="partition") as dc:
            dc.push_device(ts.data_ptr(), None, len(skew))
            st = dc.finish()
            keys, cnts = dc.result()
            assert st["kmers"] == m.total() and np.array_equal(keys, want_k) and np.array_equal(cnts, want_c), tag
            got_batches[tag] = st["part_batches"]
    assert got_batches["room"] == 1 and got_batches["retry"] >= 1


@pytest.mark.parametrize("k,minq", [(21, None), (19, 20), (31, None), (25, 20)], ids=["k21", "k19q20", "k31", "k25q20"])
@pytest.mark.parametrize("b2", [3, 40, 96, 520, 640, 1000])
def test_tables_of_1024_x_b2_regions(K, monkeypatch, b2, k, minq):
    """A table of 1024 x b2 regions for b2 that are NOT powers of two (KMERHIP_TABLE_REGIONS forces the geometry a large
    input would get from round_cap): region = p1 * b2 + ((x * b2) >> 32), start from the product's low word.  b2 = 3: the
    exact level 2 (count -> scan -> unit scatter); 40, 96: the arena kernel's 512-bucket form; 520, 640, 1000: its 1024-bucket
    form with the bins shared out among b2 buckets.  Both payload widths, with and without -Q.  Counted by the partitioned path in three pushes
    (fresh pass, then passes over a filled table) and, into a second table of the same geometry, by the direct path;
    then every consumer of the layout: results (through the 8-byte image where there is one), histogram, lookups, min-count,
    and growth by rehash into a larger table (a push through the direct path that doubles it)."""
    monkeypatch.setenv("KMERHIP_TABLE_REGIONS", str(1024 * b2))
    n_reads = 60_000 if b2 < 500 else 200_000
    bases, qual = O.synth_reads(SEED + b2, 1 << 21, 150, 0, n_reads)
    m = O.OracleMap()
    total = m.scan_flat(bases, k, qual=qual if minq is not None else None, min_quality=minq, nthreads=NCPU)
    want_k, want_c = m.arrays()
    import torch
    tb, tq = torch.from_numpy(bases).cuda(), torch.from_numpy(qual).cuda()
    torch.cuda.synchronize()
    for path in ("partition", "direct"):
        with K.DeviceCounter(k, min_quality=minq, capacity_hint=len(m), path=path) as dc:
            cut = [0, (n_reads // 2) * 151, (n_reads // 2 + 1) * 151, n_reads * 151]
            for a, b in zip(cut, cut[1:]):
                dc.push_device(tb.data_ptr() + a, tq.data_ptr() + a if minq is not None else None, b - a)
            st = dc.finish()
            assert st["table_slots"] == 1024 * b2 * 4096, st
            assert st["kmers"] == total and st["distinct"] == len(m)
            keys, cnts = dc.result()
            assert np.array_equal(keys, want_k) and np.array_equal(cnts, want_c), f"{path}: map differs from the oracle"
            assert dc.histogram() == m.histogram()
            probe = np.concatenate([want_k[:: max(1, len(want_k) // 5000)], np.array([0, (1 << (2 * k)) - 1 if k < 32 else 2**64 - 1], dtype=np.uint64)])
            got = dc.lookup(probe)
            exp = np.array([m.get(int(x)) for x in probe], dtype=np.uint64)
            assert np.array_equal(got, exp)
            k2, c2 = dc.result(min_count=3)
            sel = want_c >= 3
            assert np.array_equal(k2, want_k[sel]) and np.array_equal(c2, want_c[sel])
    # growth: a table of 1024 x 3 regions (12.6 M slots) that these keys outgrow -- rehash into 1024 x 6, from the direct path
    # and from the partitioned one (whose regions overflow first: the failed buckets' re-insert)
    # (both payload widths; the -Q twins of the same two would differ only in the mask, which every counting test above draws)
    if b2 == 640 and minq is None:
        bases2, qual2 = O.synth_reads(SEED + 77, 1 << 24, 150, 0, 250_000)
        m2 = O.OracleMap()
        total2 = m2.scan_flat(bases2, k, qual=qual2 if minq is not None else None, min_quality=minq, nthreads=NCPU)
        w2k, w2c = m2.arrays()
        assert len(m2) > 0.8 * 1024 * 3 * 4096
        tb2, tq2 = torch.from_numpy(bases2).cuda(), torch.from_numpy(qual2).cuda()
        monkeypatch.setenv("KMERHIP_TABLE_REGIONS", str(1024 * 3))
        monkeypatch.setenv("KMERHIP_ESTIMATE", "0")   # (the sample would size the table before the region pass: no overflow to handle)
        for path in ("direct", "partition"):
            with K.DeviceCounter(k, min_quality=minq, capacity_hint=1000, path=path) as dc:
                dc.push_device(tb2.data_ptr(), tq2.data_ptr() if minq is not None else None, tb2.numel())
                st = dc.finish()
                assert st["grows"] >= 1 and st["table_slots"] > 1024 * 3 * 4096
                assert st["kmers"] == total2 and st["distinct"] == len(m2)
                keys, cnts = dc.result()
                assert np.array_equal(keys, w2k) and np.array_equal(cnts, w2c), path


@pytest.mark.parametrize("pow2,room_mb", [("0", 1), ("1", 48), ("1", 1)], ids=["b2-steps-1MB", "pow2-tables-48MB", "pow2-tables-1MB"])
def test_sample_sized_table_that_does_not_fit_the_room(K, monkeypatch, pow2, room_mb):
    """when the table the level-1 sample asks for exceeds a third of the free memory, the size is stepped
    down -- `round_cap(0.8 x)` -- and for every power-of-two size (all tables up to 2^28 slots, and every table with
    KMERHIP_POW2_TABLE=1) that rounds straight back up: the loop never ended.  KMERHIP_TABLE_ROOM_MB (test build) plays the nearly
    full device: the batch must come back -- with a smaller table that then grows by rehash -- and with the oracle's map."""
    monkeypatch.setenv("KMERHIP_TABLE_ROOM_MB", str(room_mb))
    monkeypatch.setenv("KMERHIP_POW2_TABLE", pow2)
    bases, _ = O.synth_reads(SEED + 77, 1 << 23, 150, 0, 140_000, with_qual=False)
    m = O.OracleMap()
    m.scan_flat(bases, 21, nthreads=NCPU)
    ok, oc = m.arrays()
    with K.DeviceCounter(21, path="partition") as dc:       # no hint: the sample decides -- ~14 M keys want a table of 2^25 slots = 512 MiB
        dc.push(bases)
        st = dc.finish()
        assert st["kmers"] == m.total() and st["distinct"] == len(ok)
        keys, cnts = dc.result()
        assert np.array_equal(keys, ok) and np.array_equal(cnts, oc)
