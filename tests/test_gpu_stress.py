"""Randomised scenarios through the partitioned path, every knob that changes HOW a batch is counted drawn at random --
never WHAT comes out: reads with repeats, N, soft-masking and qualities, pushed through the three input calls of the ABI
in several pieces into a table that is hinted well, badly or not at all, with the hot-bucket threshold, the survival
estimate, the region pass's workgroup shape, the 8-byte image, the arena level 2 and the batch size forced this way or that.
The map must equal the oracle's (bit-exact), whatever route the batches took.

Run with `pytest -m gpu` on an MI355X.  Nothing here reads /root/reference."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
NCPU = max(1, min(os.cpu_count() or 1, 16))

KNOBS = {
    "KMERHIP_HOT_CUT": [None, None, "0", "3000", "60000"],
    "KMERHIP_SURVIVAL": [None, None, "0.03", "1", "0.6"],
    "KMERHIP_REGION_NT": [None, "512", "1024"],
    "KMERHIP_NARROW": [None, None, "0"],
    "KMERHIP_L2_ARENA": [None, None, "0"],
    "KMERHIP_L2_SKEW_X": [None, None, "0", "1"],
    "KMERHIP_PART_BUDGET_GB": [None, None, "0.07", "0.2"],
    "KMERHIP_P2_LINES": [None, None, "0"],
    "KMERHIP_GENERIC_K": [None, None, None, "1"],
    "KMERHIP_OVF_AGG": [None, "1", "1", "0"],
    # round 4: the table's size comes from the level-1 sample unless told otherwise -- "0" brings back tables that are too
    # small for what they get (regions that fill up, failed buckets re-inserted, growth); and tables of 1024 x b2 regions
    "KMERHIP_ESTIMATE": [None, None, "0"],
    "KMERHIP_TABLE_REGIONS": [None, None, None, None, "3072", "10240", "40960"],
}


def scenario(seed):
    rng = np.random.default_rng(90_000 + seed)
    k = int(rng.choice([12, 13, 15, 16, 17, 19, 20, 21, 21, 22, 23, 25, 27, 31, 31, 32]))
    n_reads = int(rng.integers(25_000, 70_000))
    glen = 1 << int(rng.integers(15, 21))
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=glen)
    minq = [None, None, 10, 20, 30][int(rng.integers(0, 5))]
    skew = float(rng.choice([0.0, 0.0, 0.02, 0.1, 0.3]))
    units = [b"A", b"AC", b"AAT", b"ACGTTGCA", b"GATTACA", bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(9, 40))).astype(np.uint8))]
    recs, quals = [], []
    for i in range(n_reads):
        n = int(rng.integers(k, 251)) if rng.random() < 0.9 else int(rng.integers(1, k + 2))
        u = rng.random()
        if u < skew:
            s = np.resize(np.frombuffer(units[int(rng.integers(0, len(units)))], dtype=np.uint8), n).copy()
        else:
            o = int(rng.integers(0, glen - 251))
            s = genome[o:o + n].copy()
        v = rng.random()
        if v < 0.08:
            a = int(rng.integers(0, n))
            s[a:a + int(rng.integers(1, 5))] = ord("N")
        elif v < 0.12:
            a, b = sorted(int(x) for x in rng.integers(0, n + 1, size=2))
            s[a:b] = np.frombuffer(s[a:b].tobytes().lower(), dtype=np.uint8)
        if i & 1:
            s = np.frombuffer(s.tobytes().translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))[::-1], dtype=np.uint8).copy()
        recs.append(s.tobytes())
        if rng.random() < 0.5:     # good up to a cut point, bad behind it
            c = int(rng.integers(0, n + 1))
            q = np.where(np.arange(n) < c, ord("I"), ord("#")).astype(np.uint8)
        else:
            q = rng.choice(np.frombuffer(b"#+5?IIII", dtype=np.uint8), size=n).astype(np.uint8)
        quals.append(q.tobytes())
    cuts = sorted({0, n_reads, *[int(x) for x in rng.integers(0, n_reads, size=int(rng.integers(0, 4)))]})
    calls = [str(rng.choice(["device", "host", "text"])) for _ in cuts[1:]]
    hint = int(rng.choice([0, 0, 3_000, 6_000_000, 60_000_000]))
    path = str(rng.choice(["partition", "partition", "partition", "auto"]))
    env = {name: vals[int(rng.integers(0, len(vals)))] for name, vals in KNOBS.items()}
    return dict(k=k, minq=minq, recs=recs, quals=quals, cuts=cuts, calls=calls, hint=hint, path=path, env=env)


@pytest.mark.parametrize("seed", range(int(os.environ.get("KMERHIP_STRESS_SEEDS", "24"))))   # (more: KMERHIP_STRESS_SEEDS=300 pytest ...)
def test_random_route_same_map(seed, monkeypatch):
    import torch
    import krust_amd as K
    K.lib()
    sc = scenario(seed)
    for name, val in sc["env"].items():
        if val is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, val)
    k, minq = sc["k"], sc["minq"]
    want = O.count_records(sc["recs"], k, quals=sc["quals"] if minq is not None else None, min_quality=minq)
    wd = want.as_dict()
    what = f"seed {seed}: k={k} minq={minq} hint={sc['hint']} path={sc['path']} calls={sc['calls']} env={ {a: b for a, b in sc['env'].items() if b} }"
    keep = []
    with K.DeviceCounter(k, min_quality=minq, capacity_hint=sc["hint"], path=None if sc["path"] == "auto" else sc["path"]) as dc:
        for (a, b), call in zip(zip(sc["cuts"], sc["cuts"][1:]), sc["calls"]):
            rs, qs = sc["recs"][a:b], sc["quals"][a:b]
            if not rs:
                continue
            if call == "text":
                text = b"".join(b"@r%d\n" % i + r + b"\n+\n" + q + b"\n" for i, (r, q) in enumerate(zip(rs, qs)))
                dc.push_text(text, "fastq")
                continue
            fb = b"\n".join(rs) + b"\n"
            fq = b"\n".join(qs) + b"\n"
            if call == "host":
                dc.push(np.frombuffer(fb, dtype=np.uint8), np.frombuffer(fq, dtype=np.uint8) if minq is not None else None)
            else:
                tb = torch.frombuffer(bytearray(fb), dtype=torch.uint8).cuda()
                tq = torch.frombuffer(bytearray(fq), dtype=torch.uint8).cuda() if minq is not None else None
                keep += [tb, tq]
                dc.push_device(tb.data_ptr(), tq.data_ptr() if tq is not None else None, len(fb))
        st = dc.finish()
        assert st["kmers"] == want.total(), what
        assert st["distinct"] == len(want), what
        keys, cnts = dc.result()
        wk, wc = want.arrays()
        assert np.array_equal(keys, wk) and np.array_equal(cnts, wc), what
        assert dc.histogram() == want.histogram(), what
        if len(wk):
            probe = np.concatenate([wk[:: max(1, len(wk) // 200)], np.array([0, (1 << (2 * k)) - 1 if k < 32 else 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)])
            assert dc.lookup(probe).tolist() == [wd.get(int(x), 0) for x in probe], what


def _np_mix64(z):
    """oracle ko_mix64 (splitmix64 finaliser), vectorised with wrap-around."""
    z = z.astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


MID_KNOBS = {
    "KMERHIP_HOT_CUT": [None, None, None, "200000"],
    "KMERHIP_SURVIVAL": [None, None, "0.1"],
    "KMERHIP_REGION_NT": [None, None, "512", "1024"],
    "KMERHIP_NARROW": [None, None, "0"],
    "KMERHIP_L2_SKEW_X": [None, None, "0"],
    "KMERHIP_L2_ARENA": [None, None, None, "0"],
    "KMERHIP_PART_BUDGET_GB": [None, None, "0.6", "2"],
    "KMERHIP_OVF_AGG": [None, None, "1", "0"],
    "KMERHIP_ESTIMATE": [None, None, None, "0"],
    "KMERHIP_TABLE_REGIONS": [None, None, None, "20480", "81920"],
}


@pytest.mark.parametrize("seed", range(int(os.environ.get("KMERHIP_STRESS_MID_SEEDS", "6"))))
def test_random_route_same_digest_at_a_few_hundred_million_windows(seed, monkeypatch):
    """The same idea at 1-3 M reads (150-450 M windows), where the DEFAULT thresholds bite: heavy level-1 partitions, hot
    buckets above a thousandth of the batch, survival estimates of real batches, several batches per push.  Reads from the
    device generator with a random share overwritten by repeats; the whole map through the order-independent digest
    sum(mix(key ^ mix(count))) of the oracle's radix formulation, plus total and distinct."""
    import torch
    import krust_amd as K
    K.lib()
    rng = np.random.default_rng(70_000 + seed)
    k = int(rng.choice([15, 17, 19, 21, 21, 23, 27, 31, 32]))
    minq = [None, None, 20][int(rng.integers(0, 3))]
    n_reads, rl = int(rng.integers(1_000_000, 3_000_001)), 150
    glen = 1 << int(rng.integers(22, 28))
    nbytes = n_reads * (rl + 1)
    tb = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    tq = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    K.synth_reads_device(tb.data_ptr(), tq.data_ptr(), 777 + seed, glen, rl, 0, n_reads)
    share = float(rng.choice([0.0, 0.01, 0.1, 0.4]))
    if share:
        units = [b"A", b"AC", b"GATTACA", bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(11, 60))).astype(np.uint8))]
        rows = torch.from_numpy(rng.choice(n_reads, size=int(share * n_reads), replace=False)).cuda()
        v = tb.view(n_reads, rl + 1)
        for j, u in enumerate(units):
            pat = torch.from_numpy(np.resize(np.frombuffer(u, dtype=np.uint8), rl).copy()).cuda()
            v[rows[j::len(units)], :rl] = pat
    torch.cuda.synchronize()
    host, hq = tb.cpu().numpy(), tq.cpu().numpy()
    total, distinct, digest = O.count_flat_radix(host, k, qual=hq if minq is not None else None, min_quality=minq, nthreads=NCPU)
    hint = int(rng.choice([0, distinct, max(1, distinct // 8), 4 * distinct]))
    env = {name: vals[int(rng.integers(0, len(vals)))] for name, vals in MID_KNOBS.items()}
    for name, val in env.items():
        if val is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, val)
    cuts = sorted({0, n_reads, *[int(x) for x in rng.integers(0, n_reads, size=int(rng.integers(0, 3)))]})
    what = f"seed {seed}: k={k} minq={minq} reads={n_reads} genome=2^{glen.bit_length() - 1} repeats={share} hint={hint} cuts={cuts} env={ {a: b for a, b in env.items() if b} }"
    with K.DeviceCounter(k, min_quality=minq, capacity_hint=hint, path="partition") as dc:
        for a, b in zip(cuts, cuts[1:]):
            if b > a:
                dc.push_device(tb.data_ptr() + a * (rl + 1), tq.data_ptr() + a * (rl + 1) if minq is not None else None, (b - a) * (rl + 1))
        st = dc.finish()
        assert st["kmers"] == total and st["distinct"] == distinct, (what, st["kmers"], total, st["distinct"], distinct)
        keys, cnts = dc.result(sort=False)
    assert keys.size == distinct, what
    with np.errstate(over="ignore"):
        got = int(_np_mix64(keys ^ _np_mix64(cnts)).sum(dtype=np.uint64))
    assert got == digest, what
