"""Device-side record scanning (kh_push_text) against the oracle.

The expected table comes from a plain Python restatement of what the reference's readers hand to
the counting path (rust-bio semantics as used in src/reader.rs:58-79: FASTA lines of a record are
joined after trimming the line end; FASTQ is id / seq / '+' / qual) followed by the oracle."""
import os

import numpy as np
import pytest

import oracle_lib as O
from krust_amd import native

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_fasta(text):
    recs, cur = [], None
    for line in text.split(b"\n"):
        line = line.rstrip(b"\r")
        if line.startswith(b">"):
            if cur is not None:
                recs.append(b"".join(cur))
            cur = []
        elif cur is not None:
            cur.append(line)
    if cur is not None:
        recs.append(b"".join(cur))
    return recs, None


def parse_fastq(text):
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    lines = [l.rstrip(b"\r") for l in lines]
    assert len(lines) % 4 == 0
    return lines[1::4], lines[3::4]


def expect(text, fmt, k, minq):
    recs, quals = (parse_fasta if fmt == "fasta" else parse_fastq)(text)
    return O.count_records(recs, k, quals=quals if minq is not None else None, min_quality=minq).as_dict()


def device(text, fmt, k, minq, path=None, chunks=None):
    with native.DeviceCounter(k, min_quality=minq, path=path) as dc:
        for part in (chunks or [text]):
            dc.push_text(part, fmt)
        st = dc.finish()
        keys, counts = dc.result()
    d = dict(zip(keys.tolist(), counts.tolist()))
    assert st["kmers"] == sum(d.values())
    return d


def rand_seq(rng, n, alphabet=b"ACGTACGTACGTacgtN"):
    return bytes(rng.choice(list(alphabet), size=n).astype(np.uint8))


def make_fastq(rng, nreads, lo, hi, eol=b"\n", final_newline=True):
    out = []
    for i in range(nreads):
        n = int(rng.integers(lo, hi + 1))
        seq = rand_seq(rng, n)
        qual = bytes(rng.choice(list(b"!#+5?I@+"), size=n).astype(np.uint8))  # '@' and '+' also start quality lines
        out.append(b"@read%d some text ACGT" % i + eol + seq + eol + b"+" + eol + qual + eol)
    text = b"".join(out)
    if not final_newline:
        text = text[: -len(eol)]
    return text


def make_fasta(rng, nrec, lo, hi, width, eol=b"\n", final_newline=True, blank_lines=False):
    out = []
    for i in range(nrec):
        n = int(rng.integers(lo, hi + 1))
        seq = rand_seq(rng, n)
        out.append(b">chr%d ACGTACGTACGTACGTACGTACGTACGTACGT > description" % i + eol)
        for o in range(0, n, width):
            out.append(seq[o:o + width] + eol)
            if blank_lines and (o // width) % 7 == 3:
                out.append(eol)
    text = b"".join(out)
    if not final_newline:
        text = text[: -len(eol)]
    return text


@pytest.mark.parametrize("k,minq", [(21, None), (21, 20), (5, None), (31, 30), (1, None), (32, None)])
@pytest.mark.parametrize("eol", [b"\n", b"\r\n"], ids=["lf", "crlf"])
def test_fastq_text_matches_oracle(k, minq, eol):
    rng = np.random.default_rng(11 + k)
    text = make_fastq(rng, 700, 0, 260, eol=eol)
    assert len(text) > 3 * 4096  # several tiles, lines straddle tile and lane boundaries
    assert device(text, "fastq", k, minq) == expect(text, "fastq", k, minq)


def test_fastq_without_final_newline_and_in_chunks():
    rng = np.random.default_rng(5)
    a = make_fastq(rng, 300, 30, 151)
    b = make_fastq(rng, 300, 30, 151, final_newline=False)
    want = expect(a + b + b"\n", "fastq", 15, 10)
    assert device(None, "fastq", 15, 10, chunks=[a, b]) == want
    assert device(a + b, "fastq", 15, 10) == want


@pytest.mark.parametrize("size", [1, 15, 16, 17, 4095, 4096, 4097, 8192, 12288 + 5])
def test_fastq_sizes_around_tile_edges(size):
    # one long record padded to an exact text size
    rng = np.random.default_rng(size)
    head = b"@r\n"
    body = (size - len(head) - 3)  # seq + '\n+\n' + qual + '\n'
    if body < 1:
        pytest.skip("too small for a record")
    n = (body - 1) // 2
    seq = rand_seq(rng, n, b"ACGT")
    text = head + seq + b"\n+\n" + b"I" * n + b"\n"
    assert device(text, "fastq", 3, None) == expect(text, "fastq", 3, None)


@pytest.mark.parametrize("k", [3, 21, 32])
@pytest.mark.parametrize("width,eol,blank", [(60, b"\n", False), (70, b"\r\n", False), (61, b"\n", True), (10 ** 9, b"\n", False)],
                         ids=["w60", "w70-crlf", "w61-blank", "single-line"])
def test_fasta_text_matches_oracle(k, width, eol, blank):
    rng = np.random.default_rng(k * 1000 + width % 997)
    text = make_fasta(rng, 40, 0, 3000, width, eol=eol, blank_lines=blank)
    got = device(text, "fasta", k, None)
    assert got == expect(text, "fasta", k, None)
    # k-mers span the line breaks of a wrapped record: same table as the unwrapped text
    recs, _ = parse_fasta(text)
    assert got == O.count_records(recs, k).as_dict()


def test_fasta_header_text_is_never_counted():
    text = b">ACGTACGTACGTACGTACGT\nAC\n>GGGGGGGGGGGG\nGT\n"
    assert device(text, "fasta", 2, None) == expect(text, "fasta", 2, None)
    assert sum(device(text, "fasta", 2, None).values()) == 2


def test_fasta_without_final_newline():
    rng = np.random.default_rng(9)
    text = make_fasta(rng, 5, 100, 500, 60, final_newline=False)
    assert device(text, "fasta", 11, None) == expect(text, "fasta", 11, None)


def _fasta_unit_cases():
    """Texts whose header / record lines lie across the 1 KiB units and 4 KiB tiles the scan carries line states over."""
    rng = np.random.default_rng(77)
    seq = lambda n: rand_seq(rng, n)
    cases = {}
    # a header of 5 KB, one of 70 KB (more than 64 units: the look-back takes a second step), records behind them
    cases["long-headers"] = (b">" + b"h ACGT>" * 730 + b"\n" + seq(300) + b"\n>" + b"ACGT" * 17500 + b"\n" + seq(5000) + b"\n"
                             + b">x\n" + seq(100) + b"\n")
    # a single-line record of 200 KB (about 200 units without a line end), then a header
    cases["long-line"] = b">a\n" + seq(200_000) + b"\n>b ACGTACGTACGTACGTACGTACGT\n" + seq(3000) + b"\n"
    # line ends at the last byte of a unit / tile, and '>' as the first byte of the next
    for edge in (1024, 4096, 8192):
        for d in (-1, 0, 1):
            head = b">r\n"
            body = seq(edge + d - len(head) - 1)
            cases[f"edge{edge}{d:+d}"] = head + body + b"\n>ACGTACGTACGTACGTACGTACGTACGT\n" + seq(700) + b"\n" + seq(50) + b"\n"
            # the same with CR LF across the edge
            cases[f"crlf{edge}{d:+d}"] = head + seq(edge + d - len(head) - 1) + b"\r\n>ACGTACGTACGTACGTACGTACGTACGT\r\n" + seq(700) + b"\r\n"
    # nothing but headers; a header without a record at the very end, without a final newline
    # '>' that does not start a line is an invalid base inside a record and plain text inside a header; a header of '>' alone
    cases["gt-inside"] = (b">r1 >ACGTACGTACGTACGTACGTACGT> >\n" + seq(500) + b">" + seq(500) + b"\n" + seq(30) + b">>" + seq(30) + b"\n>\n"
                          + seq(1500) + b"\n>>ACGTACGTACGTACGTACGTACGTACGTACGT\n" + seq(2000) + b">\n" + seq(100) + b"\n")
    # short records, headers and lines of every length around the 16 bytes a lane takes
    parts = []
    for i in range(400):
        parts.append(b">" + b"h" * (i % 37) + b"\n")
        for _ in range(i % 4):
            parts.append(seq(1 + (i * 7) % 45) + b"\n")
    cases["ragged"] = b"".join(parts)
    cases["headers-only"] = b"".join(b">ACGTACGTACGTACGTACGT%d\n" % i for i in range(600))
    cases["open-header-end"] = b">r\n" + seq(2000) + b"\n>ACGTACGTACGTACGTACGTACGT"
    return cases


_FASTA_UNIT_CASES = _fasta_unit_cases()


@pytest.mark.parametrize("name", sorted(_FASTA_UNIT_CASES))
def test_fasta_line_states_across_units(name):
    text = _FASTA_UNIT_CASES[name]
    for k in (4, 21):
        assert device(text, "fasta", k, None) == expect(text, "fasta", k, None)


@pytest.mark.parametrize("seed", range(24))
def test_fasta_random_layouts_across_units(seed):
    """Random FASTA layouts of 20-120 KB: line and header lengths from 0 to several units, LF or CR LF, '>' inside lines,
    empty records, with or without the final line end -- the line states carried across 1 KiB units, waves and 4 KiB tiles
    against the Python line parser + oracle."""
    rng = np.random.default_rng(4200 + seed)
    eol = b"\r\n" if seed % 3 == 0 else b"\n"
    out = []
    for r in range(int(rng.integers(3, 60))):
        hl = int(rng.choice([0, 1, 7, 15, 16, 17, 60, 300, 1023, 1024, 5000])) if rng.random() < 0.5 else int(rng.integers(0, 200))
        out.append(b">" + rand_seq(rng, hl, alphabet=b"ACGT >xyz|0123") + eol)
        for _ in range(int(rng.integers(0, 12))):
            ll = int(rng.choice([0, 1, 15, 16, 17, 60, 61, 70, 1023, 1024, 1025, 4095, 4096, 9000])) if rng.random() < 0.4 else int(rng.integers(1, 120))
            line = bytearray(rand_seq(rng, ll))
            if ll > 3 and rng.random() < 0.1:
                line[int(rng.integers(1, ll))] = ord(">")  # not at the line's start: an invalid base, not a header
            out.append(bytes(line) + eol)
    text = b"".join(out)
    if seed % 2 and text.endswith(eol):
        text = text[: -len(eol)]
    k = [4, 21, 11, 31][seed % 4]
    assert device(text, "fasta", k, None) == expect(text, "fasta", k, None)


@pytest.mark.parametrize("edge", [1024, 4096])
@pytest.mark.parametrize("what", ["space-lf", "tab-crlf", "bare-cr", "cr-at-end-of-text"])
def test_fasta_line_end_rules_across_units(edge, what):
    """The blank-before-a-line-end and bare-CR rules look one byte ahead: across a lane, a wave and a tile."""
    rng = np.random.default_rng(edge)
    head = b">r\n"
    body = rand_seq(rng, edge - len(head) - 1, alphabet=b"ACGT")
    if what == "cr-at-end-of-text":  # a CR as the text's last byte is a line-end byte: accepted, and not a base
        text = head + body + b"\r"
        assert len(text) == edge
        assert device(text, "fasta", 5, None) == expect(text, "fasta", 5, None)
        return
    tail = {"space-lf": b" \n", "tab-crlf": b"\t\r\n", "bare-cr": b"\rA"}[what]
    text = head + body + tail + rand_seq(rng, 100, alphabet=b"ACGT") + b"\n"
    assert text[edge - 1] == tail[0]
    _format_error(text, "fasta")


@pytest.mark.parametrize("name", ["simple.fa", "soft_masked.fa", "with_n.fa", "simple.fq", "with_n.fq", "low_quality.fq"])
def test_reference_fixtures_as_text(name):
    with open(os.path.join(ROOT, "tests", "fixtures", name), "rb") as f:
        text = f.read()
    fmt = "fasta" if name.endswith(".fa") else "fastq"
    for k, minq in ((3, None), (4, 20 if fmt == "fastq" else None)):
        assert device(text, fmt, k, minq) == expect(text, fmt, k, minq)


@pytest.mark.parametrize("path", ["direct", "partition"])
def test_large_fastq_text_both_paths(path):
    rng = np.random.default_rng(77)
    genome = rand_seq(rng, 200_000, b"ACGT")
    starts = rng.integers(0, len(genome) - 150, size=60_000)
    recs = [genome[s:s + 150] for s in starts]
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * 150) for i, r in enumerate(recs))
    got = device(text, "fastq", 21, None, path=path)
    m = O.OracleMap()
    m.process(b"\n".join(recs), 21)
    assert got == m.as_dict()


def _format_error(text, fmt, k=5, minq=None):
    with native.DeviceCounter(k, min_quality=minq) as dc:
        with pytest.raises(native.KmerHipError) as e:
            dc.push_text(text, fmt)
        assert e.value.status == native.KH_ERR_FORMAT
        assert dc.finish()["kmers"] == 0  # nothing was counted; the context stays usable
        dc.push(b"ACGTACGT")
        assert dc.finish()["kmers"] == 4


@pytest.mark.parametrize("text,fmt", [
    (b"@r\nACGT\nACGT\n+\nIIII\nIIII\n", "fastq"),              # wrapped FASTQ (6 lines)
    (b"@r\nACGT\nACGT\n+\nIIII\nIIII\n@q\nAC\n", "fastq"),      # 8 lines, markers in the wrong places
    (b"@r\nACGT\n+\nIII\n", "fastq"),                           # |seq| != |qual|
    (b"r\nACGT\n+\nIIII\n", "fastq"),                           # no '@'
    (b"@r\nACGT\n-\nIIII\n", "fastq"),                          # no '+'
    (b"@r\nACGT\n+\nIIII\n\n@q\nAC\n+\nII\n", "fastq"),         # blank line between records
    (b"ACGT\n>r\nACGT\n", "fasta"),                             # text before the first header
    (b">r\nACGT \nACGT\n", "fasta"),                            # blank at a line end inside a record
    (b">r\nACGT\t\r\nACGT\n", "fasta"),
    (b">r\nACGTAC\rGTACGT\nACGT\n", "fasta"),                   # bare CR in the middle of a line: the line parsers keep
    (b">r\nACGTACGT\r\r\nACGT\n", "fasta"),                     # it as an invalid base, dropping it would join AC|GT
], ids=["wrapped", "wrapped8", "lens", "no-at", "no-plus", "blank-line", "no-header", "trailing-space", "trailing-tab-crlf",
        "mid-line-cr", "double-cr"])
def test_unsupported_layouts_are_reported_not_miscounted(text, fmt):
    _format_error(text, fmt)


def test_push_text_device_needs_alignment_and_matches_host_text():
    import torch
    rng = np.random.default_rng(3)
    text = make_fastq(rng, 500, 50, 150)
    buf = torch.zeros(len(text) + 32, dtype=torch.uint8, device="cuda")
    buf[16:16 + len(text)] = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    with native.DeviceCounter(21, min_quality=20) as dc:
        with pytest.raises(native.KmerHipError) as e:
            dc.push_text_device(buf.data_ptr() + 1, len(text), "fastq")
        assert e.value.status == native.KH_ERR_BAD_ARG
        assert buf.data_ptr() % 16 == 0
        dc.push_text_device(buf.data_ptr() + 16, len(text), "fastq")
        dc.finish()
        keys, counts = dc.result()
    assert dict(zip(keys.tolist(), counts.tolist())) == expect(text, "fastq", 21, 20)


def test_text_counted_under_the_next_call_is_never_lost_or_doubled():
    """kh_push_text returns when its text is on the device and scanned; the COUNTING is left to the next call that enters
    the context (the next kh_push_text copies its text meanwhile).  Whatever comes next -- another text, plain bases, a
    lookup, the results, a refused text, a reset -- the pending text is counted exactly once, or forgotten with the reset."""
    import torch
    rng = np.random.default_rng(31)
    a, b, c3 = (make_fastq(rng, 700, 40, 160) for _ in range(3))
    fa = make_fasta(rng, 30, 100, 3000, 60)
    plain = rand_seq(rng, 40_000, b"ACGT")
    k, minq = 21, 20

    def add(d, e):
        for kk, v in e.items():
            d[kk] = d.get(kk, 0) + v
        return d
    ea, eb, ec = (expect(t, "fastq", k, minq) for t in (a, b, c3))
    efa = expect(fa, "fasta", k, minq)
    eplain = O.count_records([plain], k).as_dict()
    with native.DeviceCounter(k, min_quality=minq) as dc:
        dc.push_text(a, "fastq")                                   # pending
        some = np.array(list(ea)[:64], dtype=np.uint64)
        assert dc.lookup(some).tolist() == [ea[int(x)] for x in some]   # a lookup counts it first
        dc.push_text(b, "fastq")                                   # pending
        dc.push_text(fa, "fasta")                                  # counts b under fa's copy; fa pending (other format, no qualities)
        t = torch.frombuffer(bytearray(plain), dtype=torch.uint8).cuda()
        dc.push_device(t.data_ptr(), None, t.numel())              # counts fa, then the plain bases
        dc.push_text(c3, "fastq")                                   # pending
        with pytest.raises(native.KmerHipError) as e:              # a refused text: c3 is counted under its copy, it is not
            dc.push_text(b"@r\nACGT\n-\nIIII\n", "fastq")
        assert e.value.status == native.KH_ERR_FORMAT
        keys, counts = dc.result()
        want = {}
        for e_ in (ea, eb, efa, eplain, ec):
            add(want, e_)
        assert dict(zip(keys.tolist(), counts.tolist())) == want
        st = dc.finish()
        assert st["kmers"] == sum(want.values())
        dc.push_text(a, "fastq")                                   # pending ...
        dc.reset()                                                 # ... and forgotten
        st = dc.finish()
        assert st["kmers"] == 0 and dc.result_size() == 0
        dc.push_text(b, "fastq")
        st = dc.finish()                                           # finish counts it
        assert st["kmers"] == sum(eb.values())


# ---------------------------------------------------------------------------
# property tests: arbitrary small texts, device scanner vs a line-by-line restatement
# ---------------------------------------------------------------------------
from hypothesis import given, settings, strategies as st, HealthCheck  # noqa: E402

_seq_line = st.text(alphabet="ACGTacgtNn", min_size=0, max_size=40).map(str.encode)
_eol = st.sampled_from([b"\n", b"\r\n"])


@st.composite
def fasta_texts(draw):
    eol = draw(_eol)
    out = []
    for i in range(draw(st.integers(1, 6))):
        out.append(b">" + draw(st.text(alphabet="ACGT >@+x", max_size=12)).encode() + eol)
        for _ in range(draw(st.integers(0, 5))):
            out.append(draw(_seq_line) + eol)
    text = b"".join(out)
    if draw(st.booleans()) and text.endswith(eol):
        text = text[: -len(eol)]
    return text


@st.composite
def fastq_texts(draw):
    eol = draw(_eol)
    out = []
    for i in range(draw(st.integers(1, 8))):
        seq = draw(_seq_line)
        qual = bytes(draw(st.lists(st.sampled_from(list(b"!#+5?I@+>")), min_size=len(seq), max_size=len(seq))))
        out.append(b"@" + draw(st.text(alphabet="ACGT @+x", max_size=8)).encode() + eol + seq + eol + b"+" + eol + qual + eol)
    text = b"".join(out)
    if draw(st.booleans()) and len(seq):  # (an EMPTY last quality line without its newline is not a line at all)
        text = text[: -len(eol)]
    return text


@pytest.fixture(scope="module")
def text_counters():
    """One context per (k, min_quality) reused across examples (context creation dominates otherwise)."""
    made = {}

    def get(k, minq):
        if (k, minq) not in made:
            made[(k, minq)] = native.DeviceCounter(k, min_quality=minq)
        dc = made[(k, minq)]
        dc.reset()
        return dc

    yield get
    for dc in made.values():
        dc.close()


def _device_dict(dc, text, fmt):
    dc.push_text(text, fmt)
    dc.finish()
    keys, counts = dc.result()
    return dict(zip(keys.tolist(), counts.tolist()))


@given(fasta_texts(), st.sampled_from([1, 2, 3, 5, 11]))
@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
def test_prop_fasta_text(text_counters, text, k):
    assert _device_dict(text_counters(k, None), text, "fasta") == expect(text, "fasta", k, None)


@given(fastq_texts(), st.sampled_from([1, 3, 7]), st.sampled_from([None, 0, 10, 20, 40]))
@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
def test_prop_fastq_text(text_counters, text, k, minq):
    assert _device_dict(text_counters(k, minq), text, "fastq") == expect(text, "fastq", k, minq)


@pytest.mark.parametrize("fmt,minq", [("fastq", 20), ("fastq", None), ("fasta", None)])
def test_text_accumulates_on_the_device_and_switches_buffers(monkeypatch, fmt, minq):
    """Round 4: scanned text accumulates in one of two device buffers and is counted when a buffer is full (or something
    looks at the table).  KMERHIP_TEXT_ACC_MB=1 makes the buffers 1 MiB: forty chunks of ~150 KB fill and switch them
    several times, the count of one buffer running while the next chunks are scanned into the other -- with a lookup in
    the middle (counts what has accumulated) and a chunk LARGER than a buffer (gets a buffer of its own size)."""
    monkeypatch.setenv("KMERHIP_TEXT_ACC_MB", "1")
    rng = np.random.default_rng(77)
    k = 21
    chunks = []
    for i in range(40):
        n = 3_000_000 if i == 25 else 150_000
        chunks.append(make_fastq(rng, n // 300, 100, 200) if fmt == "fastq" else make_fasta(rng, max(2, n // 5000), 2000, 8000, 70))
    want = {}
    with native.DeviceCounter(k, min_quality=minq) as dc:
        for i, part in enumerate(chunks):
            dc.push_text(part, fmt)
            for kk, v in expect(part, fmt, k, minq).items():
                want[kk] = want.get(kk, 0) + v
            if i == 13:
                some = np.array(list(want)[:200], dtype=np.uint64)
                assert dc.lookup(some).tolist() == [want[int(x)] for x in some]
        st = dc.finish()
        keys, counts = dc.result()
    assert st["kmers"] == sum(want.values())
    assert dict(zip(keys.tolist(), counts.tolist())) == want


def test_deferred_text_scan_reports_one_call_late_and_counts_the_same():
    """KH_FLAG_DEFER_TEXT_SCAN (what the kmerust command line sets): kh_push_text returns when its text is on the device;
    the PREVIOUS text is scanned beside the transfer, this one during the next call (kh_finish at the latest).  Same table
    as without the flag; a refusal arrives one call late and concerns the previous text -- the text of the call that
    reports it is dropped too, nothing of either is counted; after a reset the context works again."""
    rng = np.random.default_rng(911)
    k, minq = 21, 20
    parts = [make_fastq(rng, 900, 60, 180) for _ in range(6)]
    want = {}
    for p in parts:
        for kk, v in expect(p, "fastq", k, minq).items():
            want[kk] = want.get(kk, 0) + v
    with native.DeviceCounter(k, min_quality=minq, defer_text_scan=True) as dc:
        for p in parts:
            dc.push_text(p, "fastq")
        st = dc.finish()                       # scans the last text, counts everything
        keys, counts = dc.result()
        assert st["kmers"] == sum(want.values()) and dict(zip(keys.tolist(), counts.tolist())) == want
        # a refused layout: reported by the NEXT call
        dc.reset()
        dc.push_text(parts[0], "fastq")
        dc.push_text(b"@r\nACGT\n-\nIIII\n", "fastq")            # returns OK: not scanned yet (parts[0] was, beside its copy)
        with pytest.raises(native.KmerHipError) as e:
            dc.push_text(parts[1], "fastq")                      # the bad text's verdict; parts[1] is dropped with it
        assert e.value.status == native.KH_ERR_FORMAT
        st = dc.finish()
        e0 = expect(parts[0], "fastq", k, minq)
        assert st["kmers"] == sum(e0.values())                   # only the first text was ever counted
        # ... or by kh_finish when nothing else comes
        dc.reset()
        dc.push_text(b"@r\nACGT\n-\nIIII\n", "fastq")
        with pytest.raises(native.KmerHipError) as e:
            dc.finish()
        assert e.value.status == native.KH_ERR_FORMAT
        dc.reset()
        dc.push_text(parts[2], "fastq")
        assert dc.finish()["kmers"] == sum(expect(parts[2], "fastq", k, minq).values())
