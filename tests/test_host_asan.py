"""The C++ host layer under AddressSanitizer + UBSan, on the CPU (GPU sanitizer runs are not available on this
pool): `make -C krust_amd/host asan` links the reader, the record cutting / chunk growth (`fasta_cut`,
`fastq_cut`, the memmove of the chunk buffer), the KMIX index code and the CLI against a recording stub of the
kh_* calls (tests/host_asan/stub_kmerhip.cpp: counts nothing, logs every push).

  * the CPU CLI tests (tests/test_host_cli.py) run again on the sanitized binary;
  * hypothesis fuzzes the parsers with arbitrary bytes (the reference fuzzes its parsers' callers the same
    way: fuzz/fuzz_targets/*.rs) -- any exit status but never a sanitizer report, never a crash;
  * property: whatever the chunk size, the text handed to the device scanner is the file cut at record
    starts -- the chunks concatenate to the file and every chunk begins with a record header; when the scanner
    refuses a chunk mid-file, the line parser pushes exactly the records `__parse` prints."""
import gzip
import os
import re
import subprocess

import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import test_host_cli as thc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_BIN = os.path.join(ROOT, "krust_amd", "host", "kmerust_asan")
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0:exitcode=97", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"}


@pytest.fixture(scope="module", autouse=True)
def _build_asan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "krust_amd", "host"), "asan"], stdout=subprocess.DEVNULL)


def run(*args, stdin=None, env=None):
    e = dict(os.environ, **SAN_ENV)
    e.pop("KH_STUB_LOG", None)
    e.pop("KH_STUB_TEXT", None)
    if env:
        e.update(env)
    r = subprocess.run([ASAN_BIN, *args], input=stdin, capture_output=True, timeout=120, env=e)
    assert b"AddressSanitizer" not in r.stderr and b"runtime error:" not in r.stderr and b"LeakSanitizer" not in r.stderr, \
        r.stderr[-3000:].decode(errors="replace")
    assert r.returncode in (0, 1, 2), (r.returncode, r.stderr[-2000:])
    return r


def test_cli_tests_on_the_sanitized_binary(monkeypatch, fixtures_dir, tmp_path):
    monkeypatch.setattr(thc, "run", lambda *a, stdin=None: run(*a, stdin=stdin))
    thc.test_help_and_version()
    thc.test_missing_and_bad_k()
    thc.test_missing_file_exit_1()
    thc.test_bad_enum_values()
    thc.test_reader_fixtures(fixtures_dir)
    (tmp_path / "a").mkdir()
    thc.test_reader_edge_cases(tmp_path / "a")
    (tmp_path / "b").mkdir()
    thc.test_query_on_index(tmp_path / "b")
    (tmp_path / "c").mkdir()
    thc.test_index_validation(tmp_path / "c")


def read_log(path):
    """[(kind, header fields, payload bytes)] of the stub's log."""
    out = []
    if not os.path.exists(path):
        return out
    data = open(path, "rb").read()
    pos = 0
    while pos < len(data):
        nl = data.index(b"\n", pos)
        head = data[pos:nl].split()
        pos = nl + 1
        if head[0] == b"RESET":
            out.append(("RESET", (), b""))
        elif head[0] == b"TEXT":
            n = int(head[1])
            out.append(("TEXT", (int(head[2]),), data[pos:pos + n]))
            pos += n
        else:
            n, q = int(head[1]), int(head[2])
            out.append(("PUSH", (q,), data[pos:pos + n * (1 + q)]))
            pos += n * (1 + q)
    return out


SEQ = st.text(alphabet="ACGTacgtNn", min_size=0, max_size=90).map(str.encode)
QCH = st.text(alphabet="#+5I@>", min_size=1, max_size=1)


@st.composite
def fastq_file(draw):
    recs = []
    for i in range(draw(st.integers(0, 25))):
        s = draw(SEQ)
        q = draw(st.text(alphabet="#+5I@>", min_size=len(s), max_size=len(s))).encode()
        recs.append(b"@r%d x\n%s\n+\n%s\n" % (i, s, q))
    return b"".join(recs)


@st.composite
def fasta_file(draw):
    recs = []
    for i in range(draw(st.integers(0, 12))):
        s = draw(st.text(alphabet="ACGTacgtNn", min_size=0, max_size=400)).encode()
        w = draw(st.integers(7, 80))
        recs.append(b">c%d\n" % i + b"".join(s[o:o + w] + b"\n" for o in range(0, len(s), w)))
    return b"".join(recs)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(data=st.binary(min_size=0, max_size=600), fmt=st.sampled_from(["fasta", "fastq"]), gz=st.booleans(), qual=st.booleans())
def test_fuzz_parsers_with_arbitrary_bytes(tmp_path, data, fmt, gz, qual):
    p = tmp_path / ("fuzz." + ("fq" if fmt == "fastq" else "fa") + (".gz" if gz else ""))
    p.write_bytes(gzip.compress(data) if gz else data)
    run("__parse", str(p), fmt, *(["--qual"] if qual else []))
    log = tmp_path / "fuzz.log"
    if log.exists():
        log.unlink()
    # the counting entry: tiny text chunks (cuts, buffer growth), scanner accepting, refusing, refusing mid-file
    for text_mode in ("1", None, "refuse:1"):
        env = {"KMERUST_TEXT_CHUNK_KB": "1", "KH_STUB_LOG": str(log)}
        if text_mode:
            env["KH_STUB_TEXT"] = text_mode
        run("5", str(p), "-i", fmt, "--quiet", "-f", "tsv", *(["-Q", "10"] if qual else []), env=env)


def _dump_records(path, fmt, qual):
    r = run("__parse", path, fmt, *(["--qual"] if qual else []))
    assert r.returncode == 0, r.stderr
    bases, quals = b"", b""
    for m in re.finditer(rb"BATCH records=\d+ bytes=(\d+)\n", r.stdout):
        n = int(m.group(1))
        bases += r.stdout[m.end():m.end() + n]
        rest = r.stdout[m.end() + n:]
        if rest.startswith(b"QUAL\n"):
            quals += rest[5:5 + n]
    return bases, quals


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(text=st.one_of(fastq_file(), fasta_file()), chunk_kb=st.sampled_from(["1", "2", "64"]), ndev=st.sampled_from([1, 3]))
def test_text_chunks_are_whole_records_and_cover_the_file(tmp_path, text, chunk_kb, ndev):
    if not text:
        return
    fastq = text.startswith(b"@")
    p = tmp_path / ("t.fq" if fastq else "t.fa")
    p.write_bytes(text)
    log = tmp_path / "t.log"
    for f in (log,):
        if f.exists():
            f.unlink()
    dev = ["--devices", ",".join(["0"] * ndev)] if ndev > 1 else []
    env = {"KMERUST_TEXT_CHUNK_KB": chunk_kb, "KH_STUB_LOG": str(log), "KH_STUB_TEXT": "1"}
    r = run("7", str(p), "--quiet", "-f", "histogram", *dev, env=env)
    assert r.returncode == 0, r.stderr
    chunks = [b for kind, _, b in read_log(str(log)) if kind == "TEXT"]
    assert all(c[:1] == (b"@" if fastq else b">") for c in chunks)          # every chunk starts at a record header
    if ndev == 1:
        assert b"".join(chunks) == text                                      # and they concatenate to the file
    else:
        assert sorted(b"".join(chunks)) == sorted(text) and sum(map(len, chunks)) == len(text)
    if len(chunks) < 2:
        return
    # scanner refuses the second chunk: reset, then the line parser pushes exactly what __parse prints
    log.unlink()
    env["KH_STUB_TEXT"] = "refuse:1"
    r = run("7", str(p), "--quiet", "-f", "histogram", "-Q", "5", env=env)
    assert r.returncode == 0, r.stderr
    entries = read_log(str(log))
    pushes = [(h, b) for kind, h, b in entries if kind == "PUSH"]
    want_b, want_q = _dump_records(str(p), "fastq" if fastq else "fasta", fastq)
    got_b = b"".join(b[: len(b) // (1 + h[0])] for h, b in pushes)
    got_q = b"".join(b[len(b) // 2:] for h, b in pushes if h[0])
    assert got_b == want_b and (not fastq or got_q == want_q)
    kinds = [kind for kind, _, _ in entries]
    assert kinds.count("TEXT") == 1 and "RESET" in kinds                      # the accepted chunk was forgotten first
    assert kinds.index("RESET") < kinds.index("PUSH")


def test_records_longer_than_the_chunk_then_short_ones(tmp_path):
    """Round 5's reader thread (three chunk buffers): a buffer that had to GROW for a record longer than the chunk hands a tail
    as long as itself to the next buffer -- which must grow with it (the first version copied it into a chunk-sized buffer:
    a segfault on the GPU box, test_gpu_cli.py::test_records_larger_than_the_text_chunk).  Under ASan, with the stub."""
    import numpy as np
    rng = np.random.default_rng(5)
    text = b""
    for i, n in enumerate([9_000, 30, 2_500, 14_000, 5, 40_000, 12]):
        s = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)[rng.choice(9, size=n)].tobytes()
        text += b">chr%d\n" % i + b"".join(s[o:o + 60] + b"\n" for o in range(0, n, 60))
    p = tmp_path / "long.fa"
    p.write_bytes(text)
    log = tmp_path / "long.log"
    for reader in ("1", "0"):
        if log.exists():
            log.unlink()
        r = run("7", str(p), "--quiet", "-f", "histogram", env={"KMERUST_TEXT_CHUNK_KB": "1", "KH_STUB_LOG": str(log), "KH_STUB_TEXT": "1",
                                                                 "KMERUST_PIPELINED_READER": reader})
        assert r.returncode == 0, r.stderr
        chunks = [b for kind, _, b in read_log(str(log)) if kind == "TEXT"]
        assert all(c[:1] == b">" for c in chunks) and b"".join(chunks) == text and len(chunks) >= 3
