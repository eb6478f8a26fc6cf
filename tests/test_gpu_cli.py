"""GPU tests of the `kmerust` command line (krust_amd/host) -- the reference's
tests/integration_tests.rs restated against our binary; expected values from tests/golden."""
import gzip
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "krust_amd", "host", "kmerust")
with open(os.path.join(ROOT, "tests", "golden", "derived_fixture_tables.json")) as f:
    DERIVED = {(r["fixture"], r["k"], r["min_quality"]): r for r in json.load(f)["tables"]}


def run(*args, stdin=None, env=None):
    return subprocess.run([BIN, *args], input=stdin, capture_output=True, timeout=300,
                          env=None if env is None else {**os.environ, **env})


def tsv(out):
    return {l.split(b"\t")[0].decode(): int(l.split(b"\t")[1]) for l in out.splitlines()}


def fx(name):
    return os.path.join(ROOT, "tests", "fixtures", name)


def test_default_fasta_output():
    r = run("3", fx("simple.fa"), "--quiet")
    assert r.returncode == 0 and r.stderr == b""                         # integration_tests.rs:233-261: quiet => empty stderr
    lines = r.stdout.splitlines()
    got = {lines[i + 1].decode(): int(lines[i][1:]) for i in range(0, len(lines), 2)}
    assert all(l.startswith(b">") for l in lines[::2])                   # ">{count}\n{kmer}" (run.rs:453-456)
    assert got == DERIVED[("simple.fa", 3, None)]["counts"]


def test_banner_goes_to_stderr():
    r = run("3", fx("simple.fa"))
    assert r.returncode == 0
    for needle in (b"k-length: 3", b"data: ", b"input-format: fasta (auto-detected)", b"output-format: fasta"):
        assert needle in r.stderr                                        # src/main.rs:75-134


@pytest.mark.parametrize("key", sorted(DERIVED, key=str), ids=lambda k: f"{k[0]}-k{k[1]}-q{k[2]}")
def test_tsv_matches_golden_tables(key):
    fixture, k, q = key
    args = [str(k), fx(fixture), "--format", "tsv", "-q"] + (["-Q", str(q)] if q is not None else [])
    r = run(*args)
    assert r.returncode == 0, r.stderr
    assert tsv(r.stdout) == DERIVED[key]["counts"]


def test_json_format():
    r = run("4", fx("simple.fa"), "-f", "json", "-q")                    # integration_tests.rs:176-188, 611-661
    data = json.loads(r.stdout)
    assert {d["kmer"]: d["count"] for d in data} == DERIVED[("simple.fa", 4, None)]["counts"]
    assert r.stdout.startswith(b"[\n  {\n    \"kmer\": ") and r.stdout.endswith(b"\n]\n")  # serde pretty, 2-space indent
    assert all(list(d) == ["kmer", "count"] for d in data)
    assert run("9", fx("simple.fa"), "-f", "json", "-q").stdout == b"[]\n"


def test_histogram_format_and_min_count():
    r = run("3", fx("simple.fa"), "-f", "histogram", "-q")               # integration_tests.rs:664-765
    rows = [tuple(map(int, l.split(b"\t"))) for l in r.stdout.splitlines()]
    assert rows == [tuple(x) for x in DERIVED[("simple.fa", 3, None)]["histogram"]]
    assert rows == sorted(rows)
    r = run("3", fx("simple.fa"), "-f", "histogram", "-q", "--min-count", "3")
    assert [tuple(map(int, l.split(b"\t"))) for l in r.stdout.splitlines()] == [(3, 1), (4, 1)]  # filter first (run.rs:447-450)
    r = run("3", "-", "--format", "histogram", "--quiet", stdin=b">seq1\nAAAAAAAA\n")
    assert b"6\t1" in r.stdout                                            # integration_tests.rs:768-799


def test_min_count_filters_output():
    r = run("3", fx("simple.fa"), "-f", "tsv", "-q", "-m", "2")          # integration_tests.rs:191-230
    assert tsv(r.stdout) == {"ACG": 4, "GTA": 3}


def test_soft_masked_and_n():
    assert b"AAA\t2" in run("3", fx("soft_masked.fa"), "-f", "tsv", "-q").stdout   # integration_tests.rs:264-281
    out = run("3", fx("with_n.fa"), "-f", "tsv", "-q").stdout
    assert b"N" not in out and tsv(out) == DERIVED[("with_n.fa", 3, None)]["counts"]


def test_stdin_default_and_dash():
    data = open(fx("simple.fa"), "rb").read()
    a = run("3", "-f", "tsv", "-q", stdin=data)                          # integration_tests.rs:47-75, 284-407
    b = run("3", "-", "-f", "tsv", "-q", stdin=data)
    assert a.returncode == 0 and tsv(a.stdout) == tsv(b.stdout) == DERIVED[("simple.fa", 3, None)]["counts"]
    fq = open(fx("simple.fq"), "rb").read()
    c = run("3", "-", "-i", "fastq", "-f", "tsv", "-q", stdin=fq)
    assert tsv(c.stdout) == tsv(a.stdout)


def test_fastq_equals_fasta_and_gzip_equals_plain():
    fa = tsv(run("4", fx("simple.fa"), "-f", "tsv", "-q").stdout)
    assert tsv(run("4", fx("simple.fq"), "-f", "tsv", "-q").stdout) == fa           # integration_tests.rs:487-523
    assert tsv(run("4", fx("simple.fa.gz"), "-f", "tsv", "-q").stdout) == fa        # integration_tests.rs:556-594
    assert tsv(run("4", fx("simple.fq.gz"), "-f", "tsv", "-q").stdout) == fa


def test_quality_flag_and_warnings():
    lq = tsv(run("4", fx("low_quality.fq"), "-f", "tsv", "-q", "-Q", "20").stdout)
    assert lq == DERIVED[("low_quality.fq", 4, 20)]["counts"]
    r = run("4", fx("simple.fa"), "-f", "tsv", "-Q", "20")               # FASTA ignores -Q (tests/quality_tests.rs:94-114)
    assert b"--min-quality is ignored for FASTA input" in r.stderr
    assert tsv(r.stdout) == DERIVED[("simple.fa", 4, None)]["counts"]
    r = run("4", "-", "-i", "fastq", "-f", "tsv", "-Q", "20", stdin=open(fx("low_quality.fq"), "rb").read())
    assert b"--min-quality is not yet supported for stdin input" in r.stderr   # src/main.rs:145-152
    assert tsv(r.stdout) == DERIVED[("low_quality.fq", 4, None)]["counts"]


def test_save_and_query(tmp_path):
    idx = str(tmp_path / "simple.kmix")                                   # integration_tests.rs:806-1083
    r = run("3", fx("simple.fa"), "-f", "tsv", "--save", idx, "-m", "4")
    assert r.returncode == 0 and b"saved: " in r.stderr and b"(6 k-mers)" in r.stderr
    assert tsv(r.stdout) == {"ACG": 4}                                    # stdout honours --min-count ...
    for kmer, cnt in DERIVED[("simple.fa", 3, None)]["counts"].items():  # ... the index holds everything (main.rs:182-205)
        assert run("query", idx, kmer).stdout == f"{cnt}\n".encode()
    assert run("query", idx, "CGT").stdout == b"4\n"                      # RC of ACG
    assert run("query", idx, "CCC").stdout == b"0\n"
    gz = str(tmp_path / "simple.kmix.gz")
    assert run("5", fx("simple.fa"), "-q", "--save", gz).returncode == 0
    assert gzip.open(gz).read()[:4] == b"KMIX"
    assert run("query", gz, "ACGTA").stdout == b"2\n"


def test_hg_like_fasta_histogram(tmp_path):
    """BASELINE.json configs[4] in miniature (hg38 itself is not on the box): a generated 'hg-like'
    FASTA -- 10 records of very different lengths up to 9 Mbp, 60-column lines, ~50 % soft-masked
    (lowercase) blocks, runs of N, a poly-A tract and a tandem repeat -- counted by the CLI with
    `--format histogram` and compared line by line with the oracle's histogram, then `tsv` on a
    1/64 key sample.  Exercises long records (k-mers across tile / workgroup / staging-chunk
    boundaries), soft-masking and N handling through the real reader."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    rng = np.random.default_rng(38)
    lens = [9_000_000, 4_000_000, 2_000_000, 1_000_000, 500_000, 200_000, 50_000, 3_000, 40, 7]
    path = tmp_path / "hg_like.fa"
    seqs = []
    with open(path, "wb") as f:
        for i, n in enumerate(lens):
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
            if n >= 200_000:
                for _ in range(max(1, n // 200_000)):            # N runs
                    a = int(rng.integers(0, n - 5_000)); s[a:a + int(rng.integers(1, 5_000))] = ord("N")
                for _ in range(max(1, n // 100_000)):            # soft-masked blocks (~half the record)
                    a = int(rng.integers(0, n - 50_000)); b = a + int(rng.integers(1_000, 50_000)); s[a:b] |= 0x20
                a = int(rng.integers(0, n - 30_000)); s[a:a + 20_000] = ord("A")          # poly-A tract
                a = int(rng.integers(0, n - 30_000)); s[a:a + 12_000] = np.frombuffer(b"ACGGTT" * 2_000, dtype=np.uint8)
            seqs.append(s.tobytes())
            f.write(b">chr%d some description\n" % (i + 1))
            for off in range(0, n, 60):
                f.write(seqs[-1][off:off + 60] + b"\n")
    m = O.OracleMap()
    for s in seqs:
        m.scan_flat(s, 21, nthreads=8)
    r = run("21", str(path), "--format", "histogram", "--quiet")
    assert r.returncode == 0, r.stderr
    got = [tuple(map(int, l.split(b"\t"))) for l in r.stdout.splitlines()]
    assert got == m.histogram(1)                                   # ascending, bit-exact
    assert sum(c * f for c, f in got) == m.total()
    r = run("21", str(path), "--format", "tsv", "--quiet", "--min-count", "3")
    want = {k: c for k, c in m.as_dict().items() if c >= 3}
    got_tsv = tsv(r.stdout)
    assert len(got_tsv) == len(want)
    import krust_amd
    assert all(want[krust_amd.pack(kmer)] == c for kmer, c in list(got_tsv.items())[:20000])


def _write_reads(tmp_path, wrapped_fastq=False):
    import numpy as np
    rng = np.random.default_rng(123)
    genome = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)[rng.choice(9, size=60_000, p=[.22, .22, .22, .22, .03, .03, .02, .02, .02])].tobytes()
    fq, fa = tmp_path / "reads.fq", tmp_path / "contigs.fa"
    with open(fq, "wb") as f:
        for i in range(4000):
            a = int(rng.integers(0, len(genome) - 200)); n = int(rng.integers(1, 200))
            q = bytes(rng.choice(list(b"#+5I@"), size=n, p=[.02, .02, .06, .8, .1]).astype(np.uint8))
            s = genome[a:a + n]
            if wrapped_fastq and i == 1234 and n > 10:
                f.write(b"@r%d\n%s\n%s\n+\n%s\n%s\n" % (i, s[:5], s[5:], q[:5], q[5:]))
            else:
                f.write(b"@r%d\n%s\n+\n%s\n" % (i, s, q))
    with open(fa, "wb") as f:
        for i in range(300):
            a = int(rng.integers(0, len(genome) - 3000)); n = int(rng.integers(0, 3000))
            f.write(b">c%d len=%d\n" % (i, n))
            for o in range(0, n, 70):
                f.write(genome[a + o:a + min(n, o + 70)] + b"\n")
    return str(fq), str(fa)


@pytest.mark.parametrize("chunk_kb", [None, "16"], ids=["one-chunk", "16KiB-chunks"])
def test_device_text_scan_equals_host_line_parser(tmp_path, chunk_kb):
    """kh_push_text (records found on the device) against the host line parser, through the CLI:
    FASTQ with / without -Q, wrapped FASTA, gzip; small chunks force cuts at record boundaries."""
    fq, fa = _write_reads(tmp_path)
    env = {} if chunk_kb is None else {"KMERUST_TEXT_CHUNK_KB": chunk_kb}
    for args in (["21", fq], ["21", fq, "-Q", "20"], ["5", fq, "-Q", "10"], ["21", fa], ["32", fa], ["2", fa]):
        args = args + ["--format", "tsv", "--quiet"]
        dev = run(*args, env=env)
        host = run(*args, env={"KMERUST_HOST_PARSE": "1"})
        assert dev.returncode == 0 and host.returncode == 0, (dev.stderr, host.stderr)
        assert tsv(dev.stdout) == tsv(host.stdout) and tsv(dev.stdout)
    gz = fq + ".gz"
    with open(fq, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    a = run("21", gz, "-Q", "20", "--format", "tsv", "--quiet", env=env)
    b = run("21", fq, "-Q", "20", "--format", "tsv", "--quiet", env={"KMERUST_HOST_PARSE": "1"})
    assert a.returncode == 0 and tsv(a.stdout) == tsv(b.stdout)


def test_wrapped_fastq_falls_back_to_the_line_parser(tmp_path):
    """A layout the device scanner refuses mid-file (after earlier chunks were already counted) must give
    the line parser's table, not a partial or doubled one."""
    fq, _ = _write_reads(tmp_path, wrapped_fastq=True)
    for env in ({"KMERUST_TEXT_CHUNK_KB": "16"}, {}):
        dev = run("11", fq, "-Q", "10", "--format", "tsv", "--quiet", env=env)
        host = run("11", fq, "-Q", "10", "--format", "tsv", "--quiet", env={"KMERUST_HOST_PARSE": "1"})
        assert dev.returncode == 0 and host.returncode == 0, (dev.stderr, host.stderr)
        assert tsv(dev.stdout) == tsv(host.stdout)


def test_malformed_input_keeps_the_parser_error(tmp_path):
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@r\nACGT\n+\nII\n")
    r = run("3", str(bad), "--quiet")
    assert r.returncode == 1 and b"unequal length" in r.stderr


def test_records_larger_than_the_text_chunk(tmp_path):
    """hg38-like shape in miniature: FASTA records far larger than the text chunk (the chunk buffer has
    to grow until it holds a whole record), a FASTQ whose reads straddle every chunk edge, no final
    newline.  Device record scanning against the host line parser."""
    import numpy as np
    rng = np.random.default_rng(5)
    fa = tmp_path / "big_records.fa"
    with open(fa, "wb") as f:
        for i, n in enumerate([700_000, 30, 250_000, 1_200_000, 5]):
            s = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)[rng.choice(9, size=n, p=[.23, .23, .23, .23, .02, .02, .01, .01, .02])].tobytes()
            f.write(b">chr%d\n" % i)
            for o in range(0, n, 60):
                f.write(s[o:o + 60] + (b"\n" if (o + 60 < n or i < 4) else b""))
    for k in ("21", "31"):
        dev = run(k, str(fa), "--format", "tsv", "--quiet", env={"KMERUST_TEXT_CHUNK_KB": "64"})
        host = run(k, str(fa), "--format", "tsv", "--quiet", env={"KMERUST_HOST_PARSE": "1"})
        assert dev.returncode == 0 and host.returncode == 0, (dev.stderr, host.stderr)
        assert tsv(dev.stdout) == tsv(host.stdout) and len(tsv(dev.stdout)) > 100_000
    fq = tmp_path / "reads.fq"
    with open(fq, "wb") as f:
        for i in range(3000):
            n = int(rng.integers(50, 3000))
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].tobytes()
            f.write(b"@r%d\n%s\n+\n%s" % (i, s, b"I" * n) + (b"\n" if i < 2999 else b""))
    dev = run("15", str(fq), "-Q", "5", "--format", "histogram", "--quiet", env={"KMERUST_TEXT_CHUNK_KB": "8"})
    host = run("15", str(fq), "-Q", "5", "--format", "histogram", "--quiet", env={"KMERUST_HOST_PARSE": "1"})
    assert dev.returncode == 0 and host.returncode == 0, (dev.stderr, host.stderr)
    assert dev.stdout == host.stdout and dev.stdout


def test_bare_cr_inside_a_fasta_line_is_an_invalid_base(tmp_path):
    """rust-bio trims only the END of a line (the reference's default reader, src/reader.rs:35-54): a bare CR in
    the middle of a sequence line stays in the record as an invalid base and breaks windows.  The device scanner
    must not join AC\\rGT to ACGT: it declines the text and the host line parser counts the file."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    fa = tmp_path / "cr.fa"
    fa.write_bytes(b">a\nACGTAC\rGTACGTTGCA\nACGTTTGA\r\n>b\nGGGCCCAAATTT\n")
    want = O.count_records([b"ACGTAC\rGTACGTTGCAACGTTTGA", b"GGGCCCAAATTT"], 4).as_str_dict(4)
    for env in ({}, {"KMERUST_HOST_PARSE": "1"}):
        r = run("4", str(fa), "--format", "tsv", "--quiet", env=env)
        assert r.returncode == 0, r.stderr
        assert tsv(r.stdout) == want


_ONE_GPU = {}


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0", "0,0,0"], ids=["2-ranks", "4-ranks", "3-ranks-pairs-route"])
def test_cli_on_several_ranks_equals_one_gpu(tmp_path, devices):
    """`kmerust --devices a,b,...` (what `--gpus N` expands to): chunks of whole records go to the ranks in turn,
    every rank counts into its own table, the library's exchange (kh_group_merge) shards the tables by hash
    range, and the output is the concatenation of the shards.  The box has one GPU, so the ranks share device 0
    (process-local transport instead of RCCL); the CLI, the chunk distribution, the merge sequence and the result
    assembly are the code a multi-GPU node runs.  Must equal the single-GPU output line for line (as a multiset)."""
    fq, fa = _write_reads(tmp_path)
    env = {"KMERUST_TEXT_CHUNK_KB": "16"}  # many chunks, so every rank gets work

    def single(*args, stdin=None):
        """The one-GPU output the ranks' must equal: the same for every `devices` (the reads are seeded), computed once."""
        key = tuple(os.path.basename(a) if os.sep in a else a for a in args)
        if key not in _ONE_GPU:
            r = run(*args, env=env, stdin=stdin)
            assert r.returncode == 0, r.stderr
            _ONE_GPU[key] = r.stdout
        return _ONE_GPU[key]

    for args in (["21", fq], ["21", fq, "-Q", "20"], ["21", fa], ["9", fa]):
        one = single(*args, "--format", "tsv", "--quiet")
        many = run(*args, "--format", "tsv", "--quiet", "--devices", devices, env=env)
        assert many.returncode == 0, many.stderr
        assert tsv(many.stdout) == tsv(one) and len(many.stdout.splitlines()) == len(one.splitlines())
        h1 = single(*args, "--format", "histogram", "--quiet")
        hn = run(*args, "--format", "histogram", "--quiet", "--min-count", "2", "--devices", devices, env=env)
        h2 = single(*args, "--format", "histogram", "--quiet", "--min-count", "2")
        assert hn.returncode == 0 and hn.stdout == h2 and h1
    # the host line parser path (stdin) on several ranks
    data = open(fa, "rb").read()
    a = run("15", "-", "--format", "tsv", "--quiet", "--devices", devices, stdin=data)
    b = single("15", "-", "--format", "tsv", "--quiet", stdin=data)
    assert a.returncode == 0 and tsv(a.stdout) == tsv(b)


def test_gpus_flag_validation():
    r = run("21", fx("simple.fa"), "--gpus", "0")
    assert r.returncode == 2 and b"at least one GPU" in r.stderr
    r = run("21", fx("simple.fa"), "--quiet", "--gpus", "1", "--format", "tsv")
    assert r.returncode == 0 and r.stdout == run("21", fx("simple.fa"), "--quiet", "--format", "tsv").stdout
