"""CPU-side tests of the C++ host layer (krust_amd/host): command-line surface, reader and KMIX
index, mirroring the reference's tests/integration_tests.rs where no counting is involved.
Everything that counts k-mers needs the GPU and lives in test_gpu_cli.py."""
import gzip
import os
import struct
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "krust_amd", "host", "kmerust")


@pytest.fixture(scope="module", autouse=True)
def _build_host():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "krust_amd", "csrc")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "krust_amd", "host")], stdout=subprocess.DEVNULL)


def run(*args, stdin=None):
    return subprocess.run([BIN, *args], input=stdin, capture_output=True, timeout=60)


def test_help_and_version():
    r = run("--help")                                   # integration_tests.rs:16-25
    assert r.returncode == 0 and b"Usage: kmerust" in r.stdout and b"--min-quality" in r.stdout
    r = run("--version")                                # integration_tests.rs:28-35
    assert r.returncode == 0 and r.stdout.startswith(b"kmerust ")


def test_missing_and_bad_k():
    assert run().returncode == 2                        # integration_tests.rs:38-44 (clap usage error)
    for bad, msg in (("0", b"k-mer length must be at least 1"), ("33", b"k-mer length must be at most 32"),
                     ("abc", b"'abc' is not a valid number"), ("-5", b"unexpected argument")):
        r = run(bad, "x.fa")                            # src/cli.rs:103-114, integration_tests.rs:78-102
        assert r.returncode == 2 and msg in r.stderr, (bad, r.stderr)


def test_missing_file_exit_1():
    r = run("3", "/nonexistent/file.fa")                # src/main.rs:58-67, integration_tests.rs:105-111
    assert r.returncode == 1 and b"File not found: /nonexistent/file.fa" in r.stderr


def test_bad_enum_values():
    assert run("3", "x.fa", "--format", "xml").returncode == 2
    assert run("3", "x.fa", "-i", "bam").returncode == 2
    assert run("3", "x.fa", "-Q", "300").returncode == 2  # u8 in the reference (src/cli.rs:68-69)


def parse(path, *extra):
    r = run("__parse", path, *extra)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_reader_fixtures(fixtures_dir):
    fa = parse(os.path.join(fixtures_dir, "simple.fa"))
    assert b"ACGTACGT\nGATTACA\n" in fa and fa.endswith(b"RECORDS 2\n")
    assert parse(os.path.join(fixtures_dir, "simple.fa.gz")) == fa          # gzip == plain (tests/gzip_tests.rs)
    fq = parse(os.path.join(fixtures_dir, "low_quality.fq"), "--qual")
    assert b"ACGTACGT\nGATTACA\nQUAL\nIIII!!!!\nIIIIIII\n" in fq
    assert b"QUAL" not in parse(os.path.join(fixtures_dir, "low_quality.fq"))
    assert parse(os.path.join(fixtures_dir, "simple.fq.gz")).count(b"\n") == fa.count(b"\n")
    assert b"NNNGATTACANNN" in parse(os.path.join(fixtures_dir, "with_n.fq"), "--qual")
    assert b"AAAa\n" in parse(os.path.join(fixtures_dir, "soft_masked.fa"))  # case preserved; masking is the kernel's job


def test_reader_edge_cases(tmp_path):
    def w(name, data):
        p = tmp_path / name
        p.write_bytes(data)
        return str(p)
    assert parse(w("empty.fa", b"")).endswith(b"RECORDS 0\n")               # library_tests.rs:178-185
    assert b"RECORDS 1" in parse(w("hdr.fa", b">seq\n"))                    # header only: one empty record (191-196)
    out = parse(w("multi.fa", b">s\nACG\nTAC\n>t\r\nGG\r\nCC\r\n"))         # multi-line + CRLF
    assert b"ACGTAC\nGGCC\n" in out
    out = parse(w("wrap.fq", b"@r1\nACGT\nACGT\n+\nIIII\nIIII\n@r2\nGG\n+r2\n##\n"), "--qual")
    assert b"ACGTACGT\nGG\nQUAL\nIIIIIIII\n##\n" in out
    out = parse(w("at.fq", b"@r1\nACGT\n+\n@@@@\n@r2\nAC\n+\nII\n"), "--qual")  # '@' as a quality char
    assert b"ACGT\nAC\nQUAL\n@@@@\nII\n" in out
    assert parse(w("unk.txt", b">s\nAC\n")).endswith(b"RECORDS 1\n")         # unknown extension -> FASTA (format.rs:66-69)
    assert parse(w("reads.FASTQ.gz", gzip.compress(b"@r\nAC\n+\nII\n"))).endswith(b"RECORDS 1\n")  # .gz stripped, case folded
    for name, data in (("bad.fa", b"ACGT\n"), ("bad.fq", b"ACGT\n"), ("short.fq", b"@r\nACGT\n+\nII\n"),
                       ("noplus.fq", b"@r\nACGT\n")):
        r = run("__parse", w(name, data))
        assert r.returncode == 1 and b"failed to parse sequence record" in r.stderr


def kmix(k, pairs, version=1, magic=b"KMIX", corrupt=False):
    body = magic + bytes([version, k]) + struct.pack("<Q", len(pairs))
    for key, cnt in pairs:
        body += struct.pack("<QQ", key, cnt)
    crc = zlib.crc32(body) ^ (1 if corrupt else 0)
    return body + struct.pack("<I", crc)


def test_query_on_index(tmp_path):
    # format: src/index.rs:7-23; query: src/main.rs:233-281
    p = tmp_path / "t.kmix"
    p.write_bytes(kmix(7, [(9156, 42), (0, 7)]))                            # GATTACA=9156 (canonical), AAAAAAA=0
    assert run("query", str(p), "GATTACA").stdout == b"42\n"
    assert run("query", str(p), "tgtaatc").stdout == b"42\n"                # reverse complement, lower case
    assert run("query", str(p), "TTTTTTT").stdout == b"7\n"
    assert run("query", str(p), "ACGTACG").stdout == b"0\n"                 # absent -> 0
    r = run("query", str(p), "ACGT")
    assert r.returncode == 1 and b"k-mer length mismatch: query has 4 bases, index has k=7" in r.stderr
    r = run("query", str(p), "ACGTNCG")
    assert r.returncode == 1 and b"invalid base 'N' (0x4e) at position 4" in r.stderr
    gz = tmp_path / "t.kmix.gz"
    gz.write_bytes(gzip.compress(kmix(7, [(9156, 42)])))                    # .gz transparently (index.rs:25-28)
    assert run("query", str(gz), "GATTACA").stdout == b"42\n"


def test_index_validation(tmp_path):
    cases = [(kmix(7, [(1, 1)], corrupt=True), b"checksum mismatch"), (kmix(7, [], magic=b"NOPE"), b"invalid magic bytes"),
             (b"KMIX\x01", b"file too small"), (kmix(7, [], version=9), b"unsupported version 9"),
             (kmix(40, []), b"invalid k-mer length")]
    for i, (data, msg) in enumerate(cases):
        p = tmp_path / f"bad{i}.kmix"
        p.write_bytes(data)
        r = run("query", str(p), "GATTACA")
        assert r.returncode == 1 and b"Failed to load index" in r.stderr and msg in r.stderr, (i, r.stderr)
    r = run("query", str(tmp_path / "missing.kmix"), "GATTACA")
    assert r.returncode == 1
    # data size mismatch: count says 2, one pair present
    body = b"KMIX" + bytes([1, 7]) + struct.pack("<Q", 2) + struct.pack("<QQ", 1, 1)
    p = tmp_path / "size.kmix"
    p.write_bytes(body + struct.pack("<I", zlib.crc32(body)))
    assert b"data size mismatch (expected 32 bytes, got 16 bytes)" in run("query", str(p), "GATTACA").stderr


def test_no_gpu_means_loud_failure(fixtures_dir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = run("3", os.path.join(fixtures_dir, "simple.fa"), "-q")
    assert r.returncode == 1 and b"no usable HIP device" in r.stderr and r.stdout == b""
