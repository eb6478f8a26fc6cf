"""CPU check of the level-1 window plan (krust_amd/csrc/window.hip.h): for every k = 11..32 and every window J = 0..15
of a lane, the bit fields the written-out window cuts out of the lane's three code words (forward strand) and their
reversed complements must be the packed k-mer of src/kmer.rs:467-471 and its reverse complement, and the level-1
digit / 32-bit payload definition must agree with the table hash.  The extraction runs on the device as v_alignbit /
v_bfe through builtins that are exactly the host arithmetic compiled here; the asm that follows it (canonical choice,
Feistel rounds, addresses) is held to the same definitions by the GPU parity tests, which run every k.  Round 6: the same
program checks that the table hash is a bijection of the 2k-bit keys for k = 1..32 (kh_unhash_n inverts kh_hash_n) and that the
arithmetic of the written-out rounds -- upper half left-aligned with the key's next bits below it, masks applied by the rounds'
own v_bitop3, v_mul_hi_u32 in rounds 2 and 4 -- modelled instruction by instruction on the host, is kh_hash_n for k = 11..32."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_window_fields_for_every_k_and_window(tmp_path):
    exe = tmp_path / "window_plan_check"
    src = os.path.join(ROOT, "tests", "window_plan_check.cpp")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    assert "WINDOW_PLAN_OK 22 x 16" in r.stdout
