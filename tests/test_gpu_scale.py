"""GPU parity at the sizes of BASELINE.json configs[3] and configs[4] (the two configurations round 1 left
untested), against the CPU oracle:

  configs[4]  k=21 on hg38 reference FASTA (long records), --format histogram
              hg38 is not on the box: `oracle_lib.synth_hg` writes an hg-shaped assembly with hg38's own
              chromosome lengths (3.09 Gbp, chr1 = 248,956,422 bp in ONE record, 60-column lines, ~50 %
              soft-masked, ~5 % N in long runs, repeat families and tandem repeats), the `kmerust` CLI counts
              the FILE with `--format histogram`, and every output line must equal the oracle's histogram of
              the same records (full CPU count in bounded-memory passes, `ko_hist_flat_radix_mt`).
  configs[3]  k=21 on 1 B x 150 bp over 8 GPUs = 125 M reads per GPU.  One GPU's shard (rank 3: reads
              375 M .. 500 M of the 1 B) on one GPU, with a capacity hint that puts the table at the
              2^31-slot sizing boundary and without any hint; then the RCCL-merge step of that size as two
              logical ranks on one device (each counts half of the shard, 32-bit heads exchange, LDS merge).

Every compute call goes through the C ABI (krust_amd.native / the kmerust binary).  Nothing reads
/root/reference."""
import os
import shutil
import subprocess
import time

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "krust_amd", "host", "kmerust")
SEED = 20260130
NCPU = max(1, min(os.cpu_count() or 1, 16))


@pytest.fixture(scope="module")
def K():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU (run through gpurun)"
    import krust_amd
    krust_amd.lib()  # ImportError if the HIP extension is missing: no silent fallback
    return krust_amd


def _roomy_dir(tmp_path, need_bytes):
    """A directory with room for the generated FASTA (the pytest tmp dir, else other scratch places)."""
    for d in (str(tmp_path), "/tmp", "/dev/shm", os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            if shutil.disk_usage(d).free > need_bytes + (1 << 30):
                return d
        except OSError:
            continue
    pytest.fail(f"no scratch directory with {need_bytes >> 20} MiB free for the generated FASTA")


# ---------------------------------------------------------------------------------------------------
# configs[4]
# ---------------------------------------------------------------------------------------------------
def test_hg38_scale_fasta_histogram_through_the_cli(tmp_path):
    lens = np.array(O.HG38_LENGTHS, dtype=np.uint64)
    total_bases = int(lens.sum())
    assert total_bases >= 1 << 31 and int(lens.max()) >= 248_000_000 and lens.size == 25
    t0 = time.time()
    flat = O.synth_hg(38, lens, nthreads=NCPU)
    # the shape the task names: ~50 % lowercase, ~5 % N
    probe = flat[:: 997]
    lower = float(np.isin(probe, np.frombuffer(b"acgt", dtype=np.uint8)).mean())
    n_share = float((probe == ord("N")).mean())
    assert 0.40 < lower < 0.55 and 0.03 < n_share < 0.08, (lower, n_share)
    d = _roomy_dir(tmp_path, total_bases + total_bases // 60 + 4096)
    path = os.path.join(d, f"hg_like_{os.getpid()}.fa")
    try:
        O.write_fasta(path, flat, lens, width=60)
        t1 = time.time()
        # (the full CPU count of the same records -- ko_hist_flat_radix_mt, eight bounded-memory passes -- has been running in
        #  a child process since the session started: tests/bg_oracle.py; a lone run of this test computes it here)
        import bg_oracle
        res = bg_oracle.collect("hg", NCPU)
        want_total, want_distinct, want_hist = res["total"], res["distinct"], [tuple(x) for x in res["hist"]]
        t2 = time.time()
        del flat
        r = subprocess.run([BIN, "21", path, "--format", "histogram", "--quiet"], capture_output=True, timeout=3000)
        t3 = time.time()
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stderr == b""
        got = [tuple(map(int, l.split(b"\t"))) for l in r.stdout.splitlines()]
        print(f"\n[hg-like] {total_bases} bases in {lens.size} records (largest {int(lens.max())}), "
              f"{want_total} k-mers, {want_distinct} distinct, {len(want_hist)} histogram lines, max count {want_hist[-1][0]}; "
              f"generate+write {t1 - t0:.1f} s, CPU oracle {res['cpu_seconds']:.1f} s on {res['threads']} threads (waited {t2 - t1:.1f} s for it), kmerust CLI {t3 - t2:.1f} s")
        assert got == sorted(got)                              # ascending by count (BTreeMap order, src/run.rs:478-480)
        assert sum(c * f for c, f in got) == want_total        # every valid window counted once
        assert sum(f for _, f in got) == want_distinct
        assert got == want_hist                                # every line, bit-exact
        assert want_hist[-1][0] > 1 << 16                      # (tandem repeats: counts beyond the dense histogram range)
        # --min-count filters BEFORE the histogram (src/run.rs:447-450)
        r = subprocess.run([BIN, "21", path, "--format", "histogram", "--quiet", "--min-count", "1000"],
                           capture_output=True, timeout=3000)
        assert r.returncode == 0, r.stderr[-2000:]
        assert [tuple(map(int, l.split(b"\t"))) for l in r.stdout.splitlines()] == [cf for cf in want_hist if cf[0] >= 1000]
    finally:
        if os.path.exists(path):
            os.remove(path)


# ---------------------------------------------------------------------------------------------------
# configs[3]
# ---------------------------------------------------------------------------------------------------
N_SHARD = 125_000_000        # reads per GPU of the 1 B-read / 8-GPU configuration
FIRST = 3 * N_SHARD          # rank 3's shard


@pytest.fixture(scope="module")
def shard_reads(K):
    """Rank 3's reads in HBM, their host copy, and the oracle's view of them: total valid windows and exact
    counts of the 1/1024 key sample."""
    import torch
    rl, k = 150, 21
    nbytes = N_SHARD * (rl + 1)
    tb = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    K.synth_reads_device(tb.data_ptr(), None, SEED, 1 << 27, rl, FIRST, N_SHARD)
    torch.cuda.synchronize()
    host = tb.cpu().numpy()
    ob, _ = O.synth_reads(SEED, 1 << 27, rl, FIRST + N_SHARD - 1000, 1000, with_qual=False)
    assert np.array_equal(host[-1000 * (rl + 1):], ob)       # the device generator at a non-zero first_read
    m = O.OracleMap()
    t0 = time.time()
    total = m.scan_flat(host, k, sample_mask=1023, nthreads=NCPU)
    print(f"\n[configs3] CPU scan of {N_SHARD} reads: {time.time() - t0:.1f} s, {total} k-mers, {len(m)} sampled keys")
    skeys, scnts = m.arrays()
    del host
    yield tb, nbytes, total, skeys, scnts
    del tb
    torch.cuda.empty_cache()


@pytest.mark.parametrize("hint", [1_350_000_000, 0], ids=["hint-2^31-boundary", "no-hint"])
def test_configs3_shard_on_one_gpu(K, shard_reads, hint):
    tb, nbytes, total, skeys, scnts = shard_reads
    with K.DeviceCounter(21, capacity_hint=hint) as dc:
        t0 = time.time()
        dc.push_device(tb.data_ptr(), None, nbytes)
        st = dc.finish()
        dt = time.time() - t0
        print(f"\n[configs3 hint={hint}] {st['kmers']} k-mers, {st['distinct']} distinct, {st['table_slots']} slots "
              f"(load {st['distinct'] / st['table_slots']:.3f}), grows={st['grows']}, batches={st['part_batches']}, "
              f"kernels {st['count_kernel_ms']:.1f} ms, wall {dt * 1e3:.0f} ms, stages "
              + ", ".join(f"{n}={v:.1f}" for n, v in st["stage_ms"].items() if v > 0))
        assert st["kmers"] == total                                   # every valid window counted once
        assert st["part_batches"] >= 1                                # the partitioned path took it
        assert st["distinct"] <= 0.8 * st["table_slots"]
        hist = dc.histogram()
        assert sum(f for _, f in hist) == st["distinct"] == dc.result_size()
        assert sum(c * f for c, f in hist) == total
        assert [c for c, _ in hist] == sorted(c for c, _ in hist)
        assert np.array_equal(dc.lookup(skeys), scnts)                # sampled keys: exact counts
        assert not dc.lookup(skeys | np.uint64(1 << 63)).any()        # (bit 63 is never set in a 21-mer: absent keys read 0)
    import torch
    torch.cuda.empty_cache()


def test_configs3_merge_step_as_two_logical_ranks(K, shard_reads, monkeypatch):
    """The RCCL merge of configs[3]-sized tables with the collective taken out: two logical ranks each count
    one half of the shard (62.5 M reads, table geometry of the full configuration), export 32-bit heads by
    owner, reset, become hash-range shards and merge the two senders' segments -- the call sequence of
    krust_amd/distributed.py with the all-to-all replaced by pointers.  The union of the two shards must hold
    the oracle's counts for the whole 125 M reads."""
    import torch
    tb, nbytes, total, skeys, scnts = shard_reads
    monkeypatch.setenv("KMERHIP_PART_BUDGET_GB", "40")  # two contexts alive at once: bound their scratch
    half = (N_SHARD // 2) * 151
    spans = [(0, half), (half, nbytes - half)]
    hint = 1_350_000_000  # the same (full-shard) hint on both: equal table sizes, as on a real node
    ranks = [K.DeviceCounter(21, capacity_hint=hint) for _ in spans]
    try:
        exports, nreg, sent = [], None, 0
        for dc, (off, n) in zip(ranks, spans):
            dc.push_device(tb.data_ptr() + off, None, n)
            st = dc.finish()
            R = st["table_slots"] // 4096
            nreg = R if nreg is None else nreg
            assert R == nreg
            heads = torch.empty(2 * st["distinct"], dtype=torch.int32, device="cuda")
            rc = torch.empty(R, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            res = dc.export_regions_heads_device(2, heads.data_ptr(), heads.numel(), rc.data_ptr(), R)
            assert res is not None, "k=21 at this table size is representable as 32-bit heads"
            parts, R2 = res
            assert R2 == R and int(parts.sum()) >= st["distinct"]
            sent += int(parts.sum())
            exports.append((heads, rc, np.concatenate([[0], np.cumsum(parts)]).astype(np.int64), st))
        assert sum(e[3]["kmers"] for e in exports) == total
        per_r = nreg // 2
        merged_distinct = merged_total = 0
        got = np.zeros(skeys.size, dtype=np.uint64)
        for o, dc in enumerate(ranks):
            dc.reset()
            dc.set_shard(o, 2)
            dc.merge_regions_heads_device(nreg, [e[0].data_ptr() + 4 * int(e[2][o]) for e in exports],
                                          [e[1].data_ptr() + 4 * per_r * o for e in exports])
            st = dc.finish()
            hist = dc.histogram()
            assert sum(f for _, f in hist) == st["distinct"]
            merged_distinct += st["distinct"]
            merged_total += sum(c * f for c, f in hist)
            mine = np.array([K.owner(int(x), 21, 2) == o for x in skeys[:20000]])
            part = dc.lookup(skeys)
            assert np.array_equal(part[:20000][~mine], np.zeros(int((~mine).sum()), dtype=np.uint64))  # not its keys
            got += part
        assert merged_total == total                      # conservation across the merge
        assert np.array_equal(got, scnts)                 # every sampled key on exactly one owner, exact count
        print(f"\n[configs3 merge] {sent} heads exchanged, {merged_distinct} distinct after the merge")
    finally:
        for dc in ranks:
            dc.close()
        torch.cuda.empty_cache()


def test_a_world_of_one_rank_merges_the_full_size_table(K, shard_reads):
    """kh_merge_across over RCCL with ONE rank: every key is the rank's own, so the merge must leave the table's content
    as it was -- through the real transport (a send to itself), in pieces, at configs[3]'s size.  Round 4: a message of
    2^30 bytes or more arrived half (the keys of the upper half of every piece were gone, silently, once a table held more
    than 2^30 keys): messages are now 256 MiB at most (exchange.hip, xp_alltoallv).  Also: the rank -- a context with a
    communicator -- still counts the shard in ONE partitioned batch (the merge gives the partition buffers back itself)."""
    tb, nbytes, total, skeys, scnts = shard_reads
    with K.DeviceCounter(21) as dc:
        dc.comm_init(1, 0, K.comm_unique_id())
        dc.push_device(tb.data_ptr(), None, nbytes)
        st = dc.finish()
        assert st["kmers"] == total and st["part_batches"] == 1, st
        assert st["distinct"] > (1 << 30)              # (the size at which the transport's limit showed)
        info = dc.merge_across()
        assert info["path"].startswith("regions") and info["owned_distinct"] == st["distinct"], info
        st2 = dc.finish()
        assert st2["distinct"] == st["distinct"]
        hist = dc.histogram()
        assert sum(f for _, f in hist) == st["distinct"] and sum(c * f for c, f in hist) == total
        assert np.array_equal(dc.lookup(skeys), scnts)
    import torch
    torch.cuda.empty_cache()
