#!/usr/bin/env python3
"""bench.py -- canonical k-mers/s at k=21 on synthetic 150 bp reads (BASELINE.json metric).

A "step" = one full pass of the counting hot path over this GPU's batch of synthetic reads,
already resident in HBM: table reset, encode + mask + canonicalise + upsert of every window
(kh_push_device), completion (kh_finish) and, for N>1, the key-partitioned RCCL merge of the
per-GPU tables (kh_merge_across).

Workload: N = 1 -> S100M = 100 M x 150 bp (the configuration the metric is quoted on; it fits one
MI355X: 15.1 GB reads + 34 GB table).  N > 1 -> BASELINE configs[3]'s per-GPU share, 125 M x 150 bp
per GPU (rank r counts reads [r x 125 M, (r + 1) x 125 M): at N = 8 that IS the 1 B-read set), weak scaling.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) -- SMALL: at most 6 KB, so that it fits whole into the
8 KB tail the driver keeps of stdout (round 4's 23.5 KB line did not, and the driver recorded `parsed: null`).  The line
(contract_line()) carries
  roofline      algorithmic HBM bytes of the counting kernels / their HIP-event duration, vs 8 TB/s; the kernels
                are NAMED from what ran (kh_stats.stage_ms), and tests/test_docs_drift.py holds the names to the
                source and to the committed rocprofv3 kernel trace
  verify        (N = 1, after the timed loop, outside it) exact counts of a 1/1024 key sample and the k-mer total
                against the CPU oracle's scan of the same reads: the line says itself whether it is bit-exact
  cpu_baseline  the krust-equivalent threaded CPU port (oracle/) timed on this host's cores on a bounded sample
                of the same reads, `optimised_value` beside it (rank 0, N=1 only).  A reported baseline only.
  configs       BASELINE configs[1..4] (and a k = 19 twin of the headline): one flat row each
  cli           the kmerust command line on S10M and S100M FASTQ files: wall seconds
The FULL report (every sub-result's roofline, phase walls, samples, explanations) goes to gpurun_out/bench_full.json
(BENCH_FULL_PATH overrides) and, as a pointer, `full_report` in the line.

  --group N     N logical ranks as THREADS of this one process on ONE device (kh_group: the in-process hub instead of RCCL,
                which refuses two ranks on one device): every line of the N > 1 accounting -- config.merge, per_rank,
                conserved, rccl_nranks -- runs on a 1-GPU box.  A dry run of the code path, not a scaling measurement.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20260130
GENOME_LEN = 1 << 27
READ_LEN = 150
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
READS_N1 = 100_000_000   # the metric's configuration
READS_NX = 125_000_000   # configs[3]: 1 B reads over 8 GPUs
# hg38's chromosome lengths (chr1..22, X, Y, M): the record lengths of the configs[4] sub-result (data, also in tests/oracle_lib.py)
HG38_LENGTHS = (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
                133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
                58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (150 bp each); 0 = 100 M at N = 1, 125 M (configs[3]) at N > 1")
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--min-quality", type=int, default=None, help="enable the quality stream + -Q masking")
    ap.add_argument("--capacity-hint", type=int, default=0, help="expected distinct k-mers per GPU (0 = estimate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--verify", action="store_true", help="(default at N = 1) check a 1/1024 key sample against the CPU oracle")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--group", type=int, default=0,
                    help="N thread ranks on ONE device through kh_group_create / kh_group_merge (the N > 1 accounting on a 1-GPU box)")
    ap.add_argument("--force-merge", action="store_true",
                    help="N = 1 only: run the N > 1 step -- count, then kh_merge_across over a REAL RCCL communicator of one rank -- and fill "
                         "config.merge / per_rank / single_gpu_same_share as a torchrun launch would (pre-flight of the 8-GPU line on one GPU)")
    ap.add_argument("--no-hint", action="store_true", help="create the context with capacity_hint = 0 (KmerMap::new() takes none): the table is sized from the level-1 sample")
    ap.add_argument("--hg", action="store_true", help="run configs[4] alone (hg-shaped FASTA text -> histogram): for profiling that workload")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the end-to-end (host buffers in, results out) figures and the configs sub-results")
    return ap.parse_args()


HINT_SOURCE = ("estimate_distinct(): genome length + reads x 150/256 substitutions x k (L-k+1)/L novel k-mers each x 0.92 -- an a-priori "
               "figure from the generator's parameters.  Since round 4 a hint only sizes the first allocation: every fresh batch "
               "sizes the table itself from the distinct keys of a few level-1 partitions (the `unhinted` twin runs without any)")


def estimate_distinct(reads, k, world, with_qual=False):
    """Genomic canonical k-mers + error k-mers (1/256 substitutions, ~k (L-k+1)/L novel each, less what N bases and
    neighbouring errors take: x 0.92 measured on S100M / S1B shards); with -Q masking only windows free of
    low-quality bases (~2.5 % of the synthetic qualities) survive."""
    novel_per_read = READ_LEN * (1.0 / 256.0) * k * (READ_LEN - k + 1) / READ_LEN
    if with_qual:
        novel_per_read *= 0.975 ** k
    return int(GENOME_LEN + reads * novel_per_read * 0.92)


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU
    box shows 256 logical CPUs but grants 16 CPUs' worth of time; more threads than that only
    add lock contention)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(host_bases, k, target_seconds):
    """krust-equivalent port (oracle/ko_count_records_mt2: one task per record, literal per-window algorithm,
    sharded lock-per-shard map with 4 x threads shards, the lock a spin-then-park mutex like DashMap's, every
    shard pre-sized so that nothing rehashes under a lock) on a bounded sample of the same reads.  The thread
    count is the best of a short probe (a sharded-lock map does not scale to every core count); `cores` is what
    the timed run used.  Reported with it: the same port on ONE thread, the share of thread time spent waiting
    for shard locks, and the lock-free radix formulation -- so the JSON says where the port's time goes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    ncpu = usable_cpus()
    stride = READ_LEN + 1
    avail = host_bases.size // stride
    per_read = READ_LEN - k + 1

    def run(n, threads):
        offs = np.arange(n, dtype=np.uint64) * stride
        lens = np.full(n, READ_LEN, dtype=np.uint32)
        m = O.OracleMap()
        st = {}
        t0 = time.perf_counter()
        # nearly every k-mer of a small sample is new (coverage << 1): size the shards for that
        cnt = m.count_records_mt(host_bases[: n * stride], offs, lens, k, nthreads=threads, expect_distinct=n * per_read, stats=st)
        return cnt, time.perf_counter() - t0, st

    probe_n = min(avail, 60_000)
    best = None
    for threads in sorted({max(1, ncpu // 2), ncpu, min(2 * ncpu, os.cpu_count() or ncpu)}):
        cnt, dt, _ = run(probe_n, threads)
        if best is None or cnt / dt > best[0]:
            best = (cnt / dt, threads)
    rate, threads = best
    n = int(min(avail, max(probe_n, 0.6 * target_seconds * rate / per_read)))
    cnt, dt, st = run(n, threads)
    n1 = int(min(avail, max(10_000, n // (2 * threads))))
    cnt1, dt1, _ = run(n1, 1)
    out = {"value": cnt / dt, "unit": "k-mers/s", "cores": threads, "kind": "port",
           "sample": f"first {n} reads of the same synthetic set ({cnt} k-mers, {dt:.1f} s) on {threads} threads "
                     f"({ncpu} usable CPUs by cgroup quota, {os.cpu_count()} visible); krust-equivalent C port "
                     "(oracle/kmer_oracle.c ko_count_records_mt2), not krust itself",
           "per_thread": cnt / dt / threads,
           "single_thread": {"value": cnt1 / dt1, "sample": f"first {n1} reads, 1 thread, {dt1:.1f} s"},
           "lock_wait_share": st["wait_ns"] / 1e9 / (dt * threads),
           "contended_acquisitions": st["contended"] / max(1, st["upserts"]),
           "rehashes_under_lock": st["rehashes_under_lock"]}
    # an optimised CPU formulation beside it, so the GPU figure is not flattered by the port's
    # allocations and locks: rolling registers + two-phase radix count (no locks, no merge)
    n2 = min(avail, 3_000_000)
    t0 = time.perf_counter()
    tot, distinct, _ = O.count_flat_radix(host_bases[: n2 * stride], k, nthreads=ncpu)
    dt2 = time.perf_counter() - t0
    out["optimised_cpu"] = {"value": tot / dt2, "unit": "k-mers/s", "cores": ncpu,
                            "sample": f"first {n2} reads ({tot} k-mers, {distinct} distinct, {dt2:.1f} s): rolling scan + "
                                      "two-phase radix count (oracle ko_count_flat_radix_mt)"}
    # the number a reader should see first when asking "how fast is a CPU at this": beside `value`, not below it
    out["optimised_value"] = out["optimised_cpu"]["value"]
    out["optimised_cores"] = ncpu
    ratio = out["optimised_cpu"]["value"] / out["value"]
    out["port_vs_optimised"] = {
        "ratio": ratio,
        "why": ("the port keeps the reference's shape on purpose: every window is copied and validated (O(k)), packed (O(k)) "
                "and canonicalised by a two-ended compare (O(k)), then upserted under a shard lock into a table far larger "
                "than the caches -- one lock cache line handed between cores and one DRAM miss per k-mer.  "
                f"On one thread it does {cnt1 / dt1 / 1e6:.2f} M k-mers/s; on {threads} threads {cnt / dt / threads / 1e6:.2f} M/s "
                f"per thread, {100 * st['wait_ns'] / 1e9 / (dt * threads):.0f} % of the thread time waiting for shard locks "
                f"({100 * st['contended'] / max(1, st['upserts']):.1f} % of the acquisitions contended, no rehash under a lock).  "
                "The radix formulation rolls the window in registers and counts partitions in private tables: no locks, "
                "cache-sized working sets.  Nothing here ties the port's contention profile to DashMap 5.5.3's own "
                "(src/run.rs:489-498): it is context, not credit.")}
    return out


# ---- which kernels ran, by name -------------------------------------------------------------------------------
# One entry per stage of kh_stats.stage_ms (krust_amd/native.py STAGES).  tests/test_docs_drift.py checks every name
# here against the __global__ kernels of krust_amd/csrc and against the committed rocprofv3 trace of this command.
def payload_bytes(k):
    """Bytes per k-mer in the partition buffers at bench table sizes (>= 2^10 regions): 4 while the 2k - 10 hash bits
    below the level-1 digit fit 32 (k <= 21), the 8-byte key otherwise (make_geom, krust_amd/csrc/kmerhip.hip)."""
    return 4 if 2 * k - 10 <= 32 else 8


def kernels_of(stages, k):
    """stage -> kernel name, for the stages that took time in this step."""
    pay = payload_bytes(k)
    arena = stages.get("level2", 0) > 0 and stages.get("level2_count", 0) == 0
    names = {"direct": "count_direct_kernel",
             "level1": "part1_bins_kernel" if pay == 4 else "part1_bins64_kernel",
             "level2_count": "part2_count_kernel",
             "level2": "part2_arena_kernel" if arena else "part2_scatter_lines_kernel",
             "region": "region_count_kernel32" if pay == 4 else "region_count_kernel64"}
    return {s: names[s] for s in stages if s in names and stages[s] > 0}


def stage_min_bytes(st, nbytes_in, k, stages):
    """HBM bytes each stage must move at the least (its own algorithmic traffic): DESIGN.md section 4.2's table."""
    pay = payload_bytes(k)
    km, slots = st["kmers"], st["table_slots"]
    return {"direct": nbytes_in + 24 * km + 8 * st["distinct"],
            "level1": nbytes_in + pay * km,
            "level2_count": pay * km,
            "level2": 2 * pay * km,
            "region": pay * km + 16 * slots}


def roofline_of(st, nbytes_in, kernel_ms, stage_ms, k):
    """HBM roofline of one step: algorithmic bytes (every input byte once +
    24 B per valid k-mer + 8 B per distinct key) over the HIP-event time of the step's counting kernels."""
    alg = nbytes_in + 24 * st["kmers"] + 8 * st["distinct"]
    achieved = alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    stages = {n: v for n, v in stage_ms.items() if v > 0}
    kern = kernels_of(stages, k)
    sb = stage_min_bytes(st, nbytes_in, k, stages)
    dom = max((n for n in stages if n in kern), key=lambda n: stages[n], default=None)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "traffic_frac": None, "alg_bytes_per_step": alg, "kernel_ms_per_step": kernel_ms,
            "formula": {"text": "bytes_in x (1 + has_qual) + 24 B x valid k-mers + 8 B x distinct keys (SURVEY.md 8d(i): encode + insert; "
                                "its 16 B x distinct compaction term is NOT included: the timed step ends at kh_finish, the compaction "
                                "kernel runs in the end_to_end / result legs)",
                        "bytes_in": int(nbytes_in), "per_kmer": 24, "kmers": int(st["kmers"]), "per_distinct": 8, "distinct": int(st["distinct"]),
                        "compaction_term_included": False},
            "kernel": " + ".join(kern[s] for s in ("direct", "level1", "level2_count", "level2", "region") if s in kern),
            "level2_path": None if "level2" not in kern else ("arena" if kern["level2"] == "part2_arena_kernel" else "exact"),
            "stages_ms": stages, "kernels": kern,
            "dominant": None if dom is None else {
                "stage": dom, "kernel": kern[dom], "ms": stages[dom], "min_bytes": sb[dom],
                "achieved": sb[dom] / (stages[dom] * 1e-3) / 1e9,
                "frac": sb[dom] / (stages[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS}}


def verify_reads(dc, st, host, hq, k, min_quality):
    """Exact counts of the 1/1024 key sample (mix(key) & 1023 == 0) and the k-mer total against the CPU oracle's
    scan of the same reads.  Outside every timed region; the oracle is the checker here, nothing else."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    t0 = time.perf_counter()
    m = O.OracleMap()
    tot = m.scan_flat(host, k, qual=hq, min_quality=min_quality, sample_mask=1023, nthreads=2 * usable_cpus())
    skeys, scnts = m.arrays()
    ok = bool(tot == st["kmers"] and np.array_equal(dc.lookup(skeys), scnts))
    return {"ok": ok, "sampled_keys": len(m), "cpu_total_kmers": int(tot), "gpu_total_kmers": int(st["kmers"]),
            "what": "every key with mix(key) & 1023 == 0: exact count equality, plus the k-mer total (CPU oracle scan_flat; "
                    "vs krust only through that restatement -- the reference's own vectors are <= 32 bases, DESIGN.md section 2)",
            "cpu_seconds": time.perf_counter() - t0}


def measured_traffic(rf, reads, k, min_quality):
    """roofline.traffic / traffic_frac of a sub-result from profiles/hbm_traffic.json "configs" (separate rocprofv3 --pmc
    passes of the same workload, committed -- tools/hbm_traffic.py --also; reads = 0: the hg-shaped text)."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            tj = json.load(f)
        for name, c in (tj.get("configs") or {}).items():
            if c.get("reads") == reads and c.get("k") == k and c.get("min_quality") == min_quality:
                ms = rf.get("kernel_ms_per_step")
                rf["traffic"] = c["bytes_per_step"]
                rf["traffic_source"] = f"profiles/{c.get('tag')}_summary.json (rocprofv3 --pmc passes of this workload, committed; not re-measured in this run)"
                rf["traffic_frac"] = (c["bytes_per_step"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms else None
                return
    except Exception:
        pass


def sub_config(krust_amd, torch, dev, local_rank, name, reads, k, min_quality, first=0, hint=None, verify=False, steps=1):
    """One BASELINE.json configuration as a sub-result: its own reads, its own context, one warm-up and `steps`
    timed steps (reset + push_device + finish), with its own roofline.  hint = 0: no capacity hint (KmerMap::new() takes
    none, src/run.rs:494-498): the table is sized from a sample of level-1 partitions."""
    stride = READ_LEN + 1
    nbytes = reads * stride
    with_qual = min_quality is not None
    tb = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    tq = torch.empty(nbytes, dtype=torch.uint8, device=dev) if with_qual else None
    krust_amd.synth_reads_device(tb.data_ptr(), tq.data_ptr() if with_qual else None, SEED, GENOME_LEN, READ_LEN, first, reads,
                                 device=local_rank, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    hint_source = "none (capacity_hint = 0)" if hint == 0 else "given" if hint is not None else HINT_SOURCE
    if hint is None:
        hint = estimate_distinct(reads, k, 1, with_qual)
    dc = krust_amd.DeviceCounter(k, min_quality=min_quality, capacity_hint=hint, device=local_rank)
    try:
        dt = 0.0
        for rep in range(1 + steps):
            dc.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dc.push_device(tb.data_ptr(), tq.data_ptr() if with_qual else None, nbytes)
            st = dc.finish()
            torch.cuda.synchronize()
            if rep:
                dt += (time.perf_counter() - t0) / steps
        rf = roofline_of(st, nbytes * (2 if with_qual else 1), st["count_kernel_ms"], st["stage_ms"], k)
        measured_traffic(rf, reads, k, min_quality)
        out = {"workload": name, "k": k, "reads": reads, "first_read": first, "min_quality": min_quality, "capacity_hint": hint,
               "capacity_hint_source": hint_source, "value": st["kmers"] / dt, "unit": "k-mers/s",
               "ms_per_step": dt * 1e3, "steps": steps, "kmers_per_step": int(st["kmers"]), "distinct": int(st["distinct"]),
               "table_load": st["distinct"] / st["table_slots"],
               "table_slots": int(st["table_slots"]), "table_grows": int(st["grows"]), "part_batches": int(st["part_batches"]),
               "dtype": "u64", "roofline": rf}
        if verify:
            out["verify"] = verify_reads(dc, st, tb.cpu().numpy(), tq.cpu().numpy() if with_qual else None, k, min_quality)
        return out
    finally:
        dc.close()
        del tb, tq
        torch.cuda.empty_cache()


def hg_like_fasta_device(torch, dev, seed=38):
    """BASELINE configs[4] without hg38 on the box: an hg-SHAPED assembly as FASTA text, generated in HBM with torch
    (plumbing: device memory and a random generator).  25 records with hg38's own chromosome lengths (3.09 Gbp, chr1 =
    248,956,422 bp in ONE record), 60-column lines, ~47 % lower case in blocks, ~4 % N in long runs (telomeres, a
    centromere block), interspersed repeats pasted from a shared 4 Mbp pool (copy numbers up to ~10^5 for the pool's
    hottest elements) and tandem repeats.  Returns (text, flat): the FASTA bytes, and the same records flat (record,
    '\\n', record, ...) for the oracle's check.  (tests/test_gpu_scale.py holds the parity test of this configuration:
    the CLI on a FILE from the oracle's own generator, every histogram line.)"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    pool = lut[torch.randint(0, 4, (4 << 20,), device=dev, generator=g)]
    ar300 = torch.arange(300, device=dev)
    texts, flats = [], []
    nl = torch.full((1,), 10, dtype=torch.uint8, device=dev)
    for r, L in enumerate(HG38_LENGTHS):
        rec = lut[torch.randint(0, 4, (L,), device=dev, generator=g)]
        if L > 100_000:
            # interspersed repeats: one 300-base element per ~2.5 kbp; source positions skewed towards the pool's start
            m = L // 2500
            dst = torch.randint(0, L - 300, (m,), device=dev, generator=g)
            u = torch.rand(m, device=dev, generator=g)
            src = ((u ** 6) * ((4 << 20) - 300)).long() // 300 * 300
            rec[(dst[:, None] + ar300).reshape(-1)] = pool[(src[:, None] + ar300).reshape(-1)]
            # tandem repeats: 6-base units x 500 at one site per ~5 Mbp
            for p in torch.randint(0, L - 3000, (max(1, L // 5_000_000),), device=dev, generator=g).tolist():
                rec[p:p + 3000] = rec[p:p + 6].repeat(500)
            # soft-masked blocks of 2 KiB (~47 %) and N runs (telomeres + a centromere block of ~3 %)
            blocks = (torch.rand((L + 2047) // 2048, device=dev, generator=g) < 0.47).repeat_interleave(2048)[:L]
            rec |= blocks.to(torch.uint8) << 5
            rec[:10_000] = ord("N")
            rec[L - 10_000:] = ord("N")
            c0 = L // 3
            rec[c0:c0 + L // 33] = ord("N")
        flats += [rec, nl]
        head = torch.tensor(list(f">chr{r + 1} hg-like {L} bp\n".encode()), dtype=torch.uint8, device=dev)
        full = L // 60
        body = torch.cat([rec[:full * 60].view(full, 60), nl.expand(full, 1)], dim=1).reshape(-1)
        texts += [head, body] + ([rec[full * 60:], nl] if L % 60 else [])
    return torch.cat(texts), torch.cat(flats)


def sub_config_hg(krust_amd, torch, dev, local_rank, k=21, verify=True):
    """configs[4]: the hg-shaped FASTA resident in HBM as TEXT -> kh_push_text_device (device-side record scan,
    wrapped lines joined) -> kh_finish -> kh_histogram (what `--format histogram` prints), with a check of the
    k-mer total and the 1/1024 key sample against the oracle's scan of the same records."""
    name = "configs[4] hg-shaped FASTA (hg38's 25 record lengths, 3.09 Gbp, 60-column lines), k=21, histogram"
    text, flat = hg_like_fasta_device(torch, dev)
    if text.data_ptr() & 15:
        raise RuntimeError("text tensor is not 16-byte aligned")
    torch.cuda.synchronize()
    dc = krust_amd.DeviceCounter(k, capacity_hint=0, device=local_rank)   # no hint: the CLI never has one
    try:
        steps = 3  # (one warm-up, then the mean of three steps: a single step's histogram read-back varied by 2 ms from run to run)
        d_count = d_hist = 0.0
        for rep in range(1 + steps):
            dc.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dc.push_text_device(text.data_ptr(), text.numel(), "fasta")
            st = dc.finish()
            t1 = time.perf_counter()
            hist = dc.histogram()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if rep:
                d_count += (t1 - t0) / steps
                d_hist += (t2 - t1) / steps
        t0, t1, t2 = 0.0, d_count, d_count + d_hist
        rf = roofline_of(st, text.numel(), st["count_kernel_ms"], st["stage_ms"], k)
        measured_traffic(rf, 0, k, None)  # (the counting kernels' share of the profiled step: scan and histogram kernels apart)
        out = {"workload": name, "k": k, "text_bytes": int(text.numel()), "records": len(HG38_LENGTHS), "largest_record": max(HG38_LENGTHS),
               "value": st["kmers"] / (t2 - t0), "unit": "k-mers/s", "ms_per_step": (t2 - t0) * 1e3, "steps": steps,
               "count_ms": (t1 - t0) * 1e3, "histogram_ms": (t2 - t1) * 1e3, "text_scan_ms": st["text_scan_ms"],
               "kmers_per_step": int(st["kmers"]), "distinct": int(st["distinct"]), "table_slots": int(st["table_slots"]),
               "table_grows": int(st["grows"]), "part_batches": int(st["part_batches"]),
               "histogram_lines": len(hist), "max_count": int(hist[-1][0]) if hist else 0,
               "histogram_consistent": bool(sum(c * f for c, f in hist) == st["kmers"] and sum(f for _, f in hist) == st["distinct"]),
               "dtype": "u64", "roofline": rf}
        del text
        if verify:
            out["verify"] = verify_reads(dc, st, flat.cpu().numpy(), None, k, None)
        return out
    finally:
        dc.close()
        del flat
        torch.cuda.empty_cache()


def end_to_end(dc, tb, torch, k):
    """host buffers on both sides, never `value`.  The rank's reads as PAGEABLE host memory ->
    kh_push (pinned staging, H2D on a copy stream overlapped with counting) -> kh_finish, then the two result
    forms: every (key, count) pair copied back into fresh host arrays (kh_result_copy), or the count-of-counts
    histogram computed on the device (kh_histogram) -- what `kmerust --format histogram` needs."""
    import numpy as np
    t0 = time.perf_counter()
    host = tb.cpu().numpy()  # (setting the stage, not timed: a host application HAS its reads in host memory)
    t_stage = time.perf_counter() - t0
    # Twice: a context's FIRST host push allocates what every later one reuses -- two accumulation buffers of up to 8 GiB on a
    # device whose memory the resident benchmark's partition buffers have just filled, 256 MiB of pinned staging -- 0.3-0.4 s that
    # rounds 1-5 booked as transfer rate (20.6-24.7 GB/s "pageable" beside 44.5 "pinned", whose run came second and found the
    # buffers there; tools/ubench/stage_probe.hip: the staging pipeline itself runs at the link's 56 GB/s from 4 threads on).
    # The rate reported is the second push's; the first one's wall time stands beside it.
    t_first = None
    for _ in range(2):
        dc.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dc.push(host)
        st = dc.finish()
        t_push = time.perf_counter() - t0
        if t_first is None:
            t_first = t_push
    t0 = time.perf_counter()
    hist = dc.histogram()
    t_hist = time.perf_counter() - t0
    t0 = time.perf_counter()
    keys, cnts = dc.result(sort=False)
    t_pairs = time.perf_counter() - t0
    ok = bool(int(cnts.sum(dtype=np.uint64)) == st["kmers"] == sum(c * f for c, f in hist) and keys.size == st["distinct"])
    kmers = int(st["kmers"])
    n_pairs = int(keys.size)
    del keys, cnts
    # the same with memory the device reaches by DMA (kh_host_alloc: what a host that reads its file into such a buffer
    # sees): no staging memcpy on the way in, no bounce and no first-touch faults on the way out
    pinned = None
    try:
        import krust_amd
        with krust_amd.PinnedArray(host.size) as pin, krust_amd.PinnedArray(n_pairs, np.uint64) as pk, krust_amd.PinnedArray(n_pairs, np.uint64) as pc:
            pin.array[:] = host   # (setting the stage, not timed)
            dc.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dc.push(pin.array)
            st2 = dc.finish()
            t_push2 = time.perf_counter() - t0
            t0 = time.perf_counter()
            k2, c2 = dc.result(sort=False, out=(pk.array, pc.array))
            t_pairs2 = time.perf_counter() - t0
            ok2 = bool(int(c2.sum(dtype=np.uint64)) == st2["kmers"] == kmers and k2.size == n_pairs)
            pinned = {"what": "the same from / into kh_host_alloc'ed (pinned) memory: DMA straight from the caller's buffer and into the caller's arrays",
                      "push_finish_s": t_push2, "push_GBps": host.size / t_push2 / 1e9, "h2d_ms": st2["h2d_ms"],
                      "count_kernel_ms": st2["count_kernel_ms"], "part_batches": int(st2["part_batches"]),
                      "result_copy_s": t_pairs2, "result_copy_GBps": 16.0 * n_pairs / t_pairs2 / 1e9,
                      "kmers_per_s_push_only": kmers / t_push2, "kmers_per_s_pairs_out": kmers / (t_push2 + t_pairs2), "consistent": ok2}
            del k2, c2
    except Exception as e:  # (never lose the line over an extra)
        pinned = {"error": repr(e)}
    return {"what": "pageable host bases -> kh_push -> kh_finish -> results on the host; reads resident in HBM is `value`, not this",
            "pinned": pinned,
            "bytes_in": int(host.size), "push_finish_s": t_push, "push_GBps": host.size / t_push / 1e9,
            "first_push_finish_s": t_first, "first_push_what": "the context's first host push: + the one-time allocation of its accumulation / staging buffers",
            "h2d_ms": st["h2d_ms"], "count_kernel_ms": st["count_kernel_ms"], "part_batches": int(st["part_batches"]),
            "pairs_out": n_pairs, "result_copy_s": t_pairs, "result_copy_GBps": 16.0 * n_pairs / t_pairs / 1e9,
            "histogram_s": t_hist, "histogram_lines": len(hist),
            "kmers_per_s_push_only": kmers / t_push,
            "kmers_per_s_histogram_out": kmers / (t_push + t_hist),
            "kmers_per_s_pairs_out": kmers / (t_push + t_pairs),
            "consistent": ok, "staging_copy_s_untimed": t_stage}


def host_room_for(nbytes):
    """True if the host has `nbytes` of available memory AND of free space in /dev/shm (where cli_leg puts its file)."""
    try:
        avail = next(int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable:"))
        lim = "/sys/fs/cgroup/memory.max"
        if os.path.exists(lim):
            v = open(lim).read().strip()
            if v != "max":
                cur = int(open("/sys/fs/cgroup/memory.current").read())
                avail = min(avail, int(v) - cur)
        st = os.statvfs("/dev/shm")
        return avail >= nbytes and st.f_bavail * st.f_frsize >= nbytes // 2
    except Exception:
        return False


def cli_leg(krust_amd, torch, dev, local_rank, reads=10_000_000, k=21):
    """north_star's drop-in IS the command line (`kmerust <k> <path>`, src/main.rs:54-231 -> src/run.rs:185-200): S10M as a
    FASTQ FILE in /dev/shm -> `kmerust 21 f.fq --format histogram -q`, wall time of the whole process with the phase walls
    the binary reports (KMERUST_TIMING=1: create / read / push / finish / result / write), and the printed histogram
    compared line by line with kh_histogram of the same reads pushed through the C ABI."""
    import subprocess
    import numpy as np
    exe = os.path.join(ROOT, "krust_amd", "host", "kmerust")
    if not os.path.exists(exe):
        return {"error": f"{exe} is missing (make -C krust_amd/host)"}
    stride, W = READ_LEN + 1, 166 + READ_LEN  # "@r%09d\n" seq "\n+\n" qual "\n"
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    path = os.path.join(d, f"kmerust_bench_{os.getpid()}.fq")
    try:
        # the file is written in slices of 10 M reads (3.2 GB of text each): host memory holds the file and ONE slice, whatever the size
        SL = 10_000_000
        with open(path, "wb") as f:
            for r0 in range(0, reads, SL):
                nr = min(SL, reads - r0)
                tb = torch.empty(nr * stride, dtype=torch.uint8, device=dev)
                tq = torch.empty(nr * stride, dtype=torch.uint8, device=dev)
                krust_amd.synth_reads_device(tb.data_ptr(), tq.data_ptr(), SEED, GENOME_LEN, READ_LEN, r0, nr, device=local_rank,
                                             stream=torch.cuda.current_stream().cuda_stream)
                rec = torch.empty((nr, W), dtype=torch.uint8, device=dev)
                rec[:, 0], rec[:, 1], rec[:, 11] = ord("@"), ord("r"), 10
                r = torch.arange(r0, r0 + nr, device=dev)
                for j in range(9):
                    rec[:, 10 - j] = ((r // 10 ** j) % 10 + 48).to(torch.uint8)
                rec[:, 12:12 + READ_LEN] = tb.view(nr, stride)[:, :READ_LEN]
                rec[:, 12 + READ_LEN], rec[:, 13 + READ_LEN], rec[:, 14 + READ_LEN] = 10, ord("+"), 10
                rec[:, 15 + READ_LEN:15 + 2 * READ_LEN] = tq.view(nr, stride)[:, :READ_LEN]
                rec[:, 15 + 2 * READ_LEN] = 10
                torch.cuda.synchronize()
                rec.cpu().numpy().tofile(f)
                del rec, tb, tq, r
        nbytes = os.path.getsize(path)
        torch.cuda.empty_cache()
        # the reference result through the C ABI: the same reads, resident
        tb = torch.empty(reads * stride, dtype=torch.uint8, device=dev)
        krust_amd.synth_reads_device(tb.data_ptr(), None, SEED, GENOME_LEN, READ_LEN, 0, reads, device=local_rank,
                                     stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        with krust_amd.DeviceCounter(k, device=local_rank) as dc:
            dc.push_device(tb.data_ptr(), None, reads * stride)
            st = dc.finish()
            want = dc.histogram()
        del tb
        torch.cuda.empty_cache()
        env = dict(os.environ, KMERUST_TIMING="1")
        runs = []
        # (this process has just freed a few hundred GB of HBM; a new process's runtime start-up has been seen to take 1-4 s
        #  instead of 0.1-0.3 right after that -- `create_s` of the runs below says which it was: give the driver a moment)
        # (measured, tools/cli_s100m_probe.py: started 0-3 s after a process that held ~230 GB of HBM has exited, a new process's
        #  hipMalloc of its partition buffers takes 1-5 s -- the driver is still reclaiming; after 10 s it takes milliseconds)
        pause = float(os.environ.get("BENCH_CLI_PAUSE_S") or (10.0 if reads >= 40_000_000 else 3.0))   # (BENCH_CLI_PAUSE_S: tests)
        subprocess.run(["cat", path], stdout=subprocess.DEVNULL)  # (the file was written slice by slice: read once, it streams at the page cache's rate)
        for rep in range(3):   # (the first run pages the binary and the ROCm libraries in)
            time.sleep(pause)
            t0 = time.perf_counter()
            p = subprocess.run([exe, str(k), path, "--format", "histogram", "-q"], capture_output=True, env=env, timeout=600)
            wall = time.perf_counter() - t0
            tj = None
            for line in p.stderr.decode(errors="replace").splitlines():
                if line.startswith('{"kmerust_timing"'):
                    tj = json.loads(line)["kmerust_timing"]
            got = [tuple(map(int, l.split(b"\t"))) for l in p.stdout.splitlines()]
            runs.append({"wall_s": wall, "rc": p.returncode, "phases": tj, "matches_c_abi_histogram": bool(got == [tuple(x) for x in want])})
        best = min(runs, key=lambda x: x["wall_s"])
        return {"what": f"kmerust {k} <S{reads // 1_000_000}M FASTQ file> --format histogram -q: process wall time, file in {d}",
                "reads": reads, "text_bytes": nbytes, "kmers": int(st["kmers"]), "runs": runs,
                "wall_s": best["wall_s"], "text_GBps": nbytes / best["wall_s"] / 1e9, "kmers_per_s": st["kmers"] / best["wall_s"],
                "ok": bool(all(x["rc"] == 0 and x["matches_c_abi_histogram"] for x in runs))}
    finally:
        if os.path.exists(path):
            os.remove(path)


def merge_dict(mi, impl):
    """kh_merge_info -> config.merge of the line (one rank's view)."""
    return {"impl": impl, "path": mi["path"], "local_distinct": mi["local_distinct"],
            "sent_pairs": mi["sent_units"], "recv_pairs": mi["recv_units"], "unit_bytes": mi["unit_bytes"],
            "owned_distinct": mi["owned_distinct"],
            # what the TRANSPORT says the world is (ncclCommCount / the hub's size) and the library's own conservation verdict
            "rccl_nranks": mi["nranks_seen"], "lib_conserved": bool(mi["conserved"]),
            "sent_count_sum": mi["sent_count_sum"], "merged_count_sum": mi["merged_count_sum"],
            "phase_ms": {"export": mi["export_ms"], "exchange_wait": mi["wait_ms"], "merge": mi["merge_ms"],
                         "total": mi["total_ms"], "pieces": mi["pieces"]}}


def _r(x, nd=4):
    """Numbers of the contract line: enough digits to recompute every ratio, not seventeen."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{nd + 2}g}") if abs(x) < 1e15 else x
    return x


LINE_LIMIT = 6144   # bytes: the whole line must fit the driver's 8 KB stdout tail
_CONTRACT_OUT = None  # the process's real stdout, once own_stdout() has moved everybody else to stderr


def own_stdout():
    """stdout carries ONE line: ours.  RCCL prints a five-line banner (version, HIP / ROCm version, host name, library path) to
    fd 1 from every process that creates a communicator -- after our line, at exit -- and gloo its own; a driver that reads
    the last line of stdout would read that.  So fd 1 is pointed at stderr for everything but emit()."""
    global _CONTRACT_OUT
    if _CONTRACT_OUT is None:
        sys.stdout.flush()
        _CONTRACT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def contract_line(full):
    """The ONE line the driver parses, from the full report: the contract's keys, `roofline` and `cpu_baseline` as the task
    statement defines them, and one flat row per sub-result.  Everything else stays in the full report."""
    rf = full.get("roofline") or {}
    dom = rf.get("dominant") or {}
    cfg = full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                      "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(line["value"], 6), _r(line["ms_per_step"], 5)
    c = {k: cfg.get(k) for k in ("workload", "k", "reads_per_gpu", "kmers_per_step_per_gpu", "distinct_per_gpu", "table_slots", "table_load",
                                  "capacity_hint", "parallelism", "mode") if cfg.get(k) is not None}
    c["table_load"] = _r(c.get("table_load"))
    mg = cfg.get("merge")
    if mg:
        c["merge"] = {k: mg.get(k) for k in ("impl", "path", "unit_bytes", "rccl_nranks", "lib_conserved", "conserved", "merged_occurrences",
                                              "merged_distinct", "sent_pairs", "fallback_from_c_abi") if mg.get(k) is not None}
        if mg.get("phase_ms"):
            c["merge"]["phase_ms"] = {k: _r(v, 3) for k, v in mg["phase_ms"].items()}
        if cfg.get("per_rank"):
            c["per_rank"] = [[r["rank"], _r(r["count_kernel_ms"], 3), _r(r["export_ms"], 3), _r(r["exchange_wait_ms"], 3), _r(r["merge_ms"], 3),
                              r["sent_units"], r["owned_distinct"]] for r in cfg["per_rank"]]
            c["per_rank_columns"] = ["rank", "count_kernel_ms", "export_ms", "exchange_wait_ms", "merge_ms", "sent_units", "owned_distinct"]
    if cfg.get("single_gpu_same_share"):
        c["single_gpu_same_share"] = {k: _r(v, 5) for k, v in cfg["single_gpu_same_share"].items() if k != "what"}
    line["config"] = c
    line["roofline"] = {"bound": rf.get("bound"), "achieved": _r(rf.get("achieved"), 5), "peak": rf.get("peak"), "unit": rf.get("unit"),
                        "frac": _r(rf.get("frac")), "traffic": rf.get("traffic"), "traffic_frac": _r(rf.get("traffic_frac")),
                        "alg_bytes_per_step": rf.get("alg_bytes_per_step"), "kernel_ms_per_step": _r(rf.get("kernel_ms_per_step"), 5),
                        "kernel": rf.get("kernel"), "stages_ms": {k: _r(v, 4) for k, v in (rf.get("stages_ms") or {}).items()},
                        "dominant": {"kernel": dom.get("kernel"), "ms": _r(dom.get("ms"), 4), "frac": _r(dom.get("frac"))} if dom else None}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb.get("value"), 5), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": (cb.get("sample") or "")[:160], "optimised_value": _r(cb.get("optimised_value"), 5),
                                "optimised_cores": cb.get("optimised_cores")}
    if full.get("verify") is not None:
        v = full["verify"]
        line["verify"] = {k: v.get(k) for k in ("ok", "sampled_keys", "cpu_total_kmers", "gpu_total_kmers") if v.get(k) is not None}
    if full.get("unhinted"):
        u = full["unhinted"]
        line["unhinted"] = {"value": _r(u.get("value"), 6), "ratio_to_headline": _r(u.get("ratio_to_headline"), 5), "verify_ok": u.get("verify_ok")}
    e2e = full.get("end_to_end") or {}
    if e2e and "error" not in e2e:
        pin = e2e.get("pinned") or {}
        line["end_to_end"] = {"pageable_push_GBps": _r(e2e.get("push_GBps")), "pageable_first_push_s": _r(e2e.get("first_push_finish_s"), 4), "pageable_kmers_per_s_push_only": _r(e2e.get("kmers_per_s_push_only"), 5),
                              "pinned_push_GBps": _r(pin.get("push_GBps")), "pinned_kmers_per_s_pairs_out": _r(pin.get("kmers_per_s_pairs_out"), 5),
                              "consistent": bool(e2e.get("consistent") and pin.get("consistent", True))}
    rows = []
    for x in full.get("configs") or []:
        row = {"workload": (x.get("workload") or "")[:72]}
        if "error" in x:
            row["error"] = x["error"][:120]
        else:
            row.update({"value": _r(x.get("value"), 5), "ms_per_step": _r(x.get("ms_per_step"), 4), "frac": _r((x.get("roofline") or {}).get("frac")),
                        "verify_ok": (x.get("verify") or {}).get("ok")})
            if (x.get("roofline") or {}).get("traffic_frac") is not None:  # (measured HBM bytes of this workload / its kernel time / peak)
                row["traffic_frac"] = _r(x["roofline"]["traffic_frac"])
        rows.append(row)
    if rows:
        line["configs"] = rows
    cli = full.get("cli")
    if cli:
        big = cli.get("s100m") or cli.get("large") or {}
        line["cli"] = {"s10m_wall_s": _r(cli.get("wall_s")), "s100m_wall_s": _r(big.get("wall_s")), "s100m_kmers_per_s": _r(big.get("kmers_per_s"), 5),
                       "s100m_reads": big.get("reads"), "ok": bool(cli.get("ok") and big.get("ok", True)) if "error" not in cli else False}
        if "error" in cli:
            line["cli"]["error"] = str(cli["error"])[:120]
    line["full_report"] = full.get("full_report")
    # never let an extra push the line over the limit: drop the optional blocks, largest first
    for drop in ("end_to_end", "configs", "cli", "unhinted"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(drop, None)
        line["dropped_for_size"] = line.get("dropped_for_size", []) + [drop]
    return line


def emit(full):
    """Full report -> gpurun_out/bench_full.json (BENCH_FULL_PATH overrides); contract line -> stdout, last and alone."""
    path = os.environ.get("BENCH_FULL_PATH") or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f)
            f.write("\n")
        full["full_report"] = os.path.relpath(path, ROOT)
    except OSError as e:
        full["full_report"] = f"not written: {e!r}"
    line = json.dumps(contract_line(full))
    assert len(line) <= LINE_LIMIT, len(line)
    print(line, file=_CONTRACT_OUT or sys.stdout, flush=True)


def main_group(args):
    """--group N: the N > 1 accounting of this file on ONE device.  N logical ranks are contexts of a kh_group (threads of this
    process, the in-process hub as transport: RCCL refuses two ranks on one device); rank r counts reads [r x reads,
    (r + 1) x reads) one after the other on the shared device, kh_group_merge runs the library's real merge sequence -- votes,
    pipeline, digests, LDS merges -- and the line carries the same config.merge / per_rank / conserved / rccl_nranks keys as a
    torchrun launch.  What it measures is the code path, not scaling: the ranks share one GPU's memory and CUs."""
    import torch
    import krust_amd
    N = args.group
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    k = args.k
    reads = args.reads or 10_000_000
    stride = READ_LEN + 1
    nbytes = reads * stride
    with_qual = args.min_quality is not None
    tbs, tqs = [], []
    for r in range(N):
        tb = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        tq = torch.empty(nbytes, dtype=torch.uint8, device=dev) if with_qual else None
        krust_amd.synth_reads_device(tb.data_ptr(), tq.data_ptr() if with_qual else None, SEED, GENOME_LEN, READ_LEN, r * reads, reads,
                                     device=0, stream=torch.cuda.current_stream().cuda_stream)
        tbs.append(tb)
        tqs.append(tq)
    torch.cuda.synchronize()
    hint = args.capacity_hint or estimate_distinct(reads, k, N, with_qual)
    with krust_amd.DeviceGroup(k, [0] * N, min_quality=args.min_quality, capacity_hint=hint) as grp:
        def step():
            sts = []
            for r in range(N):
                grp[r].reset()
                grp[r].push_device(tbs[r].data_ptr(), tqs[r].data_ptr() if with_qual else None, nbytes)
                sts.append(grp[r].finish())
            return sts, grp.merge()

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sts, infos = step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        total_kmers = sum(int(st["kmers"]) for st in sts)
        occ = dist_ = 0
        for r in range(N):   # the key-sharded tables together hold every occurrence exactly once
            hist = grp[r].histogram()
            occ += sum(c * f for c, f in hist)
            dist_ += sum(f for _, f in hist)
        mg = merge_dict(infos[0], "kh_group_merge (C ABI; in-process hub: N thread ranks on one device)")
        mg.update(merged_occurrences=occ, merged_distinct=dist_,
                  conserved=bool(occ == total_kmers and all(i["conserved"] for i in infos)
                                 and sum(i["sent_count_sum"] for i in infos) == sum(i["merged_count_sum"] for i in infos) == total_kmers))
        if any(i["nranks_seen"] != N for i in infos):
            raise RuntimeError(f"the transport saw {[i['nranks_seen'] for i in infos]} ranks, the group has {N}")
        per_rank = [{"rank": r, "count_kernel_ms": sts[r]["count_kernel_ms"], "export_ms": i["export_ms"], "exchange_wait_ms": i["wait_ms"],
                     "merge_ms": i["merge_ms"], "local_distinct": int(i["local_distinct"]), "sent_units": int(i["sent_units"]),
                     "owned_distinct": int(i["owned_distinct"])} for r, i in enumerate(infos)]
        st = sts[0]
        rf = roofline_of(st, nbytes * (2 if with_qual else 1), st["count_kernel_ms"], st["stage_ms"], k)
        verify = None
        if not args.no_verify and reads <= 20_000_000:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            tot = 0
            for r in range(N):
                m = O.OracleMap()
                tot += m.scan_flat(tbs[r].cpu().numpy(), k, qual=tqs[r].cpu().numpy() if with_qual else None, min_quality=args.min_quality,
                                   sample_mask=1023, nthreads=2 * usable_cpus())
            verify = {"ok": bool(tot == total_kmers == occ), "cpu_total_kmers": int(tot), "gpu_total_kmers": total_kmers,
                      "what": "the k-mer total of all ranks' reads against the CPU oracle's scan, and against the merged shards' histograms"}
    out = {"metric": "canonical k-mers/s at k=21, 100M x 150bp reads; bit-exact vs krust CPU",
           "value": total_kmers * args.steps / elapsed, "unit": "k-mers/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
           "data": "synthetic",
           "config": {"workload": f"--group {N}: {N} logical ranks x {reads} x {READ_LEN} bp reads on ONE GPU, k={k}: a dry run of the N > 1 accounting, not a scaling figure",
                      "mode": f"group{N}", "k": k, "reads_per_gpu": reads, "kmers_per_step_per_gpu": int(st["kmers"]),
                      "distinct_per_gpu": int(st["distinct"]), "table_slots": int(st["table_slots"]),
                      "table_load": st["distinct"] / st["table_slots"], "capacity_hint": int(hint),
                      "parallelism": f"reads sharded x{N} (thread ranks sharing one device); kh_group_merge",
                      "merge": mg, "per_rank": per_rank},
           "roofline": rf}
    if verify is not None:
        out["verify"] = verify
    emit(out)


def main():
    args = parse_args()
    own_stdout()
    if args.group > 1:
        return main_group(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    import torch
    import torch.distributed as dist

    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # BENCH_BACKEND=gloo (tests only): several ranks share the visible GPU(s) and the collectives are
    # host-staged, so the whole N > 1 code path of this file can be exercised on a 1-GPU box.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    merged = world > 1 or args.force_merge     # the step ends in kh_merge_across (and the line carries config.merge)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL
        else:
            dist.init_process_group(backend)
    elif args.force_merge:   # a world of one, for the reductions below (the merge itself talks RCCL through the library)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
        backend = "gloo"
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where small reduction operands live

    import krust_amd
    from krust_amd.distributed import merge_across_ranks

    if args.hg:   # configs[4] alone, as the headline of this invocation (profiling runs; the default line carries it as a sub-result)
        sub = None
        for _ in range(max(1, args.steps)):
            sub = sub_config_hg(krust_amd, torch, dev, local_rank, verify=not args.no_verify)
        out = {"metric": "canonical k-mers/s, hg-shaped FASTA text -> histogram (BASELINE configs[4])", "value": sub["value"], "unit": "k-mers/s", "n_gpus": 1,
               "steps": 1, "warmup": 1, "ms_per_step": sub["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
               "data": "synthetic", "config": {"workload": sub["workload"], "k": sub["k"], "reads_per_gpu": 0, "kmers_per_step_per_gpu": sub["kmers_per_step"],
                                               "distinct_per_gpu": sub["distinct"], "table_slots": sub["table_slots"],
                                               "table_load": sub["distinct"] / sub["table_slots"], "capacity_hint": 0},
               "roofline": sub["roofline"], "verify": sub.get("verify"), "hg": {k2: sub[k2] for k2 in ("count_ms", "histogram_ms", "text_scan_ms", "histogram_lines")}}
        emit(out)
        return

    k = args.k
    reads = args.reads or (READS_NX if merged else READS_N1)
    stride = READ_LEN + 1
    nbytes = reads * stride
    with_qual = args.min_quality is not None
    tb = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    tq = torch.empty(nbytes, dtype=torch.uint8, device=dev) if with_qual else None
    first = rank * reads  # weak scaling: every GPU gets its own `reads` reads of the same genome
    krust_amd.synth_reads_device(tb.data_ptr(), tq.data_ptr() if with_qual else None, SEED, GENOME_LEN, READ_LEN,
                                 first, reads, device=local_rank, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()

    hint = 0 if args.no_hint else (args.capacity_hint or estimate_distinct(reads, k, world, with_qual))
    # stream=None: the context launches on its own non-blocking stream (the merge pipeline overlaps it with
    # RCCL's); every hand-over between torch work and the counter is fenced by a device-wide synchronize here
    dc = krust_amd.DeviceCounter(k, min_quality=args.min_quality, capacity_hint=hint, device=local_rank, stream=None)

    # N > 1: the exchange runs INSIDE the library (kh_merge_across: RCCL send/recv groups on its own stream,
    # pipelined against the export and LDS-merge kernels) -- the route a Rust / C++ host takes.  torch.distributed
    # only carries the 128-byte communicator id and the final timing reductions.  BENCH_MERGE=python (or the gloo
    # backend of the 1-GPU tests, where ranks share a device and RCCL cannot be used) selects the
    # torch.distributed harness krust_amd/distributed.py instead; both leave identical shard tables.
    merge_impl = os.environ.get("BENCH_MERGE", "c" if (backend == "nccl" or args.force_merge) else "python")
    merge_note = None
    if merged and merge_impl == "c":
        ok = 1
        try:
            box = [krust_amd.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            dc.comm_init(world, rank, box[0])
        except Exception as e:  # (an RCCL failure inside the library: every rank falls back together, and says so)
            ok, merge_note = 0, f"kh_comm_init failed on rank {rank}: {e!r}"
        flag = torch.tensor([ok], dtype=torch.int64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            merge_impl = "python"
            merge_note = merge_note or "kh_comm_init failed on another rank"

    def merge():
        if merge_impl == "c":
            # A failed merge raises on EVERY rank from this same call (kh_merge_across: status gathers, bounded waits,
            # KH_ERR_PEER; since round 5 also when what arrived is not what was sent); the process then exits non-zero.
            return merge_dict(dc.merge_across(), "kh_merge_across (C ABI, RCCL)")
        out = dict(merge_across_ranks(dc, phase_times=True), impl="krust_amd.distributed (torch.distributed)")
        if merge_note:
            out["fallback_from_c_abi"] = merge_note
        return out

    def step():
        dc.reset()
        dc.push_device(tb.data_ptr(), tq.data_ptr() if with_qual else None, nbytes)
        st = dc.finish()
        mg = merge() if merged else None  # (phase walls of rank 0 go into the JSON)
        return st, mg

    def fence():
        torch.cuda.synchronize()
        if merged:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    launches = 0
    stage_ms = {}
    st = mg = None
    for _ in range(args.steps):
        st, mg = step()
        kernel_ms += st["count_kernel_ms"]
        launches += st["launches"]
        for name, ms in st["stage_ms"].items():
            stage_ms[name] = stage_ms.get(name, 0.0) + ms
    fence()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if merged:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        km = torch.tensor([st["kmers"]], dtype=torch.int64, device=cdev)
        dist.all_reduce(km, op=dist.ReduceOp.SUM)
        total_kmers = int(km.item())
        # conservation across the merge (outside the timed region): the key-sharded tables together hold
        # every occurrence exactly once, and the shards' key sets are disjoint by construction
        hist = dc.histogram()
        chk = torch.tensor([sum(c * f for c, f in hist), sum(f for _, f in hist)], dtype=torch.int64, device=cdev)
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        mg = dict(mg, merged_occurrences=int(chk[0].item()), merged_distinct=int(chk[1].item()),
                  conserved=bool(int(chk[0].item()) == total_kmers))
        # the world RCCL itself reports (ncclCommCount inside the library) must be the world this run was launched with
        if merge_impl == "c" and mg.get("rccl_nranks") != world:
            raise RuntimeError(f"RCCL saw {mg.get('rccl_nranks')} ranks, the launcher {world}")
        # every rank's own figures of the last step: counting kernels, merge phases, what it sent and owns
        mine = torch.tensor([kernel_ms / args.steps, mg["phase_ms"].get("export", 0.0) if mg.get("phase_ms") else 0.0,
                             mg["phase_ms"].get("exchange_wait", 0.0) if mg.get("phase_ms") else 0.0,
                             mg["phase_ms"].get("merge", 0.0) if mg.get("phase_ms") else 0.0,
                             float(mg["local_distinct"]), float(mg["sent_pairs"]), float(mg["owned_distinct"])],
                            dtype=torch.float64, device=cdev)
        rows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        per_rank = [{"rank": r, "count_kernel_ms": float(v[0]), "export_ms": float(v[1]), "exchange_wait_ms": float(v[2]),
                     "merge_ms": float(v[3]), "local_distinct": int(v[4]), "sent_units": int(v[5]), "owned_distinct": int(v[6])}
                    for r, v in enumerate(rows)]
        # The same share on ONE GPU, measured in this run: the rank's own reads counted with no merge (reset + push + finish),
        # all ranks at once, rank 0's wall time.  The driver's N = 1 point is S100M (100 M reads); this is the N = 1 point of THIS
        # workload (125 M reads per GPU), so that a 1 -> N curve can be read without mixing the two.
        same_share = None
        fence()
        t1 = time.perf_counter()
        for _ in range(2):
            dc.reset()
            dc.push_device(tb.data_ptr(), tq.data_ptr() if with_qual else None, nbytes)
            st1 = dc.finish()
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / 2
        same_share = {"value": st1["kmers"] / dt1, "ms_per_step": dt1 * 1e3, "steps": 2, "kernel_ms_per_step": st1["count_kernel_ms"],
                      "what": "rank 0's share counted alone on its GPU (no merge), measured after the timed loop of this run"}
        fence()
    else:
        total_kmers = int(st["kmers"])
        same_share = None

    verify = None
    do_verify = (args.verify or (not merged and not args.no_verify))
    if do_verify and rank == 0:
        host = tb.cpu().numpy()
        hq = tq.cpu().numpy() if with_qual else None
        if not merged:
            verify = verify_reads(dc, st, host, hq, k, args.min_quality)
        else:  # (the table is a shard by now: only the rank's k-mer total can be checked here)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            m = O.OracleMap()
            tot = m.scan_flat(host, k, qual=hq, min_quality=args.min_quality, sample_mask=1023, nthreads=2 * usable_cpus())
            verify = {"ok": bool(tot == st["kmers"]), "cpu_total_kmers": int(tot), "what": "rank 0's k-mer total only (N > 1)"}
        del host, hq

    if rank == 0:
        # Roofline: HBM-bound.  Algorithmic bytes of one step: every
        # input byte once + 24 B per valid k-mer (slot key+count read, count write-back) + 8 B per
        # distinct key written once; achieved = that / the HIP-event time of the counting kernels of
        # the step (events recorded on the launch stream inside the library, kh_stats.stage_ms).
        kernel_ms_step = kernel_ms / args.steps
        stages = {name: ms / args.steps for name, ms in stage_ms.items()}
        rf = roofline_of(st, nbytes * (2 if with_qual else 1), kernel_ms_step, stages, k)
        traffic = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):  # measured in separate rocprofv3 --pmc passes (profiles/README.md), not in this run
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("reads_per_gpu") == reads and tj.get("k") == k and not with_qual:
                    traffic = tj.get("bytes_per_step")
                    traffic_source = f"profiles/{tj.get('tag')}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not re-measured in this run)"
            except Exception:
                traffic = traffic_source = None
        rf.update({"traffic": traffic, "traffic_source": traffic_source,
                   # measured HBM bytes / kernel time / peak: what the memory system really moved (frac prices the SURVEY formula)
                   "traffic_frac": (traffic / (kernel_ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and kernel_ms_step) else None,
                   "frac_of_measured_copy_peak": rf["achieved"] / 6290.0,  # MI355X_MICROARCH.md: 6.29 TB/s copy
                   "launches_per_step": int(launches / args.steps),
                   "kernel_kmers_per_s": st["kmers"] / (kernel_ms_step * 1e-3) if kernel_ms_step else None})
        shape = ("S100M" if reads == READS_N1 else "configs[3] share (S1B / 8)" if reads == READS_NX else f"S{reads // 1_000_000}M")
        out = {
            "metric": "canonical k-mers/s at k=21, 100M x 150bp reads; bit-exact vs krust CPU",
            "value": total_kmers * args.steps / elapsed,
            "unit": "k-mers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"{shape}: {reads} x {READ_LEN} bp reads per GPU (rank r: reads [r x {reads}, (r + 1) x {reads})), k={k}"
                                   + (f", -Q {args.min_quality}" if with_qual else "")
                                   + f", genome 2^27 bp, seed {SEED}; resident in HBM",
                       "k": k, "reads_per_gpu": reads, "kmers_per_step_per_gpu": int(st["kmers"]),
                       "distinct_per_gpu": int(st["distinct"]), "table_slots": int(st["table_slots"]),
                       "table_load": st["distinct"] / st["table_slots"],
                       "capacity_hint": int(hint), "capacity_hint_source": "--capacity-hint" if args.capacity_hint else HINT_SOURCE,
                       "table_grows": int(st["grows"]),
                       "parallelism": f"reads sharded x{world}" + ("; RCCL all-to-all table merge (kh_merge_across)" if merged else "")},
            "roofline": rf,
        }
        if mg is not None:
            out["config"]["merge"] = mg
            out["config"]["per_rank"] = per_rank
            out["config"]["single_gpu_same_share"] = same_share
        if verify is not None:
            out["verify"] = verify
        if not merged and not args.no_cpu_baseline:
            sample_reads = min(reads, 6_000_000)
            host = tb[: sample_reads * stride].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(host, k, args.cpu_seconds)
            del host
        if not merged and not args.no_extras:
            try:
                out["end_to_end"] = end_to_end(dc, tb, torch, k)
            except Exception as e:  # (never lose the headline line over an extra)
                out["end_to_end"] = {"error": repr(e)}
            dc.close()
            del tb, tq
            torch.cuda.empty_cache()
            subs = []
            full = reads >= READS_N1
            plan = [("headline twin WITHOUT a capacity hint (S100M, k=21)", dict(reads=reads, k=21, min_quality=None, hint=0, steps=3, verify=True)),
                    ("configs[1] S10M: 10 M x 150 bp, k=21", dict(reads=10_000_000, k=21, min_quality=None, steps=3)),
                    ("configs[1] twin WITHOUT a capacity hint (S10M, k=21)", dict(reads=10_000_000, k=21, min_quality=None, hint=0, steps=3)),
                    ("configs[2] S100M: 100 M x 150 bp, k=31, -Q 20 (no capacity hint)", dict(reads=100_000_000, k=31, min_quality=20, hint=0)),
                    ("k=19 twin of the headline (S100M, k=19, no capacity hint): the level-1 window is generated for every k", dict(reads=100_000_000, k=19, min_quality=None, hint=0)),
                    ("k=25 twin of the headline (S100M, k=25, no capacity hint): 8-byte payloads, narrowed by level 2 to the 4 bytes below the region index",
                     dict(reads=100_000_000, k=25, min_quality=None, hint=0, verify=True)),
                    ("configs[3] rank 3's share of S1B: 125 M x 150 bp, k=21, with the a-priori capacity hint",
                     dict(reads=READS_NX, k=21, min_quality=None, first=3 * READS_NX, steps=2)),
                    ("configs[3] the same share, no capacity hint",
                     dict(reads=READS_NX, k=21, min_quality=None, first=3 * READS_NX, hint=0, verify=True, steps=2))]
            for name, kw in plan:
                if not full and kw["reads"] > reads:
                    continue  # (a reduced --reads run: keep the extras proportionate)
                try:
                    subs.append(sub_config(krust_amd, torch, dev, local_rank, name, **kw))
                except Exception as e:
                    subs.append({"workload": name, "error": repr(e)})
            if full:
                try:
                    subs.append(sub_config_hg(krust_amd, torch, dev, local_rank))
                except Exception as e:
                    subs.append({"workload": "configs[4] hg-shaped FASTA", "error": repr(e)})
            out["configs"] = subs
            # the twins without a capacity hint, beside what they are twins of (the reference API has no hint)
            def _twin(tag):
                return next((x for x in subs if x.get("workload", "").startswith(tag) and "value" in x), None)
            tw = _twin("headline twin")
            if tw:
                out["unhinted"] = {"value": tw["value"], "ms_per_step": tw["ms_per_step"], "steps": tw["steps"], "table_slots": tw["table_slots"],
                                   "table_load": tw["table_load"], "ratio_to_headline": tw["value"] / out["value"],
                                   "verify_ok": (tw.get("verify") or {}).get("ok")}
            c1, c1u = _twin("configs[1] S10M"), _twin("configs[1] twin")
            if c1 and c1u:
                c1u["ratio_to_hinted"] = c1u["value"] / c1["value"]
            try:
                out["cli"] = cli_leg(krust_amd, torch, dev, local_rank, reads=min(10_000_000, reads))
                # The same on a file four times the size (S40M, 12.6 GB): what the command line sustains once the process's
                # fixed costs (runtime start-up, pinned buffers, exit: ~0.45 s) are spread thinner.  Only where the host has
                # the memory for the file twice (tensor -> numpy -> /dev/shm).
                # ... and on the file the METRIC is quoted on: S100M, 31.6 GB of FASTQ text (where /dev/shm and the host's memory hold
                # it; else the largest multiple of 10 M reads that fits, stated in `reads`)
                if full:
                    big_reads = next((r for r in (100_000_000, 70_000_000, 40_000_000) if host_room_for(r * 316 + (12 << 30))), 0)
                    if big_reads:
                        big = cli_leg(krust_amd, torch, dev, local_rank, reads=big_reads)
                        key = "s100m" if big_reads == 100_000_000 else "large"
                        out["cli"][key] = {k: big.get(k) for k in ("what", "reads", "text_bytes", "wall_s", "text_GBps", "kmers_per_s", "ok", "error")}
                        out["cli"][key]["phases"] = [r.get("phases") for r in big.get("runs", [])]
            except Exception as e:
                out["cli"] = dict(out.get("cli") or {}, error=repr(e))
        emit(out)

    dc.close()
    if merged:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
