"""Drift guards between the documents, the committed measurements and the source (CPU only, like
tests/test_ffi_drift.py for the FFI): the numbers a reader is given must be the ones in profiles/, and the kernels the
documents name must exist."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(*p):
    with open(os.path.join(ROOT, *p)) as f:
        return f.read()


def _bench(tag):
    return json.loads([l for l in _read("profiles", f"{tag}_bench.json").splitlines() if l.startswith("{")][-1])


def test_hbm_traffic_json_is_what_the_pmc_summary_says():
    tj = json.loads(_read("profiles", "hbm_traffic.json"))
    s = json.loads(_read("profiles", f"{tj['tag']}_summary.json"))
    steps = max(k["calls"] for k in s["kernels"] if "region_count" in k["name"])
    fetch = write = 0.0
    for name, ctrs in s["pmc"].items():
        if "synth_reads" in name or "table_init" in name:
            continue
        fetch += ctrs.get("FETCH_SIZE", {}).get("sum", 0.0) * 1024 / steps
        write += ctrs.get("WRITE_SIZE", {}).get("sum", 0.0) * 1024 / steps
    assert tj["bytes_per_step"] == int(2 * fetch + write)
    assert tj["reads_per_gpu"] == 100_000_000 and tj["k"] == 21


def test_hbm_traffic_of_the_other_configurations_is_what_their_summaries_say():
    """hbm_traffic.json "configs" (round 5: configs[2], configs[3]'s share, configs[4]) -- what bench.py reports as the
    sub-results' roofline.traffic -- recomputed from the committed PMC summaries of their tags."""
    tj = json.loads(_read("profiles", "hbm_traffic.json"))
    assert {"k31q20", "s125", "hg"} <= set(tj.get("configs", {}))
    for name, c in tj["configs"].items():
        s = json.loads(_read("profiles", f"{c['tag']}_summary.json"))
        steps = max(k["calls"] for k in s["kernels"] if "region_count" in k["name"])
        fetch = write = 0.0
        for kn, ctrs in s["pmc"].items():
            if "kh::" not in kn or "synth_reads" in kn or "table_init" in kn:
                continue
            if any(x in kn for x in ("fasta_", "fastq_", "raw_", "scan_", "hist", "lookup")):
                continue
            fetch += ctrs.get("FETCH_SIZE", {}).get("sum", 0.0) * 1024 / steps
            write += ctrs.get("WRITE_SIZE", {}).get("sum", 0.0) * 1024 / steps
        assert c["bytes_per_step"] == int(2 * fetch + write), name
        assert (c["reads"], c["k"]) in ((100_000_000, 31), (125_000_000, 21), (0, 21))


def test_headline_numbers_of_the_documents_are_the_committed_bench_line():
    tag = json.loads(_read("profiles", "hbm_traffic.json"))["tag"]
    b = _bench(tag)
    g = b["value"] / 1e9
    readme = _read("README.md")
    m = re.search(r"\*\*≈ (\d+) G canonical k-mers/s\*\*", readme)
    assert m and abs(int(m.group(1)) - g) / g < 0.02, (m and m.group(1), g)
    design = _read("DESIGN.md")
    m = re.search(r"= \*\*([\d.]+) G k-mers/s\*\*, bit-exact", design)
    assert m and abs(float(m.group(1)) - g) / g < 0.01, (m and m.group(1), g)
    m = re.search(r"= ([\d.]+) of the HBM peak, measured traffic ([\d.]+) GB", design)
    assert m and abs(float(m.group(1)) - b["roofline"]["frac"]) < 0.005
    assert abs(float(m.group(2)) * 1e9 - json.loads(_read("profiles", "hbm_traffic.json"))["bytes_per_step"]) < 0.5e9
    prof = _read("profiles", "README.md")
    assert f"## {tag} -- " in prof and f"{g:.1f} G k-mers/s" in prof


def test_kernels_named_in_design_exist_in_the_source():
    src = "".join(_read("krust_amd", "csrc", f) for f in ("partition.hip.h", "level1.hip.h", "kernels.hip.h", "rawparse.hip.h", "shard.hip.h"))
    design = _read("DESIGN.md")
    sec = design[design.index("### 4.2"):design.index("### 4.3")]
    table = [l for l in sec.splitlines() if l.startswith("| `")]
    names = set()
    for l in table:
        names.update(re.findall(r"`(\w+_kernel\w*)", l.split("|")[1]))
    assert {"part1_bins_kernel", "part2_arena_kernel", "region_count_kernel32"} <= names
    for n in names:
        assert re.search(r"\bvoid " + n + r"\(", src), n
    # the profile of the same tag lists them too
    tag = json.loads(_read("profiles", "hbm_traffic.json"))["tag"]
    stats = _read("profiles", f"{tag}_kernel_stats.csv")
    for n in ("part1_bins_kernel", "part2_arena_kernel", "region_count_kernel32"):
        assert n in stats


def test_kernels_named_by_the_bench_line_exist_and_were_profiled():
    """bench.py names the kernels of a step from what ran (kh_stats.stage_ms -> kernels_of): every name it can emit must
    be a __global__ kernel of krust_amd/csrc, and the ones of the headline configuration must be in the committed
    rocprofv3 kernel trace of that command (the r02 line named kernels that no longer existed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    src = "".join(_read("krust_amd", "csrc", f) for f in ("partition.hip.h", "level1.hip.h", "kernels.hip.h"))
    emitted = set()
    for k in (15, 21, 31):
        for stages in ({"direct": 1.0}, {"level1": 1.0, "level2": 1.0, "region": 1.0}, {"level1": 1.0, "level2_count": 1.0, "level2": 1.0, "region": 1.0}):
            emitted.update(bench.kernels_of(stages, k).values())
    assert {"part1_bins_kernel", "part1_bins64_kernel", "part2_arena_kernel", "part2_count_kernel", "region_count_kernel64"} <= emitted
    for n in emitted:
        assert re.search(r"\bvoid " + n + r"\(", src), n
    tag = json.loads(_read("profiles", "hbm_traffic.json"))["tag"]
    stats = _read("profiles", f"{tag}_kernel_stats.csv")
    headline = bench.kernels_of({"level1": 1.0, "level2": 1.0, "region": 1.0, "misc": 0.1}, 21)
    assert headline and all(n in stats for n in headline.values()), headline
    # the stage names of the binding are the ones the bench line looks up
    from krust_amd import native
    assert set(headline) <= set(native.STAGES)
    b = _bench(tag)
    if "kernels" in b["roofline"]:     # (lines written since round 3 carry the names themselves)
        assert all(n in stats for n in b["roofline"]["kernels"].values())


def test_the_bench_line_describes_its_own_roofline_and_hint():
    """the line says which terms `roofline.achieved` prices (compaction excluded), what the memory
    system moved per peak second beside it (`traffic_frac`), and where the capacity hint came from -- bench.py builds
    these keys, and BASELINE.md states the same formula."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    st = {"kmers": 1000, "distinct": 100, "table_slots": 4096}
    rf = bench.roofline_of(st, 5000, 1.0, {"level1": 0.4, "level2": 0.3, "region": 0.3}, 21)
    assert rf["formula"]["compaction_term_included"] is False and rf["formula"]["per_kmer"] == 24 and rf["formula"]["per_distinct"] == 8
    assert rf["alg_bytes_per_step"] == 5000 + 24 * 1000 + 8 * 100
    assert "traffic_frac" in rf and "traffic" in rf
    src = _read("bench.py")
    for key in ('"capacity_hint"', '"capacity_hint_source"', '"unhinted"', '"table_load"', '"traffic_frac"'):
        assert key in src, key
    assert "8·N_distinct" in _read("BASELINE.md") and "24·N_distinct" not in _read("BASELINE.md")


CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline", "verify"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_frac", "alg_bytes_per_step", "kernel_ms_per_step", "kernel", "dominant"}


def _check_contract_line(text, n1=True):
    assert len(text) <= 6144, len(text)          # the driver keeps an 8 KB tail of stdout: the line must fit it whole
    d = json.loads(text)
    assert CONTRACT_KEYS <= set(d), CONTRACT_KEYS - set(d)
    assert ROOFLINE_KEYS <= set(d["roofline"]) and {"kernel", "ms", "frac"} <= set(d["roofline"]["dominant"])
    assert {"workload", "k", "reads_per_gpu", "table_load", "capacity_hint"} <= set(d["config"])
    assert {"value", "unit", "cores", "kind", "sample", "optimised_value"} <= set(d["cpu_baseline"])
    assert "ok" in d["verify"]
    for row in d.get("configs", []):
        assert {"workload", "value", "ms_per_step", "frac", "verify_ok"} <= set(row) or "error" in row
    return d


def test_the_contract_line_of_a_full_report_fits_the_drivers_tail():
    """VERDICT r4 (row d regressed): round 4's line was 23.5 KB and the driver stored `parsed: null`.  bench.py now writes the
    full report to a side file and prints contract_line(full): here, the committed full report of a driver-style run goes
    through it -- every contract key present, at most 6 KB."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    tag = json.loads(_read("profiles", "hbm_traffic.json"))["tag"]
    full = _bench(tag) if not os.path.exists(os.path.join(ROOT, "profiles", f"{tag}_bench_full.json")) else json.loads(_read("profiles", f"{tag}_bench_full.json"))
    d = _check_contract_line(json.dumps(bench.contract_line(full)))
    assert {"s10m_wall_s", "s100m_wall_s", "s100m_kmers_per_s", "ok"} <= set(d["cli"]) and {"value", "ratio_to_headline"} <= set(d["unhinted"])
    assert len(d["configs"]) >= 7


@pytest.mark.gpu
def test_bench_prints_one_small_line_and_a_full_report(tmp_path):
    """The same on a real (reduced) run: `bench.py --reads 2000000` prints exactly ONE stdout line of <= 6 KB with every contract
    key, and the full report lands in the side file."""
    path = str(tmp_path / "full.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "2000000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, BENCH_FULL_PATH=path, BENCH_CLI_PAUSE_S="0.3"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = _check_contract_line(lines[0])
    assert d["verify"]["ok"] is True and d["n_gpus"] == 1 and d["full_report"]
    full = json.load(open(path))
    assert full["roofline"]["formula"]["compaction_term_included"] is False and "end_to_end" in full
