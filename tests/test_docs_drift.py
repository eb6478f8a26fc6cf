"""Drift guards between the documents, the committed measurements and the source (CPU only, like
tests/test_ffi_drift.py for the FFI): the numbers a reader is given must be the ones in profiles/, and the kernels the
documents name must exist."""
import json
import os
import re

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
    src = _read("krust_amd", "csrc", "partition.hip.h") + _read("krust_amd", "csrc", "kernels.hip.h") + \
        _read("krust_amd", "csrc", "rawparse.hip.h") + _read("krust_amd", "csrc", "shard.hip.h")
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
