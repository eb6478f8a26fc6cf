"""Generates tests/golden/derived_fixture_tables.json.

These tables are DERIVED (output of the oracle restatement on the byte-exact
re-created fixtures), not reference-run: the reference cannot be built here.
They are consistent with every assertion the reference's tests make on these
fixtures and were cross-checked against the survey's
hand-derived tables.  Run from the repo root:  python tests/golden/make_derived.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
from test_oracle_golden import _parse_fixture  # noqa: E402

ROWS = [("simple.fa", 3, None), ("simple.fa", 4, None), ("simple.fa", 5, None), ("simple.fa", 7, None),
        ("with_n.fa", 3, None), ("with_n.fa", 4, None), ("with_n.fa", 5, None),
        ("simple.fq", 4, None), ("with_n.fq", 3, 20),
        ("low_quality.fq", 4, 20), ("low_quality.fq", 4, None), ("low_quality.fq", 4, 0),
        ("soft_masked.fa", 3, None)]

tables = []
for fx, k, q in ROWS:
    recs, quals = _parse_fixture(os.path.join(os.path.dirname(HERE), "fixtures", fx))
    m = O.count_records(recs, k, quals=quals, min_quality=q)
    tables.append({"fixture": fx, "k": k, "min_quality": q, "label": "derived",
                   "distinct": len(m), "total": m.total(),
                   "counts": dict(sorted(m.as_str_dict(k).items())),
                   "histogram": m.histogram(1)})
with open(os.path.join(HERE, "derived_fixture_tables.json"), "w") as f:
    json.dump({"_comment": "DERIVED by tests/golden/make_derived.py from the oracle; see that script.",
               "tables": tables}, f, indent=1)
print("wrote", len(tables), "tables")
