#!/bin/bash
# SQ counter passes for the hot kernels (run through gpurun).  usage: bash tools/sq_probe.sh <tag>
set -u
TAG=${1:-sq}
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-verify ${BENCH_ARGS:-}"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -- python3 bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -- python3 bench.py $ARGS > $OUT/p2.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("p1","p2"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv"%d):
        acc=collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"].split("(")[0].replace("void kh::","")
            if any(t in n for t in ("part1","part2","region_count")):
                acc[n][r["Counter_Name"]]=acc[n].get(r["Counter_Name"],0)+float(r["Counter_Value"])
        for n,c in acc.items():
            print(n[:48], {k:(f"{v:.3g}") for k,v in c.items()})
PY
