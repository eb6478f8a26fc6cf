#!/bin/bash
# SQ counter passes for kernels matching a pattern.  usage: bash tools/sq_probe2.sh <tag> "<kernel name substrings, |-separated>" <bench args...>
set -u
TAG=$1; PAT=$2; shift 2
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -- python3 bench.py "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -- python3 bench.py "$@" > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_GDS --output-format csv -d $OUT/p3 -- python3 bench.py "$@" > $OUT/p3.log 2>&1
python3 - "$PAT" <<PY | tee $OUT/summary.txt
import csv,glob,collections,sys
pats=sys.argv[1].split("|")
for d in ("p1","p2","p3"):
    for f in glob.glob("$OUT/%s/*/*_counter_collection.csv"%d):
        acc=collections.defaultdict(dict); calls=collections.Counter()
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"].split("(")[0].replace("void kh::","")
            if any(t in n for t in pats):
                acc[n][r["Counter_Name"]]=acc[n].get(r["Counter_Name"],0)+float(r["Counter_Value"])
                calls[(n,r["Counter_Name"])]+=1
        for n,c in acc.items():
            print(n[:56], "calls", max(v for (nn,_),v in calls.items() if nn==n), {k:(f"{v:.3g}") for k,v in c.items()})
PY
find $OUT -name "*.csv" -size +2M -delete
