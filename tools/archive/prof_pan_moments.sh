#!/bin/bash
# Counters of the young-pixel moments launch under the bench's pan (separate PMC passes, no tracing): tools/archive/prof_pan_moments.sh <tag>
TAG=${1:-pan}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--motion pan --no-cpu --no-extra --steps 5 --warmup 1 --windows 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc2 -- python3 $R/bench.py $ARGS > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("pmc1", "pmc2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            k = "moments_young_kernel" if "moments_young" in name else "temporal_kernel" if "temporal_kernel" in name else None
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(k, {n: round(sum(v) / len(v), 1) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "moments" in r["Name"] or "temporal" in r["Name"]:
            print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
