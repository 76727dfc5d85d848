#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   (run on the GPU box through gpurun)
# kernel-trace stats and PMC counters are collected in SEPARATE rocprofv3 runs.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-extra "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra "$@" > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra "$@" > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra "$@" > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_lds -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra "$@" > $OUT/bench_pmc_lds.log 2>&1
find $OUT -name "*.csv" | head -30
tail -2 $OUT/bench_trace.log | cut -c1-300
