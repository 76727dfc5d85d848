#!/bin/bash
# SQ issue/wait counters of the kernels (separate PMC passes, no tracing).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra "$@" > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq3 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra "$@" > $OUT/c.log 2>&1
python3 $R/tools/summarize_prof.py $OUT | grep -v "at::native\|rocclr"
tail -3 $OUT/c.log | cut -c1-200
