#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_gpu.sh <tag> [workload]
# Writes gpurun_out/prof_<tag>/: kernel-trace stats, then SEPARATE --pmc passes (never combined with
# trace domains other than --kernel-trace), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-r01}
WL=${2:-c2}
OUT=gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 50 --warmup 10 --workload $WL --no-cpu-baseline --sustain-seconds 0"
python bench.py --workload $WL > $OUT/bench.json 2> $OUT/bench.err
# the trace pass runs bench.py's default step counts so that its per-kernel average is the same mix of warm-up and
# steady launches the bench line is measured on
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --workload $WL --no-cpu-baseline --sustain-seconds 0 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES -d $OUT/pmc_mfma -o pmc -- $CMD > $OUT/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -d $OUT/pmc_lds -o pmc -- $CMD > $OUT/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES -d $OUT/pmc_co -o pmc -- $CMD > $OUT/pmc_co.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM -d $OUT/pmc_lds2 -o pmc -- $CMD > $OUT/pmc_lds2.log 2>&1
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
