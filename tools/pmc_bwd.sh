#!/bin/bash
# PMC counters of the chunked kernels (one --pmc group per run, kernel-trace only): bash tools/pmc_bwd.sh <outdir> [iters]
OUT=$1; N=${2:-12}
ROOT=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $ROOT/$OUT/pmc$i -- python3 $ROOT/tools/run_bwd.py $N > $ROOT/$OUT/pmc$i.log 2>&1 || echo "pmc group $i failed: $grp"
done
cd $ROOT
python3 tools/pmc_agg.py $OUT
