#!/bin/bash
# SQ counter pass (one rocprofv3 --pmc run with --kernel-trace only) over the layer-shape kernels incl. conv3x3h, final binary
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_sq -o p -- python $GRAFT_REPO_ROOT/scripts/pmc_kernels_r2.py > /tmp/pmc_sq.log 2>&1
echo "pmc sq exit $?"
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/pmc/r2f_sq.csv
cd $GRAFT_REPO_ROOT && python scripts/pmc_summary.py gpurun_out/pmc/r2f_sq.csv "conv_dma|conv3x3h|wgrad" | tee gpurun_out/pmc/r2f_sq_summary.txt
