#!/bin/bash
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  n=$1; shift
  rm -rf /tmp/pmc_$n
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$n -o p -- python $GRAFT_REPO_ROOT/$SCRIPT > /tmp/pmc_$n.log 2>&1
  echo "pmc $n exit $?"
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/pmc/${TAG}_$n.csv
}
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
ls -la $GRAFT_REPO_ROOT/gpurun_out/pmc
