#!/bin/bash
# HBM traffic of the training step per kernel: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, with
# --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) over the bench command -> gpurun_out/pmc_bench/
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/pmc_bench
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcb_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eval --no-profile --no-fp32-path > /tmp/pmcb_$c.log 2>&1
  echo "pmc $c exit $?"
  f=$(find /tmp/pmcb_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" /tmp/pmcb_$c.csv
done
python $GRAFT_REPO_ROOT/scripts/pmc_traffic.py /tmp/pmcb_FETCH_SIZE.csv /tmp/pmcb_WRITE_SIZE.csv $GRAFT_REPO_ROOT/gpurun_out/pmc_bench/traffic.json
# r4: the same two passes over the fp32 parity path (bench.py --dtype fp32) -> traffic_fp32.json (the fp32_path roofline block of bench.py)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcf_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcf_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --dtype fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-eval --no-profile --no-fp32-path > /tmp/pmcf_$c.log 2>&1
  echo "pmc fp32 $c exit $?"
  f=$(find /tmp/pmcf_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" /tmp/pmcf_$c.csv
done
python $GRAFT_REPO_ROOT/scripts/pmc_traffic.py /tmp/pmcf_FETCH_SIZE.csv /tmp/pmcf_WRITE_SIZE.csv $GRAFT_REPO_ROOT/gpurun_out/pmc_bench/traffic_fp32.json
