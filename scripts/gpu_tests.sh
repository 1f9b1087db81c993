#!/bin/bash
# Run on the GPU box via gpurun: full -m gpu suite (no -x) + smoke, logs into gpurun_out/.
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
export PYTHONUNBUFFERED=1
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 300 ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
