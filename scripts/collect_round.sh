#!/bin/bash
# Copy the artefacts of `scripts/gpu_job.sh final` from gpurun_out/ (scratch) into profiles/ under the round's prefix:  bash scripts/collect_round.sh r04
set -u
r=${1:?round prefix, e.g. r04}; G=gpurun_out; P=profiles
cpf() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
cpf $G/bench.json                   $P/${r}_bench_n1.json
cpf $G/prof/train_kernel_stats.csv  $P/${r}_train_bs12_bf16_kernel_stats.csv
cpf $G/prof/eval_kernel_stats.csv   $P/${r}_eval_pyramid_kernel_stats.csv
cpf $G/pmc_bench/traffic.json       $P/${r}_pmc_traffic.json
cpf $G/pmc_bench/traffic_fp32.json  $P/${r}_pmc_traffic_fp32.json
cpf $G/layer_table.json             $P/${r}_layer_table.json
cpf $G/layer_table.md               $P/${r}_layer_table.md
cpf $G/eval_layer_table.json        $P/${r}_eval_layer_table.json
cpf $G/eval_layer_table.md          $P/${r}_eval_layer_table.md
cpf $G/timeline.txt                 $P/${r}_step_timeline.txt
cpf $G/parity_report_full.txt       $P/${r}_parity_report.txt
cpf $G/gridbar.txt                  $P/${r}_gridbar.txt
cpf $G/pytest_gpu.log               $P/${r}_pytest_gpu.log
cpf $G/stamp.json                   $P/${r}_stamp.json
python scripts/kernel_resources.py > $P/${r}_kernel_resources.txt 2>/dev/null && echo "  $P/${r}_kernel_resources.txt"
