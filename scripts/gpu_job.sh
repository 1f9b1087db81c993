#!/bin/bash
# ONE parameterised entry for everything that runs on the GPU box through gpurun (replaces the one-shot gpu_round2_*.sh / gpu_r3_*.sh files):
#
#   gpurun --timeout 900 -- 'bash scripts/gpu_job.sh <job> [args]'        (several jobs: 'bash scripts/gpu_job.sh tests; bash scripts/gpu_job.sh bench')
#
#   tests [pytest args]        pytest -m gpu (default: the whole suite)           -> gpurun_out/pytest_gpu.log, parity_report.txt
#   bench [bench.py args]      the driver's bench line                            -> gpurun_out/bench.json
#   ab "ENV=.." "ENV=.." ...   alternating A/B of bench.py variants on THIS box    (scripts/gpu_ab.sh; STEPS=20)
#   prof [train|eval]          rocprofv3 --kernel-trace --stats of the bench cmd   -> gpurun_out/prof/<leg>_kernel_stats.csv
#   timeline [ENV=..]          kernel trace of 3 steps -> per-queue timeline       -> gpurun_out/timeline[_tag].txt
#   pmc-traffic                FETCH_SIZE / WRITE_SIZE passes over the bench cmd   -> gpurun_out/pmc_bench/traffic.json (scripts/gpu_pmc_bench.sh)
#   pmc-sq SCRIPT              one SQ counter pass over `python SCRIPT`            -> gpurun_out/pmc/sq_summary.txt
#   pmc-all SCRIPT             SQ + FETCH_SIZE + WRITE_SIZE passes over `python SCRIPT` (TAG names the files) -> gpurun_out/pmc/<TAG>_{sq,fetch,write}.csv
#   ubench SCRIPT [args]       a micro-benchmark under the kernel tracer: TRUE kernel durations (TAG, PAT=kernel-name regex) -> gpurun_out/ub/<TAG>.txt
#   eval-lanes                 A/B of the evaluation pyramid over lane count / enqueue order -> gpurun_out/eval_lanes.txt
#   layer-table                bench.py --layer-table (every MFMA launch bracketed)-> gpurun_out/layer_table.{json,md}
#   eval-table                 the same table for the configs[1] pyramid (per image)-> gpurun_out/eval_layer_table.{json,md}
#   contention | floor | nms   the round-3 micro-benchmarks (scripts/contention.py, floor.py, nms_bench.py under the tracer)
#   dist-smoke                 RCCL 1-rank group, 2 gloo ranks on one GPU, 2 nccl ranks on one GPU (expected refusal) -> gpurun_out/dist_smoke.txt
#   final                      tests + bench + prof + pmc-traffic + layer-table + timeline + eval prof + smoke: the artefacts of a round
set -u
job=${1:-help}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
export PYTHONUNBUFFERED=1
BARGS="--no-cpu-baseline --no-eval --no-profile --no-fp32-path"

trace_cmd() {   # trace_cmd <outdir> <extra rocprof flags> -- <cmd...>   (run from /tmp: rocprofv3 writes beside its cwd otherwise)
  local out=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf "$out" && timeout ${TRACE_TIMEOUT:-300} rocprofv3 --kernel-trace --output-format csv -d "$out" -o t "$@" ) > "$R/gpurun_out/rocprof_last.log" 2>&1
  echo "rocprofv3 exit $?"
}

case "$job" in
  tests)
    rm -f "$R/gpurun_out/parity_report.txt"
    sel=tests; case "${1:-}" in tests/*) sel=; esac          # explicit test files replace the whole suite
    timeout ${TEST_TIMEOUT:-900} python -m pytest $sel -q -m gpu -p no:cacheprovider --durations=8 "$@" > "$R/gpurun_out/pytest_gpu.log" 2>&1
    echo "pytest exit $?" | tee -a "$R/gpurun_out/pytest_gpu.log"; tail -12 "$R/gpurun_out/pytest_gpu.log" ;;
  bench)
    timeout ${BENCH_TIMEOUT:-900} python bench.py "$@" > "$R/gpurun_out/bench.json" 2> "$R/gpurun_out/bench.err"
    echo "bench exit $?"; tail -2 "$R/gpurun_out/bench.err"; python -c "
import json; d=json.load(open('$R/gpurun_out/bench.json')); r=d.get('roofline',{})
print(d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'fwd', r.get('forward_pass',{}).get('ms'), 'eval', d.get('eval',{}).get('ms_per_image'), 'hard', d.get('eval_hard',{}).get('ms_per_image'), 'fp32', d.get('fp32_path',{}).get('img_s'), 'cpu', d.get('cpu_baseline',{}).get('value'))" ;;
  ab)
    STEPS=${STEPS:-20} bash scripts/gpu_ab.sh "$@" 2>&1 | tee "$R/gpurun_out/ab.txt" ;;
  prof)
    leg=${1:-train}
    if [ "$leg" = eval ]; then cmd="python $R/bench.py --eval-only"; else cmd="python $R/bench.py --steps 7 --warmup 3 $BARGS"; fi
    trace_cmd /tmp/prof_$leg --stats -- $cmd
    mkdir -p "$R/gpurun_out/prof"; f=$(find /tmp/prof_$leg -name "*kernel_stats.csv" | head -1); cp "$f" "$R/gpurun_out/prof/${leg}_kernel_stats.csv"; head -12 "$f" | cut -c1-150 ;;
  timeline)
    tag=${TAG:-}; trace_cmd /tmp/tl -- env "$@" python $R/bench.py --steps 3 --warmup 2 $BARGS
    f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1); python scripts/trace_timeline.py "$f" > "$R/gpurun_out/timeline${tag:+_$tag}.txt" 2>&1; head -12 "$R/gpurun_out/timeline${tag:+_$tag}.txt" ;;
  pmc-traffic) bash scripts/gpu_pmc_bench.sh ;;
  pmc-sq)
    mkdir -p "$R/gpurun_out/pmc"
    ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_sq && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_sq -o p -- python $R/$1 > /tmp/pmc_sq.log 2>&1 ); echo "pmc exit $?"
    f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1); python scripts/pmc_summary.py "$f" "${2:-conv|wgrad}" | tee "$R/gpurun_out/pmc/sq_summary.txt" ;;
  pmc-all)
    mkdir -p "$R/gpurun_out/pmc"; tag=${TAG:-pmc}
    for pass in "sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
      set -- $pass; n=$1; shift       # separate passes, --kernel-trace only (MI355X_MICROARCH.md; gpurun refuses --pmc beside the sys / hip traces)
      ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_$n && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$n -o p -- python $R/${SCRIPT:?SCRIPT=scripts/...} > /tmp/pmc_$n.log 2>&1 ); echo "pmc $n exit $?"
      f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" "$R/gpurun_out/pmc/${tag}_$n.csv"
    done ;;
  ubench)
    mkdir -p "$R/gpurun_out/ub"; tag=${TAG:-ub}
    ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ub_$tag && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ub_$tag -o ub -- python $R/"$@" > "$R/gpurun_out/ub/$tag.log" 2>&1 )
    f=$(find /tmp/ub_$tag -name "*kernel_trace.csv" | head -1); python scripts/trace_summary.py "$f" "${PAT:-conv_dma|wgrad}" | tee "$R/gpurun_out/ub/$tag.txt" ;;
  eval-lanes)
    out="$R/gpurun_out/eval_lanes.txt"; : > "$out"
    for rep in 1 2 3; do for cfg in "1 big" "2 big" "3 big" "2 small" "3 small"; do
      set -- $cfg
      ms=$(TINYFACES_EVAL_LANES=$1 TINYFACES_EVAL_LANES_ORDER=$2 python bench.py --eval-only 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['eval']['ms_per_image'])")
      echo "[$rep] lanes=$1 order=$2 -> $ms ms/image" | tee -a "$out"
    done; done ;;
  layer-table)
    timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-eval --no-fp32-path --layer-table "$R/gpurun_out/layer_table.json" > "$R/gpurun_out/bench_layer.json" 2> "$R/gpurun_out/bench_layer.err"; echo "layer-table exit $?"; head -12 "$R/gpurun_out/layer_table.md" ;;
  eval-table)
    timeout 300 python bench.py --eval-only --layer-table "$R/gpurun_out/eval_layer_table.json" > "$R/gpurun_out/bench_eval_layer.json" 2> "$R/gpurun_out/bench_eval_layer.err"; echo "eval-table exit $?"; head -40 "$R/gpurun_out/eval_layer_table.md" ;;
  contention) timeout 600 python scripts/contention.py 2>&1 | grep -v amdgpu.ids | tee "$R/gpurun_out/contention.txt" | tail -40 ;;
  floor)
    python scripts/floor.py 2>&1 | grep -v amdgpu.ids | tee "$R/gpurun_out/floor.txt"
    trace_cmd /tmp/fl --stats -- python $R/scripts/floor.py; f=$(find /tmp/fl -name "*kernel_stats.csv" | head -1); head -4 "$f" | cut -c1-160 | tee -a "$R/gpurun_out/floor.txt" ;;
  nms) bash scripts/gpu_nms.sh ;;
  dist-smoke) bash scripts/gpu_dist_smoke.sh ;;
  final)
    # the profile passes FIRST: bench.py quotes the committed kernel-stats / PMC files, so they are refreshed in the box's copy of profiles/
    # (and come home through gpurun_out/) before the bench line is taken with the same binary
    rp=${ROUND:-r06}
    bash "$0" prof train; cp "$R/gpurun_out/prof/train_kernel_stats.csv" "$R/profiles/${rp}_train_bs12_bf16_kernel_stats.csv"
    python scripts/stamp_profiles.py "$R/profiles/${rp}_stamp.json" > /dev/null; cp "$R/profiles/${rp}_stamp.json" "$R/gpurun_out/stamp.json"     # the library these profiles belong to
    bash "$0" pmc-traffic; cp "$R/gpurun_out/pmc_bench/traffic.json" "$R/profiles/${rp}_pmc_traffic.json"; cp "$R/gpurun_out/pmc_bench/traffic_fp32.json" "$R/profiles/${rp}_pmc_traffic_fp32.json"
    bash "$0" bench
    bash "$0" tests; cp "$R/gpurun_out/parity_report.txt" "$R/gpurun_out/parity_report_full.txt" 2>/dev/null
    bash "$0" layer-table; bash "$0" timeline; bash "$0" prof eval; bash "$0" eval-table
    python scripts/gridbar.py 2>&1 | grep -v amdgpu.ids > "$R/gpurun_out/gridbar.txt"; tail -5 "$R/gpurun_out/gridbar.txt"
    timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 ;;
  *) sed -n 2,22p "$0" ;;
esac
