#!/bin/bash
# Round-end evidence on the GPU box (after tools/gpu_round.sh): PMC traffic of the dominant kernels, SQ counters of the cross-term
# GEMM under tools/x8_bench.py, the race screen, and the bench lines of the other workloads. Outputs under gpurun_out/.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
bash tools/pmc_traffic.sh r03 > gpurun_out/r03_pmc_traffic.stdout 2>&1 || echo "pmc traffic failed"
bash tools/pmc_attn.sh gemm_x8 gpurun_out/r03_pmc_x8.txt -- python3 tools/x8_bench.py > gpurun_out/r03_pmc_x8.stdout 2>&1 || echo "pmc x8 failed"
timeout -k 10 400 python tools/race_screen.py 100 > gpurun_out/r03_race_screen.log 2>&1; echo "race rc=$?"
for w in video clip vstream refine5; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r03_w_$w.log 2>&1; tail -1 gpurun_out/r03_w_$w.log | cut -c1-200; done
