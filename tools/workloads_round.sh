#!/bin/bash
# Bench lines of the other workloads (whole-clip driver, one window, streaming, v5 refiner, ViT-g) on the GPU box -> gpurun_out/r03_w_*.log
cd "$GRAFT_REPO_ROOT"
for w in video clip vstream refine5; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r03_w_$w.log 2>&1; tail -1 gpurun_out/r03_w_$w.log | cut -c1-160; done
timeout -k 10 300 python bench.py --encoder vitg --no-cpu-baseline --no-pcie --no-kernel-events > gpurun_out/r03_w_vitg.log 2>&1; tail -1 gpurun_out/r03_w_vitg.log | cut -c1-160
