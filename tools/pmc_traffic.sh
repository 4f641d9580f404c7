#!/bin/bash
# HBM traffic of the dominant kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only)
# over one single-lane step of the default bench workload. Usage (GPU box): tools/pmc_traffic.sh r02
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --lanes 1 --steps 1 --warmup 7 \
    --no-cpu-baseline --no-pcie --no-kernel-events > gpurun_out/pmc_$c.log 2>&1 || { tail -5 gpurun_out/pmc_$c.log; exit 1; }
done
python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE f16x3 gpurun_out/${TAG}_pmc_traffic.json gpurun_out/${TAG}_pmc_traffic_f16x3.md
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cat gpurun_out/${TAG}_pmc_traffic_f16x3.md
