#!/bin/bash
# One single-lane rocprofv3 kernel profile of the default bench workload, printing the encoder kernels (A/B of an env switch). Usage (GPU box): tools/prof_one.sh TAG
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --lanes 1 --steps 5 --warmup 7 --no-cpu-baseline --no-pcie --no-kernel-events > gpurun_out/${TAG}_prof.log 2>&1
ST=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
python3 tools/prof_summary.py "$ST" gpurun_out/${TAG}_kernel_stats.md "$TAG" > /dev/null
rm -rf gpurun_out/prof_$TAG
grep -E "192, 102|100, 256|flash_attn2" gpurun_out/${TAG}_kernel_stats.md | cut -c1-150
