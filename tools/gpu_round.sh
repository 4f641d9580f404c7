#!/bin/bash
# One GPU-box visit: the -m gpu tests, the default bench line, and a single-lane rocprofv3 kernel profile whose
# summaries land in gpurun_out/ (copy what should be judged into profiles/). Usage: tools/gpu_round.sh TAG [pytest-args]
TAG=$1; shift
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s "$@" > gpurun_out/${TAG}_gputest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_gputest.log | tail -2
grep -E "FAILED|Error" gpurun_out/${TAG}_gputest.log | head -5
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1 || exit 1
tail -1 gpurun_out/${TAG}_bench.log | cut -c1-400
rm -rf gpurun_out/prof_${TAG}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --lanes 1 --steps 5 --warmup 7 --no-cpu-baseline --no-pcie --no-kernel-events > gpurun_out/${TAG}_prof.log 2>&1 || exit 1
ST=$(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1); TR=$(find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" | head -1)
python3 tools/prof_summary.py "$ST" gpurun_out/${TAG}_kernel_stats.md "${TAG}: bench.py --lanes 1 --steps 5 --warmup 7 (DepthAnythingV2 vitl, batch 8, f16x3), single lane under rocprofv3"
python3 tools/prof_by_grid.py "$TR" gpurun_out/${TAG}_by_grid.md "${TAG} single-lane by grid" 58
rm -rf gpurun_out/prof_${TAG}
head -30 gpurun_out/${TAG}_kernel_stats.md
