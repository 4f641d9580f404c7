#!/bin/bash
# Same-box A/B of the current tree against a checkout of an earlier commit placed under .abtest/ (git-ignored; it travels with
# the gpurun snapshot): `git worktree add -f .abtest <commit> && (cd .abtest && python __graft_entry__.py build)`, then on the
# GPU box `bash tools/ab_prev.sh`. Alternates the two builds per workload, two rounds.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for w in clip stream; do
  (cd .abtest && timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('e5m2', '$w', d['value'], d['ms_per_step'])")
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('x6  ', '$w', d['value'], d['ms_per_step'])"
done; done
