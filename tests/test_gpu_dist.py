"""The N > 1 path with the real kernels: R ranks share this box's one GPU under gloo — vdn/dist.py
stages device tensors through the host for gloo groups — and run the product's `infer_video_depth_sharded` (schedule,
subgroups, encoder tap exchange, TemporalEngine.run_sharded's frames<->pixels all-to-all on the HIP engine, gather,
device stitcher). Rank 0 runs the same clip alone through `infer_video_depth`; tools/dist_rehearsal.py fails unless the
two depth videos agree within 2e-5 rel-L2 (the sharded temporal modules see other GEMM shapes: fp32 summation order)
and every rank received the same stitched clip. RCCL itself needs more than one GPU and is the driver's run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks,frames,schedule", [
    (2, 50, "[(0, 0, 1), (1, 1, 1), (2, 0, 2)]"),   # two whole windows, then one window frame-sharded over both ranks
    (4, 20, "[(0, 0, 4)]"),                         # one window, 8 frames per rank
    (3, 20, "[(0, 0, 2)]"),                         # rank 2 owns no head job (idle-rank path), still encodes its frames
    (4, 130, "[(0, 0, 1), (1, 1, 1), (2, 2, 1), (3, 3, 1), (4, 0, 2), (5, 2, 2)]"),   # hybrid: 4 whole windows, then 2 windows on 2 ranks each
    (4, 256, "[(0, 0, 1), (1, 0, 1), (2, 0, 1), (3, 1, 1), (4, 1, 1), (5, 1, 1), (6, 2, 1), (7, 2, 1), (8, 2, 1), (9, 3, 1), (10, 3, 1), (11, 3, 1)]"),   # configs[3]'s 12-window clip: contiguous runs of 3 windows per rank, tap exchange between neighbours
])
def test_sharded_clip_equals_single_process_on_real_kernels(ranks, frames, schedule):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_rehearsal.py"), "--ranks", str(ranks),
                        "--frames", str(frames)], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("[rehearsal]") or l.startswith("REHEARSAL")]
    print("\n".join(lines))
    assert p.returncode == 0 and "REHEARSAL OK" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
    assert f"schedule {schedule}" in p.stdout
