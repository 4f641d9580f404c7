"""The oracle (oracle/ref_cpu.py) against fixtures produced by the IMPORTED reference
(tools/make_golden.py, build container only). This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from common import GOLD, inputs, rel_l2, sample_idx, synth_sd
from oracle import ref_cpu as O

TOL = 2e-5


def _run_A(name, enc, check_steps=None, which="A"):
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    B, steps, H, W, sub, _ = [int(v) for v in g["meta"]]
    sd = synth_sd(which, enc)
    x = inputs(B * steps, H, W).reshape(steps, B, 3, H, W)
    mem = O.MemoryState(6)
    kept = sorted(int(k.split("_")[1]) for k in g.files if k.startswith("pre_") and not k.startswith("pre_stats"))
    last = max(kept) if check_steps is None else max(check_steps)
    with torch.no_grad():
        for t in range(last + 1):
            tr = {}
            pre = O.depth_anything_v2_forward(sd, x[t], mem, enc, pre_relu=True, trace=tr)
            if t in kept and (check_steps is None or t in check_steps):
                assert rel_l2(pre[:, ::sub, ::sub], g[f"pre_{t}"]) < TOL, (name, t)
                for i, (pt, _) in enumerate(tr["taps"]):
                    idx = sample_idx(pt.numel())
                    assert rel_l2(pt.reshape(-1)[idx], g[f"tap{i}_samp_{t}"]) < TOL
                mf = mem.items[-1]["memory_feature"]
                assert rel_l2(mf.reshape(-1)[sample_idx(mf.numel())], g[f"memfeat_samp_{t}"]) < TOL
                if f"blk0_in_samp_{t}" in g.files:  # stage fixtures G2 / G4 (forward hooks on the reference), NHWC samples
                    st = {"blk0_in": tr["tokens0"], "blk0_out": tr["block0"]}
                    st.update({f"path{k}": tr[f"path_{k}"].permute(0, 2, 3, 1) for k in (4, 3, 2, 1)})
                    for k, v in st.items():
                        ref = g[f"{k}_samp_{t}"]
                        assert rel_l2(v.reshape(-1)[sample_idx(v.numel(), len(ref))], ref) < TOL, (name, t, k)


def test_oracle_A_vits_stream_fills_and_evicts():
    # 8-frame stream: memory depth 0..6, then eviction at frame 7 (memory_bank.py:17-20)
    _run_A("A_vits_518", "vits")


def test_oracle_A_vits_batch2_nonstandard_grid():
    # 266x266 (19x19 grid): bicubic pos-embed path (dinov2.py:179-210), batch of 2 streams
    _run_A("A_vits_b2_266", "vits")


def test_oracle_A_vitb_and_stage_fixtures():
    _run_A("A_vitb_266", "vitb")
    _run_A("G_vits_392", "vits")


def _run_B(name, enc, which="B"):
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    _, T, H, W, sub, _ = [int(v) for v in g["meta"]]
    sd = synth_sd(which, enc)
    x = inputs(T, H, W).reshape(1, T, 3, H, W)
    tr = {}
    with torch.no_grad():
        pre = O.video_depth_anything_forward(sd, x, enc, pre_relu=True, trace=tr)[0]
    for k in g.files:
        if k.startswith("pre_") and k != "pre_stats_all":
            t = int(k.split("_")[1])
            assert rel_l2(pre[t, ::sub, ::sub], g[k]) < TOL, (name, t)
    for i, nm in enumerate(["layer_3", "layer_4", "path_4", "path_3"]):
        v = tr[nm]
        assert rel_l2(v.reshape(-1)[sample_idx(v.numel())], g[f"mm{i}_samp"]) < TOL
    means = np.array([pre[t].mean().item() for t in range(T)])
    assert np.allclose(means, g["pre_stats_all"][:, 0], rtol=1e-3, atol=1e-4)


def test_oracle_A_vitg_swiglu():
    """ViT-g (run_video.py:32): 40 blocks, 24 heads, SwiGLU FFN (dinov2_layers/swiglu_ffn.py), DPT features 384."""
    _run_A("A_vitg_266", "vitg")


def test_oracle_use_bn_and_use_clstoken():
    """The two constructor flags no shipped configuration enables (dpt.py:81-88,119-123; util/blocks.py:49-51,71-77):
    fixtures from the imported reference built with use_bn=True, use_clstoken=True and non-trivial BatchNorm statistics."""
    _run_A("Af_vits_266", "vits", which="Af")
    _run_B("Bf_vits_266", "vits", which="Bf")


def test_oracle_B_pe_rope():
    _run_B("Br_vits_266", "vits", which="Br")


def test_oracle_B_vits_full_window():
    _run_B("B_vits_518", "vits")


def test_oracle_B_vits_nonsquare():
    _run_B("B_vits_392x518", "vits")


def test_oracle_checkpoint_like_weights():
    """The heavy fixtures (vdn/synth.heavy_overlay: outlier channels, LayerScale over two decades, peaked heads, 1e4 MLP
    pre-activations): the oracle stays on the imported reference there too (ViT-S clip in full, ViT-L stream frame 0)."""
    _run_B("B_vits_518_heavy", "vits", which="Bh")
    _run_A("A_vitl_518_heavy", "vitl", check_steps=[0], which="Ah")


def test_oracle_A_vitl():
    _run_A("A_vitl_518", "vitl", check_steps=[0, 1])  # S = 0 and the first cross-attention over a stored frame


def test_oracle_host_pieces():
    g = np.load(os.path.join(GOLD, "host.npz"))
    for w, h, nw, nh in g["get_size"]:
        assert O.get_size(int(w), int(h)) == (int(nw), int(nh))
    assert np.array_equal(np.array(O.window_inputs(50)), g["windows_50"])
    from vdn import synth
    pe = torch.from_numpy(synth.synth_param(1234, "pretrained.pos_embed", (1, 1370, 384)))
    for (h, w) in [(224, 224), (392, 518), (266, 266)]:
        out = O.interpolate_pos_encoding(pe, (h // 14) * (w // 14), h, w)
        assert rel_l2(out.reshape(-1)[sample_idx(out.numel(), 512)], g[f"pos_{h}x{w}_samp"]) < 1e-6


def test_oracle_streaming_mode():
    g = np.load(os.path.join(GOLD, "S_vits_266.npz"))
    _, n, H, W, _, _ = [int(v) for v in g["meta"]]
    sd = synth_sd("B", "vits")
    x = inputs(n, H, W)
    st = O.StreamState()
    with torch.no_grad():
        for t in range(n):
            pre = O.video_depth_stream_step(sd, x[t][None, None], st, "vits", pre_relu=True)
            if f"pre_{t}" in g.files:
                assert rel_l2(pre, g[f"pre_{t}"]) < TOL
    assert len(st.frame_cache_list) == 42 and st.frame_id_list[0] == 0


@pytest.mark.parametrize("version,name", [(5, "R5_vits"), (4, "R4_vits"), (5, "R5f_vits")])
def test_oracle_depth_refiner_v4_v5(version, name):
    """models/video_depth_model_v{4,5}.VideoDepthAnything.forward: per-frame median scale,
    Sobel normals, encoder + temporal head, ReLU before the resize, scalar shift + residual."""
    from vdn import synth
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    v, S, H, W, seed = [int(t) for t in g["meta"]]
    assert v == version
    x = torch.from_numpy(synth.depth_clip(seed, S, H, W))[None]
    tr = {}
    with torch.no_grad():
        which = f"R{version}f" if name.startswith(f"R{version}f") else f"R{version}"   # "f": use_bn + use_clstoken
        out = O.depth_refiner_forward(synth_sd(which, "vits"), x, "vits", version=version, trace=tr)
    assert rel_l2(tr["median"], g["median"]) < 1e-6 and rel_l2(tr["scale"], g["scale"]) < 1e-6
    assert rel_l2(tr["net_depth"][0], g["net_depth"]) < TOL
    assert rel_l2(out[0], g["out"]) < TOL
