"""End-to-end parity on the MI355X: the HIP path (through the C-ABI) against the oracle on the same
seeded inputs and against the committed fixtures from the imported reference.
Tolerance: 1e-3 relative on the fp32 depth map — BASELINE.json north_star — applied twice: to the rel-L2 of the whole map and
to the WORST pixel (max |got - ref| <= 1e-3 max |ref|), on the pre-ReLU map as well as on the returned (post-ReLU) one."""
import os

import numpy as np
import pytest
import torch

from common import GOLD, inputs, rel_l2, sample_idx, stats, synth_sd, worst_px

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _product(which, enc, precision=None):
    import vdn
    cls = vdn.DepthAnythingV2 if which[0] == "A" else vdn.VideoDepthAnything
    flags = dict(use_bn=True, use_clstoken=True) if which.endswith("f") else {}   # "Af" / "Bf": the two optional constructor flags
    if which == "Br":
        flags = dict(pe="rope")
    m = cls(**dict(vdn.MODEL_CONFIGS[enc], **flags))
    m.load_state_dict(synth_sd(which, enc), strict=True)
    if precision:
        m.set_precision(precision)
    return m.to("cuda").eval()


def _stream_A(name, enc, oracle_steps, which="A"):
    from oracle import ref_cpu as O
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    B, steps, H, W, sub, _ = [int(v) for v in g["meta"]]
    model = _product(which, enc)
    sd = synth_sd(which, enc)
    x = inputs(B * steps, H, W).reshape(steps, B, 3, H, W)
    kept = sorted(int(k.split("_")[1]) for k in g.files if k.startswith("pre_") and not k.startswith("pre_stats"))
    mem = O.MemoryState(6)
    worst = 0.0
    for t in range(max(kept) + 1):
        pre = model.forward(x[t].cuda(), _pre_relu=True).cpu()
        assert torch.isfinite(pre).all()
        if t in kept:
            e = rel_l2(pre[:, ::sub, ::sub], g[f"pre_{t}"])
            e_post = rel_l2(torch.relu(pre[:, ::sub, ::sub]), np.maximum(g[f"pre_{t}"], 0))
            w_pre = worst_px(pre[:, ::sub, ::sub], g[f"pre_{t}"])
            mf = model._eng["rt"].hbuf("mem_feat", (B * (H // 14) * (W // 14), model.pretrained.embed_dim)).float().cpu()
            e_mf = rel_l2(mf.reshape(-1)[sample_idx(mf.numel())], g[f"memfeat_samp_{t}"])
            print(f"[{name}] frame {t}: vs reference fixture pre-ReLU {e:.2e} post-ReLU {e_post:.2e} worst pixel {w_pre:.2e} memory feature {e_mf:.2e}")
            worst = max(worst, e_post)
            assert e_post < TOL and e < TOL and w_pre < TOL and e_mf < 2e-3, (name, t, e, e_post, w_pre, e_mf)
        if t < oracle_steps:
            with torch.no_grad():
                ref = O.depth_anything_v2_forward(sd, x[t], mem, enc, pre_relu=True)
            e, w_full = rel_l2(torch.relu(pre), torch.relu(ref)), worst_px(pre, ref)
            print(f"[{name}] frame {t}: vs oracle (full map) post-ReLU {e:.2e}, worst pixel of the full pre-ReLU map {w_full:.2e}")
            assert e < TOL and w_full < TOL
    return worst


def test_A_vits_stream_fill_and_evict():
    """8-frame stream on one memory bank: depth 0..6 then eviction (memory_bank.py:17-20)."""
    _stream_A("A_vits_518", "vits", oracle_steps=3)


def test_A_vits_batch2_266():
    """Two streams in one batch on a 19x19 grid (bicubic pos-embed path, dinov2.py:179-210)."""
    _stream_A("A_vits_b2_266", "vits", oracle_steps=3)


def test_A_vitl_518_stream_fill_and_evict():
    """BASELINE configs[1] at batch 1: ViT-L, 8 frames on one memory bank — every depth S = 0..6, then one eviction —
    against the fixture written by the imported reference (frames 0, 1, 6, 7 kept)."""
    _stream_A("A_vitl_518", "vitl", oracle_steps=1)


def test_A_vitg_266_stream():
    """ViT-g (run_video.py:32): 40 blocks of 24 heads with the SwiGLU FFN (gated epilogue with a SiLU gate, halves swapped at
    packing), DPT features 384 / out_channels 1536: two frames on one memory bank against the reference fixture."""
    _stream_A("A_vitg_266", "vitg", oracle_steps=1)


def test_A_use_bn_and_use_clstoken():
    """DepthAnythingV2(use_bn=True, use_clstoken=True): BatchNorm (non-trivial running statistics) folded into the fusion
    blocks' convolutions, cls-token readout on taps 0-2 after the encoder and on the memory block's output for tap 3
    (dpt.py:81-88,119-123; util/blocks.py:49-51,71-77), three frames on one memory bank against the reference fixture."""
    _stream_A("Af_vits_266", "vits", oracle_steps=2, which="Af")


def test_A_vitb_266_stream():
    """ViT-B (12 blocks, 12 heads, taps 2/5/8/11, DPT features 128): three frames on one memory bank against the fixture
    written by the imported reference."""
    _stream_A("A_vitb_266", "vitb", oracle_steps=2)


@pytest.mark.parametrize("name,enc", [("G_vits_392", "vits"), ("A_vitb_266", "vitb")])
def test_stage_fixtures_vit_block_and_dpt_paths(name, enc):
    """SURVEY.md §8c fixtures G2 and G4, taken with forward hooks on the imported reference: the token stream entering and
    leaving ViT block 0 and leaving the last block (patch embed + pos-embed, one full block, the whole stack), the four
    FeatureFusionBlock outputs path_4..path_1 and output_conv1 — so a regression in one DPT stage shows at that stage
    and not only as an error of the final map. 1024 sampled values (NHWC order) and mean/std per tensor."""
    import vdn
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    B, steps, H, W, _, _ = [int(v) for v in g["meta"]]
    model = _product("A", enc)
    e = model._engines()
    rt, toks = e["rt"], {}
    e["enc"].probe = lambda i, tok: toks.__setitem__(i, tok.float().cpu().clone())
    x = inputs(B * steps, H, W).reshape(steps, B, 3, H, W)
    depth, F, ph = vdn.modules.ENCODERS[enc]["depth"], vdn.MODEL_CONFIGS[enc]["features"], H // 14
    for t in range(steps):
        model.forward(x[t].cuda())
        if f"blk0_in_samp_{t}" not in g.files:
            continue
        got = {"blk0_in": toks[-1], "blk0_out": toks[0], "blkL_out": toks[depth - 1]}
        for k, s_ in ((4, ph), (3, 2 * ph), (2, 4 * ph), (1, 8 * ph)):
            got[f"path{k}"] = rt.hbuf(f"path{k}", (B * s_ * s_, F)).float().cpu()
        got["oc1"] = rt.fbuf("out1_f32", (B * 64 * ph * ph, F // 2)).cpu()
        for k, v in got.items():
            ref = g[f"{k}_samp_{t}"]
            err = rel_l2(v.reshape(-1)[sample_idx(v.numel(), len(ref))], ref)
            st, rs = stats(v), g[f"{k}_stats_{t}"]
            print(f"[{name}] frame {t} stage {k} {tuple(v.shape)}: samples rel-L2 {err:.2e}, mean {st[0]:.5f} / {rs[0]:.5f}, std {st[1]:.5f} / {rs[1]:.5f}")
            assert err < TOL and abs(st[1] - rs[1]) <= 1e-3 * rs[1] and abs(st[0] - rs[0]) <= 1e-3 * rs[1], (k, t, err, st, rs)
    e["enc"].probe = None


def _lanes_against_single_lane_and_fixture(name, enc, B, monkeypatch, which="A"):
    """The dispatch bench.py times: a batch of B independent streams dealt to two HIP-stream lanes (own workspace,
    cu_hint = 128, one shared memory-bank ring) for 8 steps — empty bank, filling, full, eviction. Every step must
    equal the single-lane run of the same batch, and batch element 0 carries the fixture's stream, so it must also
    match the reference fixture: lanes, batching and bank slicing change nothing."""
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    _, steps, H, W, sub, _ = [int(v) for v in g["meta"]]
    kept = sorted(int(k.split("_")[1]) for k in g.files if k.startswith("pre_") and not k.startswith("pre_stats"))
    pool = inputs(steps + B - 1, H, W)  # element b sees the fixture's stream delayed by b frames
    model = _product(which, enc)
    runs = {}
    for lanes in ("2", "1"):
        monkeypatch.setenv("VDN_STREAMS", lanes)
        model.clear_memory()
        outs = []
        for t in range(steps):
            xb = torch.stack([pool[t + b] for b in range(B)])
            outs.append(model.forward(xb.cuda(), _pre_relu=True).cpu())
            assert torch.isfinite(outs[-1]).all()
        runs[lanes] = outs
        if lanes == "2":
            assert model._lanes is not None and len(model._lanes) == 2, "the two-lane path did not run"
    worst_l, worst_f = 0.0, 0.0
    for t in range(steps):
        e = rel_l2(runs["2"][t], runs["1"][t])
        worst_l = max(worst_l, e)
        # same arithmetic; only the split-K slicing follows the lane's CU share — unless the lane's half batch falls under the
        # row count from which the encoder linears run on the 8-bit cross-term kernel (engine.py: M >= 4096) while the whole
        # batch does not: two fp32-faithful kernels then differ by their cross-term rounding (2^-14 per term)
        N = (H // 14) * (W // 14) + 1
        assert e < (3e-6 if (B // 2 * N >= 4096) == (B * N >= 4096) else 3e-4), (t, e)
        if t in kept:
            for lanes in ("2", "1"):
                ef = rel_l2(torch.relu(runs[lanes][t][0, ::sub, ::sub]), np.maximum(g[f"pre_{t}"][0], 0))
                worst_f = max(worst_f, ef)
                assert ef < TOL, (lanes, t, ef)
    print(f"[{name} B={B}] two lanes vs one lane: worst step {worst_l:.2e}; element 0 vs reference fixture: worst {worst_f:.2e}")


def test_A_vitl_batch8_two_lanes_benchmarked_dispatch(monkeypatch):
    """BASELINE configs[1] exactly as bench.py runs it: DepthAnythingV2(vitl), batch 8, two lanes, S = 0..6 + eviction."""
    _lanes_against_single_lane_and_fixture("A_vitl_518", "vitl", 8, monkeypatch)


def test_A_vits_batch4_two_lanes(monkeypatch):
    _lanes_against_single_lane_and_fixture("A_vits_518", "vits", 4, monkeypatch)


def test_memory_bank_rejects_a_batch_change_until_cleared():
    """One ring for the whole batch: a frame whose batch differs from the stored memories is an error, as in the
    reference (memory_attention.py:135-137), not a silent reset."""
    model = _product("A", "vits")
    x = inputs(2, 266, 266)
    model.forward(x[:1].cuda())
    with pytest.raises(RuntimeError, match="clear_memory"):
        model.forward(x.cuda())
    model.clear_memory()
    assert torch.isfinite(model.forward(x.cuda())).all()


def _clip_B(name, enc, use_oracle, which="B"):
    from oracle import ref_cpu as O
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    _, T, H, W, sub, _ = [int(v) for v in g["meta"]]
    model = _product(which, enc)
    x = inputs(T, H, W).reshape(1, T, 3, H, W)
    pre = model.forward(x.cuda(), _pre_relu=True)[0].cpu()
    assert torch.isfinite(pre).all()
    for k in g.files:
        if k.startswith("pre_") and k != "pre_stats_all":
            t = int(k.split("_")[1])
            e = rel_l2(pre[t, ::sub, ::sub], g[k])
            e_post = rel_l2(torch.relu(pre[t, ::sub, ::sub]), np.maximum(g[k], 0))
            w_pre = worst_px(pre[t, ::sub, ::sub], g[k])
            print(f"[{name}] frame {t}: vs reference fixture pre-ReLU {e:.2e} post-ReLU {e_post:.2e} worst pixel {w_pre:.2e}")
            assert e_post < TOL and e < TOL and w_pre < TOL
    means = np.array([pre[t].mean().item() for t in range(T)])
    assert np.allclose(means, g["pre_stats_all"][:, 0], rtol=5e-3, atol=2e-3)
    if use_oracle:
        with torch.no_grad():
            ref = O.video_depth_anything_forward(synth_sd(which, enc), x, enc, pre_relu=True)[0]
        e, w_full = rel_l2(torch.relu(pre), torch.relu(ref)), worst_px(pre, ref)
        print(f"[{name}] all {T} frames vs oracle post-ReLU {e:.2e}, worst pixel of the full pre-ReLU maps {w_full:.2e}")
        assert e < TOL and w_full < TOL


def test_B_vits_full_window():
    _clip_B("B_vits_518", "vits", use_oracle=True)


def test_B_use_bn_and_use_clstoken():
    """VideoDepthAnything(use_bn=True, use_clstoken=True): the readout is per frame, so every tap gets it right after the
    encoder (also in the tap cache, the streaming and the sharded drivers); 4-frame clip against the reference fixture."""
    _clip_B("Bf_vits_266", "vits", use_oracle=True, which="Bf")


def test_A_vitl_checkpoint_like_weights():
    """Range test (fixture from the imported reference with vdn/synth.heavy_overlay): residual-stream outlier channels at
    +-150..550, LayerScale over two decades, attention logits spread over +-170, MLP pre-activations of 1e4 — what a trained
    DINOv2 checkpoint does to the fp16 planes and the e5m2 cross terms. Frames 0-2 (memory depth 0, 1, 2), ViT-L 518 x 518."""
    _stream_A("A_vitl_518_heavy", "vitl", oracle_steps=0, which="Ah")


def test_A_vitl_checkpoint_like_weights_batch8_two_lanes(monkeypatch):
    """The same heavy weights at batch 8 (two lanes of M = 5480 rows: the encoder linears run on the 8-bit cross-term kernel,
    whose e5m2 planes see the 1e4 activations and the +-550 outlier channels)."""
    _lanes_against_single_lane_and_fixture("A_vitl_518_heavy", "vitl", 8, monkeypatch, which="Ah")


def test_fp16_plane_range_is_wide_and_its_overflow_is_loud():
    """Range of the 16-bit operand planes (INTEGRATION.md 'Range'): hi = RTZ fp16 saturates at 65 504 and lo carries the
    rest, so an MLP activation of ~1e5 is still represented (parity with the fp32 oracle holds); at ~1e6 the planes
    overflow and the image / clip drivers raise instead of returning a NaN map."""
    import vdn
    from oracle import ref_cpu as O
    from vdn import synth
    sd = {k: v.clone() for k, v in synth_sd("Ah", "vits").items()}
    x = inputs(1, 266, 266)
    img = np.ascontiguousarray(synth.frames_u8(1234, 1, 266, 266)[0][:, :, ::-1])
    for gain, ok in ((9.0, True), (100.0, False)):   # the 1e4 unit of block 1 -> ~1e5 (inside) / ~1e6 (outside)
        sd2 = {k: v.clone() for k, v in sd.items()}
        sd2["pretrained.blocks.1.mlp.fc1.weight"][77] *= gain
        sd2["pretrained.blocks.1.mlp.fc1.bias"][77] *= gain
        model = vdn.DepthAnythingV2(**vdn.MODEL_CONFIGS["vits"])
        model.load_state_dict(sd2, strict=True)
        model = model.to("cuda").eval()
        if ok:
            got = model.forward(x.cuda(), _pre_relu=True).cpu()
            with torch.no_grad():
                ref = O.depth_anything_v2_forward(sd2, x, O.MemoryState(6), "vits", pre_relu=True)
            e, w = rel_l2(got, ref), worst_px(got, ref)
            print(f"[range] MLP activations ~1e5 (hi plane saturated, lo carries the rest): rel-L2 {e:.2e}, worst pixel {w:.2e}")
            assert e < TOL and w < TOL
            assert np.isfinite(model.infer_image(img, 266)).all()
        else:
            with pytest.raises(FloatingPointError, match="fp16 range"):
                model.infer_image(img, 266)


def test_B_vits_checkpoint_like_weights():
    _clip_B("B_vits_518_heavy", "vits", use_oracle=True, which="Bh")


def test_B_pe_rope():
    """VideoDepthAnything(pe='rope') (motion_module.py:236-240,279-282): the temporal attention's q / k rotated by the frame
    index on load (vdn_temporal_attn rope_cs) instead of the additive table; 8-frame clip against the imported reference."""
    _clip_B("Br_vits_266", "vits", use_oracle=True, which="Br")


def test_B_vits_nonsquare_short_clip():
    _clip_B("B_vits_392x518", "vits", use_oracle=True)


def test_B_vitl_4_frames():
    _clip_B("B_vitl_518", "vitl", use_oracle=False)


def test_B_vitl_full_32_frame_window():
    """BASELINE configs[2]: VideoDepthAnything(vitl) on the full 32-frame 518x518 window, against the fixture written
    by the imported reference (frames 0, 13, 31 and every frame's mean)."""
    _clip_B("B_vitl_518_T32", "vitl", use_oracle=False)


def test_determinism_and_memory_reset():
    """Size-independent properties: same stream after clear_memory() is bit-identical; a batch of two
    identical streams gives identical rows; frames are independent of batch position."""
    model = _product("A", "vits")
    x = inputs(3, 518, 518)
    outs = []
    for _ in range(2):
        model.clear_memory()
        outs.append([model.forward(x[t:t + 1].cuda()).cpu() for t in range(3)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    model.clear_memory()
    two = [model.forward(torch.stack([x[t], x[t]]).cuda()).cpu() for t in range(3)]
    for t in range(3):
        assert torch.equal(two[t][0], two[t][1])
        assert rel_l2(two[t][0], outs[0][t][0]) < 1e-5


def test_video_clip_batch_equals_single():
    """B=2 clips in one call == each clip alone (frames only mix inside their own clip)."""
    model = _product("B", "vits")
    x = inputs(8, 266, 266).reshape(2, 4, 3, 266, 266)
    both = model.forward(x.cuda()).cpu()
    for b in range(2):
        one = model.forward(x[b:b + 1].cuda()).cpu()
        assert rel_l2(both[b], one[0]) < 1e-5


def test_infer_video_depth_windows_and_stitch():
    """Driver plumbing at small size: 40 frames -> 2 windows -> aligned/blended output of the right shape."""
    from vdn import synth
    model = _product("B", "vits")
    frames = synth.frames_u8(1234, 40, 140, 140)
    d, fps = model.infer_video_depth(frames, 24, input_size=140)
    assert d.shape == (40, 140, 140) and fps == 24 and np.isfinite(d).all() and (d >= 0).all()
    # the on-device stitcher against the host restatement applied to the same per-window outputs
    from vdn import util
    net = model.preprocess_frames(frames, 140)
    per_window = []
    for idxs in util.window_table(40):
        w = model.forward(net[torch.tensor(idxs, device=net.device)][None])[0].cpu().numpy()
        per_window += [w[i] for i in range(32)]
    host = util.stitch(per_window, 40)
    assert np.allclose(d, host, rtol=2e-5, atol=1e-6), float(np.abs(d - host).max())


def test_infer_video_depth_256_frames_vitl_518_full_size():
    """BASELINE configs[3] at FULL size on one GPU (video_depth.py:88-156): a 256-frame ViT-L 518x518 clip = 12 windows,
    384 window slots over 256 distinct frames (the padding repeats the last frame). Size-independent properties of the driver:
      (1) the clip-level tap cache (5.7 GB) and the three-run slot copies: windows 0, 5 and 11 from the cache equal a fresh
          `forward()` of the same 32 window slots (every slot encoded again);
      (2) the 11-push device stitcher chain equals the host restatement `util.stitch` applied to the 12 per-window outputs."""
    import vdn
    from vdn import synth, util
    model = vdn.VideoDepthAnything(**vdn.MODEL_CONFIGS["vitl"])
    sd = model.state_dict()
    sd.update(synth.fast_state_dict([(k, tuple(v.shape)) for k, v in model.named_parameters()], 1234))
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda").eval()
    n = 256
    frames = synth.frames_u8(1234, n, 518, 518)
    d, fps = model.infer_video_depth(frames, 24, input_size=518)
    assert d.shape == (n, 518, 518) and fps == 24 and np.isfinite(d).all() and (d >= 0).all()
    d = d.copy()   # the driver's result aliases a reused pinned buffer
    table = util.window_table(n)
    assert len(table) == 12 and sum(len(w) for w in table) == 384 and len({f for w in table for f in w}) == 256   # 384 window slots, every distinct frame encoded once
    net = model.preprocess_frames(frames, 518)
    per_window = []
    for w, dw in enumerate(model.window_depths(net, table)):
        dw = dw.clone()   # the generator yields a view of the head's output buffer, which forward() below reuses
        if w in (0, 5, 11):
            fresh = model.forward(net[torch.tensor(table[w], device=net.device)][None])[0]
            e = rel_l2(dw, fresh)
            worst = float((dw - fresh).abs().max() / fresh.abs().max())
            print(f"[clip256] window {w}: tap cache vs fresh forward rel-L2 {e:.2e}, worst pixel {worst:.2e} of max")
            assert e < 1e-5 and worst < 1e-4, (w, e, worst)
        hw = dw.cpu().numpy()
        per_window += [hw[i] for i in range(32)]
    host = util.stitch(per_window, n)
    err = float(np.abs(d - host).max() / np.abs(host).max())
    print(f"[clip256] device stitcher chain (11 pushes) vs host util.stitch: worst pixel {err:.2e} of max")
    assert np.allclose(d, host, rtol=5e-5, atol=1e-5 * float(np.abs(host).max())), err


def test_clip_result_through_pinned_memory_is_never_overwritten_while_held():
    """vdn.util.to_host: the drivers' device-to-host copy reuses one pinned buffer, but only after the caller dropped the
    previous result."""
    from vdn import util
    a = torch.arange(6, dtype=torch.float32, device="cuda").reshape(2, 3)
    r1 = util.to_host(a)
    r2 = util.to_host(a + 10)          # r1 is still alive: a second buffer
    assert r1.tolist() == [[0, 1, 2], [3, 4, 5]] and r2.tolist() == [[10, 11, 12], [13, 14, 15]]
    p2 = r2.ctypes.data
    del r2
    r3 = util.to_host(a + 20)          # the dropped result's buffer is reused
    assert r3.ctypes.data == p2 and r3.tolist() == [[20, 21, 22], [23, 24, 25]] and r1.tolist() == [[0, 1, 2], [3, 4, 5]]
    assert util.to_host(torch.ones(2)).tolist() == [1, 1]   # CPU tensors pass through


def test_image2tensor_and_infer_image_against_oracle():
    """a1 + a12 (depth_anything_v2.py:57-92): BGR u8 image -> cubic resize to the 14-multiple lower bound -> normalise ->
    forward -> bilinear back to the image size, against the oracle's restatement of the same pipeline (its cubic resize
    restates cv2's published algorithm; cross-checked against torch's bicubic in tests/test_host.py)."""
    from oracle import ref_cpu as O
    from vdn import synth
    model = _product("A", "vits")
    img = np.ascontiguousarray(synth.frames_u8(1234, 1, 240, 240)[0][:, :, ::-1])
    x, (h, w) = model.image2tensor(img, input_size=266)
    xr, (hr, wr) = O.image2tensor(img, 266)
    assert (h, w) == (hr, wr) == (240, 240) and tuple(x.shape) == tuple(xr.shape) == (1, 3, 266, 266)
    assert float((x.cpu() - xr).abs().max()) < 5e-5   # fp32 cubic on the device against the float64 restatement, values O(1)
    d = model.infer_image(img, input_size=266)
    with torch.no_grad():
        ref = O.infer_image(synth_sd("A", "vits"), img, O.MemoryState(6), "vits", 266)
    e = rel_l2(d, ref)
    print(f"[infer_image] 240x240 -> 266 -> 240: rel-L2 vs oracle {e:.2e}")
    assert d.shape == (240, 240) and np.isfinite(d).all() and e < TOL


@pytest.mark.parametrize("precision,limit", [("f16", 3e-3), ("bf16", 3e-2), ("bf16x3", 1e-3)])
def test_single_pass_modes_error_is_reported_and_bounded(precision, limit):
    """The other precision modes: the full-rate single-product ones are not fp32-faithful, bf16x3
    (split bf16 planes, ~16 significant bits, the round-1 attention kernel) is; their distance from the reference is
    printed and bounded so that a regression (a wrong kernel) cannot hide in it."""
    g = np.load(os.path.join(GOLD, "A_vitl_518.npz"))
    sub = int(g["meta"][4])
    model = _product("A", "vitl", precision)
    x = inputs(2, 518, 518).reshape(2, 1, 3, 518, 518)
    for t in range(2):
        pre = model.forward(x[t].cuda(), _pre_relu=True).cpu()
        e = rel_l2(torch.relu(pre[:, ::sub, ::sub]), np.maximum(g[f"pre_{t}"], 0))
        print(f"[A_vitl_518 {precision}] frame {t}: vs reference fixture post-ReLU {e:.2e}")
        assert e < limit


@pytest.mark.parametrize("pv,limit", [(1, 2e-4), (2, 4e-5), (3, 3e-5)])   # 3: fp16 lo planes of Q / K must be written (runtime.qk_dst)
@pytest.mark.parametrize("name,enc", [("A_vitl_518", "vitl"), ("A_vits_518", "vits")])
def test_pv_product_modes_error_is_reported_and_bounded(name, enc, pv, limit):
    """pv_products of vdn_flash_attn (per call; model.set_attention_pv): 1 (the default) = V enters the attention as ONE fp16
    plane rounded to nearest (P~ V_hi: 20 instead of 28 MFMAs per 64-key tile), 2 = P~ (V_hi + V_lo), 3 = P split into planes
    too (reads the fp16 lo planes of Q / K, which the projections then write). The whole 8-frame stream — memory depth 0..6
    and the eviction — against the reference fixture: measured 2e-5..9e-5 (1) and 7e-6..2e-5 (2) against the 1e-3 bar."""
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    _, steps, H, W, sub, _ = [int(v) for v in g["meta"]]
    kept = sorted(int(k.split("_")[1]) for k in g.files if k.startswith("pre_") and not k.startswith("pre_stats"))
    model = _product("A", enc)
    model.set_attention_pv(pv)
    assert model._engines()["rt"].pv_products == pv
    x = inputs(steps, H, W).reshape(steps, 1, 3, H, W)
    for t in range(max(kept) + 1):
        pre = model.forward(x[t].cuda(), _pre_relu=True).cpu()
        if t in kept:
            e = rel_l2(torch.relu(pre[:, ::sub, ::sub]), np.maximum(g[f"pre_{t}"], 0))
            print(f"[{name} pv={pv}] frame {t}: vs reference fixture post-ReLU {e:.2e}")
            assert e < limit, (t, e)


def test_frame_sharded_forward_matches_plain_on_one_rank():
    """vdn/dist.py path with a world of 1 (the exchange is the identity): same kernels, same result as
    forward(); the 2-rank exchange itself is covered on CPU/gloo in tests/test_dist.py."""
    model = _product("B", "vits")
    x = inputs(4, 266, 266).reshape(1, 4, 3, 266, 266)
    a = model.forward(x.cuda()).cpu()
    b = model.forward_sharded(x.cuda()).cpu()
    assert rel_l2(b, a) < 1e-6


def test_window_sharded_driver_single_rank():
    from vdn import synth
    from vdn.dist import infer_video_depth_sharded
    model = _product("B", "vits")
    frames = synth.frames_u8(1234, 40, 140, 140)
    d0, _ = model.infer_video_depth(frames, 24, input_size=140)
    d1, _ = infer_video_depth_sharded(model, frames, 24, input_size=140)
    assert np.allclose(d0, d1, rtol=1e-5, atol=1e-6)


def test_streaming_mode_against_reference_fixture():
    """infer_video_depth_one semantics: 14 frames, window slides after frame 10 (video_depth_stream.py:155-158)."""
    from oracle import ref_cpu as O
    g = np.load(os.path.join(GOLD, "S_vits_266.npz"))
    _, n, H, W, _, _ = [int(v) for v in g["meta"]]
    model = _product("B", "vits")
    model.reset_stream()
    x = inputs(n, H, W)
    st = O.StreamState()
    sd = synth_sd("B", "vits")
    for t in range(n):
        pre = model.stream_step(x[t][None, None].cuda(), _pre_relu=True).cpu()
        if f"pre_{t}" in g.files:
            e = rel_l2(torch.relu(pre), np.maximum(g[f"pre_{t}"], 0))
            print(f"[S_vits_266] frame {t}: vs reference fixture post-ReLU {e:.2e}")
            assert e < TOL
        if t < 3:
            with torch.no_grad():
                ref = O.video_depth_stream_step(sd, x[t][None, None], st, "vits", pre_relu=True)
            assert rel_l2(torch.relu(pre), torch.relu(ref)) < TOL
    assert len(model._stream["cache"]) == 42


@pytest.mark.parametrize("version,name", [(5, "R5_vits"), (4, "R4_vits"), (5, "R5f_vits")])
def test_depth_refiner_v4_v5_against_reference_fixture(version, name):
    """the v4 / v5 wrappers (median radix select, scale, Sobel normals, temporal network, shift +
    residual) against the fixture written by the imported reference model; tolerance 1e-3 on the refined depth."""
    import importlib
    import vdn
    from vdn import synth
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    v, S, H, W, seed = [int(t) for t in g["meta"]]
    cls = importlib.import_module(f"vdn.video_depth_model_v{version}").VideoDepthAnything
    flagged = name.startswith(f"R{version}f")   # built with use_bn=True, use_clstoken=True
    m = cls(**dict(vdn.MODEL_CONFIGS["vits"], **(dict(use_bn=True, use_clstoken=True) if flagged else {})))
    m.load_state_dict(synth_sd(f"R{version}f" if flagged else f"R{version}", "vits"), strict=True)
    m = m.to("cuda").eval()
    x = torch.from_numpy(synth.depth_clip(seed, S, H, W))[None]
    out = m.forward(x.cuda())[0].cpu()
    e = rel_l2(out, g["out"])
    print(f"[{name}] refined depth vs reference fixture {e:.2e}")
    assert torch.isfinite(out).all() and e < TOL
    # two clips in one batch are independent
    both = m.forward(torch.cat([x, x * 0.5 + 100.0]).cuda()).cpu()
    assert rel_l2(both[0], g["out"]) < TOL


def test_depth_refiner_v5_vitl_64_frames_1024():
    """BASELINE configs[4] as the reference implements it: video_depth_model_v5 (ViT-L, num_frames = 64) on a
    [1, 64, 1024, 1024] raw depth clip (the network itself runs at 224 x 224), against the imported reference's
    fixture: a 16-strided sample of every refined frame plus per-frame statistics of the full maps."""
    import vdn
    from vdn import synth
    from vdn.video_depth_model_v5 import VideoDepthAnything
    g = np.load(os.path.join(GOLD, "R5_vitl_T64.npz"))
    v, S, H, W, seed = [int(t) for t in g["meta"]]
    sub, nf = int(g["sub"]), int(g["num_frames"])
    m = VideoDepthAnything(num_frames=nf, **vdn.MODEL_CONFIGS["vitl"])
    m.load_state_dict(synth_sd("R5", "vitl"), strict=True)
    m = m.to("cuda").eval()
    x = torch.from_numpy(synth.depth_clip(seed, S, H, W))[None]
    out = m.forward(x.cuda())[0].cpu()
    assert torch.isfinite(out).all()
    e = rel_l2(out[:, ::sub, ::sub], g["out"])
    per_frame = max(rel_l2(out[t, ::sub, ::sub], g["out"][t]) for t in range(S))
    means = np.array([out[t].mean().item() for t in range(S)])
    print(f"[R5_vitl_T64] refined depth vs reference fixture {e:.2e} (worst frame {per_frame:.2e})")
    assert e < TOL and per_frame < TOL
    assert np.allclose(means, g["out_stats"][:, 0], rtol=1e-3)


@pytest.mark.parametrize("use_residual,input_normal", [(False, True), (True, False)])
def test_depth_refiner_flags_against_oracle(use_residual, input_normal):
    """The v5 wrapper's constructor switches (video_depth_model_v5.py:135-136,172-189): no shift/residual,
    depth broadcast to 3 channels instead of Sobel normals — against the oracle on a 2-frame clip."""
    import vdn
    from oracle import ref_cpu as O
    from vdn import synth
    from vdn.video_depth_model_v5 import VideoDepthAnything
    sd = synth_sd("R5", "vits")
    m = VideoDepthAnything(use_residual=use_residual, input_normal=input_normal, **vdn.MODEL_CONFIGS["vits"])
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").eval()
    x = torch.from_numpy(synth.depth_clip(77, 2, 70, 95))[None]
    with torch.no_grad():
        ref = O.depth_refiner_forward(sd, x, "vits", version=5, use_residual=use_residual, input_normal=input_normal)
    out = m.forward(x.cuda()).cpu()
    e = rel_l2(out, ref)
    print(f"[refiner v5 residual={use_residual} normals={input_normal}] vs oracle {e:.2e}")
    assert e < TOL
