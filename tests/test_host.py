"""Host logic of the product package + the C-ABI surface (no GPU compute)."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from common import GOLD, ROOT, schema


def test_get_size_table_matches_reference(golden_dir):
    from vdn import util
    g = np.load(os.path.join(golden_dir, "host.npz"))
    for w, h, nw, nh in g["get_size"]:
        assert util.get_size(int(w), int(h)) == (int(nw), int(nh))


def test_window_table_and_stitch_match_reference(golden_dir):
    from vdn import synth, util
    g = np.load(os.path.join(golden_dir, "host.npz"))
    assert np.array_equal(np.array(util.window_table(50)), g["windows_50"])
    s = np.load(os.path.join(golden_dir, "stitch.npz"))
    n, h, w, seed = [int(v) for v in s["meta"]]
    x = synth.normalize_frames(synth.frames_u8(seed, n, h, w))
    dl = []
    for wi, idxs in enumerate(util.window_table(n)):
        dep = np.abs(x[idxs].mean(1) * (1.0 + 0.25 * wi) + 0.1 * (wi + 1))
        dl += [dep[i] for i in range(32)]
    out = util.stitch(dl, n)
    assert np.abs(out - s["out"]).max() / np.abs(s["out"]).max() < 1e-5


@pytest.mark.parametrize("n", [1, 22, 23, 32, 33, 256])
def test_window_table_structure(n):
    """slot 0 is always frame 0; window count = ceil(n/22); 256 frames -> 12 windows."""
    from vdn import util
    t = util.window_table(n)
    assert len(t) == -(-n // 22)
    assert all(len(w) == 32 for w in t)
    assert all(w[0] == 0 for w in t)
    assert max(max(w) for w in t) == n - 1
    if n == 256:
        assert len(t) == 12


def test_scale_shift_and_blend(golden_dir):
    from vdn import synth, util
    g = np.load(os.path.join(golden_dir, "host.npz"))
    pred = synth.normal(1234, "ss_pred", (2, 40, 50)) * 3 + 5
    targ = 1.7 * pred - 0.3 + 0.1 * synth.normal(1234, "ss_noise", (2, 40, 50))
    s, sh = util.compute_scale_and_shift(np.concatenate(list(pred)), np.concatenate(list(targ)),
                                         np.concatenate(np.ones_like(targ) == 1))
    assert np.allclose([s, sh], g["scale_shift"], rtol=1e-6)
    pre = [synth.normal(1234, f"ip{i}", (6, 7)) for i in range(8)]
    post = [synth.normal(1234, f"iq{i}", (6, 7)) for i in range(8)]
    assert np.array_equal(np.stack(util.get_interpolate_frames(pre, post)), g["blend"])
    # degenerate system -> identity (utils/util.py:52-60)
    z = np.zeros((4, 4), np.float32)
    assert util.compute_scale_and_shift(z, z, np.zeros_like(z)) == (1, 0)


def test_library_exports_every_declared_symbol():
    """include/vdn.h <-> libvdn_hip.so <-> vdn/_abi.py agree (load + symbols only, no launches)."""
    from vdn import _abi
    hdr = open(os.path.join(ROOT, "include", "vdn.h")).read()
    declared = set(re.findall(r"\b(vdn_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vdn_stream"}
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(_abi.lib, name), f"{name} declared in vdn.h but not exported"
    assert declared == set(_abi.EXPORTS), (declared ^ set(_abi.EXPORTS))
    assert _abi.lib.vdn_sizeof_gemm_desc() == ctypes.sizeof(_abi.GemmDesc)
    assert _abi.lib.vdn_offsetof_gemm_zeros() == _abi.GemmDesc.zeros.offset
    assert b"gfx950" in _abi.lib.vdn_version()


def test_gemm_argument_validation_without_gpu():
    """vdn_gemm rejects malformed descriptors before touching the device."""
    from vdn import _abi
    d = _abi.GemmDesc()
    assert _abi.lib.vdn_gemm(ctypes.byref(d), None) == -1            # empty
    d.M, d.N, d.K, d.A, d.W, d.zeros = 8, 8, 12, 16, 16, 16          # K % 8 != 0
    d.ldb = 64
    assert _abi.lib.vdn_gemm(ctypes.byref(d), None) == -3
    d.K, d.lda, d.dt = 16, 16, 2                                      # f32 operands unsupported
    assert _abi.lib.vdn_gemm(ctypes.byref(d), None) == -2


@pytest.mark.parametrize("which,enc", [("A", "vits"), ("A", "vitb"), ("A", "vitl"), ("A", "vitg"), ("B", "vits"), ("B", "vitl"), ("Af", "vits"),
                                       ("Bf", "vits")])
def test_state_dict_schema_matches_reference(which, enc):
    """Drop-in contract: same parameter/buffer keys and shapes as the reference classes; "Af" / "Bf" = the
    same classes built with use_bn=True, use_clstoken=True (BatchNorm parameters and buffers, readout_projects)."""
    import vdn
    cls = vdn.DepthAnythingV2 if which[0] == "A" else vdn.VideoDepthAnything
    m = cls(**dict(vdn.MODEL_CONFIGS[enc], **(dict(use_bn=True, use_clstoken=True) if which.endswith("f") else {})))
    sch = schema(which, enc)
    assert {k: tuple(v.shape) for k, v in m.named_parameters()} == {k: tuple(s) for k, s in sch["params"]}
    assert {k: tuple(v.shape) for k, v in m.named_buffers()} == {k: tuple(s) for k, s in sch["buffers"]}
    sd = m.state_dict()
    m.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("version", [4, 5])
def test_refiner_state_dict_schema_matches_reference(version):
    """v4 / v5 depth-refiner wrappers: same keys and shapes as models/video_depth_model_v{4,5}.VideoDepthAnything."""
    import importlib
    import vdn
    cls = importlib.import_module(f"vdn.video_depth_model_v{version}").VideoDepthAnything
    m = cls(**vdn.MODEL_CONFIGS["vits"])
    sch = schema(f"R{version}", "vits")
    assert {k: tuple(v.shape) for k, v in m.named_parameters()} == {k: tuple(s) for k, s in sch["params"]}
    assert {k: tuple(v.shape) for k, v in m.named_buffers()} == {k: tuple(s) for k, s in sch["buffers"]}
    m.load_state_dict(m.state_dict(), strict=True)


def test_reference_import_paths_exist():
    """A reference user's imports, with the package name swapped: depth_anything_v2.depth_anything_v2, video_depth_anything.
    video_depth, video_depth_anything.video_depth_stream, models.video_depth_model_v4 / _v5."""
    import importlib
    for mod, cls, methods in (("vdn.depth_anything_v2", "DepthAnythingV2", ["forward", "infer_image", "image2tensor", "clear_memory"]),
                              ("vdn.video_depth", "VideoDepthAnything", ["forward", "infer_video_depth"]),
                              ("vdn.video_depth_stream", "VideoDepthAnything", ["forward", "infer_video_depth_one"]),
                              ("vdn.video_depth_model_v4", "VideoDepthAnything", ["forward"]),
                              ("vdn.video_depth_model_v5", "VideoDepthAnything", ["forward"])):
        c = getattr(importlib.import_module(mod), cls)
        assert all(callable(getattr(c, m)) for m in methods), (mod, methods)


def test_product_refuses_cpu():
    import vdn
    m = vdn.DepthAnythingV2(**vdn.MODEL_CONFIGS["vits"])
    with pytest.raises(Exception):
        m.forward(torch.zeros(1, 3, 28, 28))


def test_rope_table_and_pack_geometry():
    """Host-visible pieces of the packing ABI (the packers themselves run on the device: tests/test_gpu_ops.py)."""
    from vdn import _abi, pack
    cs = pack.rope_table(3, 3, 64)
    assert cs.shape == (9, 32, 2) and torch.allclose(cs[0, :, 0], torch.ones(32))
    L = _abi.lib
    assert L.vdn_pack_ldb(_abi.PACK_LINEAR, 1024, 588, 0) == 640 and L.vdn_pack_rows(_abi.PACK_LINEAR, 1024, 588, 0) == 1024
    assert L.vdn_pack_ldb(_abi.PACK_CONV3X3, 256, 64, 0) == 576 and L.vdn_pack_ldb(_abi.PACK_CONV3X3_TAPS, 32, 32, 0) == 320
    assert L.vdn_pack_rows(_abi.PACK_CONVT, 256, 256, 4) == 4096 and L.vdn_pack_ldb(_abi.PACK_CONVT, 256, 256, 4) == 256
    assert L.vdn_pack_rows(_abi.PACK_GEGLU, 48, 64, 0) < 0 and L.vdn_pack_rows(_abi.PACK_ROPE, 100, 64, 0) < 0  # bad row counts
    assert L.vdn_pack_weight(_abi.F16, 99, None, 1, 1, 0, None, None, 64, None) == -1
    assert L.vdn_gemm_workspace_bytes(None) == 0 and L.vdn_groupnorm_workspace_bytes(4, 32, 16) == 4 * 16 * 32 * 2 * 4


def test_oracle_cubic_resize_matches_torch_bicubic():
    """oracle.resize_cubic restates cv2.INTER_CUBIC (absent here: parity unpinned against cv2 itself); torch's bicubic is an
    independent implementation of the same A = -0.75 half-pixel kernel with replicated borders: up- and down-scaling agree."""
    import torch
    import torch.nn.functional as F
    from oracle import ref_cpu as O
    img = np.random.default_rng(0).random((120, 100, 3))
    for nw, nh in [(140, 168), (266, 266), (56, 70)]:
        a = O.resize_cubic(img, nw, nh)
        b = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(nh, nw), mode="bicubic", align_corners=False)[0]
        assert a.shape == (nh, nw, 3) and np.abs(a - b.permute(1, 2, 0).numpy()).max() < 1e-12
    raw = (img * 255).astype(np.uint8)
    x, (h, w) = O.image2tensor(raw, 140)
    assert (h, w) == (120, 100) and tuple(x.shape) == (1, 3, 168, 140) and x.dtype == torch.float32
    same = np.zeros((28, 28, 3), np.uint8) + np.array([10, 20, 30], np.uint8)          # constant image: BGR -> RGB order, mean/std
    x, _ = O.image2tensor(same, 28)
    want = (np.array([30, 20, 10]) / 255.0 - np.array([0.485, 0.456, 0.406])) / np.array([0.229, 0.224, 0.225])
    assert np.allclose(x[0].mean((1, 2)).numpy(), want, atol=1e-6)


@pytest.mark.parametrize("inc,args,n_mfma", [("attn2_stream.inc", [], 28), ("attn2_stream_pv1.inc", ["--pv", "1"], 20),
                                             ("attn2_stream_stamps.inc", ["--stamps"], 28)])
def test_generated_attention_stream_is_current_and_well_formed(inc, args, n_mfma):
    """csrc/attn2_stream*.inc are GENERATED (tools/gen_attn_stream.py): the committed files equal the generator's output for
    the documented arguments, and the stream obeys the rules the kernel relies on: every MFMA's operand fragment is loaded
    into its ring buffer at least one MFMA EARLIER (or before the first one) and not overwritten before the MFMA issues, every softmax piece appears exactly
    once with F_p before X_p before C_p, the row maximum / rescale test precede the first F, each LDS-DMA piece once."""
    import re
    import subprocess
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_attn_stream.py")] + args,
                         capture_output=True, text=True, check=True).stdout
    with open(os.path.join(ROOT, "video-depth-normal-v2_amd", "csrc", inc)) as f:
        assert f.read() == gen, f"{inc} is stale: regenerate with tools/gen_attn_stream.py"
    calls = re.findall(r"A2_(\w+)\(([^)]*)\)", gen)
    live4, live8, seen, gap, loaded_gap = {}, {}, [], -1, {}
    mfmas = 0
    for name, a in calls:
        arg = [int(x) for x in a.split(",")] if a.strip() else []
        if name in ("LDV", "LDK"):
            live4[arg[-1]] = (name, tuple(arg[:-1])); loaded_gap[("4", arg[-1])] = gap
        elif name == "LDK8":
            live8[arg[-1]] = tuple(arg[:-1]); loaded_gap[("8", arg[-1])] = gap
        elif name == "PV":
            c, db, buf = arg
            assert live4[buf][0] == "LDV" and live4[buf][1][:2] == (c, db) and (loaded_gap[("4", buf)] < gap or loaded_gap[("4", buf)] == -1)
            mfmas += 1; gap += 1
        elif name == "QK":
            kb, ks, buf, first = arg
            assert live4[buf] == ("LDK", (kb, ks)) and (loaded_gap[("4", buf)] < gap or loaded_gap[("4", buf)] == -1)
            mfmas += 1; gap += 1
        elif name == "QX":
            kb, lo, buf, first = arg
            assert live8[buf] == (kb, lo) and (loaded_gap[("8", buf)] < gap or loaded_gap[("8", buf)] == -1)
            mfmas += 1; gap += 1
        elif name in ("F", "X", "C", "MAX", "DMA"):
            seen.append((name, arg[0]))
        elif name in ("XH", "BP", "LR"):
            seen.append((name, -1))
    assert mfmas == n_mfma
    order = {k: i for i, k in enumerate(seen)}
    assert len(order) == len(seen)                                            # nothing twice
    for p_ in range(16):
        assert order[("F", p_)] < order[("X", p_)] < order[("C", p_)]
        assert order[("BP", -1)] < order[("F", p_)]
    assert all(order[("MAX", q)] < order[("XH", -1)] < order[("BP", -1)] for q in range(4)) and order[("C", 15)] < order[("LR", -1)]
    dma = sorted(a for n, a in seen if n == "DMA")
    assert dma == (list(range(8)) if "--pv" not in args else [0, 2, 4, 5, 6, 7])
