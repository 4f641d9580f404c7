#!/usr/bin/env python3
"""Per-layer precision budget (GPU box): end-to-end error of DepthAnythingV2(vitl), batch 8 (the 8-bit cross-term kernel's
path), against the reference fixtures when one cross term of one encoder linear is dropped (VDN_X8_TERMS, include/vdn.h
x8_terms). Element 0 of the batch carries the fixture's stream. Writes a markdown table to stdout."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(ROOT, "video-depth-normal-v2_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch, vdn
    from common import GOLD, inputs, rel_l2, synth_sd, worst_px
    res = {}
    for name, which, steps in (("A_vitl_518", "A", 3), ("A_vitl_518_heavy", "Ah", 3)):
        g = np.load(os.path.join(GOLD, f"{name}.npz"))
        _, _, H, W, sub, _ = [int(v) for v in g["meta"]]
        m = vdn.DepthAnythingV2(**vdn.MODEL_CONFIGS["vitl"]); m.load_state_dict(synth_sd(which, "vitl"), strict=True); m = m.to("cuda").eval()
        B = 8
        pool = inputs(steps + B - 1, H, W)
        e_max = w_max = 0.0
        for t in range(steps):
            xb = torch.stack([pool[t + b] for b in range(B)])
            pre = m.forward(xb.cuda(), _pre_relu=True).cpu()
            if f"pre_{t}" in g.files:
                e_max = max(e_max, rel_l2(torch.relu(pre[0, ::sub, ::sub]), np.maximum(g[f"pre_{t}"][0], 0)))
                w_max = max(w_max, worst_px(pre[0, ::sub, ::sub], g[f"pre_{t}"][0]))
        res[name] = (e_max, w_max)
        del m
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(res))
    sys.exit(0)

configs = [("both terms (shipped)", "")]
for lin in ("qkv", "proj", "fc1", "fc2"):
    for term, label in ((1, "no A_lo W_hi"), (2, "no A_hi W_lo")):
        configs.append((f"{lin}: {label}", f"{lin}={term}"))
configs += [("proj + fc2: no A_lo W_hi", "proj=1,fc2=1"), ("qkv + fc1: no A_lo W_hi", "qkv=1,fc1=1"), ("all four: no A_lo W_hi", "qkv=1,proj=1,fc1=1,fc2=1"),
            ("all four: no A_hi W_lo", "qkv=2,proj=2,fc1=2,fc2=2"), ("proj + fc2, blocks 12-23: no A_lo W_hi", "proj=1@12-23,fc2=1@12-23")]
print("| configuration (VDN_X8_TERMS) | A_vitl_518 rel-L2 | worst px | A_vitl_518_heavy rel-L2 | worst px |\n|---|---:|---:|---:|---:|", flush=True)
for label, spec in configs:
    env = dict(os.environ, VDN_X8_TERMS=spec)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print(f"| {label} (`{spec}`) | failed: {p.stderr[-200:]} |", flush=True)
        continue
    r = json.loads(line[0][7:])
    a, h = r["A_vitl_518"], r["A_vitl_518_heavy"]
    print(f"| {label} (`{spec or '-'}`) | {a[0]:.2e} | {a[1]:.2e} | {h[0]:.2e} | {h[1]:.2e} |", flush=True)
