#!/usr/bin/env python3
"""Per-launch HBM traffic of the dominant GEMM from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
gfx950: FETCH_SIZE under-reports wide coalesced reads by exactly 2x (MI355X_MICROARCH.md §HBM), so the
read side is doubled; both counters are in KiB. Writes profiles/r01_pmc_traffic.json + a .md summary."""
import csv, glob, json, re, sys, collections
fetch_csv = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
write_csv = glob.glob(sys.argv[2] + "/*/*_counter_collection.csv")[0]
prec = sys.argv[3]
out_json, out_md = sys.argv[4], sys.argv[5]

def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::|vdn_gemm_impl::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        acc[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc

F, W = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
rows = []
for key in sorted(F, key=lambda k: -sum(F[k])):
    f = F[key]; w = W.get(key, [0.0])
    rows.append((key[0], key[1], len(f), 2 * 1024 * sum(f) / len(f), 1024 * sum(w) / len(w)))
# encoder-linear launches (single lane, batch 8: M = 8*1370): the plain-A 8-wave kernels at 232 (proj, fc2),
# 688 (fc1, BM 256) and 696 (qkv) workgroups of 512 threads
enc = [r for r in rows if (r[0].startswith("gemm_x3_big_kernel<0, 0,") or r[0].startswith("gemm_x3_p8_kernel<0, 0,") or r[0].startswith("gemm_x8_kernel<"))
       and r[1] in (232 * 512, 688 * 512, 696 * 512, 928 * 512, 516 * 512, 172 * 512)]
n = sum(r[2] for r in enc)
rd = sum(r[3] * r[2] for r in enc) / max(n, 1)
wr = sum(r[4] * r[2] for r in enc) / max(n, 1)
C, M = 1024, 8 * 1370
planes = 2 if prec.endswith("x3") else 1
alg = planes * 2 * (M * 12 * C * 2 + 12 * C * C) / 4  # mean over qkv/proj/fc1/fc2: read A + write C (+W once), 16-bit planes
res = {prec: {"enc_linear_bytes_per_launch": round(rd + wr), "read_bytes": round(rd), "write_bytes": round(wr),
              "launches": n, "algorithmic_bytes_per_launch": round(alg),
              "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB->bytes, mean over encoder/memory linear launches at M=8*1370"}}
try:
    old = json.load(open(out_json))
except Exception:
    old = {}
old.update(res)
json.dump(old, open(out_json, "w"), indent=1)
with open(out_md, "w") as f:
    f.write(f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 7, precision {prec}\n\n")
    f.write("read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB; mean bytes per launch\n\n")
    f.write("| kernel | grid threads | launches | read MB | write MB |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:24]:
        f.write(f"| `{r[0][:70]}` | {r[1]} | {r[2]} | {r[3]/1e6:.1f} | {r[4]/1e6:.1f} |\n")
    f.write(f"\nencoder-linear GEMM launches: {n}, mean read {rd/1e6:.1f} MB + write {wr/1e6:.1f} MB = {(rd+wr)/1e6:.1f} MB per launch; "
            f"algorithmic (A read + C write + W once, {planes} planes) {alg/1e6:.1f} MB\n")
print(json.dumps(res))
