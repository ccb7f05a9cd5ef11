#!/bin/bash
# HBM traffic of the decode step's launches IN SITU (the captured graph replayed by tools/decode_probe.py): separate rocprofv3 --pmc
# passes for FETCH_SIZE and WRITE_SIZE (--kernel-trace only), averaged per kernel name.  On gfx950 FETCH_SIZE reports half the bytes of
# a wide streaming read (MI355X_MICROARCH.md): traffic = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB.
# usage (GPU box, repo root): bash tools/pmc_decode_pass.sh [context=128] > gpurun_out/<tag>_decode_pmc.txt
AT=${1:-128}
R=$PWD
python tools/decode_probe.py --steps 4 --at 8 --reps 1 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dp1 /tmp/dp2
LLAMAHIP_WITH_TORCH=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/dp1 -o p -- python $R/tools/decode_probe.py --steps 16 --at $AT --reps 1 > /tmp/dp1.log 2>&1
LLAMAHIP_WITH_TORCH=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/dp2 -o p -- python $R/tools/decode_probe.py --steps 16 --at $AT --reps 1 > /tmp/dp2.log 2>&1
cd $R
python - "$AT" $(find /tmp/dp1 -name "*counter_collection.csv") $(find /tmp/dp2 -name "*counter_collection.csv") <<'PY'
import collections, csv, sys
at = int(sys.argv[1])
def agg(path, name):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"]; k = k[:k.index("(")] if "(" in k else k
            a[k.replace("void ", "")].append(float(r["Counter_Value"]))
    return a
f, w = agg(sys.argv[2], "FETCH_SIZE"), agg(sys.argv[3], "WRITE_SIZE")
d, F, V, T = 4096, 11008, 32000, at + 8
gb = lambda M, K: M * (K // 32) * 20 + (K // 32) * 20 + 4 * M
algo = {"lh::k_qkv_attn": gb(3 * d, d) + 2 * T * d * 4, "lh::k_gemv<0, 1, 16": gb(d, d), "lh::k_gemv<4, 2, 4": gb(2 * F, d), "lh::k_gemv<0, 1, 10": gb(d, F), "lh::k_gemv<4, 0, 4": gb(V, d)}
print(f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over tools/decode_probe.py --steps 16 --at {at}: the captured 7B decode step")
print(f"# traffic = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB per launch (gfx950 FETCH_SIZE correction); algorithmic = weights (+ K and V rows at context ~{T} for the fused launch)")
print(f"{'kernel':44s} {'launches':>8s} {'traffic_MB':>11s} {'algo_MB':>9s} {'ratio':>6s}")
for k in sorted(f, key=lambda k: -sum(f[k])):
    if len(f[k]) < 16 or not k.startswith("lh::k_"):
        continue
    tr = (2 * sum(f[k]) / len(f[k]) + (sum(w[k]) / len(w[k]) if k in w else 0.0)) * 1024
    al = next((v for p, v in algo.items() if k.startswith(p)), None)
    print(f"{k:44s} {len(f[k]):8d} {tr / 1e6:11.2f} {(al / 1e6 if al else float('nan')):9.2f} {(tr / al if al else float('nan')):6.3f}")
PY
