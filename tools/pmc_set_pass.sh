#!/bin/bash
# HBM traffic of the SET step's launches in situ (the captured graph replayed by tools/set_probe.py): separate rocprofv3 --pmc passes for
# FETCH_SIZE and WRITE_SIZE (--kernel-trace only), averaged per kernel and matrix.  On gfx950 FETCH_SIZE reports half the bytes of a wide
# streaming read (MI355X_MICROARCH.md): traffic = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB.
# usage (GPU box, repo root): bash tools/pmc_set_pass.sh [seqs=4] > gpurun_out/<tag>_set_pmc.txt
S=${1:-4}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sp1 /tmp/sp2
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/sp1 -o p -- python $R/tools/set_probe.py --seqs $S --steps 12 > /tmp/sp1.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/sp2 -o p -- python $R/tools/set_probe.py --seqs $S --steps 12 > /tmp/sp2.log 2>&1
cd $R
python - "$S" $(find /tmp/sp1 -name "*counter_collection.csv") $(find /tmp/sp2 -name "*counter_collection.csv") <<'PY'
import collections, csv, sys
B = int(sys.argv[1])
def agg(path, name):
    a = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    seen = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"]; k = k[:k.index("(")] if "(" in k else k
        k = k.replace("void ", "")
        if k.startswith("lh::k_gemv_set") and k.endswith(", 1>"):        # wo and w2 share the kernel: they alternate in dispatch order
            seen[k] += 1
            k += " wo" if seen[k] % 2 == 1 else " w2"
        a[k].append(float(r["Counter_Value"]))
    return a
f, w = agg(sys.argv[2], "FETCH_SIZE"), agg(sys.argv[3], "WRITE_SIZE")
d, F, V = 4096, 11008, 32000
gb = lambda M, K: M * (K // 32) * 20 + B * (K // 32) * 20 + 4 * M * B          # SURVEY 8d with B activation rows
def algo(k):
    if not k.startswith("lh::k_gemv_set"): return None
    if k.endswith(" wo"): return gb(d, d)
    if k.endswith(" w2"): return gb(d, F)
    if k.endswith(", 3>"): return gb(3 * d, d)
    if k.endswith(", 7>"): return gb(2 * F, d)
    if k.endswith(", 0>"): return gb(V, d)
    return None
print(f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over tools/set_probe.py --seqs {B} --steps 12: the captured 7B set step ({B} sequences)")
print("# traffic = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB per launch (gfx950 FETCH_SIZE correction); algorithmic = Q4_0 weights + B operand rows + B fp32 output rows (SURVEY 8d)")
print(f"{'kernel':52s} {'launches':>8s} {'traffic_MB':>11s} {'algo_MB':>9s} {'ratio':>6s}")
for k in sorted(f, key=lambda k: -sum(f[k])):
    if len(f[k]) < 8 or not k.startswith("lh::k_"):
        continue
    tr = (2 * sum(f[k]) / len(f[k]) + (sum(w[k]) / len(w[k]) if k in w else 0.0)) * 1024
    al = algo(k)
    print(f"{k:52s} {len(f[k]):8d} {tr / 1e6:11.2f} {(al / 1e6 if al else float('nan')):9.2f} {(tr / al if al else float('nan')):6.3f}")
PY
