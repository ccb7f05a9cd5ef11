"""Measurement / fall-back variants of the library that the GPU suite re-runs its parity tests under (switches are read once per
process, so each variant is a nested pytest run), the ONE list both the GPU tests and the host-only plan walk use, and the helper that
runs a nested selection and reports it in one line.

`python tests/variants.py walk` walks the few-row kernel's plan for every LLaMA shape and row count under the CURRENT environment
(host only, no device): tests/test_host.py runs it once per entry of SET_PLAN_VARIANTS, so a forced plan that would send a shape to a
generic kernel is caught in the GPU-less container (a forced LLAMAHIP_SET_PLAN_SMALL=2,4 did exactly that on 13B / 65B w2)."""
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

# (environment, pytest id) of test_few_row_kernel_selectable_epilogues_and_plans
SET_PLAN_VARIANTS = [
    ({"LLAMAHIP_SET_W13_BLOCKS": "0"}, "w13_half_blocks_where_they_apply"),
    ({"LLAMAHIP_SET_W13_BLOCKS": "1"}, "w13_whole_blocks_at_every_row_count"),
    ({"LLAMAHIP_SET_PLAN": "3,1"}, "unshared_groups_of_3_everywhere"),
    ({"LLAMAHIP_SET_PLAN_SMALL": "2,4"}, "shared_rings_on_the_small_matrices"),
]

EPI_STORE, EPI_RESID, EPI_SILU_QA, EPI_ROPE_KV, EPI_SILU_QAH = 0, 1, 2, 3, 7
SET_PAIRS = {(1, 2), (1, 3), (1, 4), (2, 1), (2, 2), (2, 3), (2, 4), (3, 1), (3, 3), (3, 4), (4, 1), (4, 2), (4, 4), (5, 1)}


def walk_set_plans(L):
    """every matrix of the four LLaMA sizes x every row count a short eval or a batched decode step can have (2 .. 60): the few-row
    kernel takes the shape with an instantiated (columns per wave, column-waves) pair, column groups that cover the rows exactly (no
    empty group), operand rows + weight ring inside the CU's 160 KB of LDS.  Returns the number of plans walked."""
    n = 0
    for d, mult in ((4096, 256), (5120, 256), (6656, 256), (8192, 256)):
        F = ((2 * (4 * d) // 3 + mult - 1) // mult) * mult                  # .mm:118-120
        mats = [("wq|wk|wv", 3 * d, d, EPI_ROPE_KV, False), ("wo", d, d, EPI_RESID, False), ("w1|w3 halves", 2 * F, d, EPI_SILU_QAH, True),
                ("w1|w3 blocks", 2 * F, d, EPI_SILU_QA, True), ("w2", d, F, EPI_RESID, False), ("output", 32000, d, EPI_STORE, False)]
        for name, M, K, epi, inter in mats:
            for N in range(2, 61):
                p = L.set_plan(M, K, N, epi, inter)
                assert p is not None, (d, name, N, "not taken: would fall to a generic kernel")
                nc, cw, ncg, rgw, lds = p
                assert (nc, cw) in SET_PAIRS and lds <= 160 * 1024 and 1 <= rgw * cw <= 16, (d, name, N, p)
                assert nc * cw * ncg >= N > nc * cw * (ncg - 1), (d, name, N, p)
                if epi in (EPI_SILU_QA, EPI_SILU_QAH):
                    assert cw == 1 and rgw == (8 if epi == EPI_SILU_QA else 4), (d, name, N, p)
                n += 1
    # odd shapes the LLaMA sizes never produce: a plan is either refused or has no empty column group
    for M, K in ((64, 256), (8, 64), (4096, 73728)):
        for N in range(2, 61):
            for epi in (EPI_STORE, EPI_RESID):
                p = L.set_plan(M, K, N, epi, False)
                if p is not None:
                    nc, cw, ncg, rgw, lds = p
                    assert (nc, cw) in SET_PAIRS and nc * cw * ncg >= N > nc * cw * (ncg - 1) and lds <= 160 * 1024, (M, K, N, epi, p)
    return n


def run_nested(env, select, files, tag, timeout=1500):
    """One nested `pytest -x -q -m gpu -k <select>` under `env`; prints ONE summary line (variant=… passed=N failed=M seconds=… and the
    first failure if any) and returns (ok, summary, tail)."""
    t0 = time.time()
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *files, "-k", select]
    r = subprocess.run(cmd, env=dict(os.environ, LLAMAHIP_NESTED="1", **env), capture_output=True, text=True, cwd=ROOT, timeout=timeout)
    out = r.stdout
    npass = sum(int(x) for x in re.findall(r"(\d+) passed", out.splitlines()[-1] if out.strip() else ""))
    nfail = sum(int(x) for x in re.findall(r"(\d+) (?:failed|error)", out.splitlines()[-1] if out.strip() else ""))
    first = next((ln for ln in out.splitlines() if ln.startswith(("FAILED", "ERROR"))), "")
    summary = f"variant={tag} env={env} passed={npass} failed={nfail} rc={r.returncode} seconds={time.time() - t0:.0f}" + (f" first_failure={first}" if first else "")
    print(summary)
    return r.returncode == 0 and npass > 0, summary, out[-2500:] + r.stderr[-1500:]


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    if sys.argv[1:] == ["walk"]:
        import llama_swift_amd as L
        print("plans walked:", walk_set_plans(L))
