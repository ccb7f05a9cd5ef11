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

PARITY = os.path.join(HERE, "test_gpu_parity.py")
PIPELINE = os.path.join(HERE, "test_pipeline.py")

# (environment, pytest id) of test_few_row_kernel_selectable_epilogues_and_plans
SET_PLAN_VARIANTS = [
    ({"LLAMAHIP_SET_W13_BLOCKS": "0"}, "w13_half_blocks_where_they_apply"),
    ({"LLAMAHIP_SET_W13_BLOCKS": "1"}, "w13_whole_blocks_at_every_row_count"),
    ({"LLAMAHIP_SET_PLAN": "3,1"}, "unshared_groups_of_3_everywhere"),
    ({"LLAMAHIP_SET_PLAN_SMALL": "2,4"}, "shared_rings_on_the_small_matrices"),
]
_FULL = "wider_models or greedy_trace_128 or tiny_model_golden or prompt_continuation or thread_splits"
_SMALL = "wider_models or tiny_model_golden or prompt_continuation or thread_splits"
# test_decode_attention_fallback_paths: (environment, -k selection, pytest id)
ATTN_FALLBACKS = [
    ({"LLAMAHIP_NO_QKV_ATTN": "1"}, _FULL, "no_qkv_attn"),
    ({"LLAMAHIP_NO_ATTN_X": "1"}, _FULL, "no_attn_x"),
    ({"LLAMAHIP_NO_W13_HALF": "1"}, "wider_models or greedy_trace_128 or ragged_contexts", "w13_block_workgroups"),
    ({"LLAMAHIP_W13_HALF": "1"}, "wider_models", "w13_half_workgroups_13b_65b_widths"),
    ({"LLAMAHIP_ATTN_TWO_FROM": "0", "LLAMAHIP_ATTN_LONG_FROM": "-1"}, _SMALL, "two_launch_everywhere"),
    ({"LLAMAHIP_ATTN_LONG_FROM": "0"}, _SMALL + " or ragged_contexts", "stream_everywhere"),          # (on the real 7B: tests/test_gpu_fullsize.py decodes behind 2048-token prompts)
    ({"LLAMAHIP_ATTN_LONG_FROM": "0", "LLAMAHIP_PV_DMA": "0"}, "ragged_contexts or thread_splits", "stream_everywhere_without_dma"),
    ({"LLAMAHIP_ATTN_TWO_FROM": "33", "LLAMAHIP_ATTN_LONG_FROM": "50", "LLAMAHIP_PV_STAGE_ROWS": "3"}, _SMALL, "three_schedules_in_one_call"),
    ({"LLAMAHIP_ATTN_LONG_FROM": "0", "LLAMAHIP_PV_STAGE_ROWS": "2", "LLAMAHIP_PV_SPLIT": "1"}, _SMALL, "stream_unsplit_short_stages"),
]
# test_production_fallbacks_and_selectable_variants: (environment, pytest id)
PRODUCTION_FALLBACKS = [
    ({"LLAMAHIP_NO_LUT_MATH": "1"}, "tables_gathered"), ({"LLAMAHIP_NORM_MODE": "0"}, "norm_two_pass"), ({"LLAMAHIP_NORM_MODE": "1"}, "norm_one_pass_in_prologues"),
    ({"LLAMAHIP_NO_HOST_IO": "1", "LLAMAHIP_HOST_SAMPLER": "1"}, "blit_copies_host_sampler"),
    ({"LLAMAHIP_MFMA_I8": "1", "LLAMAHIP_MFMA_MIN": "32"}, "int8_matrix_core_gemm"), ({"LLAMAHIP_EAGER_PREFILL_COPY": "1"}, "eager_prefill_copies"),
]
_PROD_SELECT = "dc_offset or thread_splits or tiny_model_golden or wider_models or prompt_continuation or runner_event or topk_candidates"

# every nested run of the GPU suite: tag -> (environment, -k selection, test files)
NESTED = {"matrix_core_prompt_gemm_forced": ({"LLAMAHIP_MFMA_MIN": "32"}, "prompt_continuation or long_prompt or multipart or wider_models", [PARITY])}
NESTED.update({"attn:" + tag: (env, sel, [PARITY]) for env, sel, tag in ATTN_FALLBACKS})
NESTED.update({"plan:" + tag: (env, "short_chunks or batched_set_steps_equal or prompt_continuation", [PARITY, PIPELINE]) for env, tag in SET_PLAN_VARIANTS})
NESTED.update({"prod:" + tag: (env, _PROD_SELECT, [PARITY]) for env, tag in PRODUCTION_FALLBACKS})
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
    """One nested `pytest -x -q -m gpu -k <select>` under `env`; returns (ok, ONE summary line: variant=… passed=N failed=M seconds=… and the
    first failure if any, tail of the inner output)."""
    t0 = time.time()
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *files, "-k", select]
    r = subprocess.run(cmd, env=dict(os.environ, LLAMAHIP_NESTED="1", **env), capture_output=True, text=True, cwd=ROOT, timeout=timeout)
    out = r.stdout
    npass = sum(int(x) for x in re.findall(r"(\d+) passed", out.splitlines()[-1] if out.strip() else ""))
    nfail = sum(int(x) for x in re.findall(r"(\d+) (?:failed|error)", out.splitlines()[-1] if out.strip() else ""))
    first = next((ln for ln in out.splitlines() if ln.startswith(("FAILED", "ERROR"))), "")
    summary = f"variant={tag} env={env} passed={npass} failed={nfail} rc={r.returncode} seconds={time.time() - t0:.0f}" + (f" first_failure={first}" if first else "")
    return r.returncode == 0 and npass > 0, summary, out[-2500:] + r.stderr[-1500:]


# The nested runs are host-bound (oracle expectations, process start-up) and independent: the first one a session asks for starts ALL the
# session selected (conftest.py sets SELECTED at collection), NESTED_JOBS at a time, each test then waits for its own (r06_a: 300 s one
# after the other).  Their kernels share the GPU with each other only -- the tests of group 4 that time polls run after the pool is drained.
SELECTED = None
NESTED_JOBS = int(os.environ.get("LLAMAHIP_NESTED_JOBS", "3"))
_pool, _futures = None, {}


def nested_result(tag):
    global _pool
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=max(1, NESTED_JOBS))
        for t in (SELECTED if SELECTED is not None else [tag]):
            env, select, files = NESTED[t]
            _futures[t] = _pool.submit(run_nested, env, select, files, t)
    if tag not in _futures:
        env, select, files = NESTED[tag]
        _futures[tag] = _pool.submit(run_nested, env, select, files, tag)
    return _futures[tag].result()


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    if sys.argv[1:] == ["walk"]:
        import llama_swift_amd as L
        print("plans walked:", walk_set_plans(L))
