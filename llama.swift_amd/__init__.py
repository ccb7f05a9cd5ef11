"""llama.swift_amd -- MI355X (gfx950) drop-in for the quantized-LLaMA hot path of
alexrozanski/llama.swift.

The product is ``csrc/libllamahip.so`` (hand-written HIP kernels behind the C ABI of
``include/llamahip.h``).  This package is only the thin host-side glue the tests and the benchmark
use: a ctypes binding of that C ABI (:mod:`.binding`) and a Python mirror of the reference's Swift
surface ``LlamaRunner`` / ``Config`` / ``RunState`` (:mod:`.runner`,
Sources/llama/LlamaRunner.swift:11-124).

There is no CPU fallback: importing works anywhere (so the symbol table can be checked without a
GPU), but every compute call fails loudly if the shared library or a HIP device is missing.

The directory is literally named ``llama.swift_amd``; import it as ``llama_swift_amd`` through the
shim module at the repository root.
"""
from .binding import (  # noqa: F401
    LIB_PATH,
    LlamaHipError,
    Model,
    Sampler,
    bench_gemv_names,
    build,
    declared_symbols,
    gemm_paths,
    lib,
    op_mul_mat_q4_0,
    op_quantize_row_q4_0,
    op_topk,
    quantize_file,
    set_plan,
    version,
)
from .runner import Config, LlamaRunner, RunState  # noqa: F401
