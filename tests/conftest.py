"""pytest configuration: the `gpu` marker and shared fixtures.

``-m "not gpu"``: oracle vs golden vectors / reference build, host logic, C-ABI symbol table.
``-m gpu``      : parity of the HIP path (through the C ABI) against the oracle.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    import reflib
    lib = os.path.join(ROOT, "llama.swift_amd", "csrc", "libllamahip.so")
    tool = os.path.join(ROOT, "llama.swift_amd", "csrc", "tools", "make_synth_model")
    if not (os.path.exists(lib) and os.path.exists(tool) and os.path.exists(reflib.ORACLE_SO)):
        subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, check=True)
    elif os.path.isdir("/root/reference") and not reflib.have_ref():
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)


@pytest.fixture(scope="session")
def built():
    _ensure_built()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    import reflib
    return reflib.OracleLib()


@pytest.fixture(scope="session")
def ref(built):
    import reflib
    if not reflib.have_ref():
        pytest.skip("oracle/_ref/libggml_ref.so not present (needs /root/reference at build time)")
    return reflib.RefLib()


@pytest.fixture(scope="session")
def L(built):
    import llama_swift_amd
    return llama_swift_amd


@pytest.fixture(scope="session")
def tmp_models(tmp_path_factory):
    return tmp_path_factory.mktemp("models")


def synth_tool(out, **kw):
    tool = os.path.join(ROOT, "llama.swift_amd", "csrc", "tools", "make_synth_model")
    args = [tool, "--out", str(out)]
    for k, v in kw.items():
        args += [f"--{k}", str(v)]
    subprocess.run(args, check=True, capture_output=True)
    return str(out)
