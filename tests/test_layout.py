"""Repository contract: the oracle is test infrastructure only, the product never falls back to it, nothing that
runs on the GPU box reads /root/reference, and every required artefact exists."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(sub):
    for d, _dirs, files in os.walk(os.path.join(ROOT, sub)):
        if "__pycache__" in d:
            continue
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_product_never_imports_oracle_or_reference():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|/root/reference", re.M)
    for path in _py_files("codeps_amd"):
        assert not pat.search(open(path).read()), f"{path} touches oracle/ or the reference"


def test_oracle_declares_itself_test_infrastructure():
    for path in _py_files("oracle"):
        head = open(path).read(600)
        assert "TEST INFRA" in head.upper(), path
    assert "parity unpinned" in open(os.path.join(ROOT, "oracle", "resnet.py")).read().lower()
    assert "parity unpinned" in open(os.path.join(ROOT, "DESIGN.md")).read().lower()


def test_gpu_box_code_does_not_read_the_reference():
    # test_boundary_cpu.py is a NON-gpu test whose reference comparisons are skipped where the tree is absent (the GPU box)
    allowed = {"make_golden.py", "_ref_import.py", "test_boundary_cpu.py"}
    for path in list(_py_files("tests")) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        if os.path.basename(path) in allowed or os.path.basename(path) == "test_layout.py":
            continue
        assert "/root/reference" not in open(path).read(), path


def test_required_artefacts_exist():
    for rel in ("include/codeps_hip.h", "DESIGN.md", "INTEGRATION.md", "bench.py", "__graft_entry__.py", "oracle/__init__.py",
                "tests/golden/make_golden.py", "profiles", "codeps_amd/csrc/Makefile"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
    assert any(f.endswith(".csv") for f in os.listdir(os.path.join(ROOT, "profiles")))


def test_no_cuda_compat_layers_in_kernels():
    bad = re.compile(r"__HIP_PLATFORM_AMD__|hipify|cuda_runtime|__CUDACC__|triton", re.I)
    csrc = os.path.join(ROOT, "codeps_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".cpp")):
            assert not bad.search(open(os.path.join(csrc, f)).read()), f
