"""CPU: the C-ABI library loads and exports every symbol include/iivision.h
declares (no compute calls -- there is no GPU here), and the host layer refuses
to run without a GPU instead of falling back."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "iivision.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(iiv_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(native):
    syms = _declared_symbols()
    assert len(syms) >= 20
    L = ctypes.CDLL(native.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), "libiivision.so does not export %s" % s
    assert set(syms) == set(native.SYMBOLS)


def test_constants_without_gpu(native):
    L = native.lib()
    assert L.iiv_version().startswith(b"iivision")
    assert (L.iiv_masked_bits(0), L.iiv_masked_bits(1)) == (14, 13)
    assert (L.iiv_masked_dots(0), L.iiv_masked_dots(1)) == (18, 10)
    assert (L.iiv_num_offsets(0), L.iiv_num_offsets(1)) == (2, 4)
    assert L.iiv_table_entries(0) == 2 << 28 and L.iiv_table_entries(1) == 4 << 26
    assert L.iiv_store_table_entries(0) == 2 << 22 and L.iiv_store_table_entries(1) == 4 << 20


def test_no_cpu_fallback(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        native.cie2000_matrix([[0, 0, 0]] * 16)
    with pytest.raises(RuntimeError):
        native.build_table(1, [0] * 256)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under ii-vision_amd/ may mention it."""
    pkg = os.path.join(ROOT, "ii-vision_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert "import oracle" not in txt and "liboracle" not in txt and "iiv_oracle.h" not in txt, fn


def test_torch_custom_operators_are_registered_over_the_c_abi(native):
    """north_star: "hand-written HIP kernels via PyTorch-ROCm custom ops".  torch_ops.py registers the C ABI's hot entry
    points as torch.ops.iivision.*; registration needs no GPU, running them does (no CPU implementation)."""
    import torch
    import torch_ops
    assert set(torch_ops.NAMES) == {"cie2000_matrix", "build_table", "build_store_table", "encode", "encode_streams",
                                    "emit_chunk", "frames_to_memory_maps"}
    for name in torch_ops.NAMES:
        op = getattr(torch.ops.iivision, name)
        assert "iivision::" + name in str(op.default._schema)
    # the launch operators mutate their output buffer and return nothing: no hidden allocation, no hidden copy
    assert "ops_out" in str(torch.ops.iivision.encode.default._schema) and "-> ()" in str(torch.ops.iivision.encode.default._schema)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            torch.ops.iivision.cie2000_matrix(torch.zeros((16, 3), dtype=torch.uint8))


def test_library_and_committed_counters_are_of_the_current_sources():
    """The build id compiled into iiv_version() is a hash of csrc/, the headers and the flags (csrc/Makefile: BUILD_ID).  The
    in-tree library must be a build of the sources as they stand, and every entry of profiles/pmc_latest.json -- the counter run
    bench.py quotes -- must carry that id: a kernel (or header) change that is not followed by tools/round_evidence.sh would
    otherwise leave bench.py printing `counters: "stale"` at the driver's end-of-round run."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = subprocess.run(["make", "-s", "-C", os.path.join(root, "ii-vision_amd", "csrc"), "build-id"],
                          capture_output=True, text=True, check=True).stdout.strip()
    assert len(want) == 12
    import _iiv_native
    assert _iiv_native.build_id() == want, "ii-vision_amd/libiivision.so is not a build of the current sources: run make -C ii-vision_amd/csrc"
    with open(os.path.join(root, "profiles", "pmc_latest.json")) as f:
        latest = json.load(f)
    stale = {k: v.get("build_id") for k, v in latest.items() if v.get("build_id") != want}
    assert not stale, "profiles/pmc_latest.json holds counter runs of another build %s (this one: %s): run tools/round_evidence.sh" % (stale, want)
