"""CPU checks of the boundary: the shared library loads without a GPU and exports exactly the header's symbols."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "adaptigraph_amd.h")


def _header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ag_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from adaptigraph_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    declared = _header_functions()
    assert declared == sorted(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ag_abi_version() == 7
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (ag_[a-z_]+)$", out, flags=re.M)))
    assert exported == declared, (exported, declared)


def test_product_library_has_no_probe_or_environment_code():
    """The clock / phase probes and the injected-failure hook live in the diagnostic build only (-DAG_DIAG); the product
    library reads the environment in ONE place (the option defaults of ag_ctx_create) and registers no exit handler."""
    from adaptigraph_amd import _lib
    strings = subprocess.run(["strings", "-n", "6", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for word in ("AG_CLOCK_PROBE", "AG_NODE_PROBE", "AG_TEST_FAIL_AT_CHUNK", "[ag clock probe]", "[ag node probe]"):
        assert word not in strings, word
    csrc = os.path.join(ROOT, "adaptigraph_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip") and f != "ag_diag.hip":
            text = open(os.path.join(csrc, f)).read()
            assert "atexit" not in text, f
            assert text.count("getenv(") == (1 if f == "ag_api.hip" else 0), (f, text.count("getenv("))
    diag = os.path.join(csrc, "libadaptigraph_hip_diag.so")
    if os.path.exists(diag):                                             # same C-ABI, probes inside
        d = subprocess.run(["nm", "-D", "--defined-only", diag], capture_output=True, text=True).stdout
        assert sorted(set(re.findall(r" T (ag_[a-z_]+)$", d, flags=re.M))) == sorted(_lib.EXPORTS)
        assert "AG_NODE_PROBE" in subprocess.run(["strings", "-n", "6", diag], capture_output=True, text=True).stdout


def test_no_cpu_fallback():
    import torch
    import adaptigraph_amd as ag
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ag.construct_edges_from_states_batch(torch.zeros(1, 4, 3), 0.5, torch.ones(1, 4, dtype=torch.bool),
                                             torch.zeros(1, 4, dtype=torch.bool))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ag.Engine("cpu")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "adaptigraph_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text, f"{f} mentions the oracle: the product path must not depend on it"


def test_host_shim_tool_layout_matches_oracle():
    """Host logic on CPU: decode_action + tool keypoint layout (forward_dynamics.py:23,42-75) vs the oracle."""
    import numpy as np
    import torch
    from adaptigraph_amd.forward_dynamics import _tool_layout
    from adaptigraph_amd.plan_utils import decode_action
    from oracle import adaptigraph_oracle as O
    rng = np.random.default_rng(0)
    a = rng.uniform(-3, 3, (5, 3, 4)).astype(np.float32)
    a[..., 3] = rng.uniform(2, 15, (5, 3))
    for pts in ([[0, 0, 0.12]], [[0, 0, 0.1], [0, 0.05, 0.1], [0, 0.025, 0.1], [0, -0.025, 0.1], [0, -0.05, 0.1]]):
        task = {"pusher_points": pts, "sim_real_ratio": 10, "push_length": 0.2}
        dec, rep = decode_action(torch.from_numpy(a), push_length=0.2)
        xz, delta = _tool_layout(dec, torch.from_numpy(a[..., 2]), task)
        odec, orep = O.decode_action(a, 0.2)
        oxz, odelta = O.tool_keypoints(odec, a[..., 2], task)
        assert np.array_equal(dec.numpy(), odec) and np.array_equal(rep.numpy(), orep)
        assert np.array_equal(xz.numpy(), oxz) and np.array_equal(delta.numpy(), odelta)


def test_host_logic_of_the_action_path_choice():
    """CPU-only host logic of dynamics(): the repeat bound comes from task_config['action_upper_lim'] (planning/*.yaml:28-29),
    absent or malformed limits mean "decode on the host"."""
    from adaptigraph_amd.forward_dynamics import _repeat_bound
    assert _repeat_bound({"action_upper_lim": [0.0, 4.5, 3.14, 15]}) == 15
    assert _repeat_bound({"action_upper_lim": [0.0, 4.5, 3.14, 10.9]}) == 10           # int(length) truncates (plan_utils.py:16)
    assert _repeat_bound({}) is None and _repeat_bound({"action_upper_lim": None}) is None
    assert _repeat_bound({"action_upper_lim": [1.0, 2.0]}) is None
    from adaptigraph_amd import _lib
    assert "device_decode" in _lib.OPTIONS and "repeat_sort" in _lib.OPTIONS
    hdr = open(HEADER).read()
    for name in _lib.OPTIONS:                                                          # every option is documented in the header
        assert f'"{name}"' in hdr, name


def test_header_is_plain_c99_and_the_c_client_compiles_against_it(tmp_path):
    """The boundary is C, not C++: the header passes gcc -std=c99 -pedantic -Werror, and the plain-C client of it
    (tools/abi_client/abi_client.c, run on the GPU by tests/test_gpu_abi_client.py) compiles against it."""
    tu = tmp_path / "t.c"
    tu.write_text('#include "adaptigraph_amd.h"\nint main(void) { return (int)(ag_abi_version() != AG_ABI_VERSION); }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", str(tu)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I",
                            os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-fsyntax-only",
                            os.path.join(ROOT, "tools", "abi_client", "abi_client.c")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_hw_queues_record_and_late_import_warning(monkeypatch):
    """adaptigraph_amd.hw_queues says who set GPU_MAX_HW_QUEUES and whether it can still take effect; a process that initialised HIP
    before the import gets a RuntimeWarning instead of silence."""
    import importlib
    import warnings
    import torch
    import adaptigraph_amd as ag
    assert ag.hw_queues["GPU_MAX_HW_QUEUES"] == __import__("os").environ["GPU_MAX_HW_QUEUES"]
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        info = ag._hw_queues_default()
    assert info == {"GPU_MAX_HW_QUEUES": "8", "set_by": "adaptigraph_amd", "in_effect": False}
    assert len(w) == 1 and "before the package was imported" in str(w[0].message)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "6")
    assert ag._hw_queues_default() == {"GPU_MAX_HW_QUEUES": "6", "set_by": "environment", "in_effect": None}
