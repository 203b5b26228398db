"""CPU-side checks of the C-ABI boundary: the library loads without a GPU and
exports every symbol include/tce_hip.h declares (no compute calls here)."""
import ctypes
import os

import pytest

from tce_rl_amd import _lib


def test_library_built():
    assert os.path.exists(_lib.LIB_PATH), \
        "run `python -m tce_rl_amd.build` (or __graft_entry__.build()) first"


def test_header_symbols_exported():
    protos = _lib._parse_header(_lib.HEADER_PATH)
    assert len(protos) >= 10
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(raw, n)]
    assert not missing, "declared in tce_hip.h but not exported: %s" % missing


def test_load_and_error_channel():
    lib = _lib.load()
    assert lib.tce_version() >= 1
    assert lib.tce_device_count() >= 0
    # argument validation happens on the host before any launch
    with pytest.raises(RuntimeError, match="moments"):
        _lib.call("tce_moments_partial_f32", None, 0, None, None)


def test_no_cpu_fallback():
    import torch
    from tce_rl_amd import ops
    r = torch.zeros(2, 4)
    v = torch.zeros(2, 5)
    d = torch.zeros(2, 4, dtype=torch.bool)
    with pytest.raises(RuntimeError, match="HIP device only"):
        ops.gae(r, v, d, d, 1.0, 0.95)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: no module of the package (nor bench.py's
    timed path) may import it; bench.py only does so inside cpu_baseline()."""
    import ast
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    for path in (root / "tce_rl_amd").rglob("*.py"):
        tree = ast.parse(path.read_text())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.")
                           for n in names), path
    bench = ast.parse((root / "bench.py").read_text())
    for fn in [n for n in bench.body if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, (ast.Import, ast.ImportFrom)) and
                   ("oracle" in (getattr(n, "module", None) or "") or
                    any("oracle" in a.name for a in n.names))
                   for n in ast.walk(fn))
        assert not uses or fn.name == "cpu_baseline", fn.name
