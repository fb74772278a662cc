"""The C-ABI shared library loads and exports every symbol include/p2i_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "p2i_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(p2i_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from p2igan_bench import _hip
    assert os.path.exists(_hip.lib_path()), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_hip.lib_path())
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/p2i_hip.h but not exported"
    missing = [n for n in names if n not in _hip.SIGNATURES and n not in ("p2i_abi_version", "p2i_last_error")]
    assert not missing, f"no ctypes signature for {missing}"
    lib.p2i_abi_version.restype = ctypes.c_int
    assert lib.p2i_abi_version() == 1


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch
    from p2igan_bench import ops
    spec = ops.ConvSpec(4, 4, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    with pytest.raises(RuntimeError):
        ops.conv_fwd(spec, torch.zeros(1, 4, 8, 8), torch.zeros(9, 4, 32))
