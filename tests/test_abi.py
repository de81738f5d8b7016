"""The C-ABI library loads and exports every symbol include/trust4_hip.h declares; without a GPU the
engine refuses to start (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from t4libs import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "trust4_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(t4_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    import trust4_amd.build as b
    path = b.build()
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    os.environ.pop("T4_LIB", None)
    import trust4_amd
    with pytest.raises(trust4_amd.T4Error):
        trust4_amd.Engine(0)


def test_rccl_is_bound_at_run_time_not_at_load_time():
    """libt4hip.so must load on a host without librccl (only t4_comm uses it, by dlopen at the first communicator): no NEEDED entry
    and no undefined nccl* symbol."""
    import subprocess
    import trust4_amd.build as b
    path = b.build()
    dyn = subprocess.run(["readelf", "-d", path], check=True, stdout=subprocess.PIPE, text=True).stdout
    needed = re.findall(r"\(NEEDED\)\s+Shared library: \[([^\]]+)\]", dyn)
    assert needed and not [n for n in needed if "rccl" in n or "nccl" in n], needed
    syms = subprocess.run(["nm", "-D", "--undefined-only", path], check=True, stdout=subprocess.PIPE, text=True).stdout
    assert "nccl" not in syms
