"""CPU: the C-ABI library loads, exports every symbol include/mi_ldu.h declares, and fails
loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "mi_ldu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-zA-Z0-9_]+)\s*\(", txt)))


def test_header_and_library_agree(pkg):
    names = declared_functions()
    assert len(names) >= 45
    lib = pkg.engine.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mi_ldu.h but not exported"
    assert sorted(pkg.engine.SYMBOLS) == names


def test_every_declaration_cites_the_reference():
    txt = open(os.path.join(ROOT, "include", "mi_ldu.h")).read()
    assert len(re.findall(r"\.[CH]:\d+", txt)) >= 25


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback(pkg):
    eng = pkg.engine
    assert not eng.device_available()
    with pytest.raises(eng.MiError, match="no CPU fallback|no HIP device"):
        eng.Context(0)


def test_product_never_imports_the_oracle():
    pk = os.path.join(ROOT, "rapidcfd-dev_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", ".C", ".H")):
                src = open(os.path.join(dp, f)).read()
                for pat in (r"(from|import)\s+oracle", r"liboracle", r"oracle/|oracle\\.py", r"\borc_"):
                    assert not re.search(pat, src), f"{f} references the oracle ({pat})"
