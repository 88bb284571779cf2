"""The C-ABI library loads and exports every symbol include/owlknn.h declares (no GPU needed)."""
import os
import re

import pytest

from owlraytracing_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"TKNN_API[^;(]*?\b(tknn\w+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    names = _declared("owlknn.h")
    assert names, "header parse found nothing"
    lib = _lib.load()
    for name in sorted(names):
        assert hasattr(lib, name), "libowl_mi355x.so lacks %s" % name
    assert names == set(_lib.SIGNATURES), "python binding and header disagree"


def test_neigh_record_is_24_bytes():
    text = open(os.path.join(ROOT, "include", "owlknn.h")).read()
    assert "int32_t pad_" in text and "int64_t intersections" in text
    from owlraytracing_amd.trueknn import NEIGH_BYTES
    assert NEIGH_BYTES == 24


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from owlraytracing_amd.trueknn import TrueKNN
    with pytest.raises(RuntimeError):
        TrueKNN()
    # and straight through the C-ABI: creation fails with a HIP error, it does not return a CPU engine
    import ctypes
    h = ctypes.c_void_p()
    rc = _lib.load().tknnCreate(ctypes.byref(h))
    assert rc != 0 and not h.value


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "owlraytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert not re.search(r'#include\s+[<"].*oracle', src), f


def test_ctypes_structs_have_the_header_layout(tmp_path):
    """Sizes and field offsets of the ctypes mirrors against include/owlknn.h compiled by gcc as C99."""
    import ctypes
    import subprocess
    pairs = {"tknnSolveInfo": _lib.SolveInfo, "tknnSolveOptions": _lib.SolveOptions,
             "tknnDbscanInfo": _lib.DbscanInfo, "tknnDbscanAutoInfo": _lib.DbscanAutoInfo, "tknnBuildInfo": _lib.BuildInfo}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "owlknn.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for field, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, field, cname, field))
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for line in out.splitlines():
        cname, what, value = line.split()
        cls = pairs[cname]
        if what == "size":
            assert ctypes.sizeof(cls) == int(value), cname
        else:
            assert getattr(cls, what).offset == int(value), (cname, what)
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in pairs.values())
