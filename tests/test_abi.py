"""CPU-side checks of the drop-in boundary: libmsvs.so loads without a GPU and exports every entry point that
include/msvs.h declares (no compute calls here)."""
import os
import re

import myscaledb_amd.capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "msvs.h")) as f:
        src = f.read()
    return sorted(set(re.findall(r"MSVS_API\s+[\w\s\*]+?\b(msvs_\w+)\s*\(", src)))


def test_header_declares_what_the_binding_lists():
    assert declared_symbols() == sorted(capi.SYMBOLS)


def test_library_loads_and_exports_every_declared_symbol():
    lib = capi.lib()
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert capi.version().startswith("msvs")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: no product source may import, include, link or dlopen it.  The only shared
    objects the product names are its own, the HIP runtime and RCCL (resolved with dlopen for the multi-GPU exchange)."""
    bad = re.compile(r"(^\s*(from|import)\s+oracle)|(#include\s*[\"<][^\">]*oracle)|(libmsvs_oracle)", re.M)
    so_name = re.compile(r"\"([^\"\s]*lib[^\"\s]*\.so[^\"\s]*)\"")
    for tree in ("myscaledb_amd", "shim"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, tree)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or fn == "Makefile":
                    with open(os.path.join(dirpath, fn)) as f:
                        src = f.read()
                    assert not bad.search(src), fn
                    for name in so_name.findall(src):
                        assert any(t in name for t in ("librccl", "libmsvs", "libamdhip64")), (fn, name)


def test_public_headers_compile_as_c99_and_cxx17(tmp_path):
    """The drop-in boundary is a C ABI: include/msvs.h and include/msvs_host.h must be consumable by a C compiler (no C++
    constructs, no torch types) and by the host's C++17."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = '#include "include/msvs.h"\n#include "include/msvs_host.h"\nint main(void) { return 0; }\n'
    c = tmp_path / "abi.c"
    c.write_text(src)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", root, str(c)], check=True)
    cpp = tmp_path / "abi.cpp"
    cpp.write_text(src)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", root, str(cpp)], check=True)


def test_concurrent_search_driver_rejects_bad_arguments():
    import ctypes as C
    import myscaledb_amd.host as host
    sec = C.c_double(0)
    assert host.lib().msvs_host_concurrent_search(None, None, C.c_size_t(0), C.c_size_t(4), 1, C.c_size_t(1), 1, b"", C.byref(sec),
                                                  None, None, None) != 0

