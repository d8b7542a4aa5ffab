"""The public header is plain C: a C99 compiler in pedantic mode must accept include/mi355pt.h on its own, and a C program that
calls every declared entry point must compile against it (no GPU needed; nothing is executed)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi355pt.h")


def test_header_is_pedantic_c99(tmp_path):
    p = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", HEADER],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def test_c_translation_unit_can_name_every_entry_point(tmp_path):
    names = sorted(set(re.findall(r"PT_API\s+[\w\s\*]+?\b(pt_\w+)\s*\(", open(HEADER).read())))
    assert len(names) >= 35
    src = tmp_path / "use_all.c"
    body = "\n".join(f"    table[{i}] = (void (*)(void)){n};" for i, n in enumerate(names))
    src.write_text(f'#include "mi355pt.h"\nvoid (*table[{len(names)}])(void);\nvoid fill(void)\n{{\n{body}\n}}\n')
    p = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                        str(tmp_path / "use_all.o")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def test_diagnostic_builds_export_the_same_abi():
    """The audit / chaos builds (hand-over audit, delay injection: tools/handover_stress + tests/test_gpu_round3.py) are the product
    sources with -DPT_AUDIT / -DPT_CHAOS: they must exist after __graft_entry__.build(), export every declared entry point and the
    audit read-out, and the product library must answer the read-out with "not an audit build" (no GPU needed: nothing is called)."""
    import ctypes as C
    import __graft_entry__ as graft
    pkg = graft.load_package()
    if not os.path.exists(pkg.native.LIB_PATH):
        graft.build()
    names = pkg.native.declared_symbols()
    for variant in pkg.native.VARIANTS:
        path = pkg.native.variant_path(variant)
        if not os.path.exists(path):
            pkg.native.build_variant(variant)
        lib = C.CDLL(path)
        missing = [n for n in names + ["pt_debug_audit_read"] if not hasattr(lib, n)]
        assert not missing, f"{path} lacks {missing}"
    assert os.path.exists(pkg.native.build_stress_tool())
