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
