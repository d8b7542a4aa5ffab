"""Which device functions differ between two commits?  (hipcc cross-compiles; no GPU needed.)

    python tools/kernel_identity.py <old commit> [<new commit or WORKTREE>] [--json out.json]

Compiles the .hip files of both trees with the product's flags to gfx950 assembly (--cuda-device-only -S) and compares every device function's
instruction stream (comments, debug labels and the compilation-unit id stripped).  Used for the provenance of profiles measured on an
earlier csrc_hash: a PMC profile of a workload stays meaningful across a commit that leaves the instructions of the kernels it launches
unchanged (bench.py still applies the hash rule literally; this is evidence, not a bypass)."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC", "-fvisibility=hidden"]
CSRC = "opentk-pathtracer_amd/csrc"


def checkout(commit, dst):
    if commit == "WORKTREE":
        return os.path.join(ROOT, CSRC), os.path.join(ROOT, "include")
    tar = subprocess.run(["git", "-C", ROOT, "archive", commit, CSRC, "include"], check=True, capture_output=True).stdout
    subprocess.run(["tar", "x", "-C", dst], input=tar, check=True)
    return os.path.join(dst, CSRC), os.path.join(dst, "include")


def functions(csrc, inc, work):
    out = {}
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(".hip"):
            continue
        s = os.path.join(work, f + ".s")
        subprocess.run(["hipcc", *FLAGS, "--cuda-device-only", "-S", "-I" + inc, "-o", s, os.path.join(csrc, f)], check=True, capture_output=True)
        txt = open(s).read()
        for m in re.finditer(r"\.type\s+(\S+),@function\n(.*?)\.Lfunc_end\d+:", txt, re.S):
            body = re.sub(r";.*", "", m.group(2))
            body = re.sub(r"\.Ltmp\d+:|\.loc.*|\.file.*", "", body)
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            out[f"{f}: {name}"] = hashlib.sha1(body.encode()).hexdigest()[:12]
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    old, new = args[0], (args[1] if len(args) > 1 else "WORKTREE")
    js = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    with tempfile.TemporaryDirectory() as ta, tempfile.TemporaryDirectory() as tb:
        a = functions(*checkout(old, ta), ta)
        b = functions(*checkout(new, tb), tb)
    res = {"old": old, "new": new, "flags": FLAGS,
           "identical": sorted(n for n in a if a[n] == b.get(n)),
           "changed": sorted(n for n in set(a) | set(b) if a.get(n) != b.get(n))}
    for n in res["changed"]:
        print("CHANGED  ", n)
    print(f"{len(res['identical'])} device functions identical, {len(res['changed'])} changed ({old} -> {new})")
    if js:
        json.dump(res, open(js, "w"), indent=1)


if __name__ == "__main__":
    main()
