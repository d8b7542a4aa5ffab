"""Register / LDS / scratch use of every kernel of the tree, from the code objects' own notes (no GPU: hipcc cross-compiles).

    python tools/kernel_resources.py [out.md]

Compiles the three .hip files with the product's flags (device only), unbundles the gfx950 code object and prints per kernel what
llvm-readelf --notes says: VGPRs (-> wavefronts per SIMD: 512 / granule-rounded VGPRs, at most 8), SGPRs, SGPR / VGPR spills, scratch bytes
per lane, static LDS (the integrator's LDS is dynamic: DESIGN 3.1), instruction counts by class from the disassembly."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC", "-fvisibility=hidden"]
CSRC = os.path.join(ROOT, "opentk-pathtracer_amd", "csrc")


def main():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for f in sorted(os.listdir(CSRC)):
            if not f.endswith(".hip"):
                continue
            o, co = os.path.join(td, f + ".o"), os.path.join(td, f + ".co")
            subprocess.run(["hipcc", *FLAGS, "--cuda-device-only", "-c", "-I" + os.path.join(ROOT, "include"), "-o", o, os.path.join(CSRC, f)], check=True, capture_output=True)
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={o}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                           check=True, capture_output=True)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
            counts = {}
            for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M):
                ops = re.findall(r"^\s+(v_|s_|ds_|global_|buffer_|scratch_|flat_)\w*", m.group(2), re.M)
                c = {}
                for op in ops:
                    c[op] = c.get(op, 0) + 1
                counts[m.group(1)] = c
            for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
                get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
                sym = get("name")
                name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().replace("(pt::FrameArgs)", "").replace("void pt::", "").replace("pt::", "")
                v = int(get("vgpr_count"))
                waves = min(8, 512 // (((v + 7) // 8) * 8)) if v else 8
                c = counts.get(sym, {})
                rows.append((f, name, v, waves, get("sgpr_count"), get("sgpr_spill_count"), get("vgpr_spill_count"), get("private_segment_fixed_size"),
                             get("group_segment_fixed_size"), c.get("v_", 0), c.get("s_", 0), c.get("ds_", 0), c.get("global_", 0) + c.get("buffer_", 0) + c.get("flat_", 0), c.get("scratch_", 0)))
    out = [f"Kernel resources of the tree (csrc_hash {g.load_package().native.csrc_hash()}; `python tools/kernel_resources.py`; code-object notes + static instruction counts)", "",
           "| file | kernel | VGPRs | waves / SIMD | SGPRs | SGPR spills | VGPR spills | scratch B | static LDS B | VALU | SALU | LDS | VMEM | scratch ops |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        out.append("| " + " | ".join(str(x) for x in r) + " |")
    text = "\n".join(out) + "\n"
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)


if __name__ == "__main__":
    main()
