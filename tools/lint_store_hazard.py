"""Build-time lint for the 16-byte buffer-store data hazard of gfx950 ("A hazard worth recording").

A `buffer_store_dwordx3/x4` whose offset operand is an SGPR reads its data registers one issue slot late: a VALU instruction
that immediately follows and writes one of those VGPRs corrupts the stored data (observed as 0.5 % of polynomials losing one element
in round 2).  LLVM (ROCm 7.2) inserts no wait state there, so every such store in the sources is followed by ALCH_STORE_GUARD
(`s_nop 0`, alchemy_amd/csrc/ntt_engine.hpp).  The guard is applied by hand; this lint makes the build fail when a store is
written without it: it disassembles the gfx950 code objects embedded in alchemy_amd/csrc/build/*.o and reports every
>8-byte buffer store with an SGPR offset whose NEXT instruction is a VALU write to one of its data registers.

  python tools/lint_store_hazard.py [objects...]      exit status 1 and one line per violation
tests/test_store_hazard_lint.py runs it on the built objects and on synthetic listings."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*([^,]+),\s*s\[\d+:\d+\],\s*(\S+)")
INSN = re.compile(r"^\s*([a-z][a-z0-9_]*)\s*(.*?)\s*(?://.*)?$")
VREG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def vgpr_range(tok):
    m = VREG.match(tok.strip())
    if not m:
        return None
    if m.group(1) is not None:
        return int(m.group(1)), int(m.group(1))
    return int(m.group(2)), int(m.group(3))


def valu_dests(mnemonic, operands):
    """VGPR ranges a VALU instruction writes: its first operand (v_swap writes both of its operands)."""
    if not mnemonic.startswith("v_") or mnemonic.startswith(("v_cmp_", "v_cmpx_", "v_readlane", "v_readfirstlane", "v_nop")):
        return []
    ops = [o.strip() for o in operands.split(",")]
    picks = ops[:2] if mnemonic.startswith("v_swap") else ops[:1]
    return [r for r in (vgpr_range(o) for o in picks) if r]


def scan(listing_lines, name="<listing>"):
    """Violations in a disassembly listing (llvm-objdump -d format): list of (name, line number, store, next instruction)."""
    out = []
    pending = None                                   # (line no, text, lo, hi) of a hazardous store awaiting its successor
    for no, raw in enumerate(listing_lines, 1):
        line = raw.rstrip("\n")
        if not line.strip() or line.lstrip().startswith(("//", ";")) or line.rstrip().endswith(":") or "file format" in line or line.startswith("Disassembly"):
            continue
        m = INSN.match(line)
        if not m:
            continue
        mnem, ops = m.group(1), m.group(2)
        if pending is not None:
            for lo, hi in valu_dests(mnem, ops):
                if lo <= pending[3] and hi >= pending[2]:
                    out.append((name, pending[0], pending[1].strip(), line.strip()))
                    break
            pending = None
        s = STORE.match(line)
        if s and re.match(r"^s\d+$", s.group(5).strip()):          # offset operand is an SGPR (not `off`, not an inline constant)
            pending = (no, line, int(s.group(2)), int(s.group(3)))
    return out


def disassemble(obj, workdir):
    """gfx950 disassembly of a HIP object / shared library (the .hip_fatbin bundle), as a list of lines; [] when it holds none."""
    fat = os.path.join(workdir, os.path.basename(obj) + ".fat")
    r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(workdir, "discard.o")],
                       capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return []
    co = fat + ".co"
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}"],
                   check=True, capture_output=True)
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout.splitlines()


def lint_objects(paths):
    bad, stores = [], 0
    with tempfile.TemporaryDirectory() as d:
        for p in paths:
            lines = disassemble(p, d)
            stores += sum(1 for l in lines if STORE.match(l))
            bad += scan(lines, os.path.basename(p))
    return bad, stores


def main():
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    paths = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "alchemy_amd", "csrc", "build", "*.o")))
    bad, stores = lint_objects(paths)
    for name, no, store, nxt in bad:
        print(f"{name}:{no}: {store}  <- next: {nxt}")
    print(f"{len(paths)} objects, {stores} 12/16-byte buffer stores, {len(bad)} unguarded", file=sys.stderr)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
