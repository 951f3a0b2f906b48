"""ISA lint of the built gfx950 library: packed fp32 arithmetic that broadcasts the HIGH half of a source into both results.

Found on MI355X in round 5 (tools/det_probe7.py): in the fused tile kernel the ONE instruction of the form
    v_pk_mul_f32 d, a, b op_sel:[0,1]          (low result = a.lo * b.HI, high result = a.hi * b.HI)
delivered  a.lo * 0  as its low result in lanes 48..63, in about 1 of 500 workgroups, whenever two workgroups shared a CU -- a
run-to-run difference of 1e-3 in one feature of one tile.  hipcc's SLP vectoriser produces the form when it pairs two different
quantities of one feature and multiplies them by a splat of a third that happens to sit in the high register of a pair.  The
kernels avoid it by doing such arithmetic on whole float4 registers (csrc/taylor_fused.inc ppsci_act_from_stash4); this lint
lists every two-source packed fp32 multiply / add whose op_sel is a bare [0,1] or [1,0] (the three-source forms -- v_pk_fma_f32
op_sel:[0,1,0], thousands of uses in the forward kernels, bitwise reproducible in every run -- are not flagged).

    python tools/isa_lint.py [path/to/libppsci_hip.so]      -> one line per hit: kernel, instruction;  exit status 1 if any"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BARE = re.compile(r"\b(v_pk_mul_f32|v_pk_add_f32)\b.*\bop_sel:\[(0,1|1,0)\]\s*(//|$)")


def scan(lib: str, only: str = ""):
    """[(kernel, instruction)] of the flagged instructions in every gfx950 code object bundled in `lib` (with `only`: in the
    code objects that define a symbol containing that string -- the others are not disassembled)."""
    hits = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        os.symlink(os.path.abspath(lib), local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        objs = sorted(f for f in os.listdir(tmp) if f.endswith("gfx950"))
        if not objs:
            raise RuntimeError(f"{lib}: no gfx950 code object found")
        for f in objs:
            if only:
                syms = subprocess.run([OBJDUMP, "-t", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
                if only not in syms:
                    continue
            out = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
            kernel = "?"
            for line in out.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                if m:
                    kernel = m.group(1)
                elif BARE.search(line):
                    hits.append((kernel, line.split("//")[0].strip()))
    return hits


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "paddlescience_amd", "libppsci_hip.so")
    hits = scan(lib)
    for k, ins in hits:
        print(k, "|", ins)
    print(f"{len(hits)} flagged instruction(s) in {lib}")
    sys.exit(1 if hits else 0)
