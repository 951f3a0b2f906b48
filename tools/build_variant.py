"""Variant build of the gfx950 library for measurements: the named translation units are recompiled with extra flags, the
rest of the objects come from the regular build (build/gfx950), and the result is paddlescience_amd/libppsci_hip.<tag>.so
(git-ignored; travels to the GPU box with gpurun).  Select it with PPSCI_HIP_LIB=<path>.
    python tools/build_variant.py <tag> <unit.hip>[,<unit.hip>...] [-DFLAG ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    tag, units, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    G.build()
    out = os.path.join(ROOT, "build", "variants", tag)
    os.makedirs(out, exist_ok=True)
    objs = []
    for src in G.SOURCES:
        if src in units:
            obj = os.path.join(out, src.replace(".hip", ".o"))
            with open(obj + ".log", "w") as f:
                subprocess.check_call([G.HIPCC] + G.FLAGS + flags + ["-Rpass-analysis=kernel-resource-usage", "-c",
                                                                    os.path.join(G.CSRC, src), "-o", obj], stdout=f, stderr=subprocess.STDOUT)
        else:
            obj = os.path.join(G.BUILD, src.replace(".hip", ".o"))
        objs.append(obj)
    lib = os.path.join(ROOT, "paddlescience_amd", f"libppsci_hip.{tag}.so")
    subprocess.check_call([G.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lhipfft"])
    print(lib)


if __name__ == "__main__":
    main()
