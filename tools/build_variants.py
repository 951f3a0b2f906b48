"""Builds ablation / tuning variants of libppsci_hip.so into build/variants/<name>.so (tanh kernels only
matter for the timing script, but every TU is built so the library links)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

VARIANTS = {
    "base": [],
    "timers": ["-DPPSCI_PHASE_TIMERS"],
    "occ2": ["-DPPSCI_BWD_MIN_WAVES=2"],
    "bwd8occ2": ["-DPPSCI_BWD_MIN_WAVES=2", "-DPPSCI_BWD_WAVES=8"],
    "fwd8": ["-DPPSCI_FWD_WAVES=8"],
    "nostash": ["-DPPSCI_ABL_NOSTASH"],
    "noatomic": ["-DPPSCI_ABL_NOATOMIC"],
    "not2n": ["-DPPSCI_ABL_NOT2N"],
    "norowsum": ["-DPPSCI_ABL_NOROWSUM"],
    "nobar": ["-DPPSCI_ABL_NOBAR"],
    "norot": ["-DPPSCI_ABL_NOROT"],
    "nobar_norot": ["-DPPSCI_ABL_NOBAR", "-DPPSCI_ABL_NOROT"],
}


def build_variant(name, extra):
    out = os.path.join(ROOT, "build", "variants", name)
    os.makedirs(out, exist_ok=True)

    def one(src):
        # only the tanh reverse kernel is rebuilt (bench shape only); everything else comes from the main build
        if src not in ("taylor_bwd_tanh.hip", "taylor_fwd_tanh.hip", "taylor_api.hip", "wgrad_reduce.hip"):
            return os.path.join(ROOT, "build", "gfx950", src.replace(".hip", ".o"))
        obj = os.path.join(out, src.replace(".hip", ".o"))
        subprocess.check_call([G.HIPCC] + G.FLAGS + extra + ["-DPPSCI_VARIANT_MIN", "-c", os.path.join(G.CSRC, src), "-o", obj],
                              stderr=subprocess.DEVNULL)
        return obj

    with ThreadPoolExecutor(8) as ex:
        objs = list(ex.map(one, G.SOURCES))
    os.makedirs(os.path.join(ROOT, "variants_out"), exist_ok=True)
    lib = os.path.join(ROOT, "variants_out", name + ".so")  # travels with gpurun (build/ does not)
    subprocess.check_call([G.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lhipfft"])
    return lib


if __name__ == "__main__":
    names = sys.argv[1:] or list(VARIANTS)
    with ThreadPoolExecutor(8) as ex:
        for n, lib in zip(names, ex.map(lambda n: build_variant(n, VARIANTS[n]), names)):
            print(n, lib, flush=True)
