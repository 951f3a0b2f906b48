"""TEST INFRASTRUCTURE: builds the kernel sources for the CPU SIMT emulator (hip_emu.h) and injects the
resulting library into paddlescience_amd._lib.  Never used by the product path."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "paddlescience_amd", "csrc")
EXTRA = os.environ.get("PPSCI_EMU_EXTRA_FLAGS", "").split()  # experiment builds (e.g. -DPPSCI_WAVE_ACC) get their own directory
OUT = os.path.join(ROOT, "tests", "_emu_build" + ("_" + "_".join(f.lstrip("-D") for f in EXTRA) if EXTRA else ""))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SOURCES = ["taylor_bwd_b_tanh.hip", "taylor_bwd_b_tanh_fourier.hip", "taylor_bwd_b_silu.hip", "taylor_bwd_b_sin.hip", "taylor_bwd_b_cos.hip", "taylor_bwd_b_sigmoid.hip", "taylor_bwd_b_gelu.hip", "taylor_bwd_b_relu.hip", "taylor_bwd_b_leaky_relu.hip", "taylor_bwd_b_elu.hip", "taylor_bwd_b_selu.hip", "taylor_bwd_b_identity.hip", "taylor_bwd_b_swish.hip", "taylor_bwd_b_stan.hip", "taylor_bwd_wx_tanh.hip", "taylor_bwd_wx_tanh_fourier.hip", "taylor_bwd_wx_silu.hip", "taylor_bwd_wx_sin.hip", "taylor_bwd_wx_cos.hip", "taylor_bwd_wx_sigmoid.hip", "taylor_bwd_wx_gelu.hip", "taylor_bwd_wx_relu.hip", "taylor_bwd_wx_leaky_relu.hip", "taylor_bwd_wx_elu.hip", "taylor_bwd_wx_selu.hip", "taylor_bwd_wx_identity.hip", "taylor_fwd_tanh.hip", "taylor_fwd_silu.hip", "taylor_fwd_sin.hip", "taylor_fwd_sigmoid.hip", "taylor_fwd_cos.hip", "taylor_fwd_gelu.hip", "taylor_fwd_swish.hip", "taylor_fwd_stan.hip", "taylor_bwd_swish.hip", "taylor_bwd_stan.hip", "taylor_fwd_tanh_fourier.hip", "taylor_bwd_tanh_fourier.hip", "reparam.hip", "taylor_bwd_tanh.hip",
           "taylor_bwd_silu.hip", "taylor_bwd_sin.hip", "taylor_bwd_sigmoid.hip", "taylor_bwd_cos.hip", "taylor_bwd_gelu.hip", "taylor_fwd_relu.hip", "taylor_bwd_relu.hip", "taylor_fwd_leaky_relu.hip", "taylor_bwd_leaky_relu.hip", "taylor_fwd_elu.hip", "taylor_bwd_elu.hip", "taylor_fwd_selu.hip", "taylor_bwd_selu.hip", "taylor_fwd_identity.hip", "taylor_bwd_identity.hip", "taylor_step_tanh.hip", "taylor_step_silu.hip", "taylor_step_sin.hip", "taylor_fused_tanh.hip", "taylor_fused_silu.hip", "taylor_fused_sin.hip", "taylor_fused_static_tanh.hip", "taylor_fused_static_silu.hip", "taylor_fused_static_sin.hip", "taylor_api.hip", "wgrad_reduce.hip", "spectral_conv.hip", "fno.hip", "field_loss.hip", "fft.hip", "spinn.hip", "pirate.hip", "epilogue_optim.hip", "comm.hip", "coupling.hip", "uno.hip", "sht.hip"]
# The four sources that carry 80 % of the suite's emulated time (PPSCI_EMU_PROFILE, hip_emu.h) are built -O1: 3x faster to run,
# under a minute to compile next to the rest at -O0.
HOT = {"taylor_fused_tanh.hip", "taylor_fwd_tanh.hip", "taylor_bwd_tanh.hip", "fno.hip"}
HEADERS = ["ppsci_common.h", "taylor_tile.h", "taylor_fwd.inc", "taylor_bwd.inc", "taylor_fwd_wide.inc", "taylor_bwd_wide.inc", "taylor_fwd_wx.inc", "taylor_bwd_wx.inc", "taylor_bwd_wx_tu.inc", "taylor_bwd_lw.inc", "taylor_fwd_body.inc", "taylor_bwd_body.inc", "taylor_step.inc", "taylor_step.h", "taylor_step_tail.h", "taylor_fused.inc", "epilogue_vm.h", "epi_static.h", "epi_static_programs.h", "dft_kept.h"]


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs)


def build() -> str:
    """Up-to-date objects are kept; the whole build runs under an exclusive file lock (the CPU suite runs on several pytest-xdist
    workers, tests/conftest.py: the first one builds, the others wait and then find everything up to date)."""
    import fcntl

    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked() -> str:
    lib = os.path.join(OUT, "libppsci_emu.so")
    deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, "include", "ppsci_hip.h"),
                                                       os.path.join(ROOT, "tests", "emu", "hip_emu.h")]
    flags = EXTRA + ["-x", "c++", "-DPPSCI_EMU", "-DPPSCI_NUM_CU=4", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
             "-I", os.path.join(ROOT, "tests", "emu")]
    opt = os.environ.get("PPSCI_EMU_OPT")

    def one(src):
        obj = os.path.join(OUT, src.replace(".hip", ".o"))
        s = os.path.join(CSRC, src)
        if not _newer(obj, [s] + deps):
            subprocess.check_call([CLANG] + flags + [opt or ("-O1" if src in HOT else "-O0"), "-c", s, "-o", obj])
        return obj

    with ThreadPoolExecutor(min(len(SOURCES), os.cpu_count() or 8)) as ex:
        objs = list(ex.map(one, sorted(SOURCES, key=lambda f: f not in HOT)))  # the slow ones first
    if not _newer(lib, objs):
        subprocess.check_call([CLANG, "-shared", "-o", lib] + objs)
    return lib


def inject():
    from paddlescience_amd import _lib

    _lib._inject_for_tests(build())
