"""Build libsfk.so (the C-ABI kernel library of include/sfk.h) with hipcc for gfx950, in-tree."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsfk.so")
SOURCES = ["conv_igemm.hip", "conv_igemm_p8.hip", "conv_halo.hip", "conv_pw.hip", "conv_wgrad.hip", "conv_wgrad_p8.hip", "conv_wgrad_band.hip", "stem_conv.hip", "bn.hip", "bn_tail.hip", "pool_head.hip",
           "optim_misc.hip", "eval_input.hip"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(CSRC, "sfk_common.h"), os.path.join(CSRC, "conv_igemm_epi.h"), os.path.join(CSRC, "conv_wgrad_common.h"), os.path.join(CSRC, "conv_wgrad_band_acc.inc"), os.path.join(ROOT, "include", "sfk.h")]
    flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    flags += os.environ.get("SFK_EXTRA_FLAGS", "").split()          # experiment builds (tools/), with SFK_LIB_OUT

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            return o, True
        return o, False

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        res = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(LIB):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose=True))
