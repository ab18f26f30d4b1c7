"""ORACLE build recipe -- test infrastructure only.

* `build_c()` compiles oracle/roi_ops_ref.c + oracle/det_ops_ref.c (our plain-C restatements) with gcc into
  oracle/_build/liboracle_roi.so.
* `build_ref()` compiles the REFERENCE's own CPU RoIPool from its sources where they lie
  under /root/reference (never copied) plus our C-ABI shim into oracle/_ref/libref_roi_pool.so.
  It needs /root/reference and the local torch headers; on the GPU box (no /root/reference)
  the prebuilt file that travelled with the snapshot is used.
Both output directories are git-ignored.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
C_LIB = os.path.join(HERE, "_build", "liboracle_roi.so")
REF_LIB = os.path.join(HERE, "_ref", "libref_roi_pool.so")


def _stale(out, srcs):
    return not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)


def build_c(force=False):
    srcs = [os.path.join(HERE, "roi_ops_ref.c"), os.path.join(HERE, "det_ops_ref.c")]
    os.makedirs(os.path.dirname(C_LIB), exist_ok=True)
    if force or _stale(C_LIB, srcs):
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", C_LIB] + srcs + ["-lm"],
                       check=True)
    return C_LIB


def build_ref(force=False):
    """Returns the library path, or None if neither the reference nor a prebuilt file is here."""
    ref_src = os.path.join(REFERENCE, "wsovod", "layers", "ROILoopPool", "ROILoopPool_cpu.cpp")
    if not os.path.exists(ref_src):
        return REF_LIB if os.path.exists(REF_LIB) else None
    shim = os.path.join(HERE, "ref_roi_pool_bind.cpp")
    os.makedirs(os.path.dirname(REF_LIB), exist_ok=True)
    if force or _stale(REF_LIB, [ref_src, shim]):
        import torch
        from torch.utils import cpp_extension

        tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
        cmd = ["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
               "-D_GLIBCXX_USE_CXX11_ABI=" + str(int(torch._C._GLIBCXX_USE_CXX11_ABI))]
        for inc in cpp_extension.include_paths():
            cmd += ["-isystem", inc]
        cmd += ["-I", os.path.join(REFERENCE, "wsovod", "layers"), ref_src, shim, "-o", REF_LIB,
                "-L", tlib, "-ltorch_cpu", "-lc10", "-ltorch", f"-Wl,-rpath,{tlib}"]
        subprocess.run(cmd, check=True)
    return REF_LIB


if __name__ == "__main__":
    print(build_c(force="--force" in sys.argv))
    print(build_ref(force="--force" in sys.argv))
