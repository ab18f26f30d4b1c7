"""Build the C-ABI shared library `libwsovod_hip.so` for gfx950 with hipcc.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU, so this runs in the authoring
container (the "does it build" check) as well as on the MI355X box.  Objects are cached by
source mtime under `wsovod_amd/csrc/build/`; the linked library lives in-tree at
`wsovod_amd/lib/libwsovod_hip.so` so that it travels with the snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libwsovod_hip.so")
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build(force=False, verbose=False):
    """Compile every .hip source for gfx950 and link libwsovod_hip.so. Returns its path."""
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "wsovod_hip.h"))
    hdr_mtime = max(os.path.getmtime(h) for h in headers)
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if (
            not force
            and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_mtime)
        ):
            continue
        cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-Wall",
               "-Wno-unused-function", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append((src, out.decode(errors="replace")))
        elif verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(f"--- {s}\n{o}" for s, o in failed))
    need_link = force or procs or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs
    )
    if need_link:
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
