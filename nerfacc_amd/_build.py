"""Build libnerfacc_hip.so (hand-written HIP, gfx950) in-tree with hipcc.

No torch.utils.cpp_extension, no pybind: the library is a plain C-ABI shared
object (include/nerfacc_hip.h) loaded with ctypes, so it compiles in seconds,
cross-compiles without a GPU, and travels to the GPU box as a file.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB_PATH = os.path.join(_HERE, "libnerfacc_hip.so")
SOURCES = ["grid.hip", "walk.hip", "gridupd.hip", "traverse2.hip", "segscan.hip", "pdf.hip"]
ARCH = os.environ.get("NERFACC_AMD_ARCH", "gfx950")
# -ffp-contract=off: the traversal must not fuse a*b+c (floating-point contract)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", f"--offload-arch={ARCH}",
         "-Wall", "-Wno-unused-function"] + os.environ.get("NERFACC_AMD_EXTRA_FLAGS", "").split()  # tuning experiments


def hipcc() -> str | None:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _deps() -> list[str]:
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in ("common.hip.h", "march.h", "walk_layout.h")]
    hdr = os.path.join(INCLUDE, "nerfacc_hip.h")
    if os.path.exists(hdr):
        deps.append(hdr)
    return deps


HASH_PATH = LIB_PATH + ".hash"


def _source_hash() -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for d in sorted(_deps()):
        if os.path.exists(d):
            h.update(os.path.basename(d).encode())
            h.update(open(d, "rb").read())
    return h.hexdigest()


def is_stale() -> bool:
    """The library is current iff it was built from exactly these sources and flags (content hash,
    not mtimes: the tree is copied to the GPU box and mtimes do not survive reliably)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    return open(HASH_PATH).read().strip() != _source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile (if stale) and return the path of libnerfacc_hip.so.  Safe to call from several
    processes at once (one rank per GPU): an exclusive lock serialises the build."""
    if not force and not is_stale():
        return LIB_PATH
    cc = hipcc()
    if cc is None:
        raise RuntimeError("nerfacc_amd: hipcc not found; cannot build libnerfacc_hip.so")
    obj_dir = os.path.join(CSRC, "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    with open(os.path.join(obj_dir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():  # another process built it while we waited
                return LIB_PATH
            tag = str(os.getpid())

            def compile_one(src: str) -> str:
                obj = os.path.join(obj_dir, f"{src}.{tag}.o")
                cmd = [cc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                subprocess.run(cmd, check=True)
                return obj

            with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
                objs = list(ex.map(compile_one, SOURCES))
            tmp = f"{LIB_PATH}.{tag}.tmp"
            subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", tmp], check=True)
            os.replace(tmp, LIB_PATH)
            with open(HASH_PATH + "." + tag, "w") as f:
                f.write(_source_hash())
            os.replace(HASH_PATH + "." + tag, HASH_PATH)
            for o in objs:
                os.remove(o)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
