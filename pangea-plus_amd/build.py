"""Build libpangea_hip.so (hipcc, gfx950 only) and the thin CLIs, in-tree.

    python pangea-plus_amd/build.py            # library + CLIs
The .so lands in pangea-plus_amd/lib/, the CLIs in pangea-plus_amd/bin/ (both git-ignored,
both travel to the GPU box with the gpurun snapshot).
"""
import concurrent.futures
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
CLI = os.path.join(HERE, "cli")
LIBDIR = os.path.join(HERE, "lib")
BINDIR = os.path.join(HERE, "bin")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libpangea_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]
BASE_FLAGS = list(FLAGS)
if os.environ.get("PGX_STAGE_PROBES"):  # measurement builds only: kernels that can be truncated after a stage
    FLAGS.append("-DPGX_STAGE_PROBES")
FLAGS += os.environ.get("PGX_EXTRA_FLAGS", "").split()  # experiments (-D switches of the kernels under study)


def hipcc():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: libpangea_hip is HIP-only and cannot be built without ROCm")
    return p


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src,) + tuple(extra))


def _compile(src):
    # (objects of a build with other -D switches -- measurement builds -- get their own names, so neither build is stale)
    tag = "" if FLAGS == BASE_FLAGS else "." + hashlib.md5(" ".join(FLAGS).encode()).hexdigest()[:8]
    obj = os.path.join(OBJDIR, os.path.basename(src) + tag + ".o")
    headers = tuple(glob.glob(os.path.join(CSRC, "*.hpp"))) + (os.path.join(HERE, "..", "include", "pangea_hip.h"),)
    if _newer(src, obj, headers):
        cmd = [hipcc()] + FLAGS + ["-x", "hip", "-c", src, "-o", obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stdout))
    return obj


def _last_link():
    try:
        return open(os.path.join(OBJDIR, "linked.txt")).read().split("\n")
    except OSError:
        return None


def build_library(verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if any(_newer(o, LIB) for o in objs) or _last_link() != objs:
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
        with open(os.path.join(OBJDIR, "linked.txt"), "w") as f:
            f.write("\n".join(objs))
    if verbose:
        print("built", LIB)
    return LIB


def build_clis(verbose=False):
    os.makedirs(BINDIR, exist_ok=True)
    out = []
    for src in sorted(glob.glob(os.path.join(CLI, "*.cpp"))):
        name = os.path.splitext(os.path.basename(src))[0]
        exe = os.path.join(BINDIR, name)
        if _newer(src, exe, (LIB,)):
            cmd = [hipcc(), "-O2", "-std=c++17", "-I", os.path.join(HERE, "..", "include"), src, "-o", exe,
                   "-L", LIBDIR, "-lpangea_hip", "-Wl,-rpath,$ORIGIN/../lib"]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError("CLI build failed on %s:\n%s" % (src, r.stdout))
        out.append(exe)
    if verbose:
        print("built", len(out), "CLIs in", BINDIR)
    return out


def build_all(verbose=False):
    build_library(verbose)
    build_clis(verbose)


if __name__ == "__main__":
    build_all(verbose=True)
