"""Builds libsmos_hip.so (gfx950) in-tree with hipcc.  ``python -m streammos_amd.build [-f]``."""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsmos_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-ffp-contract=on", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every csrc/*.hip to an object (parallel) and link them into lib/libsmos_hip.so."""
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(PKG), "include", "smos.h")]
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (src, out.decode()))
        elif verbose and out:
            print(out.decode())
    if failed:
        raise RuntimeError("libsmos_hip.so: compilation failed")
    if force or procs or _stale(LIB_PATH, objs):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_PATH


CPU_LIB_PATH = os.path.join(LIB_DIR, "libsmos_cpu.so")


def build_cpu(force=False, verbose=False):
    """g++ build of the host twin (csrc/cpu/*.cpp -> lib/libsmos_cpu.so); no HIP involved."""
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "cpu", "*.cpp")))
    deps = srcs + [os.path.join(os.path.dirname(PKG), "include", "smos_cpu.h")]
    if force or _stale(CPU_LIB_PATH, deps):
        cmd = [os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", CPU_LIB_PATH] + srcs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CPU_LIB_PATH


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
    print(build_cpu(force="-f" in sys.argv, verbose=True))
