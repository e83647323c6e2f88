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
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + [os.path.join(os.path.dirname(PKG), "include", "smos.h")]
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


SHIMS = (("point_deep_cuda_kernel", "SMOS_SHIM_POINT_DEEP"), ("MultiScaleDeformableAttention", "SMOS_SHIM_MSDA"))


def shim_path(name):
    return os.path.join(LIB_DIR, "pybind", name + ".so")


def build_pybind_shims(force=False, verbose=False):
    """g++ build of the two pybind11 modules a reference maintainer would keep (csrc/shim/pybind_shims.cpp, INTEGRATION.md
    section 3): the reference's function names and argument lists over libsmos_hip.so's C ABI.  Host code only; torch
    supplies headers and the current HIP stream.  Optional: the shipped Python path binds the C ABI through ctypes."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension
    build()
    src = os.path.join(CSRC, "shim", "pybind_shims.cpp")
    deps = [src, os.path.join(os.path.dirname(PKG), "include", "smos.h")]
    os.makedirs(os.path.join(LIB_DIR, "pybind"), exist_ok=True)
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    out = []
    procs = []
    for name, macro in SHIMS:
        target = shim_path(name)
        out.append(target)
        if not (force or _stale(target, deps)):
            continue
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-deprecated-declarations",
               "-D" + macro, "-DTORCH_EXTENSION_NAME=" + name, "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
               "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
               "-I" + os.path.join(os.path.dirname(PKG), "include"), "-I" + sysconfig.get_paths()["include"]]
        cmd += ["-I" + p for p in cpp_extension.include_paths("cuda")]
        cmd += [src, "-o", target, "-L" + torch_lib, "-L" + LIB_DIR, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip",
                "-ltorch", "-ltorch_python", "-lsmos_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + torch_lib]
        if verbose:
            print(" ".join(cmd))
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for name, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("pybind shim %s: g++ failed:\n%s" % (name, log.decode()))
    return out


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
    print(build_cpu(force="-f" in sys.argv, verbose=True))
    if "--shims" in sys.argv:
        print(build_pybind_shims(force="-f" in sys.argv, verbose=True))
