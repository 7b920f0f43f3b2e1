"""Build recipes for the native parts (hipcc for gfx950, gcc for the host C code).

Everything is built in-tree so that the shared objects travel to the GPU box with the
repository snapshot; nothing is installed into site-packages.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")

LIB_GPU = os.path.join(HERE, "librb3gpu.so")
LIB_GPU_HOOKS = os.path.join(HERE, "librb3gpu_hooks.so")
LIB_HOST = os.path.join(HERE, "librb3host.so")
BIN_CLI = os.path.join(HERE, "ropebwt3-amd")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.exists(s))


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def hipcc_path():
    for p in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if p and os.path.exists(p):
            return p
    raise RuntimeError("hipcc not found; the HIP engine cannot be built (there is no CPU fallback)")


GPU_UNITS = ("rb3gpu.hip", "rb3gpu_sort.hip", "rb3gpu_fmdenc.hip", "rb3gpu_comm.hip")


def build_gpu(force=False):
    """librb3gpu.so: the HIP engine + C ABI (include/rb3gpu.h), gfx950 only -- and librb3gpu_hooks.so, the same sources
    with -DRB3GPU_TEST_HOOKS (the test hooks of rb3gpu_tune are compiled out of the release library).  Translation units
    are compiled in parallel into build/ and linked; force=True recompiles everything (what the driver's build() does)."""
    from concurrent.futures import ThreadPoolExecutor
    units = [u for u in GPU_UNITS if os.path.exists(os.path.join(CSRC, u))]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "rb3gpu.h")]
    odir = os.path.join(HERE, "build")
    os.makedirs(odir, exist_ok=True)
    hipcc = hipcc_path()
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC]
    jobs = []
    for u in units:
        src = os.path.join(CSRC, u)
        variants = [("", [])]
        if u == "rb3gpu.hip":
            variants.append(("_hooks", ["-DRB3GPU_TEST_HOOKS"]))
        for tag, defs in variants:
            obj = os.path.join(odir, u.replace(".hip", tag + ".o"))
            if force or not _newer(obj, [src] + hdrs):
                jobs.append(base + defs + ["-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(_run, jobs))
    for lib, tag in ((LIB_GPU, ""), (LIB_GPU_HOOKS, "_hooks")):
        objs = [os.path.join(odir, u.replace(".hip", (tag if u == "rb3gpu.hip" else "") + ".o")) for u in units]
        if force or jobs or not _newer(lib, objs):
            _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return LIB_GPU


def build_host(force=False):
    """librb3host.so + the `ropebwt3-amd` CLI: host-side C (I/O, suffix sorting, FMD/FMR)."""
    hdir = os.path.join(CSRC, "host")
    if not os.path.isdir(hdir):
        return None
    csrcs = sorted(os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".c"))
    hdrs = sorted(os.path.join(hdir, f) for f in os.listdir(hdir) if f.endswith(".h")) + [os.path.join(INCLUDE, "rb3gpu.h")]
    lib_srcs = [s for s in csrcs if not s.endswith("main.c")]
    flags = ["-O3", "-g", "-Wall", "-fopenmp", "-fPIC", "-I" + INCLUDE, "-I" + hdir]
    if force or not _newer(LIB_HOST, lib_srcs + hdrs):
        _run(["gcc"] + flags + ["-shared", "-o", LIB_HOST] + lib_srcs + ["-lz", "-lm", "-lpthread"])
    main_c = os.path.join(hdir, "main.c")
    if os.path.exists(main_c) and (force or not _newer(BIN_CLI, csrcs + hdrs + [LIB_GPU])):
        build_gpu()
        _run(["gcc"] + flags + ["-o", BIN_CLI] + csrcs +
             ["-L" + HERE, "-lrb3gpu", "-Wl,-rpath,$ORIGIN", "-lz", "-lm", "-lpthread", "-ldl"])
    return LIB_HOST


def build_oracle(force=False):
    """oracle/liboracle.so (+ oracle/_ref when /root/reference is present): test infrastructure."""
    odir = os.path.join(ROOT, "oracle")
    _run(["make", "-C", odir, "-j8"] + (["-B"] if force else []))
    return os.path.join(odir, "liboracle.so")


def build_all(force=False):
    build_gpu(force)
    build_host(force)
    build_oracle(force)
