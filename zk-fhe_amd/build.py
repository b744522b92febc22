"""Builds libzkfhe_hip.so (HIP kernels + C ABI, gfx950 only) in-tree with plain hipcc.

    python zk-fhe_amd/build.py [--force] [--jobs N]

Objects go to zk-fhe_amd/_build/, the library to zk-fhe_amd/libzkfhe_hip.so (git-ignored, but it
travels to the GPU box with the snapshot).  hipcc cross-compiles without a GPU.
"""
import argparse
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libzkfhe_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=default", "-Wno-unused-value",
         "-mllvm", "-pragma-unroll-threshold=1000000", "-I", os.path.join(HERE, "..", "include")]
FLAGS += os.environ.get("ZKFHE_EXTRA_FLAGS", "").split()   # A/B builds of experiments (e.g. -DZK_MAD_C): build a copy of this directory with it
TILE_SIZES = list(range(3, 13))   # the 2^13 tile is csrc/ntt13.hip


def hipcc():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return p


def units():
    u = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") and f != "ntt_tile_inst.hip":
            u.append((os.path.join(CSRC, f), os.path.join(OUT, f[:-4] + ".o"), []))
    host = os.path.join(HERE, "host")
    for f in sorted(os.listdir(host)):
        if f.endswith((".cpp", ".hip")) and not f.endswith("_main.cpp"):
            u.append((os.path.join(host, f), os.path.join(OUT, "host_" + f.rsplit(".", 1)[0] + ".o"), []))
    for k in TILE_SIZES:
        u.append((os.path.join(CSRC, "ntt_tile_inst.hip"), os.path.join(OUT, "ntt_tile_%d.o" % k), ["-DZK_TILE_LOGN=%d" % k]))
    return u


def newest_header(host):
    """csrc/ units depend on csrc/ + include/ headers only; host/ units on every header"""
    t = 0.0
    dirs = (CSRC, os.path.join(HERE, "..", "include")) + ((os.path.join(HERE, "host"),) if host else ())
    for d in dirs:
        for f in os.listdir(d):
            if f.endswith((".hpp", ".h", ".inc")):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def compile_one(src, obj, extra):
    if src.endswith(".cpp"):
        # host-only translation units (C ABI glue, verifier, the AVX-512 Poseidon core): plain C++, no device pass
        cmd = [hipcc(), "-x", "c++"] + [f for f in FLAGS if not f.startswith("--offload-arch")] + extra + ["-c", src, "-o", obj]
    else:
        cmd = [hipcc()] + FLAGS + extra + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (os.path.basename(obj), r.stdout[-4000:]))
    return obj


def build(force=False, jobs=None, verbose=True):
    os.makedirs(OUT, exist_ok=True)
    hdr_csrc, hdr_host = newest_header(False), newest_header(True)
    todo = []
    objs = []
    for src, obj, extra in units():
        objs.append(obj)
        hdr_t = hdr_host if os.path.basename(obj).startswith("host_") else hdr_csrc
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append((src, obj, extra))
    if todo:
        jobs = jobs or max(1, min(len(todo), (os.cpu_count() or 4)))
        if verbose:
            print("[zkfhe build] compiling %d unit(s) for %s with %d job(s)" % (len(todo), ARCH, jobs), flush=True)
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            for f in [ex.submit(compile_one, *t) for t in todo]:
                f.result()
    if todo or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout[-4000:])
        if verbose:
            print("[zkfhe build] linked", LIB, flush=True)
    return LIB


def build_cli(verbose=True):
    """the `bfv` command-line driver (zk-fhe_amd/bfv), linked against libzkfhe_hip.so with an $ORIGIN rpath"""
    build(verbose=verbose)
    exe = os.path.join(HERE, "bfv")
    src = os.path.join(HERE, "host", "bfv_main.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(LIB)):
        cmd = ["g++", "-O2", "-std=c++17", src, "-o", exe, "-L" + HERE, "-lzkfhe_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("bfv CLI link failed:\n" + r.stdout[-4000:])
        if verbose:
            print("[zkfhe build] built", exe, flush=True)
    return exe


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--cli", action="store_true", help="also build the bfv command-line driver")
    a = ap.parse_args()
    try:
        build(a.force, a.jobs)
        if a.cli:
            build_cli()
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
