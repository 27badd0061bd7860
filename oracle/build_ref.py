#!/usr/bin/env python3
"""Build the reference's Fortran computational core into oracle/_ref/libfast_ref.so.

TEST INFRASTRUCTURE ONLY.  Nothing under fast_amd/ may load the result; it is the
checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

The sources are compiled where they lie under /root/reference (never copied into
this repository): each `.for` file named in the reference's own source lists
(Fast/FastC/srcs.py and Fast/FastS/srcs.py, read here as *text*, not executed) is
run through the external C preprocessor with the macro set of the reference's
build (Fast/FastS/setup.scons:83-98) and compiled with AMD flang.  Objects go to a
scratch directory outside the repository; only the linked shared library lands in
oracle/_ref/ (git-ignored, but shipped to the GPU box with the tree).

No stand-in is written for anything: `omp_lib.h` comes from flang's own include
directory, and the one symbol no compiled source defines (`ldurelaxscal_`, used by
the GMRES path which is outside the scope of this build) is left undefined in the
shared object, which is legal for a .so as long as the routine is never called.
To keep the loader from resolving it eagerly the library is linked `-z lazy` and
must be opened with RTLD_LAZY (see oracle/ref_lib.py).

Usage:  python oracle/build_ref.py [--jobs N] [--variant generic|v3] [--force]
"""
import argparse
import concurrent.futures as cf
import hashlib
import os
import re
import subprocess
import sys

REF = "/root/reference/Fast"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
FLANG = "/opt/rocm/bin/amdflang"
FLANG_INC = "/opt/rocm/lib/llvm/include/flang"

CPP_DEFS = [
    "-DE_SCALAR_COMPUTER", "-D_OPENMP4", "-D_OPENMP", "-DE_OMP_SOUS_DOMAIN",
    "-DCACHELINE=64", "-DNB_SOCKET=1", "-DCORE_PER_SOCK=8",
    "-DINTEGER_E=INTEGER*4", "-DREAL_E=REAL*8",
]


def source_list(module):
    """Fortran sources of one module, parsed from the text of its srcs.py."""
    txt = open(os.path.join(REF, module, "srcs.py")).read()
    # drop commented-out entries, then take every quoted *.for path
    lines = [ln.split("#", 1)[0] for ln in txt.splitlines()]
    body = "\n".join(lines)
    m = re.search(r"for_srcs\s*=\s*\[", body)
    if not m:
        return []
    paths = re.findall(r"['\"]([^'\"]+\.for)['\"]", body[m.end():])
    seen, out = set(), []
    for p in paths:
        if p not in seen:
            seen.add(p)
            out.append(p)
    return out


def compile_one(args):
    module, rel, objdir, fflags = args
    src = os.path.join(REF, module, rel)
    tag = hashlib.md5((module + "/" + rel).encode()).hexdigest()[:10]
    base = os.path.join(objdir, os.path.basename(rel)[:-4] + "_" + tag)
    fsrc, obj = base + ".f", base + ".o"
    if os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src):
        return obj, None
    incs = ["-I" + os.path.join(REF, "FastS"), "-I" + os.path.join(REF, "FastC"),
            "-I" + os.path.join(REF, module), "-I" + REF]
    r = subprocess.run(["cpp", "-P", "-traditional-cpp"] + CPP_DEFS + incs + [src],
                       capture_output=True, text=True)
    if r.returncode != 0:
        return None, "cpp failed for %s:\n%s" % (src, r.stderr[-2000:])
    with open(fsrc, "w") as f:
        f.write(r.stdout)
    r = subprocess.run([FLANG, "-c", "-fPIC", "-ffixed-form", "-fopenmp",
                        "-fdefault-real-8", "-fdefault-double-8",
                        "-I" + FLANG_INC] + fflags + ["-o", obj, fsrc],
                       capture_output=True, text=True)
    if r.returncode != 0:
        return None, "flang failed for %s:\n%s" % (src, r.stderr[-2000:])
    return obj, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--variant", default="generic", choices=["generic", "v3"],
                    help="generic: baseline x86-64, no FMA contraction (parity oracle); "
                         "v3: -march=x86-64-v3 -O3 (AVX2+FMA, CPU timing baseline only)")
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()

    if not os.path.isdir(REF):
        print("reference tree not present; keeping any prebuilt oracle/_ref library")
        return 0
    if a.variant == "generic":
        fflags = ["-O2", "-ffp-contract=off"]
        libname = "libfast_ref.so"
    else:
        fflags = ["-O3", "-march=x86-64-v3"]
        libname = "libfast_ref_v3.so"
    out = os.path.join(OUT_DIR, libname)
    if os.path.exists(out) and not a.force:
        print("up to date:", out)
        return 0
    objdir = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fast_ref_obj_" + a.variant)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(OUT_DIR, exist_ok=True)

    jobs = []
    for module in ("FastC", "FastS"):
        for rel in source_list(module):
            jobs.append((module, rel, objdir, fflags))
    print("compiling %d Fortran sources with %d jobs -> %s" % (len(jobs), a.jobs, objdir))
    objs, errs = [], []
    with cf.ThreadPoolExecutor(max_workers=a.jobs) as ex:
        for i, (obj, err) in enumerate(ex.map(compile_one, jobs)):
            if err:
                errs.append(err)
            else:
                objs.append(obj)
            if (i + 1) % 100 == 0:
                print("  %d/%d" % (i + 1, len(jobs)), flush=True)
    if errs:
        print("\n".join(errs[:5]))
        print("%d sources failed" % len(errs))
        return 1
    rsp = os.path.join(objdir, "objs.rsp")
    with open(rsp, "w") as f:
        f.write("\n".join(objs))
    r = subprocess.run([FLANG, "-shared", "-fopenmp", "-Wl,-z,lazy", "-o", out, "@" + rsp],
                       capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr[-4000:])
        return 1
    print("built", out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
