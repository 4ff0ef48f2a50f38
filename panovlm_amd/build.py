"""Builds libpvlm.so (HIP kernels + C ABI, gfx950) in-tree with hipcc.  No JIT cache: the .so sits
next to this file so that it travels to the GPU box with the repo snapshot."""
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpvlm.so")
ARCH = "gfx950"
# sources whose float / double decisions must equal a non-FMA x86-64 build of the reference bit for bit
NO_CONTRACT = ("pvlm_assoc.hip", "pvlm_lines.hip", "pvlm_linegrow.hip", "pvlm_mvs.hip", "pvlm_ring.hip", "pvlm_undistort.hip")


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "pvlm.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    common = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
    procs = []
    for src in sources():
        obj = os.path.join(bdir, os.path.basename(src) + ".o")
        flags = list(common) + os.environ.get("PVLM_DEFINES", "").split()
        # association kernels make accept/reject decisions that must be bit-identical to a
        # non-FMA x86-64 build of the reference: no contraction there.
        if os.path.basename(src) in NO_CONTRACT:
            flags.append("-ffp-contract=off")
        # the candidate loop of the k-NN search: SLP packs two of its three float subtractions / multiplications into v_pk ops and pays
        # three register moves to assemble the operands — measured 2 % slower than the scalar form (23.6 vs 23.15 ms per 134 M queries)
        if os.path.basename(src) == "pvlm_assoc.hip":
            flags.append("-fno-slp-vectorize")
        cmd = [_hipcc()] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("hipcc failed on " + src)
        if verbose and out.strip():
            print(out)
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-pthread"]
    subprocess.check_call(cmd)
    return LIB


def build_variant(tag, defines):
    """A measured variant of the library: build/var/libpvlm_<tag>.so with extra -D switches (e.g. ["-DPVLM_NT_LOADS=0"]), selected at
    run time through PVLM_LIB (tools/ab_eval_loads.sh).  A source is recompiled when it, or any header under csrc/, mentions one of
    the switches (a switch consumed in a header reaches every source that may include it); the other objects are those of the
    in-tree build.  Same flags as the in-tree build (PVLM_DEFINES included).  A switch nobody mentions is an error: the variant
    would silently equal the baseline.  python -m panovlm_amd.build --variant temporal -DPVLM_NT_LOADS=0"""
    build(force=False)
    vdir = os.path.join(HERE, "..", "build", "var")
    os.makedirs(vdir, exist_ok=True)
    names = [d[2:].split("=")[0] for d in defines if d.startswith("-D")]
    headers = {f: open(os.path.join(CSRC, f)).read() for f in os.listdir(CSRC) if f.endswith(".h")}
    in_header = [n for n in names if any(n in t for t in headers.values())]
    mentioned = set(in_header)
    objs = []
    for src in sources():
        base = os.path.basename(src)
        obj = os.path.join(HERE, "build", base + ".o")
        text = open(src).read()
        mentioned.update(n for n in names if n in text)
        if any(n in text for n in names) or in_header:
            obj = os.path.join(vdir, base + "." + tag + ".o")
            flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
            flags += os.environ.get("PVLM_DEFINES", "").split() + list(defines)
            if base in NO_CONTRACT:
                flags.append("-ffp-contract=off")
            if base == "pvlm_assoc.hip":
                flags.append("-fno-slp-vectorize")
            subprocess.check_call([_hipcc()] + flags + ["-c", src, "-o", obj])
        objs.append(obj)
    missing = [n for n in names if n not in mentioned]
    if missing:
        raise RuntimeError("no source or header under csrc/ mentions %s: the variant would equal the baseline" % ", ".join(missing))
    out = os.path.join(vdir, "libpvlm_%s.so" % tag)
    subprocess.check_call([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-pthread"])
    return out


HOST_LIB = os.path.join(HERE, "libpvlm_host.so")
HOST_DRIVER = os.path.join(HERE, "build", "pvlm_host_driver")


def build_host(force=False):
    """libpvlm_host.so (C++ mirror of PanoVLM's interfaces above the C ABI) and its test driver."""
    build(force=False)
    hdir = os.path.join(HERE, "host")
    # one translation unit per mirrored reference file (pvlm_host_*.cpp), the feature extractor (pvlm_features.cpp, pvlm_lines.cpp) and the
    # core (pvlm_host.cpp); -ffp-contract=off everywhere: the float threshold decisions of the extractor and the double chains of the
    # solver must not pick up FMAs the oracle does not have
    srcs = sorted(glob.glob(os.path.join(hdir, "*.cpp")))
    hdrs = sorted(glob.glob(os.path.join(hdir, "*.hpp"))) + [os.path.join(HERE, "csrc", "pvlm_workers.h"), os.path.join(HERE, "..", "include", "pvlm.h")]
    drv = os.path.join(HERE, "..", "tests", "cpp", "pvlm_host_driver.cpp")
    odir = os.path.join(HERE, "build", "host")
    os.makedirs(odir, exist_ok=True)
    newest_hdr = max(os.path.getmtime(x) for x in hdrs)
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(odir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_hdr):
            jobs.append(["g++", "-std=c++17", "-O2", "-fPIC", "-Wall", "-ffp-contract=off", "-c", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            list(pool.map(subprocess.check_call, jobs))
    if jobs or not os.path.exists(HOST_LIB) or os.path.getmtime(HOST_LIB) < os.path.getmtime(LIB):
        subprocess.check_call(["g++", "-shared", "-o", HOST_LIB] + objs + ["-L" + HERE, "-lpvlm", "-pthread", "-Wl,-rpath,$ORIGIN"])
    adapter = os.path.join(HERE, "..", "integration", "pvlm_ceres.hpp")
    if os.path.exists(drv) and (force or not os.path.exists(HOST_DRIVER) or
                                os.path.getmtime(HOST_DRIVER) < max(os.path.getmtime(drv), os.path.getmtime(HOST_LIB), os.path.getmtime(adapter))):
        root = os.path.join(HERE, "..")
        # integration/pvlm_ceres.hpp is compiled into the driver against tests/cpp/ceres_double: an interface-only stand-in
        # for the handful of Ceres classes the adapter touches (this image has no Ceres)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "tests", "cpp", "ceres_double"),
                               drv, "-o", HOST_DRIVER, "-L" + HERE, "-lpvlm_host", "-lpvlm", "-Wl,-rpath," + HERE])
    return HOST_LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:
        k = sys.argv.index("--variant")
        print(build_variant(sys.argv[k + 1], [a for a in sys.argv[k + 2:] if a.startswith("-D")]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
        print(build_host(force="--force" in sys.argv))
