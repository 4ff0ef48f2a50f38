#!/usr/bin/env python3
"""Lists the kernels of the built library whose ISA holds chains of `load -> s_waitcnt vmcnt(0)` pairs — a global (or scratch) load whose
result is waited for before the next load is issued.  A guarded `cond ? p[i] : 0` or `if (..) p[i] -= v` per element compiles into exactly
that, and in a latency-bound kernel every pair is one memory round trip (round 4: K10's update / backward step / panel, 14.1 -> 11.1 ms per
Floor solve after their reads were batched).  Needs the object files of `python -m panovlm_amd.build` and the ROCm LLVM tools.
usage: python tools/isa_serial_loads.py [min_pairs]   ->  (longest chain, pairs, loads, object, kernel) per line, worst first"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    min_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for obj in sorted(glob.glob(os.path.join(ROOT, "panovlm_amd", "build", "pvlm_*.hip.o"))):
            tag = os.path.basename(obj)[5:-6]
            fat, co = os.path.join(d, tag + ".fatbin"), os.path.join(d, tag + ".co")
            subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                "--input=" + fat, "--output=" + co], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co):
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
            parts = re.split(r"\n[0-9a-f]+ <([^>]+)>:\n", txt)
            for k in range(1, len(parts), 2):
                name, body, ops = parts[k], parts[k + 1], []
                for line in body.split("\n"):
                    m = re.match(r"\s+(\S+)", line)
                    if not m:
                        continue
                    op = m.group(1)
                    if op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load")):
                        ops.append("L")
                    elif op.startswith("s_waitcnt") and "vmcnt(0)" in line:
                        ops.append("W")
                seq = "".join(ops)
                pairs = len(re.findall(r"(?<!L)LW", seq))
                longest = max((len(m.group(0)) // 2 for m in re.finditer(r"(?:LW){2,}", seq)), default=0)
                if pairs >= min_pairs:
                    rows.append((longest, pairs, seq.count("L"), tag, name[:90]))
    for r in sorted(rows, reverse=True):
        print("%3d in a row  %4d pairs  %4d loads  %-7s %s" % r)


if __name__ == "__main__":
    main()
