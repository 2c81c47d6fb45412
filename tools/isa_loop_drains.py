#!/usr/bin/env python
"""Loops that drain the memory queue while they have stores in flight: for every kernel of a .hip file, every loop (backward
branch) that holds global loads, global stores AND an `s_waitcnt vmcnt(0)`.  On gfx9 stores count in vmcnt like loads, so a
full drain inside a software-pipelined loop also waits for the write acknowledgements of the trip's own stores (~2 us); the
usual cause is a store behind a divergent condition (the wait-count pass must be right on the path that skips it).
Usage: python tools/isa_loop_drains.py ao_amd/csrc/gva_bwd_point.hip [name filter] [--all]   (--all: enclosing loops too)"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-fno-gpu-rdc", "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "--cuda-device-only", "-S"]


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0][-90:]


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    asm = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", "-"], capture_output=True, text=True).stdout
    cur, body, kernels = None, [], []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*; @", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur and line.startswith(".Lfunc_end"):
            kernels.append((cur, body))
            cur = None
            continue
        if cur is not None:
            t = line.strip().split(";")[0].rstrip()
            if t and not (t.startswith(".") and not t.startswith(".LBB")):
                body.append(t)
    for name, body in kernels:
        dn = demangle(name)
        if flt and flt not in dn:
            continue
        labels = {m.group(1): n for n, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        for n, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if not (m and m.group(1) in labels and labels[m.group(1)] < n):
                continue
            blk = body[labels[m.group(1)]:n]
            # innermost loops only: a loop that contains another backward branch is reported through that one
            inner = False
            for k, x in enumerate(blk):
                mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", x)
                if mm and mm.group(1) in labels and labels[m.group(1)] < labels[mm.group(1)] <= labels[m.group(1)] + k:
                    inner = True
                    break
            if inner and "--all" not in sys.argv:
                continue
            loads = sum(1 for x in blk if re.match(r"(global|buffer|flat)_load", x))
            stores = sum(1 for x in blk if re.match(r"(global|buffer|flat)_(store|atomic)", x))
            drains = [i for i, x in enumerate(blk) if re.match(r"s_waitcnt vmcnt\(0\)", x)]
            if loads and stores and drains:
                print("%-92s loop of %4d instr: %2d loads %2d stores, vmcnt(0) at %s" % (dn, len(blk), loads, stores, drains[:6]))


if __name__ == "__main__":
    main()
