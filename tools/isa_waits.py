"""Static check of the compiler's memory waits: for every kernel of a .hip file, how many of its global loads are
followed by a full drain of the memory queue (`s_waitcnt vmcnt(0)`) within a few instructions -- the signature of a load that
sits in a basic block of its own (a `cond ? *p : 0` the compiler could not speculate) or that was scheduled next to its use.
Each such pair is one exposed memory round trip per loop trip.  Usage: python tools/isa_waits.py ao_amd/csrc/gva_bwd_point.hip [name filter]
(found the 12 - 48 serialised g_A operand loads of attention_bwd_point_kernel in round 3: DESIGN.md section 4)."""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-fno-gpu-rdc", "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "--cuda-device-only", "-S"]


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    asm = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", "-"], capture_output=True, text=True).stdout
    cur, body = None, []
    kernels = []
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
            body.append(line)
    names = subprocess.run(["c++filt"] + [k for k, _ in kernels], capture_output=True, text=True).stdout.splitlines() if kernels else []
    print("%-78s %6s %6s %7s %9s" % ("kernel", "insts", "loads", "drains", "immediate"))
    for (mangled, body), name in zip(kernels, names):
        name = re.sub(r"\(.*", "", name)
        if flt and flt not in name:
            continue
        insts = [l.strip() for l in body if l.startswith("\t") and not l.strip().startswith((";", "."))]
        in_loop, loads, drains, imm, since = True, 0, 0, 0, 99  # (whole kernel: fully unrolled bodies have no loop markers)
        for l in body:
            s = l.strip()
            if not l.startswith("\t") or s.startswith((";", ".")):
                continue
            if s.startswith(("global_load", "flat_load", "buffer_load")):
                loads += in_loop
                since = 0
            elif s.startswith("s_waitcnt") and "vmcnt(0)" in s:
                if in_loop:
                    drains += 1
                    imm += since <= 12
                since = 99
            else:
                since += 1
        print("%-78s %6d %6d %7d %9d" % (name[-78:], len(insts), loads, drains, imm))


if __name__ == "__main__":
    main()
