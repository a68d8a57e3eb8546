#!/usr/bin/env python3
"""Listing check of the asynchronous scalar loads that csrc/kmeans.hip issues from inline assembly (pk_issue / pk_wait: one pair record
= s_load_dwordx16 + s_load_dwordx8 + s_load_dwordx4 from one address, waited for by a SEPARATE asm statement).

The compiler believes the destination SGPRs are defined as soon as the issuing statement is over; the data arrives later.  Two things
would read garbage or fault, and neither is visible in the source:
  (1) a destination range that overlaps the address pair of its own block (the second and third request would then read an address
      the first one is overwriting) or another destination of the block -- the fault of round 4, fixed with early-clobber outputs;
  (2) any instruction between the requests and the `s_waitcnt lgkmcnt(0)` that covers them which reads, copies, spills or overwrites
      one of the destination registers (the allocator moving a "live" value that is not there yet), or control flow that leaves the
      straight line before the wait.
This walks the compiler's own listing (hipcc -S --offload-device-only) and reports both.

    python tools/isa_sload_window.py kmeans.s [kernel-name-substring ...]      exit code 1 if a problem was found
"""
import re
import sys

SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")
LOAD16 = re.compile(r"^s_load_dwordx16 s\[(\d+):(\d+)\], s\[(\d+):(\d+)\], 0x0$")
LOAD8 = re.compile(r"^s_load_dwordx8 s\[(\d+):(\d+)\], s\[(\d+):(\d+)\], 0x40$")
LOAD4 = re.compile(r"^s_load_dwordx4 s\[(\d+):(\d+)\], s\[(\d+):(\d+)\], 0x60$")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def code(line):
    return line.split(";", 1)[0].strip()


def check(lines, name_filters):
    problems, blocks, kernels = [], 0, set()
    kernel = None
    i = 0
    n = len(lines)
    while i < n:
        raw = lines[i]
        if raw.startswith("_ZN") and raw.rstrip().endswith(":") or (raw.startswith("_ZN") and ":" in raw.split(";")[0]):
            kernel = raw.split(":")[0]
        c = code(raw)
        m16 = LOAD16.match(c)
        if m16 and i + 2 < n and (not name_filters or (kernel and any(f in kernel for f in name_filters))):
            m8, m4 = LOAD8.match(code(lines[i + 1])), LOAD4.match(code(lines[i + 2]))
            if m8 and m4 and m8.group(3, 4) == m16.group(3, 4) == m4.group(3, 4):
                blocks += 1
                kernels.add(kernel)
                addr = set(range(int(m16.group(3)), int(m16.group(4)) + 1))
                d = [set(range(int(m.group(1)), int(m.group(2)) + 1)) for m in (m16, m8, m4)]
                dest = d[0] | d[1] | d[2]
                where = f"{kernel} line {i + 1}"
                if dest & addr:
                    problems.append(f"{where}: a destination overlaps the address pair s[{min(addr)}:{max(addr)}]")
                if len(dest) != 16 + 8 + 4:
                    problems.append(f"{where}: the three destinations overlap each other")
                j = i + 3
                done = False
                while j < n:
                    cj = code(lines[j])
                    if not cj or cj.startswith("."):
                        if cj.startswith(".LBB") or cj.startswith(".Lfunc"):
                            problems.append(f"{where}: a label (line {j + 1}) before the wait that covers the loads")
                            break
                        j += 1
                        continue
                    if cj.startswith("s_waitcnt") and "lgkmcnt(0)" in cj:
                        done = True
                        break
                    if cj.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
                        problems.append(f"{where}: control flow (line {j + 1}: {cj}) before the wait that covers the loads")
                        break
                    ops = cj.split(None, 1)[1] if " " in cj else ""
                    hit = sregs(ops) & dest
                    if hit:
                        problems.append(f"{where}: line {j + 1} touches s{sorted(hit)} while the load is in flight: {cj}")
                    j += 1
                if not done and j >= n:
                    problems.append(f"{where}: no s_waitcnt lgkmcnt(0) follows")
                i += 3
                continue
        i += 1
    return blocks, kernels, problems


if __name__ == "__main__":
    lines = open(sys.argv[1]).read().split("\n")
    blocks, kernels, problems = check(lines, sys.argv[2:])
    for p in problems:
        print("PROBLEM", p)
    print(f"{blocks} asynchronous pair-record request(s) in {len(kernels)} kernel(s) checked, {len(problems)} problem(s)")
    sys.exit(1 if problems or not blocks else 0)
