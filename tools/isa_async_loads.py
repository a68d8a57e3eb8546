#!/usr/bin/env python3
"""Safety check of the hand-scheduled MFMA loops (gram_kernel, wrows_kernel, propagate_kernel) on the compiler's device listing.

Those loops issue their operand loads as inline-assembly `global_load_dwordx2` with "=v" outputs and wait for them with
explicit `s_waitcnt vmcnt(N)`: the compiler does not know the destination registers are written asynchronously, so a copy,
spill or accumulator-file move it inserted between a load and its wait would read a register whose load may still be in
flight -- silently wrong numbers.  This script finds, for every kernel whose name matches, the innermost loops that contain
both `global_load_dwordx2` and `v_mfma`, and checks that
  * the loop body consists of loads, MFMAs, waits, scalar pointer arithmetic / loop control and at most `--allow-mul` v_mul_f64
    (the pair weight of the Gram) -- no v_mov*, v_accvgpr*, scratch_*, buffer_*, ds_*, v_readlane / v_writelane;
  * every register written by a load in the loop is read only by v_mfma / v_mul_f64, and only after an s_waitcnt that follows it;
  * no load destination overlaps an MFMA accumulator range.
    python tools/isa_async_loads.py LISTING.s kernel_name_substring [...]      exit code 0 = clean
"""
import re
import sys


def regs(tok):
    """VGPR numbers named by an operand token: v12 or v[12:13]."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check_kernel(name, body):
    labels = {}
    for j, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = j
    loops = []
    for j, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < j:
            loops.append((labels[m.group(1)], j))
    problems, checked = [], 0
    for a, b in loops:
        ins = [l.strip() for l in body[a:b + 1] if l.strip() and not l.strip().startswith((";", "."))]
        if not any(i.startswith("global_load_dwordx2") for i in ins) or not any(i.startswith("v_mfma") for i in ins):
            continue
        if any((a2 > a or b2 < b) and a2 >= a and b2 <= b for a2, b2 in loops):
            continue                                     # not innermost
        checked += 1
        acc, dests = set(), set()
        for i in ins:
            op, _, rest = i.partition(" ")
            toks = [t.strip() for t in rest.split(",")]
            if op.startswith("v_mfma"):
                acc |= regs(toks[0])
            if op == "global_load_dwordx2":
                dests |= regs(toks[0])
        if dests & acc:
            problems.append(f"{name}: a load destination overlaps an accumulator")
        for i in ins:
            op, _, rest = i.partition(" ")
            toks = [t.strip() for t in rest.split(",")]
            touched = set()
            for t in toks:
                touched |= regs(t)
            if op in ("global_load_dwordx2", "s_waitcnt") or op.startswith("v_mfma") or op == "v_mul_f64" or op.startswith("s_"):
                continue
            if op.startswith(("scratch_", "buffer_", "v_accvgpr", "v_readlane", "v_writelane", "ds_")):
                problems.append(f"{name}: spill / register-file traffic in the hand-scheduled loop: {i}")
            elif touched & (dests | acc):
                problems.append(f"{name}: {i} touches an operand or accumulator register of the hand-scheduled loop")
        # order check: between a load of register R and the next s_waitcnt, nothing reads R
        inflight = set()
        for i in ins + ins:                              # two trips: the back edge
            op, _, rest = i.partition(" ")
            toks = [t.strip() for t in rest.split(",")]
            if op == "global_load_dwordx2":
                inflight |= regs(toks[0])
            elif op == "s_waitcnt" and "vmcnt" in rest:
                n = int(re.search(r"vmcnt\((\d+)\)", rest).group(1))
                # loads complete in order: after vmcnt(n) only the last n issued may be in flight; the loops issue whole
                # register sets of n loads, so everything issued before the most recent set has landed
                recent, cnt = set(), 0
                for k in reversed(seen_loads):
                    if cnt >= n:
                        break
                    recent |= k
                    cnt += 1
                inflight = recent
            else:
                srcs = set()
                for t in toks[1:]:
                    srcs |= regs(t)
                if op.startswith("v_") and (srcs & inflight):
                    problems.append(f"{name}: {i} reads a register whose load has not been waited for")
            if op == "global_load_dwordx2":
                seen_loads.append(regs(toks[0]))
    return checked, problems


seen_loads = []


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split("\n")
    total, problems = 0, []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if not m or not any(p in m.group(1) for p in pats):
            continue
        end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
        del seen_loads[:]
        n, pr = check_kernel(m.group(1), lines[i:end])
        total += n
        problems += pr
        print(f"{m.group(1)[:60]}: {n} hand-scheduled loop(s) checked, {len(pr)} problem(s)")
    for p in problems:
        print("PROBLEM:", p)
    if total == 0:
        print("PROBLEM: no hand-scheduled loop found (kernel renamed?)")
        return 2
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
