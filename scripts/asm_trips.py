#!/usr/bin/env python3
"""Compressed memory-dependency trace of one kernel from `make asm`'s gfx950 assembly: runs of loads / stores / MFMAs, every s_waitcnt
and barrier, labels and branches -- to count the DEPENDENT round trips a workgroup makes.  usage: asm_trips.py <kernel substring> [file]"""
import re, sys
pat = sys.argv[1]
f = sys.argv[2] if len(sys.argv) > 2 else "/tmp/ltg_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
lines = open(f).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l]
for s in starts:
    name = lines[s][:-1]
    print("==", name[:150])
    out = []
    def push(tag):
        if out and out[-1][0] == tag: out[-1][1] += 1
        else: out.append([tag, 1])
    for l in lines[s + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end"): break
        if re.match(r"^\.LBB\d+_\d+:", t): out.append(["\n" + t.split(":")[0] + ":", 1]); continue
        op = t.split(" ")[0] if t else ""
        if op.startswith(("global_load", "buffer_load", "flat_load")): push("L" + ("lds" if "lds" in t else ""))
        elif op.startswith("scratch_"): push("SCRATCH")
        elif op.startswith(("global_store", "buffer_store", "flat_store")): push("St")
        elif op.startswith(("global_atomic", "buffer_atomic", "flat_atomic")): push("At")
        elif op.startswith("s_load") or op.startswith("s_buffer_load"): push("sL")
        elif op.startswith("ds_"): push("ds")
        elif op.startswith("v_mfma"): push("M")
        elif op == "s_waitcnt": out.append(["W(" + t[len("s_waitcnt"):].strip() + ")", 1])
        elif op == "s_barrier": out.append(["BAR", 1])
        elif op.startswith("s_cbranch") or op == "s_branch": out.append(["->" + t.split()[-1], 1])
        elif op == "s_endpgm": out.append(["END", 1])
    print(" ".join(k if n == 1 else "%s*%d" % (k, n) for k, n in out))
