#!/usr/bin/env python3
"""Instruction-class census of one or more kernels in the -save-temps assembly (make -C long-tail-gan_amd/csrc asm):
usage: python scripts/asm_count.py /tmp/ltg_kernels-hip-amdgcn-amd-amdhsa-gfx950.s fks_d_l1 fk_d_l2 ..."""
import collections
import re
import sys

src = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    m = re.search(r"^(_Z\w*\d%s(?:I|E)\w*):[^\n]*\n(.*?)s_endpgm" % name, src, re.S | re.M)
    if not m:
        print(name, "not found")
        continue
    body = m.group(2)
    ops = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ops)
    valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    print("%s: %d instructions, VALU %d, MFMA %d, vmem %d, lds %d, s_waitcnt %d" % (
        name, len(ops), valu, sum(v for k, v in c.items() if k.startswith("v_mfma")),
        sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_"))), sum(v for k, v in c.items() if k.startswith("ds_")), c["s_waitcnt"]))
    print("   ", dict(c.most_common(30)))
