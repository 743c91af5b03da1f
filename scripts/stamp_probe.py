#!/usr/bin/env python3
"""Per-workgroup phase timeline of ONE latency kernel (measurement build -DLTG_STAMP=<id>: 1 fk_d_l1, 2 fk_d_l2, 3 fk_d_bwd1 job A,
4 job B, 5 fk_d_bwd2, 11 fk_enc1, 12 fk_dec0, 15 fk_dz, 16 fk_dh1; csrc/ltg_rgemm.h).  Runs one D sub-epoch of the headline workload through bench.py and reads the stamps of the LAST
launch.  usage: LTG_HIP_LIB=ab_live/libltg_stamp2.so python scripts/stamp_probe.py <n_workgroups>"""
import ctypes, os, runpy, sys
import numpy as np

nwg = int(sys.argv[1]) if len(sys.argv) > 1 else 580
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--sub-epochs", "1", "--no-probe", "--no-cpu-baseline", "--no-other-workloads"]
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
lib = ctypes.CDLL(os.environ["LTG_HIP_LIB"])
buf = np.zeros(nwg * 8, dtype=np.uint64)
rc = lib.ltg_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(nwg * 8))
assert rc == 0
s = buf.reshape(nwg, 8)
t = s[:, :6].astype(np.int64)
ok = t[:, 0] > 0
t = t[ok]
hw = s[ok, 6]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
cu = (hw.astype(np.int64) >> 8) & 0xF
se = (hw.astype(np.int64) >> 13) & 0x7
t0 = t[:, 0].min()
us = (t - t0) / 100.0      # 100 MHz
names = ["entry", "requests issued", "mid done", "last MFMA issued", "K slices met", "epilogue done"]
print("workgroups with stamps: %d of %d; times in us from the FIRST workgroup's entry (wave 0 of each workgroup)" % (ok.sum(), nwg))
print("%-18s %8s %8s %8s %8s %8s" % ("stamp", "min", "p10", "median", "p90", "max"))
for j, n in enumerate(names):
    c = us[:, j]
    print("%-18s %8.2f %8.2f %8.2f %8.2f %8.2f" % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
print("phase lengths per workgroup (us):")
for j in range(1, 6):
    d = us[:, j] - us[:, j - 1]
    print("  %-34s %8.2f %8.2f %8.2f %8.2f %8.2f" % (names[j - 1] + " -> " + names[j], d.min(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max()))
d = us[:, 5] - us[:, 0]
print("  %-34s %8.2f %8.2f %8.2f %8.2f %8.2f" % ("entry -> epilogue done", d.min(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max()))
print("per XCC: workgroups, median entry, median end")
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("  xcc %d: %4d  %6.2f  %6.2f" % (x, m.sum(), np.median(us[m, 0]), np.median(us[m, 5])))
order = np.argsort(us[:, 0])
print("entry time of workgroup k in dispatch order: k=0,64,128,...:", [round(float(us[order[k], 0]), 2) for k in range(0, len(order), 64)])
