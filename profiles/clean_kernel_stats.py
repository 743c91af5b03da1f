#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats summary -> the tracked profiles/r<N>_<workload>_kernel_stats.csv: the same rows and columns plus
`PercentageOfWork`, the kernel's share of the time of the kernels that DO work.  The one-wave gate kernels (k_gate_wait / k_gate_set, and the
stream-pair probe of the start-up) sit parked on the aux / side queues for as long as the kernel they wait for runs -- 15.6 % of the summed kernel
time of the round-5 headline summary -- and made every percentage read that much low; their rows stay (Calls / durations are facts), their
PercentageOfWork is empty.
usage: python profiles/clean_kernel_stats.py <rocprof kernel_stats.csv> <out.csv>"""
import csv
import sys

NOT_WORK = ("k_gate_wait", "k_gate_set", "k_pipe_probe")


def base(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].strip()


def main(src, dst):
    rows = list(csv.DictReader(open(src)))
    work = sum(float(r["TotalDurationNs"]) for r in rows if base(r["Name"]) not in NOT_WORK)
    fields = list(rows[0].keys()) + ["PercentageOfWork"]
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=fields)
        w.writeheader()
        for r in rows:
            r["PercentageOfWork"] = "" if base(r["Name"]) in NOT_WORK else "%.4f" % (100.0 * float(r["TotalDurationNs"]) / work)
            w.writerow(r)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
