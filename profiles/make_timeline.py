"""Timeline of the last G steps of a run from a rocprofv3 --kernel-trace csv:

    python profiles/make_timeline.py <kernel_trace.csv> [first kernel of a step = k_q0_touch_unique] [steps = 2]

Prints kernel, queue, start / end / duration in us from the start of the window (the last `steps` whole steps of the trace)."""
import csv
import sys


def main():
    path = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "k_q0_touch_unique"
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))

    def name(r):
        return r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    idx = [i for i, r in enumerate(rows) if name(r).startswith(first)]
    a, b = idx[-(steps + 3)], idx[-3]          # (the very last steps of a phase are followed by the flush: stay clear of it)
    t0 = int(rows[a]["Start_Timestamp"])
    print("%-36s %-5s %9s %9s %8s %8s" % ("kernel", "queue", "start", "end", "dur", "grid"))
    for r in rows[a:b + 1]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print("%-36s q%-4s %9.2f %9.2f %8.2f %8s" % (name(r)[:36], r.get("Queue_Id", "?"), s, e, e - s, r.get("Grid_Size_X", r.get("Grid_Size", ""))))


if __name__ == "__main__":
    main()
