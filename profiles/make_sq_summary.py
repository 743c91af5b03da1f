"""Per-kernel averages of an SQ counter pass (scripts/run_pmc_sq.sh) -> r2_sq_summary.json (SQ_SUMMARY_OUT overrides the name).
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (1024 * kernel duration in shader cycles); the duration
comes from the kernel-trace stats of the same build (r1_v9_*_kernel_stats.csv), the clock is the 2.4 GHz peak."""
import collections
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_pmc_traffic import key  # noqa: E402


def durations(path):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        k = key(r["Name"])
        tot[k] += int(r["TotalDurationNs"])
        cnt[k] += int(r["Calls"])
    return {k: tot[k] / cnt[k] for k in tot}


def main():
    out = {}
    for wl, sq_csv, stats in ((a.split("=")[0], *a.split("=")[1].split(",")) for a in sys.argv[1:]):
        dur = durations(stats)
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for r in csv.DictReader(open(sq_csv)):
            k = key(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                n[k] += 1
        out[wl] = {}
        for k in sorted(agg, key=lambda k: -agg[k]["SQ_WAVE_CYCLES"]):
            if k.startswith("at::") or k.startswith("__amd") or k not in dur:
                continue
            a = {c: v / n[k] for c, v in agg[k].items()}
            cyc = dur[k] * 2.4                         # ns * 2.4 cycles/ns
            out[wl][k] = {"avg_us": round(dur[k] / 1e3, 2), "mfma_busy_cycles": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)),
                          "mfma_util": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc), 4),
                          "wait_any_over_wave_cycles": round(a.get("SQ_WAIT_ANY", 0.0) / max(a.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3),
                          "launches": n[k]}
    json.dump(out, open(os.path.join(HERE, os.environ.get("SQ_SUMMARY_OUT", "r2_sq_summary.json")), "w"), indent=1)
    for wl in out:
        print(wl)
        for k, v in list(out[wl].items())[:12]:
            print("  %-16s %8.1f us  mfma_util %.3f  wait/wave %.2f" % (k, v["avg_us"], v["mfma_util"], v["wait_any_over_wave_cycles"]))


if __name__ == "__main__":
    main()
