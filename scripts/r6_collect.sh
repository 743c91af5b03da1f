# copies the judged summaries of scripts/r6_evidence.sh from gpurun_out/r6_evidence/ into profiles/ (run in the container, after the GPU call)
set -e
cd $(dirname $0)/..
E=gpurun_out/r6_evidence
cp $E/r6_bench_*.json $E/r6_*_kernel_stats.csv $E/r6_askubuntu_g_step_timeline.txt $E/r6_d_step_timeline.txt profiles/
mkdir -p profiles/pmc
for wl in askubuntu c4 ml20m custom_25024; do for c in FETCH_SIZE WRITE_SIZE; do cp $E/r6_${wl}_$c.csv profiles/pmc/; done; done
rm -f profiles/r6_pmc_traffic.json
python profiles/make_pmc_traffic.py --out r6_pmc_traffic.json askubuntu=profiles/pmc/r6_askubuntu_FETCH_SIZE.csv,profiles/pmc/r6_askubuntu_WRITE_SIZE.csv \
    c4=profiles/pmc/r6_c4_FETCH_SIZE.csv,profiles/pmc/r6_c4_WRITE_SIZE.csv ml20m=profiles/pmc/r6_ml20m_FETCH_SIZE.csv,profiles/pmc/r6_ml20m_WRITE_SIZE.csv \
    custom:25024=profiles/pmc/r6_custom_25024_FETCH_SIZE.csv,profiles/pmc/r6_custom_25024_WRITE_SIZE.csv
grep -E "PURE fp32|max rel err probs" $E/forward_parity.log | sed 's/^\.*//' > profiles/r6_bf16_vs_pure_fp32_gap.txt
python - <<'PY'
import collections, csv, re
E = "gpurun_out/r6_evidence"
dur = {}
for r in csv.DictReader(open(E + "/r6_askubuntu_kernel_stats.csv")):
    m = re.search(r"(fk_d_\w+|fkt_d_tower)", r["Name"])
    if m: dur[m.group(1)] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
agg = collections.defaultdict(dict)
for line in open(E + "/r6_d_step_counters_raw.txt"):
    m = re.match(r"(\w+)\s+launches (\d+) (\{.*\})", line)
    if m: agg[m.group(1)].update(eval(m.group(3)))
out = []
out.append("Discriminator step at config.ini's sizes on Askubuntu_Sample (h = 100/150/250/300, ~1 840 pair rows per step), round 6: fk_d_l2 / fk_d_bwd1 / fk_d_bwd2 form\n"
           "their fp32 products as six bf16 cross terms of split operands (d_arith bf16x6; fk_d_l1 stays on the fp32 matrix pipe), and the one-kernel forward-only tower\n"
           "fkt_d_tower (93 k pair rows per launch).  rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0 --sub-epochs 1 --no-probe` (scripts/r6_evidence.sh), averages\n"
           "per launch.  SQ_* wave counters are in quad-cycles summed over all waves of a launch; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the 1 024 SIMDs.  avg us =\n"
           "rocprofv3 --kernel-trace --stats of the same workload (both launches of fk_d_bwd1 are one row).  Round 5's table of the fp32-MFMA kernels: r5_d_step_counters.txt.\n")
out.append("%-12s %7s | %9s %9s %9s | %8s %11s | %9s %9s | %9s %9s" % ("kernel", "avg us", "parked", "issue-", "issuing", "MFMA", "MFMA us/", "VMEM rd", "LDS", "L2 hit", "L1 hit"))
out.append("%-12s %7s | %9s %9s %9s | %8s %11s | %9s %9s | %9s %9s" % ("", "", "WAIT_ANY", "stalled", "", "util", "SIMD", "insts", "insts", "rate", "rate"))
for k in ("fk_d_l1", "fk_d_l2", "fk_d_bwd1", "fk_d_bwd2", "fk_d_adam", "fkt_d_tower"):
    c = agg.get(k)
    if not c or "SQ_WAVE_CYCLES" not in c: continue
    wc = c["SQ_WAVE_CYCLES"]
    us = dur.get(k, (0, 0))[0]
    mfma_us = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / 2100.0          # cycles per SIMD at ~2.1 GHz
    hit = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    l1 = 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(1.0, c["TCP_TOTAL_CACHE_ACCESSES_sum"])
    out.append("%-12s %7.1f | %8.0f%% %8.0f%% %8.0f%% | %7.0f%% %11.2f | %9d %9d | %8.0f%% %8.0f%%" % (
        k, us, 100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc,
        100 * mfma_us / max(us, 1e-9), mfma_us, c["SQ_INSTS_VMEM_RD"], c["SQ_INSTS_LDS"], 100 * hit, 100 * l1))
out.append("\nRaw per-launch averages:\n")
out.append(open(E + "/r6_d_step_counters_raw.txt").read())
open("profiles/r6_d_step_counters.txt", "w").write("\n".join(out))
print("\n".join(out[:12]))
PY
ls profiles | grep r6_
