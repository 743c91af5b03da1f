# copies the judged summaries of scripts/r5_evidence.sh from gpurun_out/r5_evidence/ into profiles/ (run in the container, after the GPU call)
set -e
cd $(dirname $0)/..
E=gpurun_out/r5_evidence
cp $E/r5_bench_*.json $E/r5_*_kernel_stats.csv $E/r5_mid25k_timeline.txt $E/r5_c3_timeline.txt profiles/
mkdir -p profiles/pmc
for wl in askubuntu c4 ml20m custom_25024; do for c in FETCH_SIZE WRITE_SIZE SQ; do cp $E/r5_${wl}_$c.csv profiles/pmc/; done; done
rm -f profiles/r5_pmc_traffic.json
python profiles/make_pmc_traffic.py --out r5_pmc_traffic.json askubuntu=profiles/pmc/r5_askubuntu_FETCH_SIZE.csv,profiles/pmc/r5_askubuntu_WRITE_SIZE.csv \
    c4=profiles/pmc/r5_c4_FETCH_SIZE.csv,profiles/pmc/r5_c4_WRITE_SIZE.csv ml20m=profiles/pmc/r5_ml20m_FETCH_SIZE.csv,profiles/pmc/r5_ml20m_WRITE_SIZE.csv \
    custom:25024=profiles/pmc/r5_custom_25024_FETCH_SIZE.csv,profiles/pmc/r5_custom_25024_WRITE_SIZE.csv
python - <<'PY'
import collections, csv, re
E = "gpurun_out/r5_evidence"
# kernel durations of the same workload (rocprofv3 --kernel-trace --stats)
dur = {}
for r in csv.DictReader(open(E + "/r5_askubuntu_kernel_stats.csv")):
    m = re.search(r"(fk_d_\w+)", r["Name"])
    if m: dur[m.group(1)] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
agg = collections.defaultdict(dict)
for line in open(E + "/r5_d_step_counters_raw.txt"):
    m = re.match(r"(\w+)\s+launches (\d+) (\{.*\})", line)
    if m: agg[m.group(1)].update(eval(m.group(3)))
out = []
out.append("Discriminator step at config.ini's sizes on Askubuntu_Sample (h = 100/150/250/300, ~1 840 pair rows per step, fp32 MFMA): what the five kernels'\n"
           "time beyond their MFMA work is made of.  rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0 --sub-epochs 1 --no-probe` (scripts/r5_evidence.sh),\n"
           "averages per launch over the 101 batches of one D sub-epoch.  SQ_* wave counters are in quad-cycles summed over all waves of a launch;\n"
           "SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the 1 024 SIMDs.  avg us = rocprofv3 --kernel-trace --stats of the same workload (both launches\n"
           "of fk_d_bwd1 -- job A on the caller's stream, jobs B / C on the aux stream -- are one row).\n")
out.append("%-12s %7s | %9s %9s %9s | %8s %11s | %9s %9s | %9s %9s" % ("kernel", "avg us", "parked", "issue-", "issuing", "MFMA", "MFMA us/", "VMEM rd", "LDS", "L2 hit", "L1 hit"))
out.append("%-12s %7s | %9s %9s %9s | %8s %11s | %9s %9s | %9s %9s" % ("", "", "WAIT_ANY", "stalled", "", "util", "SIMD", "insts", "insts", "rate", "rate"))
for k in ("fk_d_l1", "fk_d_l2", "fk_d_bwd1", "fk_d_bwd2", "fk_d_adam"):
    c = agg[k]
    wc = c["SQ_WAVE_CYCLES"]
    us = dur.get(k, (0, 0))[0]
    mfma_us = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / 2100.0          # cycles per SIMD at ~2.1 GHz
    hit = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    l1 = 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(1.0, c["TCP_TOTAL_CACHE_ACCESSES_sum"])
    out.append("%-12s %7.1f | %8.0f%% %8.0f%% %8.0f%% | %7.0f%% %11.2f | %9d %9d | %8.0f%% %8.0f%%" % (
        k, us, 100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc,
        100 * mfma_us / us if us else 0, mfma_us, c["SQ_INSTS_VMEM_RD"], c["SQ_INSTS_LDS"], 100 * hit, 100 * l1))
out.append("")
out.append("parked = wave cycles in s_waitcnt / barriers (memory round trips, the LDS meeting of the K slices); issue-stalled = SQ_WAIT_INST_ANY (a wave\n"
           "has an instruction but the pipe is taken or its accumulator not ready: fp32 MFMA 16x16x4 issues every 32 cycles per SIMD, 40 dependent);\n"
           "MFMA util = MFMA busy time per SIMD / kernel duration.\n")
out.append("Reading: the products are 1-3 us of MFMA time per SIMD inside kernels of 6-15 us.  The forward / backward GEMM kernels spend 27-38 % of their\n"
           "wave cycles parked and 35-51 % issue-stalled behind the (few) waves that hold the matrix pipe: the work per workgroup is one dependent chain\n"
           "request operands -> MFMA -> LDS meeting -> epilogue, and the grid is one to two rounds of workgroups, so the kernel lasts about one chain plus\n"
           "launch ramp and drain whatever the MFMA rate.  fk_d_adam is pure latency (92 % parked: one round trip for the slabs, one for theta / m / v).\n"
           "L2 hit rates are 80-92 % (fk_d_bwd2 54 %, fk_d_adam 38 %: first touch of the gradient slabs another kernel just wrote).\n"
           "Raw per-launch averages:\n")
out += [l.rstrip() for l in open(E + "/r5_d_step_counters_raw.txt")]
open("profiles/r5_d_step_counters.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:14]))
PY
