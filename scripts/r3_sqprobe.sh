# one rocprofv3 --pmc pass (SQ block only, no trace domains) with issue / wait counters: where do the waves of the discriminator kernels spend their cycles?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_sqprobe
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 $R/bench.py --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline --no-other-workloads > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/p2 -- python3 $R/bench.py --steps 1 --warmup 0 --sub-epochs 1 --no-probe --no-cpu-baseline --no-other-workloads > $O/p2.log 2>&1
python3 - $O <<'PY'
import csv, sys, glob, collections
O = sys.argv[1]
for p in ("p1", "p2"):
    f = glob.glob(O + "/" + p + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(p, "no output"); print(open(O + "/" + p + ".log").read()[-1500:]); continue
    tot = collections.defaultdict(lambda: collections.Counter()); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"): cnt[k] += 1
    for k in ("fk_d_bwd1", "fk_d_l2", "fk_d_l1", "fk_d_bwd2", "fk_g_tail<true>", "fk_dh2<true>", "fk_enc1<false>"):
        if k in tot:
            n = max(1, cnt[k])
            print(p, "%-18s" % k, {c: round(v / n) for c, v in sorted(tot[k].items())})
PY
rm -rf $O/p1 $O/p2
