"""Builds <round>_pmc_traffic.json (default r2) from rocprofv3 --pmc passes (one counter per pass, no trace domains):

    python profiles/make_pmc_traffic.py [--out r2_pmc_traffic.json] <workload>=<FETCH_SIZE csv>,<WRITE_SIZE csv> ...

Per kernel: average FETCH_SIZE / WRITE_SIZE (KB) per launch and traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 --
gfx950 tallies the 128-B requests of wide coalesced reads at 64 B (MI355X_MICROARCH.md, HBM section).  Keys are
bench.py's probe names."""
import collections
import csv
import json
import os
import re
import sys

ALIAS = {"dh2_partial": "dh2", "dh2_stream": "dh2", "dec1_fwd_stream": "dec1_fwd", "dec1_fwd_stream2": "dec1_fwd", "dec1_bwd_adam_stream": "dec1_bwd_adam",
         "dense_fwd<0>": "enc1", "dense_fwd<1>": "dec0", "row_partial_seg": "row_partial", "dec1": "dec1_fwd", "enc0_grad_rows": "enc0_grad", "q0_touch_slice": "q0_touch_unique"}


def key(kernel_name):
    n = kernel_name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    base = re.sub(r"<.*>", "", n)
    if base == "k_dense_fwd":
        base = n
    base = base[3:] if base.startswith("fk_") else (base[2:] if base.startswith("k_") else base)     # round-2 kernels: fk_*
    return ALIAS.get(base, base)


def averages(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = key(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main():
    args = sys.argv[1:]
    name = "r2_pmc_traffic.json"
    if args and args[0] == "--out":
        name, args = args[1], args[2:]
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    doc = json.load(open(out_path)) if os.path.exists(out_path) else {"workloads": {}}
    for arg in args:
        wl, files = arg.split("=")
        f_csv, w_csv = files.split(",")
        fetch, n = averages(f_csv, "FETCH_SIZE")
        write, _ = averages(w_csv, "WRITE_SIZE")
        doc["workloads"][wl] = {k: {"fetch_size_kb_raw": round(fetch[k], 1), "write_size_kb": round(write.get(k, 0.0), 1),
                                    "traffic_bytes": int((2 * fetch[k] + write.get(k, 0.0)) * 1024), "launches": n[k]}
                                for k in sorted(fetch) if not k.startswith("at::") and not k.startswith("__amd")}
    json.dump(doc, open(out_path, "w"), indent=1)
    print("wrote", out_path, {w: len(v) for w, v in doc["workloads"].items()})


if __name__ == "__main__":
    main()
