#!/usr/bin/env python3
"""profiles/pmc_traffic.json from rocprofv3 counter passes of bench.py (tools/pmc_run.sh; the FETCH_SIZE GRBM_GUI_ACTIVE
pass).

    python tools/make_pmc_traffic.py <dir of the default (16-vector) run> [<dir of the --block 8 run> ...] > profiles/pmc_traffic.json

Per streaming-kernel kind (cprod / prod / cprod_stats, wide_cprod / wide_prod for the three-column-block launches) and
number of column blocks the record keeps the FULL-SIZE launches only (duration within
30 % of the longest: the warm-start launches touch 1/16 of the variants), their kernel name as rocprofv3 prints it,
FETCH_SIZE (KiB, doubled on gfx950 as MI355X_MICROARCH.md prescribes) and the clock GRBM_GUI_ACTIVE / 8 XCDs / time.
bench.py quotes an entry as roofline.traffic only when the name equals the instantiation the running library launches
and `matvec_sha256` equals the hash of bigsnpr_amd/csrc/matvec.hip it was built from."""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kind_of(name):
    """(column blocks, kind) of a streaming kernel; kinds as bench.py names them (the three-block launches of the
    precision schedule are timed apart: wide_prod / wide_cprod)"""
    m = re.match(r"(?:void )?bsn::(k_c?prodT?)<(.*)>$", name)
    if not m:
        return None
    args = [a.strip() for a in m.group(2).split(",")]
    nb = int(args[0])
    if m.group(1) in ("k_prod", "k_prodT"):   # k_prodT: the product on the sample-major copy (two / three column blocks)
        return nb, ("wide_prod" if nb == 3 else "prod")
    if args[4] == "true":                      # k_cprod<NB, NPLANE, KC, RAW0, STATS, ...>
        return nb, "cprod_stats"
    return nb, ("wide_cprod" if nb == 3 else "cprod")


def collect(d):
    ctr = defaultdict(lambda: defaultdict(dict))   # name -> dispatch id -> counter -> value
    for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"])
            if kind_of(name):
                ctr[name][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    dur = defaultdict(dict)
    for f in sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"])
            if kind_of(name):
                dur[name][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out = defaultdict(dict)   # column blocks -> kind -> record
    for name, disp in ctr.items():
        rows = [(dur[name].get(i), c) for i, c in disp.items() if "FETCH_SIZE" in c and dur[name].get(i)]
        if not rows:
            continue
        longest = max(t for t, _ in rows)
        full = [(t, c) for t, c in rows if t > 0.7 * longest]
        fs = sum(c["FETCH_SIZE"] for _, c in full) / len(full)
        ms = sum(t for t, _ in full) / len(full)
        rec = {"name": name, "fetch_size_kib": fs, "hbm_read_bytes": fs * 1024 * 2, "profiled_ms": ms, "dispatches": len(full)}
        if all("GRBM_GUI_ACTIVE" in c for _, c in full):
            rec["effective_GHz"] = sum(c["GRBM_GUI_ACTIVE"] for _, c in full) / len(full) / 8 / (ms * 1e-3) / 1e9
        nb, k = kind_of(name)
        if k not in out[nb] or rec["profiled_ms"] * rec["dispatches"] > out[nb][k]["profiled_ms"] * out[nb][k]["dispatches"]:
            out[nb][k] = rec
    return out


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    res = {"_comment": "HBM read traffic per launch from rocprofv3 PMC (a --pmc FETCH_SIZE GRBM_GUI_ACTIVE pass on its own with "
                       "--kernel-trace only, tools/pmc_run.sh; made by tools/make_pmc_traffic.py). FETCH_SIZE is in KiB and is doubled "
                       "on gfx950 as MI355X_MICROARCH.md prescribes. `kernels`: one MFMA column block (8 vectors x 2 slices; the "
                       "counting pass of the default solve: 16 vectors x 1 slice), `kernels_nb2`: two (16 vectors x 2 slices), "
                       "`kernels_nb3`: three (16 vectors x 3 slices: the early steps of the default solve at k = 20). bench.py quotes "
                       "the entry of the dominant kernel as roofline.traffic only when `name` and `matvec_sha256` match the running build.",
           "workload": {"n": int(os.environ.get("PMC_N", 400000)), "m_per_gpu": int(os.environ.get("PMC_M", 1000000))},
           "matvec_sha256": hashlib.sha256(open(os.path.join(ROOT, "bigsnpr_amd", "csrc", "matvec.hip"), "rb").read()).hexdigest(),
           "kernels": {}, "kernels_nb2": {}, "kernels_nb3": {}}
    for d in sys.argv[1:]:   # (earlier directories win: the default run first)
        for nb, recs in collect(d).items():
            dst = res["kernels" if nb == 1 else "kernels_nb%d" % nb]
            for k, rec in recs.items():
                dst.setdefault(k, rec)
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
