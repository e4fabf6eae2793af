#!/usr/bin/env python3
"""profiles/pmc_traffic.json from rocprofv3 counter passes of bench.py (tools/pmc_run.sh; the FETCH_SIZE GRBM_GUI_ACTIVE
pass).

    python tools/make_pmc_traffic.py <dir of the default (16-vector) run> [<dir of the --block 8 run>] > profiles/pmc_traffic.json

Per streaming-kernel kind (cprod / prod / cprod_stats) the record keeps the FULL-SIZE launches only (duration within
30 % of the longest: the warm-start launches touch 1/16 of the variants), their kernel name as rocprofv3 prints it,
FETCH_SIZE (KiB, doubled on gfx950 as MI355X_MICROARCH.md prescribes) and the clock GRBM_GUI_ACTIVE / 8 XCDs / time.
bench.py quotes an entry as roofline.traffic only when the name equals the instantiation the running library launches
and `matvec_sha256` equals the hash of bigsnpr_amd/csrc/matvec.hip it was built from."""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kind_of(name):
    m = re.match(r"(?:void )?bsn::(k_c?prodT?)<(.*)>$", name)
    if not m:
        return None
    args = [a.strip() for a in m.group(2).split(",")]
    if m.group(1) in ("k_prod", "k_prodT"):   # k_prodT: the product on the sample-major copy (two column blocks)
        return "prod"
    return "cprod_stats" if args[4] == "true" else "cprod"     # k_cprod<NB, NPLANE, KC, RAW0, STATS, ...>


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
    out = {}
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
        k = kind_of(name)
        if k not in out or rec["profiled_ms"] * rec["dispatches"] > out[k]["profiled_ms"] * out[k]["dispatches"]:
            out[k] = rec
    return out


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    res = {"_comment": "HBM read traffic per launch from rocprofv3 PMC (a --pmc FETCH_SIZE GRBM_GUI_ACTIVE pass on its own with "
                       "--kernel-trace only, tools/pmc_run.sh; made by tools/make_pmc_traffic.py). FETCH_SIZE is in KiB and is doubled "
                       "on gfx950 as MI355X_MICROARCH.md prescribes. `kernels`: one MFMA column block (8 vectors x 2 slices), "
                       "`kernels_nb2`: two (16 vectors x 2 slices, the library default at k = 20). bench.py quotes the entry of the "
                       "dominant kernel as roofline.traffic only when `name` and `matvec_sha256` match the running build.",
           "workload": {"n": int(os.environ.get("PMC_N", 400000)), "m_per_gpu": int(os.environ.get("PMC_M", 1000000))},
           "matvec_sha256": hashlib.sha256(open(os.path.join(ROOT, "bigsnpr_amd", "csrc", "matvec.hip"), "rb").read()).hexdigest(),
           "kernels_nb2": collect(sys.argv[1]),
           "kernels": collect(sys.argv[2]) if len(sys.argv) > 2 else {}}
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
