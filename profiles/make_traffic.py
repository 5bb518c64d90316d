#!/usr/bin/env python3
"""profiles/traffic_latest.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, counters only)
over profiles/pmc_workload.py:

    cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d <out>/pmc_fetch -- python3 profiles/pmc_workload.py
    cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d <out>/pmc_write -- python3 profiles/pmc_workload.py
    python profiles/make_traffic.py <out>/pmc_fetch <out>/pmc_write [profiles/rNN]

The workload launches the SpMM three times on a PERMUTATION graph whose compulsory bytes are known (calibration of
what FETCH_SIZE reports for this access pattern: the guide's x2 on gfx950) and three times on the bench workload.
Counter values are KiB per dispatch; per step = main + fix-up + empty-rows kernels.  The spmm.hip the counters were
collected with is recorded by sha256: bench.py reports the traffic only while that still matches."""
import csv, datetime, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_step(dirname, counter):
    f = glob.glob(os.path.join(dirname, "*", "*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "spmm_" in r["Kernel_Name"]]
    # dispatch order: 3 x (main, fixup[, empty]) on the permutation graph, then 3 x (main, fixup, empty) on RMAT
    steps, cur = [], None
    for r in rows:
        if "spmm_main_kernel" in r["Kernel_Name"]:
            cur = {"main": 0.0, "fixup": 0.0, "empty": 0.0}
            steps.append(cur)
        key = "main" if "main" in r["Kernel_Name"] else "fixup" if "fixup" in r["Kernel_Name"] else "empty"
        cur[key] += float(r["Counter_Value"]) * 1024.0
    return steps, f


fetch, f_file = per_step(sys.argv[1], "FETCH_SIZE")
write, w_file = per_step(sys.argv[2], "WRITE_SIZE")
assert len(fetch) == 6 and len(write) == 6, (len(fetch), len(write))
N, E, d = 5_000_000, 100_000_000, 128
cal_f = sum(sum(s.values()) for s in fetch[:3]) / 3
cal_w = sum(sum(s.values()) for s in write[:3]) / 3
known_read = N * d * 4 + N * 4 + N * 4 + 8                                 # X rows once 2.56 GB + col 20 MB + row ids 20 MB
known_write = N * d * 4
ratio = known_read / cal_f
corr = 2.0 if 1.8 <= ratio <= 2.2 else 1.0
run_f = [sum(s.values()) for s in fetch[3:]]
run_w = [sum(s.values()) for s in write[3:]]
fb, wb = sum(run_f) / 3 * corr, sum(run_w) / 3
out = {
    "workload": [N, E, d, "f32"],
    "hbm_bytes_per_step": fb + wb,
    "fetch_bytes_corrected": fb,
    "write_bytes": wb,
    "fetch_size_correction": corr,
    "per_kernel_bytes": {k: sum(s[k] for s in fetch[3:]) / 3 * corr + sum(s[k] for s in write[3:]) / 3 for k in ("main", "fixup", "empty")},
    "calibration": {"graph": "random permutation |E|=|V|=5M d=128 (every X row read once)", "known_read_bytes": known_read,
                    "FETCH_SIZE_reported_bytes": cal_f, "ratio": ratio, "WRITE_SIZE_reported_bytes": cal_w,
                    "known_write_bytes": known_write},
    "spmm_hip_sha256": hashlib.sha256(open(os.path.join(ROOT, "link-prediction-gnn_amd", "csrc", "spmm.hip"), "rb").read()).hexdigest(),
    "collected": datetime.date.today().isoformat() + (" (" + sys.argv[3] + ")" if len(sys.argv) > 3 else ""),
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, counters only) over profiles/pmc_workload.py; "
              "profiles/make_traffic.py",
    "note": "FETCH_SIZE/WRITE_SIZE count the L2's memory-side requests; Infinity-Cache hits are included, so this is an "
            "upper bound on HBM bytes",
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    import shutil
    os.makedirs(sys.argv[3], exist_ok=True)
    for src, name in ((f_file, "pmc_fetch_size_spmm.csv"), (w_file, "pmc_write_size_spmm.csv")):
        rows = [r for r in csv.DictReader(open(src)) if "spmm_" in r["Kernel_Name"]]
        with open(os.path.join(sys.argv[3], name), "w", newline="") as fo:
            wtr = csv.writer(fo)
            wtr.writerow(["Dispatch_Id", "Grid_Size", "Kernel_Name", "VGPR_Count", "Counter_Name", "Counter_Value_KiB"])
            for r in rows:
                wtr.writerow([r["Dispatch_Id"], r["Grid_Size"], r["Kernel_Name"][:110], r["VGPR_Count"], r["Counter_Name"], r["Counter_Value"]])
