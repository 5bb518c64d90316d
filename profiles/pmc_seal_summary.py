#!/usr/bin/env python3
"""Average per-launch SQ counters of the SEAL kernels from rocprofv3 --pmc passes (counters only) over
profiles/seal_stages.py:  python profiles/pmc_seal_summary.py <dir with pass*/ subdirectories> > summary"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "pass*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for key in ("seal_conv_t2_kernel<3, float, false>", "seal_conv_t2_kernel<1, unsigned short, false>", "seal_subgraph_kernel",
                    "seal_conv_t_kernel"):
            if key in name:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c in sorted(cs):
        v = cs[c]
        print(f"  {c:28s} {sum(v) / len(v):14.4g}   ({len(v)} launches)")
