#!/usr/bin/env python3
"""Per-kernel dynamic instruction mix from a rocprofv3 --pmc SQ_INSTS_* pass (csv): instructions per wave and per MFMA.
usage: pmc_inst_mix.py counter_collection.csv"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
print("%-52s %9s %7s %7s %7s %6s %6s %6s %7s" % ("kernel", "per wave:", "MFMA", "VALU-M", "SALU", "LDS", "VMEM", "BR", "nonM/M"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    w = a.get("SQ_WAVES", 0)
    if w == 0 or a.get("SQ_INSTS_VALU", 0) == 0:
        continue
    m = a.get("SQ_INSTS_MFMA", 0)
    valu = a["SQ_INSTS_VALU"] - m
    vm = a.get("SQ_INSTS_VMEM_RD", 0) + a.get("SQ_INSTS_VMEM_WR", 0)
    tot = valu + a.get("SQ_INSTS_SALU", 0) + a.get("SQ_INSTS_LDS", 0) + vm + a.get("SQ_INSTS_BRANCH", 0)
    print("%-52s %9s %7.0f %7.0f %7.0f %6.0f %6.0f %6.0f %7.2f" % (k[:52], "", m / w, valu / w, a.get("SQ_INSTS_SALU", 0) / w, a.get("SQ_INSTS_LDS", 0) / w,
                                                               vm / w, a.get("SQ_INSTS_BRANCH", 0) / w, tot / m if m else float("nan")))
