#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_* pass (csv) per kernel: MfmaUtil, wave-time breakdown, LDS conflicts, residency.
usage: pmc_sq_summary.py counter_collection.csv [out.csv]   (clock assumed 2.19 GHz = SQ_BUSY_CYCLES/32 SE/us measured)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt, dur, seen = collections.Counter(), collections.defaultdict(float), set()
for r in rows:
    k = r["Kernel_Name"].replace("void ", "").split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        cnt[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = ["kernel,calls,us_per_call,MfmaUtil_pct,wave_active_pct,wave_issue_stall_pct,wave_wait_pct,avg_waves_per_CU,"
       "lds_bank_conflict_pct_of_lds_cycles"]
for k in sorted(agg, key=lambda k: -dur[k]):
    a = agg[k]
    wc = a["SQ_WAVE_CYCLES"]
    if wc == 0:
        continue
    out.append('"%s",%d,%.1f,%.1f,%.1f,%.1f,%.1f,%.2f,%.2f' % (
        k, cnt[k], dur[k] / cnt[k], 100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur[k] * 2190 * 1024),
        100 * a["SQ_ACTIVE_INST_ANY"] / wc, 100 * a["SQ_WAIT_INST_ANY"] / wc, 100 * a["SQ_WAIT_ANY"] / wc,
        4 * wc / (dur[k] * 2190) / 256, 100 * a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1)))
txt = "\n".join(out) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
sys.stdout.write(txt)
