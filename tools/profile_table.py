#!/usr/bin/env python3
"""Join the passes of one profiled config into the per-kernel fraction table DESIGN.md section 6 quotes.

    profile_table.py <layers.json> <kernel_stats.csv> <pmc_traffic.json> <pmc_sq.csv> [calibration.json] > table.txt

layers.json: tools/config_run.py --layers (algorithmic FLOPs / bytes per launch); kernel_stats.csv: rocprofv3 --kernel-trace --stats (time);
pmc_traffic.json: tools/pmc_traffic.py (FETCH_SIZE / WRITE_SIZE, gfx950-corrected); pmc_sq.csv: tools/pmc_sq_summary.py (MfmaUtil, LDS conflicts).
Fractions: MFMA = algorithmic FLOP/s / 2.5 PFLOP/s; HBM = MEASURED bytes / time / 8 TB/s (and / the measured streaming ceiling when given)."""
import csv
import json
import sys

PEAK_TF, PEAK_TBS = 2500.0, 8.0


def short(name):
    name = name.replace("void ", "")
    return name[:name.index("(")] if "(" in name else name


def main():
    layers = json.load(open(sys.argv[1]))
    stats = {}
    for r in csv.DictReader(open(sys.argv[2])):
        stats[short(r["Name"])] = r
    traffic = json.load(open(sys.argv[3]))["kernels"] if len(sys.argv) > 3 and sys.argv[3] != "-" else {}
    sq = {}
    if len(sys.argv) > 4 and sys.argv[4] != "-":
        for r in csv.DictReader(open(sys.argv[4])):
            sq[r["kernel"]] = r
    calib = json.load(open(sys.argv[5])) if len(sys.argv) > 5 else None
    ceil = min(calib["copy_1_1_tbs"], calib["copy_1_3_tbs"]) if calib else None
    total = sum(float(r["TotalDurationNs"]) for r in stats.values()) or 1.0
    print("# config %s, %d maps per launch; time = rocprofv3 --kernel-trace --stats; traffic = --pmc FETCH_SIZE / WRITE_SIZE (separate passes)"
          % (layers["config"], layers["maps_per_launch"]))
    print("# hbm MB = WRITE_SIZE + FETCH_SIZE x the kernel's factor (tools/pmc_traffic.py): x 2, the guide's gfx950 correction for WIDE reads, unless calibrated (round 6: the ConvGRU kernel, x 1.52).")
    print("#   For the other STREAMED kernels (stream8p / 8q / 8g / 8, s2g: their patch fills are 64-byte segment reads, tallied at ~0.9-1.0 of their bytes) the x 2 makes `hbm MB` an UPPER bound:")
    print("#   part of their apparent over-fetch against `alg MB` is the counter, as profiles/r06_s2g_traffic.txt shows for conv2_1 (profiles/r06_fetch_calibration.txt).")
    print("%-58s %5s %9s %6s %8s %6s %9s %9s %6s %7s %6s %6s" % ("kernel", "calls", "us/launch", "share", "TFLOP/s", "mfma", "alg MB", "hbm MB", "TB/s", "hbm/8.0",
                                                          "Mfma%", "LDSc%"))
    for k, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
        us = float(r["AverageNs"]) / 1e3
        share = 100.0 * float(r["TotalDurationNs"]) / total
        if share < 0.3:
            continue
        a = layers["kernels"].get(k)
        t = traffic.get(k)
        q = sq.get(k)
        tf = a["alg_flops_per_launch"] / us / 1e6 if a and a["alg_flops_per_launch"] else None
        mb = t["hbm_bytes_per_launch"] / 1e6 if t else None
        tbs = t["hbm_bytes_per_launch"] / us / 1e6 if t else None
        print("%-58s %5d %9.1f %5.1f%% %8s %6s %9s %9s %6s %7s %6s %6s" % (
            k[:58], int(r["Calls"]), us, share, "%.0f" % tf if tf else "-", "%.2f" % (tf / PEAK_TF) if tf else "-",
            "%.1f" % (a["alg_bytes_per_launch"] / 1e6) if a else "-", "%.1f" % mb if mb else "-", "%.2f" % tbs if tbs else "-",
            "%.2f" % (tbs / PEAK_TBS) if tbs else "-", q["MfmaUtil_pct"] if q else "-", q["lds_bank_conflict_pct_of_lds_cycles"] if q else "-"))
    if ceil:
        print("# measured streaming ceiling of this box (calibration): 1:1 %.2f TB/s, 1:3 %.2f TB/s" % (calib["copy_1_1_tbs"], calib["copy_1_3_tbs"]))


if __name__ == "__main__":
    main()
