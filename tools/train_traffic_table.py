#!/usr/bin/env python3
"""Row f-3's roofline evidence: per kernel of ONE training step -- launches, time (rocprofv3 --kernel-trace --stats), HBM bytes read and
written (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, corrected as tools/pmc_traffic.py: KiB -> bytes, FETCH_SIZE x 2 on gfx950) -- and the
step's totals against both roofs: 3 x forward FLOPs / time / 2.5 PFLOP/s and measured bytes / time / 8 TB/s.
usage: train_traffic_table.py <kernel_stats.csv> <steps traced> <fetch counter_collection.csv> <write counter_collection.csv> <steps counted> <maps> <out.json>"""
import collections
import csv
import json
import sys


def short(name):
    name = name.replace("void ", "")
    depth = 0
    if name.rstrip().endswith(")"):
        for i in range(len(name.rstrip()) - 1, -1, -1):
            if name[i] == ")":
                depth += 1
            elif name[i] == "(":
                depth -= 1
                if depth == 0:
                    return name[:i]
    return name


def pmc(path, counter):
    d = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    return d


def main():
    stats, steps_t, fpath, wpath, steps_c, maps, out = sys.argv[1], float(sys.argv[2]), sys.argv[3], sys.argv[4], float(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    rows = list(csv.DictReader(open(stats)))
    f, w = pmc(fpath, "FETCH_SIZE"), pmc(wpath, "WRITE_SIZE")
    table = []
    for r in rows:
        k = short(r["Name"])
        us = float(r["TotalDurationNs"]) / steps_t / 1e3
        rd, wr = 2.0 * 1024.0 * f.get(k, 0.0) / steps_c, 1024.0 * w.get(k, 0.0) / steps_c
        table.append((k, int(r["Calls"]) / steps_t, us, rd, wr))
    table.sort(key=lambda t: -t[2])
    tot_us = sum(t[2] for t in table)
    tot_rd, tot_wr = sum(t[3] for t in table), sum(t[4] for t in table)
    fwd_gflop = 155.8 / 5.0 * maps           # FaFNet: encoder + decoder + heads, SURVEY 8a
    print("# one FaFNet training step at %d maps (forward, loss, backward, Adam; eager launches): per kernel -- launches, time, HBM bytes (PMC)" % maps)
    print("%-64s %7s %10s %10s %10s %7s" % ("kernel", "x/step", "us/step", "read MB", "write MB", "TB/s"))
    for k, n, us, rd, wr in table[:40]:
        print("%-64s %7.1f %10.1f %10.1f %10.1f %7.2f" % (k[:64], n, us, rd / 1e6, wr / 1e6, (rd + wr) / max(us, 1e-9) / 1e6))
    print("# total: %.2f ms of kernel time per step, %.2f GB read + %.2f GB written = %.2f GB -> %.2f TB/s = %.2f of the 8 TB/s HBM roof"
          % (tot_us / 1e3, tot_rd / 1e9, tot_wr / 1e9, (tot_rd + tot_wr) / 1e9, (tot_rd + tot_wr) / tot_us / 1e6, (tot_rd + tot_wr) / tot_us / 1e6 / 8.0))
    print("# MFMA roof: 3 x %.1f GFLOP forward = %.0f GFLOP per step -> %.0f TFLOP/s = %.3f of 2.5 PFLOP/s.  The step is HBM-side: batch-statistics BatchNorm alone is"
          " 8 passes over every convolution's output (statistics, apply; gradient sums, dx: reads of x, dy and writes of y, dx), each layer's maps are written and read by"
          " forward, data-gradient and weight-gradient launches; the MFMA fraction is what that traffic leaves." % (fwd_gflop, 3 * fwd_gflop, 3 * fwd_gflop / (tot_us / 1e3), 3 * fwd_gflop / (tot_us / 1e3) / 2500.0))
    json.dump({"maps": maps, "kernel_ms_per_step": tot_us / 1e3, "hbm_read_bytes_per_step": tot_rd, "hbm_write_bytes_per_step": tot_wr,
               "hbm_tb_s": (tot_rd + tot_wr) / tot_us / 1e6, "frac_of_hbm_peak": (tot_rd + tot_wr) / tot_us / 1e6 / 8.0,
               "frac_of_mfma_peak": 3 * fwd_gflop / (tot_us / 1e3) / 2500.0,
               "source": "rocprofv3 --kernel-trace --stats (%d eager steps) and --pmc FETCH_SIZE / WRITE_SIZE (separate passes, %d steps; KiB -> bytes, FETCH_SIZE x 2)" % (steps_t, steps_c)},
              open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
