#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs, csv output) into per-kernel
HBM bytes per launch.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md section HBM: counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of a wide coalesced read stream -> doubled; WRITE_SIZE as is
(both calibrated here on bits_to_nhwc_bf16_kernel: known 10.5 MB read / 167.8 MB written per launch).
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"]][0] += 1
            d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d


def short(name):
    name = name.replace("void ", "")
    return name[:name.index("(")] if "(" in name else name


# Round 6: the x 2 holds for WIDE reads; kernels whose raw count was calibrated on their own access pattern carry their own factor (profiles/r06_fetch_calibration.txt;
# the same table as bench.py's FETCH_FACTOR).  Everything else keeps the guide's x 2 and says so ("fetch_factor_calibrated": false).
FETCH_FACTOR = {"conv3x3_stream8g_kernel<96, 2, false>": (168.6 * 1.10 + 142.2 * 2.0) / 309.6}


def main():
    f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in f:
        if k not in w or f[k][0] == 0 or w[k][0] == 0:
            continue
        raw = 1024.0 * f[k][1] / f[k][0]
        ff = FETCH_FACTOR.get(short(k), 2.0)
        fetch = ff * raw
        write = 1024.0 * w[k][1] / w[k][0]
        out[short(k)] = {"launches_profiled": f[k][0], "hbm_read_bytes_per_launch": fetch,
                         "hbm_write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
                         "fetch_size_raw_bytes_per_launch": raw, "fetch_factor": ff, "fetch_factor_calibrated": short(k) in FETCH_FACTOR}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --graph 0, default workload (128 frames/step as two 64-frame half-batches: every launch covers 320 (agent,frame) items; the gather kernel's 256x256 layer runs as two launches)",
               "corrections": "KiB -> bytes; FETCH_SIZE x fetch_factor: 2 (gfx950: wide reads are tallied at half their bytes) unless calibrated on the kernel's own access pattern (round 6: 64-byte segment reads are tallied at 0.91-1.0 of their bytes -- profiles/r06_fetch_calibration.txt); WRITE_SIZE x1",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print("%-60s read %8.1f MB  write %8.1f MB" % (k[:60], v["hbm_read_bytes_per_launch"] / 1e6, v["hbm_write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
