#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace) into a per-kernel stats table
(the same columns as rocprofv3's kernel_stats.csv).  usage: rocpd_kernel_stats.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPR,AGPR,SGPR,LDS"]
    for r in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s,%s' % (r[0], r[1], r[2], r[3], r[4], r[5],
                                                                  100.0 * r[2] / total, r[6], r[7], r[8], r[9]))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
