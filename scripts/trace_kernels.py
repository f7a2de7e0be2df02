#!/usr/bin/env python3
"""Per-kernel totals and a timeline out of a rocprofv3 --kernel-trace database (rocpd sqlite, what this ROCm writes by default).
usage: python scripts/trace_kernels.py <results.db> [--timeline [substring]] [--from-ms T]"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, sgpr_count, stream_id from kernels order by start"))
    if not rows:
        print("no kernels")
        return
    t0 = rows[0][1]
    short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        a = agg[short(r[0])]
        a[0] += 1
        a[1] += (r[2] - r[1]) / 1e6
    print("%10s %6s %10s  kernel" % ("total ms", "calls", "avg ms"))
    for k, (c, ms) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("%10.2f %6d %10.3f  %s" % (ms, c, ms / c, k))
    if "--timeline" in sys.argv:
        i = sys.argv.index("--timeline")
        sub = sys.argv[i + 1] if len(sys.argv) > i + 1 and not sys.argv[i + 1].startswith("--") else ""
        frm = float(sys.argv[sys.argv.index("--from-ms") + 1]) if "--from-ms" in sys.argv else 0.0
        print()
        for r in rows:
            at = (r[1] - t0) / 1e6
            if at >= frm and sub in r[0]:
                print("%10.2f +%9.3f ms  wgs %8d lds %6d scratch %4d vgpr %3d stream %2d  %s" % (at, (r[2] - r[1]) / 1e6, r[3] // max(1, r[4]), r[5], r[6], r[7], r[9], short(r[0])))


if __name__ == "__main__":
    main()
