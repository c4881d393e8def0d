#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats` result database (rocpd sqlite) into the
per-kernel table that is committed under profiles/.  Usage:
    tools/rocprof_summary.py gpurun_out/r1_grm/grm_results.db [more.db ...] > profiles/xxx.txt
"""
import sqlite3
import sys


def short(name):
    name = name.replace("snpgpu::", "")
    cut = name.find("(")
    name = name[:cut] if cut > 0 else name
    if len(name) > 70:
        name = name[:67] + "..."
    return name


def main():
    for path in sys.argv[1:]:
        c = sqlite3.connect(path)
        rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        print("# %s  (rocprofv3 --kernel-trace --stats; durations in microseconds)" % path)
        print("%-72s %8s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        other = [0, 0.0, 0.0]
        for name, calls, tot, avg, pct in rows:
            if "snpgpu" in name:
                print("%-72s %8d %14.1f %14.1f %7.2f" % (short(name), calls, tot, avg, pct))
            else:
                other[0] += calls; other[1] += tot; other[2] += pct
        print("%-72s %8d %14.1f %14s %7.2f" % ("(torch data generation / memcpy / memset kernels)", other[0],
                                               other[1], "-", other[2]))
        print()


if __name__ == "__main__":
    main()
