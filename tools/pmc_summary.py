#!/usr/bin/env python3
"""Per-kernel mean of one PMC counter from a rocprofv3 `--pmc X` result database (rocpd sqlite).
Usage: tools/pmc_summary.py <results.db> [--schema]   -> JSON {kernel: {counter: {mean, launches}}}"""
import json
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    if "--schema" in sys.argv:
        for name, typ in c.execute("select name,type from sqlite_master where type in ('view','table')"):
            cols = [r[1] for r in c.execute("pragma table_info('%s')" % name)]
            print(typ, name, cols)
        return
    views = [r[0] for r in c.execute("select name from sqlite_master where type='view'")]
    if "counters_collection" not in views:
        raise SystemExit("no counters_collection view in %s" % db)
    cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    q = "select %s, counter_name, dispatch_id, sum(value) from counters_collection group by %s, counter_name, dispatch_id" % (kcol, kcol)
    per = {}
    for k, cn, did, v in c.execute(q):
        if "snpgpu" not in k:
            continue
        k = k.replace("snpgpu::", "")
        k = k[:k.find("(")] if "(" in k else k
        per.setdefault((k, cn), []).append(v)
    for (k, cn), vals in per.items():
        out.setdefault(k, {})[cn] = {"mean": sum(vals) / len(vals), "launches": len(vals)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
