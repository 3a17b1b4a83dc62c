#!/usr/bin/env python3
"""Per-kernel totals of PMC counters over a whole run (e.g. bench.py): one rocprofv3 --pmc pass per counter, then
   tools/pmc_by_kernel.py <dir-of-counter-A> <dir-of-counter-B> ...   -> one row per kernel (timeline.py's short names), one column per counter.
Used to find LDS bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) kernel by kernel."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from timeline import short


def main():
    table, counters = {}, []
    for d in sys.argv[1:]:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        for r in csv.DictReader(open(files[0])):
            c = r["Counter_Name"]
            if c not in counters:
                counters.append(c)
            row = table.setdefault(short(r["Kernel_Name"]), {})
            row[c] = row.get(c, 0.0) + float(r["Counter_Value"])
            row.setdefault("_n_" + c, set()).add(r["Dispatch_Id"])
    print("%-34s %8s " % ("kernel", "launches") + " ".join("%22s" % c for c in counters))
    for k, row in sorted(table.items(), key=lambda kv: -kv[1].get(counters[0], 0.0)):
        n = max(len(row.get("_n_" + c, ())) for c in counters)
        print("%-34s %8d " % (k, n) + " ".join("%22.0f" % row.get(c, 0.0) for c in counters))


if __name__ == "__main__":
    main()
