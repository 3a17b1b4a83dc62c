#!/usr/bin/env python3
"""Per-launch average of one PMC counter for one kernel from a rocprofv3 --pmc output directory.
usage: tools/pmc_avg.py <dir> <kernel-name-substring> <COUNTER>"""
import csv
import glob
import os
import sys


def main():
    d, kern, counter = sys.argv[1:4]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("%s %s: no counter_collection.csv under %s" % (kern, counter, d))
        return
    vals = {}
    dur = {}
    for r in csv.DictReader(open(files[0])):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
            # one row per (dispatch, counter[, dimension]): sum the dimensions of a dispatch
            vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if not vals:
        print("%s %s: kernel or counter not found" % (kern, counter))
        return
    v = list(vals.values())
    tail = ""
    if dur:                 # duration of the profiled launches themselves (the clock a kernel held = GRBM_GUI_ACTIVE / 8 XCDs / this)
        tail = ", launch %.1f us under the profiler" % (sum(dur.values()) / len(dur))
    print("%s %s: launches %d, mean %.1f, min %.1f, max %.1f%s" % (kern, counter, len(v), sum(v) / len(v), min(v), max(v), tail))


if __name__ == "__main__":
    main()
