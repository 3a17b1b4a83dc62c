#!/usr/bin/env python3
"""Share of the GPU time of a run that is spent in kernels that are not this repository's (PyTorch elementwise / copy / pad / resize
launches, e.g. the KITTI pre/post-processing inside the captured graph): tools/kernel_share.py <rocprofv3 kernel_stats.csv>"""
import csv
import sys


def main():
    ours = theirs = 0.0
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        n, t = r["Name"], float(r["TotalDurationNs"])
        foreign = n.startswith("void at::") or "at::native" in n or "__amd_rocclr" in n or "hipcub" in n or "rocprim" in n
        rows.append((t, foreign, n))
        if foreign:
            theirs += t
        else:
            ours += t
    tot = ours + theirs
    print("kernel time %.1f ms: this repository's kernels %.1f %%, PyTorch / runtime kernels %.1f %%" % (tot / 1e6, 100 * ours / tot, 100 * theirs / tot))
    for t, foreign, n in sorted(rows, reverse=True)[:12]:
        print("  %6.2f %%  %s%s" % (100 * t / tot, "[torch] " if foreign else "", n[:110]))


if __name__ == "__main__":
    main()
