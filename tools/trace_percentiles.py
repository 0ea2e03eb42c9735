"""Distribution of one kernel's launch durations in a rocprofv3 --kernel-trace csv directory.
usage: trace_percentiles.py <dir> <kernel name substring>"""
import csv
import glob
import os
import sys


def main():
    d, key = sys.argv[1], sys.argv[2]
    durs = []
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if key in row.get("Kernel_Name", ""):
                    durs.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    durs.sort()
    if not durs:
        print("no launches of", key)
        return
    ns = [x for _, x in durs]
    srt = sorted(ns)
    pick = lambda q: srt[min(len(srt) - 1, int(q * len(srt)))] / 1e3
    print("%s: %d launches; us: mean %.1f  min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (
        key, len(ns), sum(ns) / len(ns) / 1e3, srt[0] / 1e3, pick(0.1), pick(0.5), pick(0.9), srt[-1] / 1e3))
    print("in launch order, means of consecutive tenths (us): " + " ".join(
        "%.1f" % (sum(ns[i * len(ns) // 10:(i + 1) * len(ns) // 10]) / max(1, len(ns[i * len(ns) // 10:(i + 1) * len(ns) // 10])) / 1e3)
        for i in range(10)))


main()
