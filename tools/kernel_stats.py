"""Calls and average duration of every vsl_ kernel in the rocprofv3 kernel statistics under a directory: tools/kernel_stats.py <dir>"""
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "vsl_" in r["Name"]:
            print("%-64s calls %6s  avg %8.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) * 1e-3))
