"""Summarise rocprofv3 --pmc csv output: mean counter value per kernel name, and (with --traffic)
the HBM bytes per launch of every kernel from a FETCH_SIZE pass and a WRITE_SIZE pass.

    python tools/pmc_summary.py DIR [DIR ...]
    python tools/pmc_summary.py --traffic KEY OUT.json FETCH_DIR WRITE_DIR

FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE tallies a 128-byte read request as 64
bytes, so the read side is doubled; both raw values are kept next to the corrected total.
"""
import collections
import csv
import glob
import json
import os
import sys


def collect(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
            name = name.split("(")[0].split("<")[0].replace("void ", "").strip()
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def means(acc):
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main():
    if sys.argv[1] == "--traffic":
        key, out, fdir, wdir = sys.argv[2:6]
        fetch, write = means(collect(fdir)), means(collect(wdir))
        entry = {}
        for k in sorted(set(fetch) | set(write)):
            f_kb = fetch.get(k, {}).get("FETCH_SIZE")
            w_kb = write.get(k, {}).get("WRITE_SIZE")
            if f_kb is None or w_kb is None:
                continue
            entry[k] = {"FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
                        "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
                        "correction": "2*FETCH_SIZE + WRITE_SIZE, KB -> bytes"}
        data = {}
        if os.path.exists(out):
            data = json.load(open(out))
        data[key] = entry
        json.dump(data, open(out, "w"), indent=1, sort_keys=True)
        print(json.dumps(entry, indent=1))
        return
    for path in sys.argv[1:]:
        for k, d in means(collect(path)).items():
            print(k)
            for c, v in sorted(d.items()):
                print("   %-28s mean %.6g" % (c, v))


if __name__ == "__main__":
    main()
