"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, the mean of every counter and the
mean dispatch duration.  Usage: pmc_summary.py <dir-or-csv> [kernel-substring]"""
import collections
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    needle = sys.argv[2] if len(sys.argv) > 2 else ""
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"),
                                                           recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-60:]
            if needle and needle not in r["Kernel_Name"]:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k in agg:
        d = list(dur[k].values())
        print(f"{k}: dispatches={len(d)} mean_ns={sum(d) / len(d):.0f}")
        for c, v in sorted(agg[k].items()):
            print(f"    {c:32s} {sum(v) / len(v):.6g}")


if __name__ == "__main__":
    main()
