#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counter_collection.csv files. usage: pmc_table.py file.csv [file.csv ...]"""
import csv, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("srh::", "").split("<")[0]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[k].add((path, row["Dispatch_Id"]))
        tot[k]["_dur_ns_" + path] += 0
for k, c in tot.items():
    print(k)
    for name, v in sorted(c.items()):
        if not name.startswith("_"):
            print("   %-24s %18.0f" % (name, v))
