#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (any counters).  usage: pmc_generic.py <csv>... """
import collections, csv, re, sys
a = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        a[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({n for c in a.values() for n in c})
print("%-34s %8s " % ("kernel", "launches") + " ".join("%24s" % n for n in names))
for k, c in sorted(a.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CU_CYCLES", kv[1].get(names[0], [0])))):
    n = max(len(v) for v in c.values())
    print("%-34s %8d " % (k[:34], n) + " ".join("%24.4g" % (sum(c[m]) / len(c[m])) if m in c else "%24s" % "-" for m in names))
