"""rocprofv3 --pmc output directory -> per-kernel table {short kernel name: {counter: mean value per launch, "launches": n}}.

    python tools/pmc_table.py <dir> [<dir> ...]      # several passes (one counter set each) are merged
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"(\w+_kernel|\w+Kernel)\w*", name)
    base = name
    if "_ZN" in name or "(anonymous" in name or "GLOBAL__N" in name:
        m = re.search(r"\d+(\w+?_kernel)I?(.*?)E*v", name)
        if m:
            base = m.group(1) + "<" + m.group(2) + ">"
    return base[:90]


def main():
    table = {}
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                e = table.setdefault(k, {})
                c = r["Counter_Name"]
                s = e.setdefault(c, [0.0, 0])
                s[0] += float(r["Counter_Value"]); s[1] += 1
    out = {}
    for k, e in table.items():
        out[k] = {c: v[0] / v[1] for c, v in e.items()}
        out[k]["launches"] = max(v[1] for v in e.values())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
