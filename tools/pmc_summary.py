#!/usr/bin/env python3
"""Average each collected counter over the dispatches of the step kernel."""
import csv
import glob
import json
import os
import sys

def main():
    out = sys.argv[1]
    match = sys.argv[2] if len(sys.argv) > 2 else "_step_kernel"
    acc = {}
    for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if match not in row.get("Kernel_Name", ""):
                    continue
                k = row["Counter_Name"]
                v = float(row["Counter_Value"])
                s = acc.setdefault(k, [0.0, 0, set()])
                s[0] += v
                s[2].add(row.get("Dispatch_Id"))
    res = {}
    for k, (tot, _, disp) in sorted(acc.items()):
        res[k] = tot / max(1, len(disp))
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main()
