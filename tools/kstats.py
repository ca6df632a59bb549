"""Print the per-kernel summary rocprofv3 --kernel-trace --stats left under a directory."""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit("no *kernel_stats.csv under %s" % sys.argv[1])
for r in list(csv.DictReader(open(files[0])))[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print("%-90s calls %6s  avg %9.1f us  total %5s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
