"""Print the top rows of a rocprofv3 --kernel-trace --stats run: python scripts/prof_stats.py <dir> [rows]"""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not files:
    print("no kernel_stats.csv under", sys.argv[1]); sys.exit(1)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in list(csv.DictReader(open(files[0])))[:n]:
    print("%-100s calls %6s avg %10.1f us  %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
