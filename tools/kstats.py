"""Print the top rows of the newest rocprofv3 *kernel_stats.csv under a directory."""
import csv, glob, os, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
for r in list(csv.DictReader(open(f)))[:n]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"]) / 1e3:9.2f}  {r["Percentage"]}%')
