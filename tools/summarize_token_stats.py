import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "qs::" in r["Name"]]
for r in rows:
    name = r["Name"].split("(")[0][:100]
    print("   %-100s calls %4s  avg %8.1f us" % (name, r["Calls"], float(r["AverageNs"]) / 1e3))
