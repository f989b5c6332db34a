import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
sel = [r for r in rows if "k_ac_tile" in r["Kernel_Name"] or "k_permute" in r["Kernel_Name"] or "k_copy_out" in r["Kernel_Name"]]
for r in sel[-60:]:
    print("%-28s start %10.1f us  dur %7.1f us" % (r["Kernel_Name"][:28], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
