"""Print per-launch durations (us) of extend / shade from a rocprofv3 --kernel-trace CSV dir."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = [r for r in csv.DictReader(open(f)) if "k_extend" in r["Kernel_Name"] or "k_shade" in r["Kernel_Name"]]
print([(r["Kernel_Name"][10:18], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), r["Grid_Size_X"]) for r in rows[:n]])
