"""Per-kernel launch durations from a rocprofv3 --kernel-trace csv directory: for every fa:: kernel the durations (us) of its
launches in order, first launch of each shape dropped by the caller's reading.  usage: kstats.py dir [substr]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "fa"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not f: print("no kernel_trace.csv"); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if sub not in n: continue
    short = n.split("(")[0].replace("void ", "").replace("fa::", "")[:48]
    per.setdefault(short, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, ds in per.items():
    half = len(ds) // 2
    a, b = sorted(ds[:half]), sorted(ds[half:])
    print(f"{n}: {a[len(a)//2]:.0f} | {b[len(b)//2]:.0f} us;")
