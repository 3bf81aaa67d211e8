#!/usr/bin/env python
"""Print the kernel timeline (start offset, duration, gap) of the dfm kernels from a rocprofv3 kernel-trace CSV."""
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        n = r["Kernel_Name"]
        if "dfm::" not in n:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("dfm::")[1].split("<")[0], r.get("Queue_Id", "")))
rows.sort()
t0 = rows[0][0]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for s, e, n, q in rows[-last:]:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:7.1f} us  end {(e - t0) / 1e3:10.1f}  q={q:>3} {n}")
