#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes of scripts/gpu_profile.sh into per-kernel HBM bytes per launch.

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B per the rocprofv3 derived-counter
definition; on gfx950 FETCH_SIZE under-counts wide (16 B/lane) streaming reads by 2x
(MI355X_MICROARCH.md, HBM section), so the read side is corrected with the factor measured on the
read-bandwidth microbenchmark (known byte count) when that calibration run is present, else x2.
Prints JSON: {kernel: {fetch_bytes, write_bytes, hbm_bytes_per_launch, launches}, "_calibration": ...}."""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def counter_rows(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def per_kernel(rows, counter):
    acc = {}
    for r in rows:
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        if "dfm::" in name:
            name = name.split("dfm::")[1].split("<")[0]
        elif "k_dma" in name or "k_plain" in name:
            name = name.split("(")[0].split("<")[0].replace("void ", "")
        else:
            continue
        a = acc.setdefault(name, [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


fetch = per_kernel(counter_rows(os.path.join(out, "pmc_fetch")), "FETCH_SIZE")
write = per_kernel(counter_rows(os.path.join(out, "pmc_write")), "WRITE_SIZE")
calib = per_kernel(counter_rows(os.path.join(out, "pmc_calib")), "FETCH_SIZE")
UNIT = 1024.0
factor, cal_note = 2.0, "no calibration run: guide's x2 for 16-B/lane streaming reads"
known = 1024 * 500 * 200 * 8
for k, (tot, n) in calib.items():
    if "k_dma" in k and n:
        # k_dma launches read 4096 x 199680 B (b=1024) or 8192 x 99328 B (b=2048): both ~ 817.9 / 813.7 MB
        per = tot / n * UNIT
        factor = (4096 * 199680) / per if per > 0 else 2.0
        cal_note = f"k_dma reads 817.9 MB per launch; FETCH_SIZE*1024 reported {per / 1e6:.1f} MB -> factor {factor:.3f}"
        break
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (source_hash: bench.py quotes this file only for the source tree it was measured on)
res = {"_calibration": dict(fetch_factor=factor, note=cal_note, unit_bytes=UNIT), "_source_hash": bench.source_hash(),
       "_workload": os.environ.get("DFM_PMC_WORKLOAD", "pass:B1024:N200:T500:r8:m0.0")}
for k in sorted(set(fetch) | set(write)):
    fb = fetch.get(k, [0.0, 0]); wb = write.get(k, [0.0, 0])
    f_per = fb[0] / fb[1] * UNIT * factor if fb[1] else None
    w_per = wb[0] / wb[1] * UNIT if wb[1] else None
    res[k] = dict(fetch_bytes=f_per, write_bytes=w_per,
                  hbm_bytes_per_launch=(f_per or 0.0) + (w_per or 0.0), launches=fb[1] or wb[1])
print(json.dumps(res, indent=1))
