#!/usr/bin/env python3
"""usage: writecal_summary.py <rocprofv3 output dir> <writecal stdout log>
WRITE_SIZE per kernel of tools/writecal.hip against the bytes each launch wrote: counter units per byte relative to the 16 B / lane
stream (the guide's calibration pattern) = the write amplification the counter reports for that pattern."""
import csv, glob, collections, re, sys
d, log = sys.argv[1], sys.argv[2]
wrote = {}
for l in open(log):
    m = re.match(r"WROTE (\S+)\s+(\d+) bytes per launch, ([\d.]+) ms", l)
    if m:
        wrote[m.group(1)] = (int(m.group(2)), float(m.group(3)))
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
key = {"k_stream16": "k_stream16", "k_stream4": "k_stream4", "k_stream1": "k_stream1", "k_tiles<0>": "k_tiles/plane", "k_tiles<1>": "k_tiles/xcd",
       "k_tiles<2>": "k_tiles/split"}
base = None
print("%-16s %14s %14s %10s %8s" % ("kernel", "bytes written", "WRITE_SIZE", "per byte", "x stream"))
for kn, name in key.items():
    v = [x for k, xs in acc.items() if kn in k.replace("void ", "") for x in xs]
    if not v or name not in wrote:
        continue
    c = sum(v) / len(v)
    per = c / wrote[name][0]
    base = base or per
    print("%-16s %14d %14.0f %10.3e %8.2f   (%.3f ms)" % (name, wrote[name][0], c, per, per / base, wrote[name][1]))
