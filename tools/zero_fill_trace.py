"""Which convolution kernels do MIOpen's zero fills (SubTensorOpWithScalar1d) belong to, and what do they cost?
Reads a rocprofv3 --kernel-trace csv of a few steps: every fill is paired with the next kernel on its queue.
usage: python tools/zero_fill_trace.py KERNEL_TRACE.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: (r.get("Queue_Id", ""), int(r["Start_Timestamp"])))
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for a, b in zip(rows, rows[1:]):
    if a["Kernel_Name"].startswith("SubTensorOpWithScalar1d") and a.get("Queue_Id") == b.get("Queue_Id"):
        fill = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
        main = (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3
        key = (b["Kernel_Name"][:110], b.get("Grid_Size", b.get("Grid_Size_X", "")))
        acc[key][0] += 1
        acc[key][1] += fill
        acc[key][2] += main
tot = sum(v[1] for v in acc.values())
print(f"zero fills: {sum(v[0] for v in acc.values())} calls, {tot / 1e3:.2f} ms in this trace")
for (name, grid), (n, fill, main) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{n:5d} x  fill {fill / n:8.1f} us  kernel {main / n:8.1f} us  grid {grid:>10s}  {name}")
