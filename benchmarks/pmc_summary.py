"""Sum rocprofv3 --pmc counter_collection.csv files per (kernel, counter): python benchmarks/pmc_summary.py OUT.json CSV [CSV ...]
Kernel names are cut at the first '(' (template arguments kept); only kernels of this library (fv3hip::) are listed."""
import csv, json, re, sys
from collections import defaultdict

out, files = sys.argv[1], sys.argv[2:]
csv.field_size_limit(1 << 30)
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for path in files:
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "fv3hip" not in name:
                continue
            name = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", "").split("(")[0]).replace("fv3hip::", "")
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[name][row["Counter_Name"]] += 1
res = {k: {"dispatches": max(calls[k].values()), **{c: v for c, v in acc[k].items()}} for k in acc}
json.dump(res, open(out, "w"), indent=1)
for k, v in res.items():
    print(k, {c: (x if c == "dispatches" else round(x / v["dispatches"])) for c, x in v.items()})
