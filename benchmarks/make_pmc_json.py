"""Turn rocprofv3 --pmc counter_collection.csv files into the recorded-measurement JSONs bench.py quotes.

  python benchmarks/make_pmc_json.py traffic OUT.json KEY::=substring[:wide][:max] ... -- CSV [CSV ...]
      per kernel whose name contains `substring`: mean FETCH_SIZE / WRITE_SIZE per dispatch (KiB), and
      traffic_bytes_per_launch = 1024 * (FETCH_SIZE * (2 if wide else 1) + WRITE_SIZE): MI355X_MICROARCH.md (HBM section):
      on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane streaming reads, WRITE_SIZE is exact for 16-byte stores;
      other access widths are uncalibrated -> no `:wide`, and traffic is reported as null.
  python benchmarks/make_pmc_json.py valu OUT.json -- CSV [CSV ...]
      per fv3hip kernel: VALU instructions per wave (= per 64 columns), lane activity and the VALU issue-bound fraction
      (SQ_INSTS_VALU * 4 + SQ_INSTS_VALU_TRANS_F32 * 12) / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs).
"""
import csv, json, re, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    return re.sub(r"^void ", "", name.replace("(anonymous namespace)::", "").split("(")[0]).replace("fv3hip::", "")


def collect(files):
    acc = defaultdict(lambda: defaultdict(list))
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if "fv3hip" in row["Kernel_Name"]:
                    acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    mode, out = sys.argv[1], sys.argv[2]
    rest = sys.argv[3:]
    split = rest.index("--")
    opts, files = rest[:split], rest[split + 1:]
    acc = collect(files)
    res = {}
    if mode == "traffic":
        for opt in opts:
            key, sub = opt.split("::=", 1)
            flags = set()
            while sub.rsplit(":", 1)[-1] in ("wide", "max"):
                sub, flag = sub.rsplit(":", 1)
                flags.add(flag)
            wide = "wide" in flags
            names = [n for n in acc if sub in n]
            if not names:
                continue
            c = acc[names[0]]
            # `:max`: the largest dispatch of that name (a kernel launched on several workload sizes); else the mean
            mean = (lambda v: max(v) if v else None) if "max" in flags else (lambda v: sum(v) / len(v) if v else None)
            fetch, write = mean(c.get("FETCH_SIZE", [])), mean(c.get("WRITE_SIZE", []))
            res[key] = {"rocprof_kernel_name": names[0], "dispatches": len(c.get("FETCH_SIZE", [])),
                        "fetch_size_kib": fetch, "write_size_kib": write, "wide_loads": wide,
                        "traffic_bytes_per_launch": (1024 * (fetch * 2 + write)) if (wide and fetch is not None and write is not None) else None}
    else:
        for name, c in acc.items():
            if "SQ_INSTS_VALU" not in c or "SQ_WAVES" not in c:
                continue
            n = len(c["SQ_INSTS_VALU"])
            m = {k: sum(v) / len(v) for k, v in c.items()}
            entry = {"dispatches": n, **{k: round(v) for k, v in m.items()},
                     "valu_instructions_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"]}
            if "SQ_THREAD_CYCLES_VALU" in m:
                entry["lane_activity"] = m["SQ_THREAD_CYCLES_VALU"] / (64 * m["SQ_INSTS_VALU"])
            if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU_TRANS_F32" in m:
                entry["valu_issue_bound_fraction"] = (m["SQ_INSTS_VALU"] * 4 + m["SQ_INSTS_VALU_TRANS_F32"] * 12) / (1024 * m["GRBM_GUI_ACTIVE"] / 8)
            res[name] = entry
    json.dump({"command": " ".join(sys.argv), "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
