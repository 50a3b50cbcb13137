"""profiles/r03_pmc_mappm.json: occupancy, stall and instruction counters of the remap sweep kernels.

  python benchmarks/make_pmc_mappm_json.py OUT.json LABEL=DIR [LABEL=DIR ...]

DIR holds the rocprofv3 --pmc passes of benchmarks/remap_sweep_timing.py (benchmarks/collect_profiles_r03.sh: pmc_mappm1/2/3);
per kernel instantiation the mean over its dispatches.  Static resources (registers, LDS) come from the code object of
fv3net_amd/csrc/remap.hip compiled here (hipcc --save-temps).  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles
(MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csv.field_size_limit(1 << 30)


def static_resources():
    d = tempfile.mkdtemp()
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value -mllvm -pragma-unroll-threshold=262144".split()
    subprocess.run(["/opt/rocm/bin/hipcc", *flags, "--save-temps=obj", "-DFV3HIP_REMAP_PART_SWEEP", "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "fv3net_amd", "csrc", "remap.hip"),
                    "-o", os.path.join(d, "x.o")], check=True, capture_output=True)
    asm = open([os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s")][0]).read()
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.vgpr_count:\s+(\d+)", asm, re.S):
        name, body, vgpr = m.group(1), m.group(2), int(m.group(3))
        t = re.search(r"mappm_sweep_kernelI([fd])Li(\d)ELi(\d)ELb([01])ELb([01])ELb([01])E", name)
        if not t:
            continue
        tf = lambda g: "true" if t.group(g) == "1" else "false"
        key = "mappm_sweep_kernel<%s, %s, %s, %s, %s, %s>" % ("float" if t.group(1) == "f" else "double", t.group(2), t.group(3), tf(4), tf(5), tf(6))
        sg = re.search(r"\.sgpr_count:\s+(\d+)", body)
        out[key] = {"vgpr": vgpr, "sgpr": int(sg.group(1)) if sg else None, "waves_per_simd_by_vgpr": min(8, 512 // (-(-vgpr // 8) * 8))}
    return out


def collect(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mappm_sweep" not in k:
                continue
            k = re.sub(r"^void fv3hip::\(anonymous namespace\)::", "", k)
            k = re.sub(r"\(.*", "", k)
            per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
            per[(k, r["Dispatch_Id"])]["_lds"] = float(r.get("LDS_Block_Size", 0) or 0)
        for (k, _), cs in per.items():
            for c, v in cs.items():
                agg[k][c].append(v)
    return agg


def main():
    out_path, sets = sys.argv[1], sys.argv[2:]
    static = static_resources()
    res = {"command": "benchmarks/collect_profiles_r03.sh (passes pmc_mappm1/2/3 of benchmarks/remap_sweep_timing.py); " + " ".join(sys.argv),
           "workload": "C384: 884736 columns (13824 waves of 64 columns), km = kn = 79, coarse-pressure target read on its own grid (TGT: the whole "
                       "target table in LDS); 1 or 4 fields per sweep, float32 / float64 inputs; means over the dispatches of an instantiation",
           "how_to_read": "waves_per_simd = SQ_WAVE_CYCLES * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8): average resident waves; wait_any / "
                          "wait_inst / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (parked at s_waitcnt, "
                          "issue stall, issuing: they add to ~1); valu_per_level = SQ_INSTS_VALU / SQ_WAVES / 79; valu_pipe_busy_4cyc = "
                          "SQ_INSTS_VALU * 4 / (1024 * GRBM_GUI_ACTIVE / 8), i.e. against ONE instruction per 4 cycles per SIMD -- "
                          "benchmarks/valu_ubench measures 2 cycles for plain 32-bit VALU and 4 for packed / 64-bit / 3-operand ones, "
                          "so the pipe itself is less busy than this figure; write_req_bytes = output bytes / TCP_TCC_WRITE_REQ: how well "
                          "the result rows leave (64 = whole 64-byte requests)",
           "data_sets": {}}
    for s in sets:
        label, d = s.split("=", 1)
        agg = collect(d)
        kernels = {}
        for k in sorted(agg):
            m = {c: sum(v) / len(v) for c, v in agg[k].items()}
            e = {"dispatches": max(len(v) for v in agg[k].values()), **{c: round(v) for c, v in m.items() if not c.startswith("_")}}
            e["static"] = static.get(k)
            # dynamic LDS of the launch (remap.hip launch_sweep2): the target table (kn + 1) x 8 floats (TGT) or the 16-row ring,
            # plus 8 result rows per field
            nf_ = 4 if ", 2, 2," in k else 1
            lds = (80 * 8 if k.endswith("true>") else 16 * 64) * 4 + nf_ * 8 * 64 * 4
            e["lds_bytes_per_wave"] = lds
            e["waves_per_simd_by_lds"] = (160 * 1024 // lds) / 4
            wc, gui = m.get("SQ_WAVE_CYCLES"), m.get("GRBM_GUI_ACTIVE")
            if wc and gui:
                e["waves_per_simd"] = wc * 4 / (1024 * gui / 8)
            if wc:
                for name, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY"),
                                ("active_valu", "SQ_ACTIVE_INST_VALU"), ("active_scalar", "SQ_ACTIVE_INST_SCA"), ("active_lds", "SQ_ACTIVE_INST_LDS")):
                    if c in m:
                        e[name] = m[c] / wc
            if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m:
                e["valu_per_level"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"] / 79
                if "SQ_INSTS_SALU" in m:
                    e["salu_per_level"] = m["SQ_INSTS_SALU"] / m["SQ_WAVES"] / 79
                if gui:
                    e["valu_pipe_busy_4cyc"] = m["SQ_INSTS_VALU"] * 4 / (1024 * gui / 8)
            if "TCP_TCC_WRITE_REQ_sum" in m:
                nf = 4 if ", 2, 2," in k else 1
                e["write_req_bytes"] = 884736 * 79 * nf * 4 / m["TCP_TCC_WRITE_REQ_sum"]
            kernels[k] = e
        res["data_sets"][label] = kernels
    json.dump(res, open(out_path, "w"), indent=1)
    for label, ks in res["data_sets"].items():
        for k, e in ks.items():
            print(label, k, {x: (round(e[x], 3) if isinstance(e.get(x), float) else e.get(x)) for x in
                             ("waves_per_simd", "wait_any", "wait_inst", "active", "valu_per_level", "valu_pipe_busy_4cyc", "write_req_bytes", "lds_bytes_per_wave")},
                  (e.get("static") or {}).get("vgpr"))


if __name__ == "__main__":
    main()
