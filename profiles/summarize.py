"""Turn the raw rocprofv3 output of profiles/collect.sh (gpurun_out/prof/<round>/) into the summaries committed under
profiles/<round>/: one kernel-stats csv per workload and pmc_traffic.json (HBM bytes per launch per kernel).

Counter handling follows the guide's HBM section: FETCH_SIZE and WRITE_SIZE come from separate passes; rocprofv3
reports them in KB summed over the XCDs' rows of one dispatch; on gfx950 FETCH_SIZE is exact only for narrow accesses
and reads half the bytes of wide (16 B per lane) coalesced streams - this kernel's accesses are 8 B per lane, so the
value is taken as reported (uncalibrated, stated in the json)."""
import csv, glob, json, os, sys
from collections import defaultdict

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.abspath(__file__))
raw = os.path.join(root, "..", "gpurun_out", "prof", rnd)
dst = os.path.join(root, rnd)
os.makedirs(dst, exist_ok=True)


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


traffic = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 "
                   "--no-also; KB per dispatch as reported, averaged over the dispatches of the run; 8-byte-per-lane accesses: "
                   "FETCH_SIZE uncalibrated for this width on gfx950 (exact for narrow, 1/2 for 16 B per lane streams)",
           "workloads": {}}
for wdir in sorted(glob.glob(os.path.join(raw, "*", ""))):
    w = os.path.basename(os.path.dirname(wdir))
    st = find(os.path.join(wdir, "stats"), "kernel_stats.csv")
    if st:
        rows = list(csv.reader(open(st)))
        with open(os.path.join(dst, f"{w.lower()}_kernel_stats.csv"), "w", newline="") as f:
            csv.writer(f).writerows(rows)
    per = defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        cc = find(os.path.join(wdir, c), "counter_collection.csv")
        if not cc: continue
        acc = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(cc)):
            if r["Counter_Name"] != c: continue
            acc[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, disp in acc.items():
            per[k][c] = sum(disp.values()) / len(disp)
    if per:
        traffic["workloads"][w] = {}
        for k, v in per.items():
            short = k.split("::")[-1]
            f_kb, w_kb = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
            traffic["workloads"][w][short] = {"fetch_KB": f_kb, "write_KB": w_kb, "hbm_bytes_per_launch": (f_kb + w_kb) * 1024.0}
import hashlib
# the kernel sources these counters belong to: bench.py reports them only for exactly this code (the C entry points --
# solve_api.hip, api.hip, mpcx_host.hpp -- are not kernel sources: editing them leaves the profiles valid)
KERNEL_SOURCES = ("solve.hip", "solve2w.hip", "solve_lds.hip", "solve_tp.hip", "solve_tp.hpp", "solve_kernel2w.hpp", "solve_common.hpp", "solve_layout.hpp", "solve_phases.hpp",
                  "solve_riccati.hpp", "solve_driver.hpp", "discretize.hip", "mpcx_device.hpp")
source_hashes = {f: hashlib.sha256(open(os.path.join(root, "..", "mpconstellation_amd", "csrc", f), "rb").read()).hexdigest()
                 for f in KERNEL_SOURCES}
build_flags = None
try:
    sys.path.insert(0, os.path.join(root, ".."))
    from mpconstellation_amd import build as _b
    build_flags = " ".join(_b.FLAGS)
except Exception:
    pass
if traffic["workloads"]:
    traffic["kernel_source_sha256"] = source_hashes
    traffic["build_flags"] = build_flags
    traffic["round"] = rnd
    json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)

# SQ passes: per kernel, counter values summed over the rows of one dispatch, averaged over dispatches
sq = {"note": "rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes), bench.py --steps 1 --warmup 1 --no-also; wave-level "
              "instruction counts; SQ_WAIT_* / SQ_ACTIVE_* / SQ_BUSY_CYCLES in quad-cycles summed over SEs/XCDs as reported",
      "workloads": {}}
for wdir in sorted(glob.glob(os.path.join(raw, "*", ""))):
    w = os.path.basename(os.path.dirname(wdir))
    per = defaultdict(dict)
    for c in ("SQ_A", "SQ_B"):
        cc = find(os.path.join(wdir, c), "counter_collection.csv")
        if not cc: continue
        acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
        for r in csv.DictReader(open(cc)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, cnt in acc.items():
            for name, disp in cnt.items():
                per[k.split("::")[-1]][name] = sum(disp.values()) / len(disp)
    keep = {k: v for k, v in per.items() if "solve_kernel" in k or "discretize_kernel" in k}
    if keep:
        for k, v in keep.items():
            if "SQ_INSTS_VALU_FMA_F64" in v:
                v["fp64_flop_64lanes"] = 64.0 * (2 * v["SQ_INSTS_VALU_FMA_F64"] + v.get("SQ_INSTS_VALU_MUL_F64", 0) + v.get("SQ_INSTS_VALU_ADD_F64", 0))
        sq["workloads"][w] = keep
if sq["workloads"]:
    sq["kernel_source_sha256"] = source_hashes
    sq["build_flags"] = build_flags
    sq["round"] = rnd
    json.dump(sq, open(os.path.join(dst, "pmc_sq.json"), "w"), indent=1)
    print(json.dumps(sq["workloads"], indent=1))
print(json.dumps(traffic["workloads"], indent=1))
for f in sorted(glob.glob(os.path.join(dst, "*_kernel_stats.csv"))):
    print("==", os.path.basename(f)); print(open(f).read())
