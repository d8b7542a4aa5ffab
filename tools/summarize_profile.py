"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the summaries committed under profiles/<round>/ and
update profiles/traffic.json (read by bench.py for roofline.traffic).

    python tools/summarize_profile.py <tag> <round-dir e.g. r01> <workload key>
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd, key = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
# the profile is only valid for the kernel sources it was measured on: bench.py compares this stamp with the built library's.
# profile_round.sh records the hash on the GPU box (gpurun_out/prof_<tag>/csrc_hash.txt); fall back to the current tree.
_h = os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "csrc_hash.txt")
CSRC_HASH = open(_h).read().strip() if os.path.exists(_h) else graft.load_package().native.csrc_hash()
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)


def counters(name):
    path = os.path.join(src, name, f"{name}_counter_collection.csv")
    acc = collections.defaultdict(list)
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if "pt_integrate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench.json")):
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
pmc = {}
for n in ("fetch", "write", "sq", "sq2", "tcc"):
    pmc.update(counters(n))
cal = {"FETCH_SIZE": counters("cal_fetch").get("FETCH_SIZE"), "WRITE_SIZE": counters("cal_write").get("WRITE_SIZE")}
kstats = [r for r in csv.DictReader(open(os.path.join(src, "stats", "stats_kernel_stats.csv"))) if "pt_integrate" in r["Name"]]
bench = json.load(open(os.path.join(src, "bench.json"))) if os.path.exists(os.path.join(src, "bench.json")) else {}
W, H = bench.get("config", {}).get("image", [1920, 1080])
pixels = W * H
# a step (frame) may be several launches (row stripes on separate streams): counters are averaged per launch above,
# so scale them to per-step values before comparing with per-frame byte counts
L = float(bench.get("roofline", {}).get("launches_per_step", 1))  # 1/64 when a launch pipelines 64 frames
pmc = {k: v * L for k, v in pmc.items()}
cal = {k: (v * L if v else v) for k, v in cal.items()}
# calibration: the depth-0 launch reads 16 B and writes 16 B per pixel, nothing else of size
known = 16.0 * pixels
fetch_factor = known / (cal["FETCH_SIZE"] * 1024.0) if cal["FETCH_SIZE"] else None
write_factor = known / (cal["WRITE_SIZE"] * 1024.0) if cal["WRITE_SIZE"] else None
summary = {
    "tag": tag, "workload": key, "csrc_hash": CSRC_HASH, "launches_per_step": L,
    "kernel_avg_ns_rocprof": float(kstats[0]["AverageNs"]) if kstats else None,
    "kernel_calls": int(kstats[0]["Calls"]) if kstats else None,
    "frames_per_launch": bench.get("roofline", {}).get("frames_per_launch", 1),
    "kernel_avg_ns_rocprof_per_frame": (float(kstats[0]["AverageNs"]) * L if kstats else None),
    "bench_kernel_ms_hip_events": bench.get("roofline", {}).get("kernel_ms"),
    "pmc_mean_per_step": pmc,
    "calibration_depth0": {"known_bytes_each_way": known, "FETCH_SIZE_KiB": cal["FETCH_SIZE"], "WRITE_SIZE_KiB": cal["WRITE_SIZE"],
                           "fetch_bytes_per_counted_byte": fetch_factor, "write_bytes_per_counted_byte": write_factor},
}
if pmc.get("FETCH_SIZE") and pmc.get("WRITE_SIZE") and fetch_factor and write_factor:
    # MI355X_MICROARCH.md section HBM: FETCH_SIZE under-counts wide coalesced reads on gfx950 (x2); the factor measured on
    # our own access pattern (calibration above) is applied instead of assuming it.
    rd = pmc["FETCH_SIZE"] * 1024.0 * fetch_factor
    wr = pmc["WRITE_SIZE"] * 1024.0 * write_factor
    summary["hbm_traffic_bytes_per_step"] = {"read": rd, "write": wr, "total": rd + wr,
                                               "algorithmic": 32.0 * pixels, "ratio_to_algorithmic": (rd + wr) / (32.0 * pixels)}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(tpath)) if os.path.exists(tpath) else {}
    t[key] = {"value": round(rd + wr), "csrc_hash": CSRC_HASH}
    json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
if pmc.get("SQ_INSTS_VALU"):
    # VALU wave-instructions per step: the numerator of bench.py's roofline.valu.issue (peak: tools/ubench2.hip)
    vpath = os.path.join(ROOT, "profiles", "valu_insts.json")
    v = json.load(open(vpath)) if os.path.exists(vpath) else {}
    v[key] = {"value": round(pmc["SQ_INSTS_VALU"]), "csrc_hash": CSRC_HASH}
    json.dump(v, open(vpath, "w"), indent=1, sort_keys=True)
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
