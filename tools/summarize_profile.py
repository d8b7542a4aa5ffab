"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the summaries committed under profiles/<round>/ and
update profiles/traffic.json + profiles/valu_insts.json (read by bench.py for roofline.traffic / roofline.valu_issue; every
entry is stamped with the hash of the kernel sources it was measured on).

    python tools/summarize_profile.py <tag> <round-dir e.g. r02> <workload key> [calibration tag]

Counters are summed over ALL integrator launches of the profiled run and divided by the frames it rendered (the library
batches frames into launches adaptively).  HBM traffic is reported twice: (a) FETCH_SIZE / WRITE_SIZE corrected with the
factors measured on a depth-0 calibration run of known traffic (the guide's prescription; `calibration tag` lets several
workloads share one calibration), (b) exact bytes from the L2's memory-side request counters by request size.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd, key = sys.argv[1], sys.argv[2], sys.argv[3]
cal_tag = sys.argv[4] if len(sys.argv) > 4 else tag
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
cal_src = os.path.join(ROOT, "gpurun_out", f"prof_{cal_tag}")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
# the profile is only valid for the kernel sources it was measured on: bench.py compares this stamp with the built library's
_h = os.path.join(src, "csrc_hash.txt")
CSRC_HASH = open(_h).read().strip() if os.path.exists(_h) else graft.load_package().native.csrc_hash()


def frames_of(d):
    p = os.path.join(d, "frames.txt")
    return int(open(p).read()) if os.path.exists(p) else None


def counters(d, name, frames):
    """per-frame value of every counter of pass `name`: sum over all integrator launches / frames"""
    path = os.path.join(d, name, f"{name}_counter_collection.csv")
    acc = collections.defaultdict(float)
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if "pt_integrate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    return {k: v / frames for k, v in acc.items()}


frames = frames_of(src)
shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench.json")):
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
pmc = {}
for n in ("fetch", "write", "rd", "wr", "sq", "sq2"):
    pmc.update(counters(src, n, frames))
cal_frames = frames_of(cal_src)
cal = {"FETCH_SIZE": counters(cal_src, "cal_fetch", cal_frames).get("FETCH_SIZE"),
       "WRITE_SIZE": counters(cal_src, "cal_write", cal_frames).get("WRITE_SIZE")}
kstats = [r for r in csv.DictReader(open(os.path.join(src, "stats", "stats_kernel_stats.csv"))) if "pt_integrate" in r["Name"]]
bench = json.load(open(os.path.join(src, "bench.json"))) if os.path.exists(os.path.join(src, "bench.json")) else {}
W, H = bench.get("config", {}).get("image", [1920, 1080])
pixels = W * H
total_ns = sum(float(r["TotalDurationNs"]) for r in kstats)
calls = sum(int(r["Calls"]) for r in kstats)
# calibration: the depth-0 launch reads 16 B and writes 16 B per pixel, nothing else of size (the calibration run's OWN image size:
# several workloads share one calibration, and they need not have its resolution)
_cb = os.path.join(cal_src, "bench.json")
_cal_img = json.load(open(_cb)).get("config", {}).get("image", [1920, 1080]) if os.path.exists(_cb) else [W, H]
known = 16.0 * _cal_img[0] * _cal_img[1]
fetch_factor = known / (cal["FETCH_SIZE"] * 1024.0) if cal["FETCH_SIZE"] else None
write_factor = known / (cal["WRITE_SIZE"] * 1024.0) if cal["WRITE_SIZE"] else None
if fetch_factor is None or write_factor is None:
    # the calibration run's raw counters are not here (gpurun_out/ does not travel to the GPU box): take the factors from the committed
    # summary of the calibration tag, which does
    _cs = os.path.join(dst, f"{cal_tag}_summary.json")
    if os.path.exists(_cs):
        _c = json.load(open(_cs)).get("calibration_depth0", {})
        fetch_factor, write_factor = _c.get("fetch_bytes_per_counted_byte"), _c.get("write_bytes_per_counted_byte")
        cal = {"FETCH_SIZE": _c.get("FETCH_SIZE_KiB"), "WRITE_SIZE": _c.get("WRITE_SIZE_KiB")}
# Consecutive pipelined launches overlap pairwise (launch chaining: two streams, ordered per pixel by tags), so the SUM of the
# launch durations counts the overlapped time twice.  The machine time per frame is the UNION of the launch intervals / frames.
launches_path = os.path.join(src, "stats", "integrator_launches.csv")
union_ns = None
if os.path.exists(launches_path):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(launches_path)))
    union_ns, cur_s, cur_e = 0, None, None
    for s_, e_ in iv:
        if cur_e is None or s_ > cur_e:
            if cur_e is not None:
                union_ns += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    if cur_e is not None:
        union_ns += cur_e - cur_s
    shutil.copy(launches_path, os.path.join(dst, f"{tag}_integrator_launches.csv"))
summary = {
    "tag": tag, "workload": key, "csrc_hash": CSRC_HASH, "frames_profiled": frames,
    "kernel_ns_per_frame_rocprof_union_of_launch_intervals": (union_ns / frames if union_ns and frames else None),
    "launch_overlap_factor": (total_ns / union_ns if union_ns else None),
    "note_sum_vs_union": "since round 3 every pipelined launch is enqueued as soon as its predecessor is resident (back-pressure chaining) and waits beside "
                         "it for the wavefront slots its drain frees; rocprofv3's start timestamp is the moment the command processor begins the dispatch, so "
                         "a launch's duration includes that wait and the SUM of durations is ~2x the elapsed time: the UNION of the launch intervals / frames "
                         "is the GPU time per frame (it agrees with bench.py's HIP-event kernel_ms)",
    "integrator_launches": calls, "kernels": [{"name": r["Name"][:96], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])} for r in kstats],
    "kernel_ns_per_frame_rocprof_sum_of_launch_durations": total_ns / frames if frames else None,
    "frames_per_launch_mean": frames / calls if calls else None,
    "bench_kernel_ms_hip_events": bench.get("roofline", {}).get("kernel_ms"),
    "bench_value_msamples": bench.get("value"),
    "pmc_per_frame": pmc,
    "calibration_depth0": {"from": cal_tag, "known_bytes_each_way": known, "FETCH_SIZE_KiB": cal["FETCH_SIZE"], "WRITE_SIZE_KiB": cal["WRITE_SIZE"],
                           "fetch_bytes_per_counted_byte": fetch_factor, "write_bytes_per_counted_byte": write_factor},
}
algorithmic = 32.0 * pixels * bench.get("config", {}).get("spp", 1) ** 0  # 32 B per pixel per FRAME whatever the spp
if pmc.get("FETCH_SIZE") and pmc.get("WRITE_SIZE") and fetch_factor and write_factor:
    rd = pmc["FETCH_SIZE"] * 1024.0 * fetch_factor
    wr = pmc["WRITE_SIZE"] * 1024.0 * write_factor
    summary["hbm_traffic_bytes_per_frame"] = {"read": rd, "write": wr, "total": rd + wr, "algorithmic": algorithmic,
                                               "ratio_to_algorithmic": (rd + wr) / algorithmic}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    t = json.load(open(tpath)) if os.path.exists(tpath) else {}
    t[key] = {"value": round(rd + wr), "csrc_hash": CSRC_HASH}
    json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
if "TCC_EA0_RDREQ_sum" in pmc and "TCC_EA0_WRREQ_sum" in pmc:
    rd = 32 * pmc["TCC_EA0_RDREQ_32B_sum"] + 64 * pmc["TCC_EA0_RDREQ_64B_sum"] + 128 * pmc["TCC_EA0_RDREQ_128B_sum"]
    wr = 64 * pmc["TCC_EA0_WRREQ_64B_sum"] + 32 * (pmc["TCC_EA0_WRREQ_sum"] - pmc["TCC_EA0_WRREQ_64B_sum"])
    summary["hbm_traffic_exact_by_request_size"] = {"read": rd, "write": wr, "total": rd + wr, "ratio_to_algorithmic": (rd + wr) / algorithmic,
                                                     "l2_hit_rate": pmc["TCC_HIT_sum"] / max(1.0, pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])}
if pmc.get("SQ_INSTS_VALU"):
    vpath = os.path.join(ROOT, "profiles", "valu_insts.json")
    v = json.load(open(vpath)) if os.path.exists(vpath) else {}
    v[key] = {"value": round(pmc["SQ_INSTS_VALU"]), "csrc_hash": CSRC_HASH}
    # the other instruction classes that take issue slots of the same wavefronts (round 4: a scalar instruction costs about what a vector one does)
    other = {k.lower()[len("sq_insts_"):]: round(pmc[k]) for k in ("SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_SMEM") if pmc.get(k)}
    if other:
        v[key]["other"] = other
    json.dump(v, open(vpath, "w"), indent=1, sort_keys=True)
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "pmc_per_frame"}, indent=1))
