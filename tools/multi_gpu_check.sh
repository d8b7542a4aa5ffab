#!/bin/bash
# Launch-readiness check for a box with N > 1 MI355X (round 5; nothing with N > 1 has run on hardware in rounds 1-4: no such box
# was available).  ONE command that exercises everything the SCALE run relies on and fails loudly:
#   tools/multi_gpu_check.sh [out-dir]
#   1. the five real-peer GPU tests (skipped on a one-GPU box): pt_create_multi over distinct devices (peer access is REQUIRED here: a group whose
#      gather is staged through the host fails the check, pt_multi_gather_is_direct), 1080p over all devices, the hand-over stress on distinct devices, bench.py's self-started
#      RCCL ranks;
#   2. bench.py --gpus N for N = 1, 2, 4, 8 (as far as the box has devices), checked per line: n_gpus == ranks == N, finite image with
#      alpha = 1, in_process_group.equals_rccl_gather_bit_for_bit (N > 1), configs[3] (4K over N) present for N > 1;
#   3. the N = 1 value against the round's committed BENCH line (profiles/<round>/bench_default.json when present): within 3 %.
# Exit code 0 only if every step passed.  Writes <out-dir>/multi_gpu_check.json (default gpurun_out/multi_gpu).
#   tools/multi_gpu_check.sh --dry-run [out-dir]    one-GPU DRY RUN (round 6; run by tests/test_gpu_round6.py so that the JSON checks below
#      cannot rot): bench.py --gpus 2 --share-gpu (two gloo ranks on cuda:0, short) through exactly the per-line checks of step 2 — n_gpus is
#      1 there by construction, which the dry run accepts and says so; steps 1 and 3 are skipped.
set -u
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
DRY=0; if [ "${1:-}" = "--dry-run" ]; then DRY=1; shift; fi
OUT=${1:-gpurun_out/multi_gpu}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
N=$(python - <<'PY'
import __graft_entry__ as g
print(g.load_package().native.load().pt_device_count())
PY
)
echo "devices: $N"
fail=0
if [ "$DRY" = 1 ]; then
  timeout 900 python bench.py --gpus 2 --share-gpu --steps 128 --warmup 64 --steady-ms 0 --no-cpu-baseline > "$OUT/bench_g2.json" 2> "$OUT/bench_g2.err" || { echo "FAIL: bench.py --gpus 2 --share-gpu (see $OUT/bench_g2.err)"; fail=1; }
elif [ "$N" -lt 2 ]; then
  echo "this box has $N HIP device(s): the multi-GPU check needs at least 2 (nothing run)"; exit 3
fi
if [ "$DRY" = 0 ]; then
timeout 3000 python -m pytest tests/test_gpu_round3.py -q -m gpu -k "distinct_devices or over_all_devices" -rs > "$OUT/peer_tests.log" 2>&1 || fail=1
tail -n 5 "$OUT/peer_tests.log"
grep -q "skipped" "$OUT/peer_tests.log" && { echo "FAIL: a real-peer test was skipped on a $N-device box"; fail=1; }
for n in 1 2 4 8; do
  [ "$n" -gt "$N" ] && continue
  timeout 1200 python bench.py --gpus $n > "$OUT/bench_g$n.json" 2> "$OUT/bench_g$n.err" || { echo "FAIL: bench.py --gpus $n (see $OUT/bench_g$n.err)"; fail=1; }
done
fi
python - "$OUT" "$N" "$DRY" <<'PY' || fail=1
import glob, json, os, sys
out, ndev, dry = sys.argv[1], int(sys.argv[2]), sys.argv[3] == "1"
if dry:
    ndev = 2  # (two ranks on one device: the checks below run on bench_g2.json; n_gpus is 1 by construction)
ok, rows = True, []
def last_line(p):
    lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None
for n in (1, 2, 4, 8):
    p = os.path.join(out, f"bench_g{n}.json")
    if n > ndev or not os.path.exists(p):
        continue
    d = last_line(p)
    if d is None:
        print(f"FAIL: no JSON line from --gpus {n}"); ok = False; continue
    errs = []
    if (d["n_gpus"] != n and not dry) or d["ranks"] != n: errs.append(f"n_gpus {d['n_gpus']} ranks {d['ranks']} != {n}")
    if dry and d["n_gpus"] != 1: errs.append(f"dry run: n_gpus {d['n_gpus']} != 1")
    if not (d["checks"]["finite"] and d["checks"]["alpha_one"]): errs.append("image not finite / alpha != 1")
    if n > 1:
        g = d.get("in_process_group") or {}
        if not g.get("equals_rccl_gather_bit_for_bit"): errs.append("in-process group handle != RCCL-gathered image (or not run)")
        if not g.get("gather_is_direct"): errs.append("the group handle's gather is staged through the host (no peer access): not a scaling configuration")
        if "configs3_4k" not in d: errs.append("no configs[3] (4K over N) measurement")
    rows.append({"gpus": n, "value": d["value"], "ms_per_step": d["ms_per_step"], "configs3_4k": (d.get("configs3_4k") or {}).get("value"), "errors": errs})
    print(f"--gpus {n}: {d['value']:.0f} Msamples/s, {d['ms_per_step']:.4f} ms per frame" + (f", 4K over {n}: {d['configs3_4k']['value']:.0f}" if "configs3_4k" in d else "") + ("  FAIL: " + "; ".join(errs) if errs else "  ok"))
    ok = ok and not errs
ref = sorted(glob.glob("profiles/r*/bench_default.json"))
if dry:
    print("DRY RUN (two ranks on one GPU): per-line checks only" + ("" if ok and rows else "  FAIL: no checked line"))
    ok = ok and bool(rows)
if rows and rows[0]["gpus"] == 1 and ref and not dry:
    try:
        want = last_line(ref[-1])["value"]
        rel = abs(rows[0]["value"] - want) / want
        print(f"N = 1 value vs {ref[-1]}: {rows[0]['value']:.0f} vs {want:.0f} ({100 * rel:.1f} %)" + ("" if rel <= 0.03 else "  FAIL: more than 3 % apart"))
        ok = ok and rel <= 0.03
    except Exception as e:
        print("could not compare with the committed BENCH line:", e)
if len(rows) > 1:
    base = rows[0]["value"]
    for r in rows[1:]:
        print(f"  scaling efficiency at {r['gpus']} GPUs (strong, the metric's 1080p image): {100 * r['value'] / (base * r['gpus']):.1f} %")
json.dump({"devices": ndev, "runs": rows, "ok": ok}, open(os.path.join(out, "multi_gpu_check.json"), "w"), indent=1)
sys.exit(0 if ok else 1)
PY
[ $fail = 0 ] && echo "multi-GPU check: ALL OK" || echo "multi-GPU check: FAILED"
exit $fail
