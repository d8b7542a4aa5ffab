#!/bin/bash
# Everything a round commits under profiles/<round>/.
#   on the GPU box (one gpurun call):   bash tools/round_profiles.sh run <round> [--quick]
#   afterwards, in the build container: bash tools/round_profiles.sh collect <round>
# `run`: GPU suite log + smoke; per workload tools/profile_round.sh (rocprofv3 --kernel-trace --stats + separate --pmc passes; the default
# workload with the depth-0 FETCH/WRITE calibration) and tools/summarize_profile.py ON THE BOX (so that profiles/traffic.json /
# valu_insts.json there carry this build's numbers) and bench.py once more; the default workload's stats pass once more with launches that
# do not overlap (serial_launches = 1: the sum of launch durations is then a time per frame); tools/bench_configs.sh; the driver's command
# line; emulated strong scaling; present rates; the 2- and 8-rank one-GPU runs of bench.py; short runs; the native stress driver on the
# four builds; the oracle-checked fuzzer.  --quick: fewer stress / fuzz cases (a re-profile after a small change).
# `collect`: summarise the merged raw counters locally into the tracked files under profiles/ and copy the logs.
# (Round 4: replaces run_round_profiles.sh, run_round_profiles_reduced.sh, profile_extra_configs.sh, unchained_stats.sh and
# collect_round_profiles.sh.)
R=/root/repo
MODE=${1:-run}; RND=${2:-r05}; QUICK=${3:-}
cd $R
# tag | workload key | bench args          (first block: all passes; second block: stats + FETCH/WRITE + SQ only, shorter)
MAIN=("default|default_1920x1080_d8_spp1_atmosphere256_g1|"
      "perframe|default_1920x1080_d8_spp1_atmosphere256_g1_fb1|--frame-batch 1"
      "C3|stress256_1920x1080_d8_spp1_atmosphere256_g1|--config C3"
      "C5|glass_1920x1080_d32_spp1_atmosphere256_g1|--config C5"
      "spp4|default_1920x1080_d8_spp4_atmosphere256_g1|--spp 4"
      "tilewave|default_1920x1080_d8_spp1_atmosphere256_g1_variant1|--variant 1")
EXTRA=("sky2048|default_1920x1080_d8_spp1_sky2048_g1|--env sky2048"
       "d13|default_1920x1080_d13_spp1_atmosphere256_g1|--depth 13"
       "C3spp4|stress256_1920x1080_d8_spp4_atmosphere256_g1|--config C3 --spp 4"
       "4k|default_3840x2160_d8_spp1_atmosphere256_g1_strong4k|--strong-4k"
       "variant14|default_1920x1080_d8_spp1_atmosphere256_g1_variant14|--variant 14"
       "C3nogrid|stress256_1920x1080_d8_spp1_atmosphere256_g1_nogrid|--config C3 --tune no_sphere_grid=1"
       "nocarry|default_1920x1080_d8_spp1_atmosphere256_g1_nocarry|--tune carry_last=0"
       "C3gridcarry|stress256_1920x1080_d8_spp1_atmosphere256_g1_gridcarry|--config C3 --tune grid_carry=1"
       "C3gridcarry2|stress256_1920x1080_d8_spp1_atmosphere256_g1_gridcarry2|--config C3 --tune grid_carry=2")

if [ "$MODE" = collect ]; then
  mkdir -p profiles/$RND
  for w in "${MAIN[@]}" "${EXTRA[@]}"; do IFS='|' read tag key args <<< "$w"
    python tools/summarize_profile.py ${RND}_$tag $RND $key ${RND}_default > /dev/null && [ -f gpurun_out/$RND/${RND}_${tag}_bench.json ] && cp gpurun_out/$RND/${RND}_${tag}_bench.json profiles/$RND/
  done
  for f in ${RND}_default_stats_unchained_timed.json pytest_gpu.log smoke.log bench_configs.jsonl driver_command_bench.json emulate_strong.json present_rate.json present_rate_group2.json \
           bench_2ranks_one_gpu.json bench_8ranks_one_gpu.json short_runs.log handover_stress.log fuzz.log profile_sections.log \
           ${RND}_default_kernel_stats_unchained.csv ${RND}_default_stats_unchained.json handover_zero_budget.log atmosphere_profile.log \
           atmosphere_pmc_1024.csv bench_default.json; do
    [ -f gpurun_out/$RND/$f ] && cp gpurun_out/$RND/$f profiles/$RND/$f
  done
  ls profiles/$RND
  exit 0
fi

export TMPDIR=/tmp
mkdir -p gpurun_out/$RND
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/$RND/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$RND/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/$RND/smoke.log 2>&1
prof() { local tag=${RND}_$1 key=$2; shift 2
  bash tools/profile_round.sh $tag "$@" > gpurun_out/$RND/${tag}_profile.log 2>&1
  python tools/summarize_profile.py $tag $RND $key ${RND}_default > gpurun_out/$RND/${tag}_summary.log 2>&1
  python bench.py $PROFILE_BENCH_EXTRA "$@" > gpurun_out/$RND/${tag}_bench.json 2> gpurun_out/$RND/${tag}_bench.err; }
first=1
for w in "${MAIN[@]}"; do IFS='|' read tag key args <<< "$w"
  if [ "$tag" = spp4 ] || [ "$tag" = tilewave ]; then export PROFILE_STEPS=192 PROFILE_WARMUP=64; fi
  prof $tag $key $args
  if [ $first = 1 ]; then first=0; export PROFILE_NO_CAL=1 PROFILE_BENCH_EXTRA="--no-cpu-baseline"; fi   # (calibration + CPU baseline: default workload only)
done
export PROFILE_PASSES="stats fetch write sq" PROFILE_STEPS=256 PROFILE_WARMUP=128
for w in "${EXTRA[@]}"; do IFS='|' read tag key args <<< "$w"; prof $tag $key $args; done
unset PROFILE_NO_CAL PROFILE_PASSES PROFILE_BENCH_EXTRA PROFILE_STEPS PROFILE_WARMUP
# the default workload's stats pass with launches that go BEHIND each other (with back-pressure chaining every launch waits beside its
# predecessor and rocprofv3's durations add up to ~2x the elapsed time)
( OUT=$R/gpurun_out/prof_${RND}_unchained; mkdir -p $OUT; cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $R/bench.py --steps 640 --warmup 320 --clock-warmup-ms 0 --steady-ms 0 --no-cpu-baseline --tune serial_launches=1 > $OUT/bench.json 2> $OUT/stats.log
  python - "$OUT" "$R/gpurun_out/$RND" "$RND" <<'PY'
import csv, glob, json, sys
out, dst, rnd = sys.argv[1:4]
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
frames = 960
f = (glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True) + glob.glob(out + "/stats/*kernel_stats.csv"))[0]
open(f"{dst}/{rnd}_default_kernel_stats_unchained.csv", "w").write(open(f).read())
r = [x for x in csv.DictReader(open(f)) if "pt_integrate" in x["Name"]][0]
bench = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
res = {"what": "default workload, rocprofv3 --kernel-trace --stats with the tuning knob serial_launches = 1: launches do not overlap, the sum of their "
               "durations / frames is comparable with bench.py's HIP-event kernel_ms of the same run", "frames": frames,
       "unchained": {"calls": int(r["Calls"]), "total_ns": int(r["TotalDurationNs"]), "avg_ns": float(r["AverageNs"]), "ns_per_frame": int(r["TotalDurationNs"]) / frames,
                     "bench_kernel_ms_same_run": bench["roofline"].get("kernel_ms"), "bench_value": bench["value"]},
       "csrc_hash": g.load_package().native.csrc_hash()}
json.dump(res, open(f"{dst}/{rnd}_default_stats_unchained.json", "w"), indent=1); print(json.dumps(res["unchained"]))
PY
  find $OUT -name "*_kernel_trace.csv" -delete )
bash tools/bench_configs.sh > gpurun_out/$RND/bench_configs.log 2>&1; cp gpurun_out/bench_configs.jsonl gpurun_out/$RND/
python bench.py --steps 20 --warmup 5 > gpurun_out/$RND/driver_command_bench.json 2> gpurun_out/$RND/driver_command_bench.err
python tools/emulate_strong.py gpurun_out/$RND/emulate_strong.json > gpurun_out/$RND/emulate_strong.log 2>&1
python tools/present_rate.py --json gpurun_out/$RND/present_rate.json > gpurun_out/$RND/present_rate.log 2>&1
python tools/present_rate.py --devices 0,0 --json gpurun_out/$RND/present_rate_group2.json > gpurun_out/$RND/present_rate_group2.log 2>&1
python bench.py --gpus 2 --share-gpu --steps 256 --warmup 128 > gpurun_out/$RND/bench_2ranks_one_gpu.json 2> gpurun_out/$RND/bench_2ranks_one_gpu.err
python bench.py --gpus 8 --share-gpu --steps 64 --warmup 64 --steady-ms 0 > gpurun_out/$RND/bench_8ranks_one_gpu.json 2> gpurun_out/$RND/bench_8ranks_one_gpu.err
bash tools/short_runs.sh > gpurun_out/$RND/short_runs.log 2>&1
bash tools/unchained_timed.sh $RND > gpurun_out/$RND/unchained_timed.log 2>&1
if [ -x tools/ab/libP.so ]; then for sc in default stress glass; do MI355PT_LIB=$R/tools/ab/libP.so timeout 200 python tools/profile_sections.py $sc 0 640 2>&1 | grep -v amdgpu; done > gpurun_out/$RND/profile_sections.log; fi
N=12000; NM=5000; F1=600; F2=400; F4=1000; if [ "$QUICK" = --quick ]; then N=3000; NM=2000; F1=200; F2=150; F4=400; fi
{ for L in "" _audit _chaos _audit_chaos; do echo "== libmi355pt$L.so"; timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so $N $((700 + ${#L})) | grep -v "^\.\.\."; done
  echo "== round 6: frame-fed launches on every image size (--tune feed_min_tiles=0), idle budget 150 us and 0"
  for L in "" _audit_chaos; do timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so $((N / 2)) $((720 + ${#L})) --tune feed_min_tiles=0 | grep -v "^\.\.\."; done
  timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit.so $((N / 4)) 731 --tune feed_min_tiles=0 --tune feed_idle_us=1 | grep -v "^\.\.\."
  timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so $((N / 4)) 732 --tune feed_min_tiles=0 --tune handover_budget_ms=0 | grep -v "^\.\.\."
  echo "== multisample focus, batch-pass kernel forced onto tiny images (--tune batch_pass_min_tiles=0)"
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit_chaos.so $NM 711 --multisample --tune batch_pass_min_tiles=0 | grep -v "^\.\.\."
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so $NM 712 --multisample --tune batch_pass_min_tiles=0 | grep -v "^\.\.\."; } > gpurun_out/$RND/handover_stress.log 2>&1
{ echo "== general"; timeout 600 python tools/fuzz_parity.py $F1 301; echo "== FUZZ_FOCUS=pipelining"; FUZZ_FOCUS=pipelining timeout 600 python tools/fuzz_parity.py $F2 302;
  echo "== FUZZ_FOCUS=pipelining FUZZ_BIAS=group_spp under the audit build"; MI355PT_LIB=$R/opentk-pathtracer_amd/libmi355pt_audit.so FUZZ_FOCUS=pipelining FUZZ_BIAS=group_spp timeout 600 python tools/fuzz_parity.py $F2 303;
  echo "== FUZZ_FOCUS=grid"; FUZZ_FOCUS=grid timeout 600 python tools/fuzz_parity.py $F4 304; } 2>&1 | grep -v amdgpu.ids > gpurun_out/$RND/fuzz.log
# round 5: the hand-over bound with a ZERO budget (a wait that outlasts two clock readings abandons its launch: the repair path as a common path) on the four builds + spp > 1
{ for L in "" _audit _chaos _audit_chaos; do echo "== libmi355pt$L.so --tune handover_budget_ms=0"; timeout 900 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt$L.so $((N / 2)) $((800 + ${#L})) --tune handover_budget_ms=0 | grep -v "^\.\.\."; done
  echo "== multisample, zero budget, batch-pass kernel forced onto tiny images"
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt_audit_chaos.so $NM 811 --multisample --tune batch_pass_min_tiles=0 --tune handover_budget_ms=0 | grep -v "^\.\.\."
  echo "== multisample, zero budget, in-lane sample chain"
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so $NM 812 --multisample --tune handover_budget_ms=0 | grep -v "^\.\.\."
  echo "== fresh handle per case, 1 ms budget"
  timeout 600 tools/handover_stress.bin opentk-pathtracer_amd/libmi355pt.so 1000 813 --fresh --tune handover_budget_ms=1 --tune handover_check_us=50 | grep -v "^\.\.\."; } > gpurun_out/$RND/handover_zero_budget.log 2>&1
# round 5: the atmosphere kernel on its own (library timer) + one PMC pass at 1024^2
{ for sz in 256 1024 2048; do python tools/atmo_profile.py $sz $((sz > 1000 ? 3 : 16)); done
  ( cd /tmp; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/prof_${RND}_atmo -o pmc -- python $R/tools/atmo_profile.py 1024 2 > /dev/null 2>&1 )
  grep atmo $R/gpurun_out/prof_${RND}_atmo/pmc_counter_collection.csv > gpurun_out/$RND/atmosphere_pmc_1024.csv; } 2>&1 | grep -v amdgpu.ids > gpurun_out/$RND/atmosphere_profile.log
cp gpurun_out/$RND/${RND}_default_bench.json gpurun_out/$RND/bench_default.json 2>/dev/null
echo "FAILURES in the stress logs (must be 0):"; grep -c "library error\|MISMATCH\|AUDIT violation" gpurun_out/$RND/handover_stress.log gpurun_out/$RND/handover_zero_budget.log
tail -3 gpurun_out/$RND/pytest_gpu.log; grep "handover_stress:\|==\|hand-over bound" gpurun_out/$RND/handover_zero_budget.log; cat gpurun_out/$RND/atmosphere_profile.log; grep "handover_stress:\|==" gpurun_out/$RND/handover_stress.log; grep "cases,\|==" gpurun_out/$RND/fuzz.log; grep "ms per" gpurun_out/$RND/present_rate.log; tail -14 gpurun_out/$RND/bench_configs.log
