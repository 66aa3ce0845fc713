# Round 6, call 5: non-temporal output stores in the dense kernels (default build) against plain stores (variant library st0):
# the frame (one sequence, pipelined + step-only) and the dense kernels at their cfg-2 shapes (mfma_utilisation.live)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_05
mkdir -p $O
for v in nt st0; do
  if [ $v = st0 ]; then export TF_MSDA_LIB=$GRAFT_REPO_ROOT/tools/bin/ablate/libtf_msda_st0.so; else unset TF_MSDA_LIB; fi
  timeout 600 python bench.py --no-cpu-baseline --no-fp32-exact --no-split3 --no-parity --no-roofline --sequences 1 > $O/bench_$v.json 2> $O/bench_$v.err
  python3 - <<PY
import json
d=json.load(open("$O/bench_$v.json"))
print("$v", "value", d["value"], "ms", d["ms_per_step"], "step_only", d.get("step_only_fps"), "host", d.get("host_frames_fps"))
live=(d.get("mfma_utilisation") or {}).get("live") or {}
for k,x in live.items():
    if isinstance(x,dict): print("   ", k, {a:b for a,b in x.items() if a in ("us","avg_us","tflops_fp32_equivalent","utilisation")})
PY
done
