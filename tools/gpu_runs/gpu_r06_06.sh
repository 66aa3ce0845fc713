# Round 6, call 6: plain stores in the dense kernels again (default build); the output-store policy of msda_fwd_f32_pquad2 IN THE FRAME
# (its output is the next GEMM's input): nt (harness winner) against plain
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_06
mkdir -p $O
for v in nt plain nt2 plain2; do
  case $v in plain*) export TF_MSDA_PQUAD="st=0";; *) unset TF_MSDA_PQUAD;; esac
  timeout 600 python bench.py --no-cpu-baseline --no-fp32-exact --no-split3 --no-parity --no-roofline --sequences 1 > $O/bench_$v.json 2> $O/bench_$v.err
  python3 -c "
import json
d=json.load(open('$O/bench_$v.json'))
print('$v', 'value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'host', d.get('host_frames_fps'))"
done
