# Round 6, call 7: the halo form of the stride-1 3 x 3 convolutions (default) against the stream form (TF_CONV_HALO=0): per-layer times,
# parity tests of the fused / backbone paths, the frame
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_07
mkdir -p $O
for v in halo stream; do
  if [ $v = stream ]; then export TF_CONV_HALO=0; else unset TF_CONV_HALO; fi
  timeout 400 python tools/bench_conv.py --iters 20 2>&1 | grep -v amdgpu.ids > $O/conv_$v.txt
  echo "== $v"; grep -E "conv2|per frame" $O/conv_$v.txt | cut -c1-120
done
unset TF_CONV_HALO
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_linear_split_gpu.py -x -q -m gpu > $O/pytest_fused.txt 2>&1; tail -3 $O/pytest_fused.txt
for v in halo stream; do
  if [ $v = stream ]; then export TF_CONV_HALO=0; else unset TF_CONV_HALO; fi
  timeout 600 python bench.py --no-cpu-baseline --no-fp32-exact --no-split3 --no-roofline --sequences 1 > $O/bench_$v.json 2> $O/bench_$v.err
  python3 -c "
import json
d=json.load(open('$O/bench_$v.json'))
print('$v', 'value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'host', d.get('host_frames_fps'), 'parity', d['parity']['max_abs_boxes'], d['parity']['max_abs_logits'], d['parity']['ids_equal'])"
done
