# Round 6, call 11: split policy of the halo form alone (the other convolutions keep theirs), then the frame
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_11
mkdir -p $O
for pol in 512,300,9,32 512,200,9,32 384,300,9,32 512,300,18,32 640,300,9,32 256,300,9,32 512,300,9,64; do
  echo "== TF_CONV_HALO_KSPLIT_POLICY=$pol"
  TF_CONV_HALO_KSPLIT_POLICY=$pol timeout 400 python tools/bench_conv.py --iters 20 2>&1 | grep -v amdgpu.ids | grep -E "conv2 |per frame" | grep -E " 3 1 |per frame" | cut -c1-100
done
for v in halo stream; do
  if [ $v = stream ]; then export TF_CONV_HALO=0; else unset TF_CONV_HALO; fi
  timeout 600 python bench.py --no-cpu-baseline --no-fp32-exact --no-split3 --no-roofline --sequences 1 > $O/bench_$v.json 2> $O/bench_$v.err
  python3 -c "
import json
d=json.load(open('$O/bench_$v.json'))
print('$v', 'value', d['value'], 'ms', d['ms_per_step'], 'step_only', d.get('step_only_fps'), 'host', d.get('host_frames_fps'), 'parity', d['parity']['max_abs_boxes'], d['parity']['max_abs_logits'], d['parity']['ids_equal'])"
done
