# Round 6, call 42: the D = 36 encoder kernel (msda_fwd_f32_pquad<...,36>, cfg 4): geometry sweep through TF_MSDA_PQUAD (bench.py --config cfg4
# --roofline-only: fused entry, pert pattern, 4 rotating input sets)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_42
mkdir -p $O
for opt in "" "lds=58" "lds=64,wgs=2" "lds=72,wgs=2" "lds=46" "npass=3,lds=52" "npass=1,lds=40,wgs=4" "hy=5,hx=8" "hy=7,hx=12,lds=58" "th=8,tw=10" "th=6,tw=12" "wgs=2"; do
  TF_MSDA_PQUAD="$opt" timeout 200 python bench.py --config cfg4 --roofline-only > $O/roof.json 2>/dev/null
  python3 -c "
import json
d=json.load(open('$O/roof.json')); o=d.get('other_patterns',{})
print('%-26s pert %6.2f us (%.3f)  init %6.2f  local %6.2f  %s' % ('$opt' or 'default', d['avg_launch_us'], d['frac'], o.get('init',{}).get('avg_launch_us',0), o.get('local',{}).get('avg_launch_us',0), d['kernel'][:40]))"
done
