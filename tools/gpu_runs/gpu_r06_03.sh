# Round 6, call 3: msda_fwd_f32_pquad2 -- nt stores as default; nt point loads; steady-state (second tile) phase stamps; the pass-major tail with
# the next tile's point loads issued early (variant libraries: pm0 = order only, pm1 / pm2 = one / both passes loaded early); tile shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_03
mkdir -p $O
B=tools/bin/msda_bench
timeout 200 $B --iters 24 --sets 4 --fused 1 --patterns pert,init,local --trace-dump $O/trace_raw.csv pquad pquad:st=0 pquad:ldnt=1 pquad:ti=1 \
   pquad:th=6,tw=16 pquad:th=7,tw=14 pquad:th=9,tw=10 pquad:th=5,tw=20 pquad:th=10,tw=10 pquad:th=4,tw=24 > $O/msda_default.txt 2>&1
grep -v "^  " $O/msda_default.txt | grep fused | cut -c1-130
for v in pm0 pm1 pm2; do
  echo "== $v"
  LD_PRELOAD=$GRAFT_REPO_ROOT/tools/bin/ablate/libtf_msda_$v.so timeout 100 $B --iters 24 --sets 4 --fused 1 --patterns pert,init,local pquad > $O/msda_$v.txt 2>&1
  grep -v "^  " $O/msda_$v.txt | grep "fused pquad" | cut -c1-130
done
