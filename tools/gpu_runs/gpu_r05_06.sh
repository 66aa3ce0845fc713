# Round 5, call 6: pquad2 with level 0 through the texture path (not staged)
mkdir -p gpurun_out/r05_06
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_06
cd $R
B=$R/tools/bin/msda_bench
for v in l0ta; do
  echo "== variant ${v:-default}"
  LD_PRELOAD=$R/tools/bin/ablate/libtf_msda_$v.so timeout 100 $B --iters 24 --sets 4 --fused 1 --patterns pert,init,local pquad 2>&1 | grep "fused pquad"
done | tee $O/variants.txt | cut -c1-150
