# Round 5, call 5: pquad2 -- staging through registers instead of LDS-DMA; phase ablations of version 2.
mkdir -p gpurun_out/r05_05
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_05
cd $R
B=$R/tools/bin/msda_bench
for v in "" stage0 abl1 abl2 abl3 abl8 abl16; do
  echo "== variant ${v:-default}"
  if [ -z "$v" ]; then timeout 100 $B --iters 24 --sets 4 --fused 1 --patterns pert,init pquad 2>&1 | grep "fused pquad"
  else LD_PRELOAD=$R/tools/bin/ablate/libtf_msda_$v.so timeout 100 $B --iters 24 --sets 4 --fused 1 --patterns pert,init pquad 2>&1 | grep "fused pquad"; fi
done | tee $O/variants.txt | cut -c1-150
LD_PRELOAD=$R/tools/bin/ablate/libtf_msda_stage0.so timeout 100 $B --iters 24 --sets 4 --fused 1 --trace --patterns pert pquad 2>&1 | tail -16 | tee $O/stage0_trace.txt | cut -c1-150
