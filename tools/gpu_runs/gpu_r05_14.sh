mkdir -p gpurun_out/r05_14
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
echo "--- zeroing kernel, copies -> A -> kernel -> B #$i: $(timeout 300 python tools/experiments/debug_wc_graph.py wceager f66 2>&1 | grep 'WRONG')"
done | tee gpurun_out/r05_14/after_fix.txt
for i in 1 2 3 4 5 6 7 8; do
echo "--- zeroing kernel, nothing between the graphs #$i: $(TF_GRAPH_DEBUG_BETWEEN=none timeout 300 python tools/experiments/debug_wc_graph.py wceager f66 2>&1 | grep 'WRONG')"
done | tee -a gpurun_out/r05_14/after_fix.txt
