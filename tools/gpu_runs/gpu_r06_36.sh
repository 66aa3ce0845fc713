# Round 6, call 36: stream priorities of the two halves of a pipelined frame (decoder half on the sequence's stream, image-only half on the
# wrapper's side stream): does a high-priority decoder half shorten the frame?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_36
mkdir -p $O
python3 -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for combo in "0 0" "-1 0" "0 -1" "-1 -1" "0 0"; do
  set -- $combo
  TF_SEQ_STREAM_PRIORITY=$1 TF_SIDE_STREAM_PRIORITY=$2 timeout 600 python bench.py --no-cpu-baseline --no-fp32-exact --no-split3 --no-roofline --no-parity --sequences 1 --no-single-sequence > $O/bench_$1_$2.json 2> $O/bench.err
  python3 -c "
import json
d=json.load(open('$O/bench_$1_$2.json')); print('seq stream priority $1, side stream priority $2: value', d['value'], 'ms', d['ms_per_step'])"
done
