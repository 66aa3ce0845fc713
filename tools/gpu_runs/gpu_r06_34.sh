# Round 6, call 34: TunableOp for the GEMMs of the cfg-3 training step: tune (writes the table), then bench with the table and without
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_34
mkdir -p $O
timeout 600 python bench.py --config cfg3 --no-cpu-baseline --no-roofline > $O/bench_cfg3_before.json 2> $O/bench_cfg3_before.err
python3 -c "
import json
d=json.load(open('$O/bench_cfg3_before.json')); print('cfg3 no table', d['value'], d['ms_per_step'])"
TF_TUNE=1 TF_TUNABLEOP_OUT=$O/tunableop_cfg3.csv timeout 1500 python bench.py --config cfg3 --no-cpu-baseline --no-roofline --steps 4 --warmup 4 --min-seconds 0.1 > $O/bench_cfg3_tuning.json 2> $O/bench_cfg3_tuning.err
ls -la $O; wc -l $O/tunableop_cfg3*.csv
f=$(ls $O/tunableop_cfg3*.csv | head -1)
mkdir -p trackformer_amd/tuning; cp $f trackformer_amd/tuning/tunableop_gfx950_cfg3.csv
timeout 600 python bench.py --config cfg3 --no-cpu-baseline --no-roofline > $O/bench_cfg3_table.json 2> $O/bench_cfg3_table.err
python3 -c "
import json
d=json.load(open('$O/bench_cfg3_table.json')); print('cfg3 with table', d['value'], d['ms_per_step'])"
grep -i "tunableop" $O/bench_cfg3_table.err | head -3
cp trackformer_amd/tuning/tunableop_gfx950_cfg3.csv $O/tunableop_gfx950_cfg3.csv
