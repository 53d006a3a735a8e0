ROOT=$GRAFT_REPO_ROOT; O=$ROOT/gpurun_out/prof_r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python3 $ROOT/bench.py"
name=c2x16
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_$name -- $B --clips 16 --steps 2 --warmup 1 --no-cpu --no-extras --inflight 1 > $O/pmc_f_$name.json 2> $O/pmc_f_$name.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$name -- $B --clips 16 --steps 2 --warmup 1 --no-cpu --no-extras --inflight 1 > $O/pmc_w_$name.json 2> $O/pmc_w_$name.err
cp $(ls $O/pmc_f_$name/*/*counter_collection.csv | tail -1) $O/pmc_fetch_size_$name.csv
cp $(ls $O/pmc_w_$name/*/*counter_collection.csv | tail -1) $O/pmc_write_size_$name.csv
python3 $ROOT/tools/pmc_traffic.py $O/pmc_fetch_size_$name.csv $O/pmc_write_size_$name.csv > $O/pmc_traffic_per_kernel_$name.json
rm -rf $O/pmc_f_$name $O/pmc_w_$name
python3 -c "
import json; t=json.load(open('$O/pmc_traffic_per_kernel_$name.json')); print(round(sum(v['hbm_mb_per_launch']*v['calls'] for v in t.values())/3), 'MB per step', {k:(v['calls'], round(v['hbm_mb_per_launch'])) for k,v in list(t.items())[:13]})"
