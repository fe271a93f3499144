p() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('$1', d['ms_per_step'], d.get('roofline',{}).get('step_frac')); oc=d.get('other_configs'); 
if oc: print('   in-bench mix4096', oc['mix4096']['ms_per_step'], 'joined', oc['mix4096_joined']['ms_per_step'], 'shard', oc['shard8192']['ms_per_step'])"; }
python bench.py --mix --channels 4096 --tiles 24 --steps 100 --warmup 20 --ramp-ms 100 --no-cpu 2>/dev/null | p "mix s100 w20 r100"
python bench.py --mix --channels 4096 --tiles 24 --steps 200 --warmup 50 --no-cpu 2>/dev/null | p "mix s200 w50 r250"
python bench.py --mix --channels 4096 --tiles 24 --steps 20 --warmup 5 --no-cpu 2>/dev/null | p "mix s20 w5 r250"
python bench.py --mix --channels 4096 --tiles 24 --steps 20 --warmup 5 --ramp-ms 100 --no-cpu 2>/dev/null | p "mix s20 w5 r100"
python bench.py --no-cpu 2>/dev/null | p "default no-cpu"
python bench.py 2>/dev/null | p "default"
