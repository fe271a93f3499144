p() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('$1', d['ms_per_step'], d['roofline']['step_frac'], d['roofline']['frac'], d['config'].get('layout'))"; }
for rep in 1 2 3; do
python bench.py --steps 200 --warmup 40 --no-cpu --no-others 2>/dev/null | p "pow2      "
python bench.py --steps 200 --warmup 40 --no-cpu --no-others --row-stride contiguous 2>/dev/null | p "contiguous"
done
python bench.py --channels 4096 --tiles 96 --steps 40 --warmup 10 --no-cpu --no-others 2>/dev/null | p "4096x96 pow2      "
python bench.py --channels 4096 --tiles 96 --steps 40 --warmup 10 --no-cpu --no-others --row-stride contiguous 2>/dev/null | p "4096x96 contiguous"
python bench.py --steps 100 --warmup 20 --ebn0 9 --no-cpu --no-others 2>/dev/null | p "9 dB pow2      "
python bench.py --steps 100 --warmup 20 --ebn0 9 --no-cpu --no-others --row-stride contiguous 2>/dev/null | p "9 dB contiguous"
python bench.py --sonde-type 3 --steps 60 --warmup 15 --no-cpu --no-others 2>/dev/null | p "M10 pow2      "
python bench.py --sonde-type 3 --steps 60 --warmup 15 --no-cpu --no-others --row-stride contiguous 2>/dev/null | p "M10 contiguous"
