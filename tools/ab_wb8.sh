for rep in 1 2; do for v in "$@"; do
  export SONDE_MI355_LIB=$PWD/tools/ab_libs/lib_$v.so
  python bench.py --wideband --wb-streams 8 --steps 60 --warmup 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernel_ms']['pfb_fft'], d['kernel_ms']['demod'])"
done; done
