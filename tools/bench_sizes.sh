for args in "--size 1000" "--size 16384" "--size 65536 --dtype f32" "--size 92672" "--size 65536 --batch 1"; do
  echo "== bench.py $args"
  python bench.py $args --no-cpu-baseline --no-denoiser 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(json.dumps({k:d[k] for k in ('value','ms_per_step','dtype')}), r['kernel'], round(r['achieved'],2), r['unit'], {k:(v['launches'], round(v['ms'],2)) for k,v in d['kernels'].items()}, 'factorize_s', round(d['setup']['factorize_s'],1))"
done
python tools/bench_configs.py twolayer 65536 64 f64 2>&1 | tail -2
python tools/bench_configs.py twolayer 65536 64 f32 2>&1 | tail -2
