# slice-ratio experiment: bench.py --lanes 2 with the first slice's share of the work set by GCDM_SLICE_FRACTIONS
for f in "" 0.4873 "" 0.4873 0.5127 0.45 0.531; do
    GCDM_SLICE_FRACTIONS=$f timeout 200 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-other-configs --no-extras --no-full-sample --no-fp32-timing 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SLICE f=%-7s ms_per_step=%.4f median=%.4f' % ('$f' or 'equal', r['ms_per_step'], r['ms_per_step_median']))"
done
