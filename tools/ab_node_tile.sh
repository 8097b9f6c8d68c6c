F="--no-cpu-baseline --no-fp32-timing --no-extras --no-full-sample --no-other-configs --steps 60 --warmup 5"
for rep in 1 2; do
for nt in 0 32 64; do
  for w in geom qm9; do
    echo -n "$w node_tile=$nt: "; GCDM_NODE_TILE=$nt timeout 300 python bench.py --workload $w $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_median'))"
  done
done
done
