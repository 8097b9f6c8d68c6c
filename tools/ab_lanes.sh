# ms/step and molecules/s of small and headline batches against the number of slices, with and without the captured step (GCDM_STEP_GRAPH)
F="--no-cpu-baseline --no-fp32-timing --no-extras --no-full-sample --no-other-configs --steps 100 --warmup 10"
run() { echo -n "$1 batch=$2 lanes=$3 graph=$4: "; GCDM_STEP_GRAPH=$4 timeout 300 python bench.py --workload $1 --batch $2 --lanes $3 $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value'],1))"; }
for b in 64 100; do for l in 1 2 3 4 6; do run qm9 $b $l 1; done; done
for l in 2 4; do run geom 32 $l 1; done
for g in 0 1 0 1; do run qm9 1024 2 $g; done
for g in 0 1; do run geom 256 2 $g; done
run qm9 1024 3 1; run qm9 1024 4 1
