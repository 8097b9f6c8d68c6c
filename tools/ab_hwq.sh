F="--no-cpu-baseline --no-fp32-timing --no-extras --no-full-sample --no-other-configs --steps 100 --warmup 10"
run() { echo -n "Q=$5 $1 batch=$2 lanes=$3 graph=$4: "; GPU_MAX_HW_QUEUES=$5 GCDM_STEP_GRAPH=$4 timeout 300 python bench.py --workload $1 --batch $2 --lanes $3 $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['value'],1))"; }
for q in 2 4 8 16; do for l in 2 3 4; do run qm9 64 $l 1 $q; done; done
run qm9 64 4 0 8
run qm9 1024 2 1 8
