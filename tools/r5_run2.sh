set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "repeat_launch_bit_identity or ragged or edge_embedding_of_both_modes or reference_golden or stage_outputs or full_size" > gpurun_out/r5_gate2.log 2>&1; tail -3 gpurun_out/r5_gate2.log
tools/ab_run.sh "qm9:64:r4bst qm9:64:k6st qm9:64:r4bst qm9:64:k6st geom:64:r4bst geom:64:k6st geom:64:r4bst geom:64:k6st"
cp gpurun_out/ab_run.log gpurun_out/r5_ab2.log
