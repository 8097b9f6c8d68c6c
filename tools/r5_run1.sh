set -u
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -k "repeat_launch_bit_identity or ragged_geom_is_bit_reproducible or edge_embedding_of_both_modes or forward_matches_reference_golden or stage_outputs" > gpurun_out/r5_gate1.log 2>&1; tail -3 gpurun_out/r5_gate1.log
tools/ab_run.sh "qm9:64:r4st qm9:64:pkst qm9:64:pkBst qm9:64:r4st qm9:64:pkst qm9:64:pkBst geom:64:r4st geom:64:pkst geom:64:pkBst qm9:64:r4 qm9:64:pk qm9:64:pkB"
cp gpurun_out/ab_run.log gpurun_out/r5_ab1.log
