#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
TF_FORK_DECODERS=1 timeout 1500 python -m pytest tests/test_model_gpu.py -q -x -k "tiny_model_losses or graph_replay_matches_eager or segmented_graphs or full_size_step_properties or bench_configuration_parity_B10 or dropout_paths or rccl" > $O/r06_fork_tests.log 2>&1; grep -v "Warn\|warn\|^$\|pin_memory" $O/r06_fork_tests.log | tail -6 | cut -c1-300
