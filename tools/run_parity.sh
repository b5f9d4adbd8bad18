#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5parity
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_model.py -q -s -k "forward_train_parity or config5_backward or depth50" > $O/parity.log 2>&1; echo "rc=$?"
grep "decisions differ\|gradients vs the fp32\|passed\|failed" $O/parity.log
timeout 600 python -m pytest tests/test_gpu_exec.py tests/test_gpu_model.py -q -s -k "executor_equals_module_path_training or iou_losses or train_step_through" > $O/exec.log 2>&1
grep "worst gradient difference\|IoU gradient\|parameters after\|passed\|failed" $O/exec.log
