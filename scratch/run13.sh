mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "batched_get_bboxes or simple_test or eval_mode or checkpoint or multiclass_nms or pcdet" 2>&1 | tail -8 > gpurun_out/t_inf.log
timeout 300 python tools/inferprof.py 2>&1 | grep -E "ms/batch|detections" > gpurun_out/inferprof2.log
