#!/bin/bash
# r5 call 1: new-op tests, executor == module path, A/B of the BatchNorm fusion, full default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g1
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "statistics or second_gradient or head_backward or bf16 or non_finite or offset_split" > $O/t_new.log 2>&1
echo "new tests rc=$?" 
tail -5 $O/t_new.log
timeout 900 python -m pytest tests/test_gpu_exec.py -x -q > $O/t_exec.log 2>&1
echo "exec tests rc=$?"
tail -5 $O/t_exec.log
for f in 1 0; do
  FC_BN_FUSE=$f timeout 300 python bench.py --no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument > $O/b_fuse$f.json 2> $O/b_fuse$f.err
  echo "bench fuse=$f rc=$?"; python -c "import json;d=json.load(open('$O/b_fuse$f.json'));print(d['value'],d['ms_per_step'],d['config']['final_loss'])"
done
FC_BN_FUSE=1 timeout 300 python bench.py --no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument --batch 2 > $O/b_fuse1_b2.json 2> $O/b_fuse1_b2.err
python -c "import json;d=json.load(open('$O/b_fuse1_b2.json'));print('B=2',d['value'],d['ms_per_step'])"
FC_BN_FUSE=0 timeout 300 python bench.py --no-cpu-baseline --infer-steps 0 --no-force-dp --no-fp32-route --no-extras --no-instrument --batch 2 > $O/b_fuse0_b2.json 2> $O/b_fuse0_b2.err
python -c "import json;d=json.load(open('$O/b_fuse0_b2.json'));print('B=2 nofuse',d['value'],d['ms_per_step'])"
timeout 900 python bench.py > $O/b_full.json 2> $O/b_full.err
echo "full bench rc=$?"
python - <<PY
import json
d=json.load(open('$O/b_full.json'))
c=d['config']
print('value',d['value'],'ms',d['ms_per_step'],'roofline',d['roofline']['achieved'],d['roofline']['frac'])
for k in ('bf16_fast_mode','literal_1cm','two_scales','sunrgbd','s3dis','config4_per_gpu','fp32_mfma_route','inference','inference_pipelined'):
    print(k, c.get(k))
PY
