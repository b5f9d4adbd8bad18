"""Generates tests/golden/config5_backward.npz: forward_train + backward of BASELINE config 5 (S3DIS-shape, 500 000 points,
12 x 10 m room, 2 cm voxels, 4 levels, 5 classes, pts_threshold pruning LIVE at the finest neck level) on the CPU
oracle (oracle/model_oracle.py) in fp32 — the yardstick the `-m gpu` test `test_full_size_config5_backward_vs_oracle_digest`
holds the HIP path against (fp32, the reference's own precision: with this test's spread class logits the focal loss
saturates — log(max(1 - p, FLT_MIN)) as in mmcv's kernel — which an fp64 run does not reproduce; and the 4-level parity
test shows the HIP path within 1e-3 of the fp32 oracle tensor by tensor, while BOTH sit up to 9e-2 from fp64 through the same
ReLU / top-k sign decisions).  The oracle needs ~1.5 minutes of CPU for this scene, which the GPU box's minutes should not
pay for on every test run, and the full gradient (70 M floats) is no fixture; so the fixture is a DIGEST per parameter
tensor: 2-norm, largest magnitude and 256 entries at fixed (seeded) positions, plus the three losses and a checksum of the initial weights (the test rebuilds them from the same seed and checks it).

Run (build container, no GPU):  python tests/golden/make_config5_golden.py
Inputs are synthetic (fcaf3d_amd/synthetic.py, seed 51) — nothing of the reference is involved."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SEED_SCENE, SEED_MODEL, N_SAMPLES = 51, 0, 256


def build():
    import fcaf3d_amd as fa
    torch.manual_seed(SEED_MODEL)
    cfg = fa.get_config('fcaf3d_s3dis-3d-5class', voxel_size=0.02)
    m = cfg.model
    model = fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg'))
    with torch.no_grad():                                     # spread the scores so that top-k has no near-ties
        model.neck_with_head.cls_conv.kernel.normal_(0, 0.5)
    return model, m


def sample_index(i, numel):
    rng = np.random.default_rng(1000 + i)
    return rng.integers(0, numel, size=min(N_SAMPLES, numel))


def main():
    from fcaf3d_amd.synthetic import WORKLOADS, make_scene
    from oracle import model_oracle as MO
    model, m = build()
    P = {k: (v.detach().clone().requires_grad_(True) if v.dtype.is_floating_point else v.detach().clone())
         for k, v in model.state_dict().items()}
    p, g, l = make_scene(SEED_SCENE, **WORKLOADS['s3dis-500k']['scene'])
    t0 = time.time()
    losses = MO.forward_train(P, m, [p], [g], [l])
    sum(losses.values()).backward()
    print(f'oracle fp32 forward_train + backward: {time.time() - t0:.0f} s', {k: float(v) for k, v in losses.items()})
    out = dict(losses=np.array([float(losses[k]) for k in ('loss_centerness', 'loss_bbox', 'loss_cls')]),
               weight_checksum=np.array([float(sum(v.detach().double().abs().sum() for k, v in P.items() if v.dtype.is_floating_point))]))
    names = [k for k, _ in model.named_parameters()]
    out['names'] = np.array(names)
    for i, k in enumerate(names):
        gr = P[k].grad.double().reshape(-1)
        idx = sample_index(i, gr.numel())
        out[f'g{i}'] = np.concatenate([[float(gr.norm()), float(gr.abs().max())], gr[idx].numpy()])
    np.savez_compressed(os.path.join(HERE, 'config5_backward.npz'), **out)
    print('wrote', os.path.join(HERE, 'config5_backward.npz'))


if __name__ == '__main__':
    main()
