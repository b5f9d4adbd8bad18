_base_ = ['fcaf3d.py']
n_points = 100000

model = dict(
    neck_with_head=dict(
        n_classes=10,
        n_reg_outs=8))

data = dict(samples_per_gpu=8, workers_per_gpu=4)
