_base_ = ['fcaf3d.py']
n_points = 100000

model = dict(
    neck_with_head=dict(
        n_classes=5,
        n_reg_outs=6,
        loss_bbox=dict(with_yaw=False)))

data = dict(samples_per_gpu=8, workers_per_gpu=4)
