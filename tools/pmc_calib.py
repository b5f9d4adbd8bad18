"""Calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on this box with a kernel of KNOWN traffic: a 1 GiB float4
device-to-device copy (reads 1 GiB, writes 1 GiB).  Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`."""
import torch
x = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device='cuda')
x.normal_()
y = torch.empty_like(x)
for _ in range(5):
    y.copy_(x)
torch.cuda.synchronize()
print('copied', x.numel() * 4 / 2**30, 'GiB x5')
