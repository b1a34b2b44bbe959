#!/usr/bin/env python3
"""The conditional-codebook BMU launch alone (64 rows x 512 codes x 4096), for counter passes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch  # noqa: E402
from qarig import ops  # noqa: E402
g = torch.Generator().manual_seed(9)
x = torch.tanh(torch.randn((64, 4, 32, 32), generator=g)).cuda()
w = torch.tanh(torch.randn((512, 4096), generator=g)).cuda()
for _ in range(10):
    ops.bmu(x, w, (32, 32))
torch.cuda.synchronize()
