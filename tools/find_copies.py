#!/usr/bin/env python3
"""Where do the device-to-device copies of one training step come from?  (run on the GPU box)
Patches Tensor.copy_ / clone / contiguous / torch.cat for one step of the bench model and prints the
call sites with counts."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import torch
import bench

sites = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "quantized-autoregression" in fr.filename or fr.filename.endswith("bench.py"):
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line}"
    return "?"
orig_copy, orig_clone, orig_contig, orig_cat = torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous, torch.cat
def copy_(self, src, *a, **k):
    if self.is_cuda: sites["copy_ " + site()] += 1
    return orig_copy(self, src, *a, **k)
def clone(self, *a, **k):
    if self.is_cuda: sites["clone " + site()] += 1
    return orig_clone(self, *a, **k)
def contiguous(self, *a, **k):
    if self.is_cuda and not self.is_contiguous(): sites["contiguous " + site()] += 1
    return orig_contig(self, *a, **k)
def cat(*a, **k):
    sites["cat " + site()] += 1
    return orig_cat(*a, **k)

sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-events"] + sys.argv[1:]
# count only the second (timed) step: patch lazily after the warm-up by counting everything and dividing
torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.contiguous, torch.cat = copy_, clone, contiguous, cat
bench.main()
for k, v in sites.most_common(40):
    print(f"{v:5d}  {k}")
