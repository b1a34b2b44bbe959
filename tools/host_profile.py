#!/usr/bin/env python3
"""cProfile of the Python side of the eager training loop (run on the GPU box):
    python tools/host_profile.py --config c4 --steps 5
prints the functions with the largest own time -- what a step costs the host per launch."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "quantized-autoregression-image-generator_amd"))
import bench
sys.argv = ["bench.py", "--warmup", "2", "--no-cpu-baseline", "--no-kernel-events"] + sys.argv[1:]
import torch
torch.autograd.set_multithreading_enabled(False)      # backward on this thread: cProfile sees it
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
