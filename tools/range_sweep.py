"""Dynamic range of the split-fp16 operand format (run on the GPU box): rel-L2 of one GeneratorBlock vs the CPU oracle with
everything that sets the conv operands' magnitude scaled by 2**k (tests/test_gpu_parity.py:_range_sweep).  The table goes
into INTEGRATION.md."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "image2video-synthesis-using-cinns_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
torch.set_grad_enabled(False)
import test_gpu_parity as t
mma = os.environ.get("I2V_DEC_MMA", "1")
print(f"# mma={mma}: scale 2^k | rel-L2 vs oracle | range flag")
for k, err, flag in t._range_sweep(list(range(-24, 19, 2))):
    print(f"2^{k:+d}  {err:.3e}  {flag}")
