import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "image2video-synthesis-using-cinns_amd")); sys.path.insert(0, REPO)
import i2v_native
if os.environ.get("FLOWTIME_LIB"):   # A/B against another build of the library (e.g. without kernarg preload)
    i2v_native.LIB_PATH = os.path.join(REPO, os.environ["FLOWTIME_LIB"])
import i2v_synth as synth
from stage2_cINN.modules.flow_blocks import ConditionalFlow
torch.set_grad_enabled(False)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.flow_state_dict(seed=7, embedding_dim=64).items()}
flow = ConditionalFlow(64, 64, 512, 2, 20, conditioning_option="None"); flow.load_state_dict(sd)
flow.linear_f16 = int(os.environ.get("FLOWTIME_F16", "0"))
flow = flow.cuda().eval()
for B in [int(v) for v in os.environ.get("FLOWTIME_B", "64,8,256").split(",")]:
    _, r, e = synth.bench_inputs(B, 64, 64); r, e = r.cuda(), e.cuda()
    for _ in range(10): flow(r, e, reverse=True)
    ts = []
    for _ in range(50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); flow(r, e, reverse=True); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    print("lib", os.environ.get("FLOWTIME_LIB"), "f16", os.environ.get("FLOWTIME_F16"), "DEV_KERNARG", os.environ.get("HIP_FORCE_DEV_KERNARG"), "B", B, "I2V_FLOW_TILE", os.environ.get("I2V_FLOW_TILE"), "PF", os.environ.get("I2V_FLOW_PF"), "NS", os.environ.get("I2V_FLOW_NS"),
          "inverse median us", round(float(np.median(ts)), 1), "min", round(float(np.min(ts)), 1), flush=True)
