"""Per-launch timeline of the cINN tile chain (run on the GPU box after tools/build_measurement_libs.sh flow):
    python tools/flow_timeline.py            -> stdout; copy into profiles/rNN_*_flow_launch_timeline.txt
Loads the -DFLOW_TIMELINE build of the library (same kernels plus wall-clock stamps: per launch the earliest workgroup
start and the latest workgroup end, per workgroup of the first 64 the phase boundaries; an explicit vmcnt(0) separates
"requests issued" from "operands landed", so the instrumented pass is a little slower than the shipped one)."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "image2video-synthesis-using-cinns_amd")); sys.path.insert(0, REPO)
import i2v_native
i2v_native.LIB_PATH = os.path.join(REPO, "tools", "_tl", "libi2v_hip_flowtl.so")
import i2v_synth as synth
from stage2_cINN.modules.flow_blocks import ConditionalFlow
torch.set_grad_enabled(False)
lib = i2v_native.lib()
lib.i2v_flow_timeline_report.restype = ctypes.c_int
lib.i2v_flow_timeline_report.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.flow_state_dict(seed=7, embedding_dim=64).items()}
flow = ConditionalFlow(64, 64, 512, 2, 20, conditioning_option="None"); flow.load_state_dict(sd); flow = flow.cuda().eval()
for B in [int(v) for v in os.environ.get("FLOWTIME_B", "64,8").split(",")]:
    _, r, e = synth.bench_inputs(B, 64, 64); r, e = r.cuda(), e.cuda()
    for _ in range(10): flow(r, e, reverse=True)
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); flow(r, e, reverse=True); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    torch.cuda.synchronize()
    lib.i2v_flow_timeline_report(0, 2, 1)          # reset the stamp buffers
    flow(r, e, reverse=True)                       # ONE instrumented pass (the graph replays the same launch numbers)
    torch.cuda.synchronize()
    print(f"B = {B}: inverse pass of the INSTRUMENTED build, HIP events, median of 30: {np.median(ts):.1f} us", flush=True)
    sys.stdout.flush()
    # launches per pass: 82 = folded chain (the default for B <= 64), 122 = round 4's chain (I2V_FLOW_FOLD=0 or B > 64)
    folded = os.environ.get("I2V_FLOW_FOLD", "1" if B <= 64 else "0") != "0"
    lib.i2v_flow_timeline_report(82 if folded else 122, 2, 0)
