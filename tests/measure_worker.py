"""Worker of test_f43_structure_switches_keep_the_bits: runs in a process of its own with I2V_LIB_PATH pointing at the MEASUREMENT
build of the library (lib/libi2v_hip_measure.so: -DI2V_MEASURE, the only build that reads the I2V_W4_* / I2V_CONVIMG_TCH switches
and carries the persistent F(4,3) kernels).  Loads the inputs and the PRODUCTION library's frames from an .npz, recomputes them
under every switch and reports which ones differ (JSON on the last stdout line)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "image2video-synthesis-using-cinns_amd")):
    sys.path.insert(0, p)
import i2v_native  # noqa: E402
import i2v_synth as synth  # noqa: E402
from stage1_VAE.modules.decoder import Generator  # noqa: E402

assert os.path.basename(i2v_native.LIB_PATH) == "libi2v_hip_measure.so", i2v_native.LIB_PATH
torch.set_grad_enabled(False)
f = np.load(sys.argv[1])
meta = json.loads(bytes(f["meta"]).decode())
x0, z, ref = (torch.from_numpy(f[k]).cuda() for k in ("x0", "z", "ref"))


def gen():
    g = Generator({"channel_factor": meta["synth"]["channel_factor"], "z_dim": 64, "upsample_s": meta["upsample_s"],
                   "upsample_t": meta["upsample_t"], "spectral_norm": True})
    g.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.decoder_state_dict(**meta["synth"]).items()})
    return g.cuda().eval()


bad = []
if not torch.equal(gen()(x0, z), ref):      # the measurement build at its defaults IS the production kernel
    bad.append(["defaults", ""])
SWITCHES = (("I2V_W4_PIPE", "1"), ("I2V_W4_PIPE", "2"), ("I2V_W4_ORDER", "0"), ("I2V_W4_ORDER", "1"), ("I2V_W4_BN", "32"),
            ("I2V_W4_NTH", "512"), ("I2V_W4_NTH", "256"), ("I2V_CONVIMG_TCH", "1"), ("I2V_CONVIMG_TCH", "2"), ("I2V_CONVIMG_TCH", "16"),
            ("I2V_W4_SKEW", "1"), ("I2V_W4_LOADER", "1"))
for env, val in SWITCHES:
    os.environ[env] = val
    if env == "I2V_W4_SKEW":
        os.environ["I2V_W4_PIPE"] = "1"
    alt = gen()(x0, z)
    os.environ.pop(env)
    os.environ.pop("I2V_W4_PIPE", None)
    if not torch.equal(alt, ref):
        bad.append([env, val, float((alt - ref).abs().max())])
for pipe in ("1", "2"):                     # 18 samples: several bricks per persistent workgroup
    os.environ["I2V_W4_PIPE"] = pipe
    big = gen()(x0.repeat(6, 1, 1, 1), z.repeat(6, 1))
    os.environ.pop("I2V_W4_PIPE")
    if not (torch.equal(big[:3], ref) and torch.equal(big[15:], ref)):
        bad.append(["I2V_W4_PIPE", pipe + " (18 samples)"])
torch.cuda.synchronize()
print(json.dumps({"checked": len(SWITCHES) + 3, "bad": bad}))
