"""Per-step wall time of the first 60 forwards after start-up (is the default warm-up long enough?)."""
import sys, time, torch
sys.path.insert(0, ".")
from audioset_convnext_inf_amd import synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56], use_speed_perturb=False)
m.load_state_dict(synth.synth_state_dict(0))
m = m.to("cuda").eval()
wav = synth.synth_waveforms(64, 320000, seed=1234).cuda()
torch.cuda.synchronize()
ts = []
for i in range(60):
    t0 = time.perf_counter(); m(wav); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join("%.2f" % t for t in ts))
time.sleep(3.0)
ts = []
for i in range(12):
    t0 = time.perf_counter(); m(wav); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("after a 3 s pause:", " ".join("%.2f" % t for t in ts))
