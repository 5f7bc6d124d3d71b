"""Sample GPU power / clocks (rocm-smi) while a child process runs forwards in a loop.

usage: python tools/power_probe.py [precision] [seconds]
The parent never touches the GPU; the child is an ordinary python process started before anything initialises HIP.
"""
import json
import subprocess
import sys
import time

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32_split"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0

child_src = r'''
import sys, time, torch
sys.path.insert(0, ".")
from audioset_convnext_inf_amd import synth
from audioset_convnext_inf_amd.pytorch.convnext import convnext_tiny
m = convnext_tiny(pretrained=False, strict=False, drop_path_rate=0.0, after_stem_dim=[252, 56], use_speed_perturb=False)
m.load_state_dict(synth.synth_state_dict(0))
m = m.to("cuda").eval().set_precision("%s")
wav = synth.synth_waveforms(64, 320000, seed=1234).cuda()
for _ in range(3): m(wav)
torch.cuda.synchronize()
print("child: warm", flush=True)
t_end = time.time() + %f
n = 0
t0 = time.time()
while time.time() < t_end:
    for _ in range(10): m(wav)
    torch.cuda.synchronize(); n += 10
dt = time.time() - t0
print("child: %%d steps, %%.2f ms/step, %%.0f clips/s" %% (n, 1e3 * dt / n, 64 * n / dt), flush=True)
''' % (prec, secs)

import glob


def smi():
    """sysfs hwmon readings (rocm-smi takes > 20 s per call on the pool's boxes)."""
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp2_input", "power1_cap"):
            try:
                out[hw.split("/")[4] + ":" + name] = int(open(hw + "/" + name).read())
            except Exception:  # noqa
                pass
    for f in glob.glob("/sys/class/drm/card*/device/gpu_busy_percent"):
        try:
            out[f.split("/")[4] + ":busy"] = int(open(f).read())
        except Exception:  # noqa
            pass
    return out


idle = smi()
child = subprocess.Popen([sys.executable, "-c", child_src], stdout=subprocess.PIPE, text=True)
line = child.stdout.readline()
print(line.strip())
samples = []
t0 = time.time()
while child.poll() is None and time.time() - t0 < secs + 30:
    s = smi()
    samples.append((time.time() - t0, s))
    time.sleep(0.25)
print(child.stdout.read().strip())
# the box shows all 8 GPUs of its node in sysfs (other tenants run on the others): ours is the one whose busy
# figure follows the child -- pick the card with the largest busy swing between the first and the loaded samples
cards = sorted({k.split(":")[0] for _, smp in samples for k in smp})
def series(card, key):
    return [smp.get(card + ":" + key) for _, smp in samples if smp.get(card + ":" + key) is not None]
idle_busy = {c: idle.get(c + ":busy", 0) for c in cards}
ours = max(cards, key=lambda c: (sum(series(c, "busy")) / max(1, len(series(c, "busy")))) - idle_busy[c]) if cards else None
print("cards seen:", cards, "-> ours:", ours)
for t, smp in samples:
    g = lambda k: smp.get("%s:%s" % (ours, k))
    pw = g("power1_input") or g("power1_average") or 0
    print("%5.1f s  %6.0f W (cap %4.0f W)  sclk %4.0f MHz  mclk %4.0f MHz  busy %3s %%  %2.0f C" %
          (t, pw / 1e6, (g("power1_cap") or 0) / 1e6, (g("freq1_input") or 0) / 1e6, (g("freq2_input") or 0) / 1e6,
           g("busy"), (g("temp2_input") or 0) / 1e3))
