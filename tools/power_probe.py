"""Board power (hwmon PPT, W) and shader clock of THIS process's GPU while a command runs: is the training step power-limited?
    python tools/power_probe.py [--ms 5] -- python bench.py --no-cpu-baseline --no-pcie --no-extras
Samples /sys/class/drm/card*/device/hwmon/hwmon*/power1_input (microwatts) and freq1_input (Hz) of the card whose PCI address is
torch's device 0, every --ms milliseconds, in the parent; prints the distribution over the busy part of the run (samples above half the maximum)."""
import glob
import os
import subprocess
import sys
import threading
import time

ms = 5.0
args = sys.argv[1:]
if args and args[0] == "--ms":
    ms = float(args[1]); args = args[2:]
if args and args[0] == "--":
    args = args[1:]
import torch  # noqa: E402

p = torch.cuda.get_device_properties(0)
bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
card = None
for c in glob.glob("/sys/class/drm/card*/device"):
    if bus in os.path.realpath(c):
        card = c
if card is None:
    raise SystemExit("no /sys/class/drm card for PCI %s" % bus)
pw = glob.glob(card + "/hwmon/hwmon*/power1_input")[0]
fq = glob.glob(card + "/hwmon/hwmon*/freq1_input")[0]
cap = int(open(glob.glob(card + "/hwmon/hwmon*/power1_cap")[0]).read()) / 1e6
samples = []
stop = False


def sampler():
    while not stop:
        try:
            samples.append((time.perf_counter(), int(open(pw).read()) / 1e6, int(open(fq).read()) / 1e6))
        except Exception:
            pass
        time.sleep(ms / 1e3)


t = threading.Thread(target=sampler)
t.start()
rc = subprocess.call(args)
stop = True
t.join()
w = sorted(s[1] for s in samples)
busy = [s for s in samples if s[1] > 0.5 * w[-1]]
bw = sorted(s[1] for s in busy)
bf = sorted(s[2] for s in busy)
q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]   # noqa: E731
print("[power] card %s (PCI %s), cap %.0f W, %d samples (%d busy): busy power W p10 %.0f median %.0f p90 %.0f max %.0f; hwmon clock MHz p10 %.0f median %.0f p90 %.0f"
      % (os.path.basename(os.path.dirname(card)), bus, cap, len(samples), len(busy), q(bw, 0.1), q(bw, 0.5), q(bw, 0.9), bw[-1], q(bf, 0.1), q(bf, 0.5), q(bf, 0.9)))
sys.exit(rc)
