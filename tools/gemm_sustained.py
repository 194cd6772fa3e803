"""Sustained (seconds, not a three-launch burst) rate, board power and clock of one GEMM shape: the engine's product (C-ABI, tuner's pick) against
torch.mm (hipBLASLt: yardstick only).  The tuner and tools/gemm_yardstick.py time sub-millisecond bursts, which run at boost clock before the power
management reacts; a training step holds the board at its power cap (tools/power_probe.py) and what counts there is the sustained rate.
    python tools/gemm_sustained.py [M N K] [seconds]"""
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from gpu_util import DEV, bf, gemm  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (65536, 3072, 768)
secs = float(sys.argv[4]) if len(sys.argv) >= 5 else 2.0
p = torch.cuda.get_device_properties(0)
bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
card = [c for c in glob.glob("/sys/class/drm/card*/device") if bus in os.path.realpath(c)][0]
pw = glob.glob(card + "/hwmon/hwmon*/power1_input")[0]
fq = glob.glob(card + "/hwmon/hwmon*/freq1_input")[0]
torch.manual_seed(0)
A = bf(torch.randn((M, K), device=DEV))
W = bf(torch.randn((N, K), device=DEV) * 0.05)
out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
Wt = W.t()


def run(name, fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    # burst: 3 launches, as the tuner times them
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    time.sleep(0.5)   # let the board cool to idle clocks first
    e0.record()
    for _ in range(3):
        fn()
    e1.record()
    torch.cuda.synchronize()
    burst = e0.elapsed_time(e1) / 3 * 1e3
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append((int(open(pw).read()) / 1e6, int(open(fq).read()) / 1e6))
            time.sleep(0.005)

    th = threading.Thread(target=sampler)
    n = max(10, int(secs / (burst * 1e-6)))
    th.start()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    stop[0] = True
    th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    tail = samples[len(samples) // 2:]
    wm = sorted(s[0] for s in tail)[len(tail) // 2]
    fm = sorted(s[1] for s in tail)[len(tail) // 2]
    fl = 2.0 * M * N * K
    print("%-28s burst %7.1f us (%6.1f TFLOP/s)   sustained over %.1f s: %7.1f us (%6.1f TFLOP/s) at %4.0f W, %4.0f MHz -> %5.1f mJ per launch"
          % (name, burst, fl / burst / 1e6, us * n / 1e6, us, fl / us / 1e6, wm, fm, wm * us * 1e-3), flush=True)


print("shape %d x %d x %d (forward layout, plain epilogue)" % (M, N, K))
run("engine (tuner's pick)", lambda: gemm(A, W, out_bf16=out))
run("torch.mm (hipBLASLt)", lambda: torch.mm(A, Wt, out=out))
run("engine (tuner's pick)", lambda: gemm(A, W, out_bf16=out))
