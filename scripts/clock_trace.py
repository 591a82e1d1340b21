#!/usr/bin/env python3
"""How launch times of the bench step drift with the card's state: run generate+apply (64 x 4K) back to back for a few seconds,
record every step's generate / apply duration (HIP events on the launch stream) and sample sclk / mclk / fclk / power / temperature
from sysfs beside it.   python scripts/clock_trace.py [seconds] > gpurun_out/clock_trace.txt"""
import ctypes as C, glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0


def find_card():
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(d + "/pp_dpm_sclk"):
            return d
    return None


def cur(path):
    try:
        for line in open(path):
            if line.strip().endswith("*"):
                return line.split(":")[1].strip().rstrip("*").strip()
    except OSError:
        pass
    return "?"


def hw(card, name):
    for f in glob.glob(card + "/hwmon/hwmon*/" + name):
        try:
            return int(open(f).read())
        except (OSError, ValueError):
            pass
    return -1


card = find_card()
samples, stop = [], False


def sampler():
    while not stop:
        t = time.perf_counter()
        if card:
            samples.append((t, cur(card + "/pp_dpm_sclk"), cur(card + "/pp_dpm_mclk"), cur(card + "/pp_dpm_fclk"),
                            hw(card, "power1_average") // 1000000 if hw(card, "power1_average") > 0 else hw(card, "power1_input") // 1000000,
                            hw(card, "temp2_input") // 1000, hw(card, "temp3_input") // 1000))
        time.sleep(0.005)


torch.cuda.set_device(0)
from libultrahdr_dev_amd import api
lib = api.init(0)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
w = bench.Batch(lib, 64, 0)
w.generate(stream); w.apply(stream, api.OUTPUT_HDR_HLG); torch.cuda.synchronize()
# the card is left alone for a moment first, as it is when bench.py's setup ends
torch.cuda.synchronize(); time.sleep(1.0)
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter()
rows = []
first = True
while time.perf_counter() - t0 < secs:
    n = 60 if first else 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * n)]
    tb = time.perf_counter() - t0
    for i in range(n):
        ev[3 * i].record(); w.generate(stream); ev[3 * i + 1].record(); w.apply(stream, api.OUTPUT_HDR_HLG); ev[3 * i + 2].record()
    torch.cuda.synchronize()
    if first:   # the first 60 steps one by one: what `cold_start` measures (5 warmup + 20 timed steps) lies in here
        acc = 0.0
        for i in range(n):
            g, a = ev[3 * i].elapsed_time(ev[3 * i + 1]), ev[3 * i + 1].elapsed_time(ev[3 * i + 2])
            rows.append((tb + acc * 1e-3, g, a)); acc += g + a
        first = False
    else:
        g = sum(ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(n)) / n
        a = sum(ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(n)) / n
        rows.append((tb, g, a))
stop = True; th.join()
print("card", card)
print("# t_s generate_ms apply_ms   (the first 60 rows: single steps from an idle card; then averages of 20)")
keep = rows[:60] + rows[60::max(1, (len(rows) - 60) // 40)]
for r in keep:
    print("%.4f %.4f %.4f" % r)
print("# t_s sclk mclk fclk power_W temp_hotspot temp_mem")
early = [x for x in samples if x[0] - t0 < 0.3]
late = [x for x in samples if x[0] - t0 >= 0.3]
for x in early + late[::max(1, len(late) // 40)]:
    print("%.3f %s %s %s %s %s %s" % ((x[0] - t0,) + x[1:]))
