#!/usr/bin/env python3
"""What clock does the card run the proof at?  Samples the amdgpu hwmon files (freq*_input, power*_average / power*_input) of every
card every ~2 ms from a thread while the main thread (1) idles 1 s, (2) proves G1ExpStark(128) for ~3 s, (3) runs the sponge
microbenchmark-like loop of back-to-back proofs with three provers in flight.  Prints min / median / max per file and phase.
usage (GPU box): clock_sampler.py"""
import glob, os, sys, threading, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq*_input") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power*_average") +
               glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power*_input"))
print("hwmon files:", files, flush=True)
for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")[:1]:
    try: print(f, open(f).read().replace("\n", " | "), flush=True)
    except Exception as e: print(f, "unreadable:", e)
samples = {}; phase = ["idle"]; stop = [False]
def sampler():
    while not stop[0]:
        for f in files:
            try: v = int(open(f).read().strip())
            except Exception: continue
            samples.setdefault((phase[0], f), []).append(v)
        time.sleep(0.002)
th = threading.Thread(target=sampler); th.start()
time.sleep(1.0)
import starky_bn254_amd as S
import bench
stark = S.G1ExpStark(128); cfg = stark.config()
ios = bench.synthetic_ios(128, 1000, "g1")
p = S.Prover(stark, cfg, 16); p.generate_trace(ios)
for _ in range(5): p.prove()
phase[0] = "one proof in flight"
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 3.0: p.prove(); k += 1
print("one in flight: %.2f ms per proof" % ((time.perf_counter() - t0) / k * 1e3), flush=True)
phase[0] = "setup"
ps = [p] + [S.Prover(stark, cfg, 16) for _ in range(2)]
for q in ps[1:]: q.generate_trace(ios); q.prove()
phase[0] = "three proofs in flight"
cnt = [0]
def run(q):
    t = time.perf_counter()
    while time.perf_counter() - t < 3.0: q.prove(); cnt[0] += 1
ths = [threading.Thread(target=run, args=(q,)) for q in ps]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
print("three in flight: %.2f ms per proof" % ((time.perf_counter() - t0) / cnt[0] * 1e3), flush=True)
phase[0] = "idle after"
time.sleep(0.5)
stop[0] = True; th.join()
for (ph, f), v in sorted(samples.items()):
    if ph == "setup": continue
    unit = 1e6 if "freq" in f else 1e6
    print("%-24s %-60s n=%5d  min %8.1f  median %8.1f  max %8.1f  (%s)" % (ph, f.split("/device/")[1], len(v), min(v) / unit, statistics.median(v) / unit, max(v) / unit, "MHz" if "freq" in f else "W"))
for q in ps: q.close()
