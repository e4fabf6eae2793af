#!/usr/bin/env python3
"""Board power and shader clock while one streaming kernel runs in a loop (run on the GPU box).

    python tools/probe_power.py --n 400000 --m 500000 --seconds 5

For each of counts / cprod(16 vectors) / prod(16 vectors) / cprod(8) / prod(8): launches the kernel back to back
for `--seconds` while a thread samples hwmon (power1_average / power1_input, freq1_input) and, once per kernel, the
text of `rocm-smi --showpower --showclocks`.  Prints one JSON line per kernel."""
import argparse, glob, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import bigsnpr_amd as ba
from bigsnpr_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=400000)
ap.add_argument("--m", type=int, default=500000)
ap.add_argument("--seconds", type=float, default=5.0)
ap.add_argument("--xkind", default="normal", help="normal | ones (0/1 panel: constant digits) | zero")
ap.add_argument("--na16", type=int, default=655, help="missing genotypes per 65 536 of the synthetic image")
ap.add_argument("--slices", type=int, default=2, help="8-bit digits per vector (3: the three-block kernels at 16 vectors)")
ap.add_argument("--only16", action="store_true", help="only the 16-vector crossproduct and product")
ap.add_argument("--tag", default="", help="copied into every line (e.g. the environment switches of this run)")
a = ap.parse_args()
L = _lib.load()


def hwmon_files():
    out = {}
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for k in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input"):
            f = os.path.join(d, k)
            if os.path.exists(f):
                out.setdefault(k, f)
    return out


HW = hwmon_files()
print("hwmon:", HW, flush=True)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.rows = []

    def run(self):
        while not self.stop:
            r = {}
            for k, f in HW.items():
                try:
                    r[k] = float(open(f).read().strip())
                except Exception:
                    pass
            self.rows.append(r)
            time.sleep(0.05)


def smi_text():
    try:
        t = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True,
                           timeout=20).stdout
        keep = [l.strip() for l in t.split("\n") if ("ower" in l or "sclk" in l or "mclk" in l or "fclk" in l)]
        return keep[:8]
    except Exception as e:
        return [repr(e)]


gb = ba.bed.synthetic(a.n, a.m, na16=a.na16)
L.bsn_device_sync()
bytes_pass = ((a.n + 3) // 4) * a.m
sc = ba.bed_scaleBinom(gb)
rng = np.random.default_rng(0)


def panel(rows, nv):
    if a.xkind == "ones":
        return (rng.random(size=(rows, nv)) < 0.5).astype(float)
    if a.xkind == "zero":
        x = np.zeros((rows, nv)); x[0, :] = 1.0
        return x
    return rng.normal(size=(rows, nv))


def loop(name, fn):
    fn(); L.bsn_device_sync()
    s = Sampler(); s.start()
    t0 = time.time(); reps = 0
    ms = C.c_double()
    L.bsn_timer_start(gb.handle)
    txt = None
    while time.time() - t0 < a.seconds:
        for _ in range(10):
            fn()
        reps += 10
        L.bsn_device_sync()
        if txt is None and time.time() - t0 > a.seconds / 2:
            # queue more work first so that the tool samples a busy GPU
            for _ in range(20):
                fn()
            reps += 20
            txt = smi_text()
            L.bsn_device_sync()
    L.bsn_timer_stop(gb.handle, C.byref(ms))
    s.stop = True; s.join()
    rows = s.rows[len(s.rows) // 4:]
    avg = {k: float(np.mean([r[k] for r in rows if k in r])) for k in HW if any(k in r for r in rows)}
    per = ms.value / reps
    print(json.dumps(dict(kernel=name, xkind=a.xkind, na16=a.na16, slices=a.slices, tag=a.tag, ms=round(per, 3), TBps=round(bytes_pass / per / 1e9, 3),
                          hwmon_avg=avg, smi=txt)), flush=True)


if not a.only16:
    loop("counts", lambda: ba.bed_counts(gb))
for nv in ((16,) if a.only16 else (16, 8)):
    op = ba.ScaledOp(gb, None, None, sc["center"], sc["scale"], slices=a.slices)
    if a.only16:
        assert gb.sample_major()   # (the product of a solve runs on the sample-major copy)
    X = ba.DeviceArray.from_numpy(panel(a.m, nv))
    R = ba.DeviceArray.from_numpy(panel(a.n, nv))
    Y = ba.DeviceArray(a.n, nv); Z = ba.DeviceArray(a.m, nv)
    loop("cprod%d" % nv, lambda: op.cprod(R, Z))
    loop("prod%d" % nv, lambda: op.prod(X, Y))
