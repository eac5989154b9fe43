#!/usr/bin/env python
"""r06 item 3: what saturates the throughput partition?  Shader clock (clock_probe.hip: s_memtime / s_memrealtime beside the workload)
and board power (hwmon) sampled WHILE (a) the codec decoder runs alone on 176 / 192 / 208 / 224 / 256 CUs with the generation
partition idle, (b) refinement alone, (c) a generation phase alone on 64 CUs, (d) the pipelined bench shape.  Also (e): decode on
192 CUs WITH generation phases running next door (the pipeline's contention, without its scheduler).
    python tools/r06/saturation_probe.py > gpurun_out/.../saturation.txt"""
import ctypes as C
import glob
import os
import subprocess
import sys
import threading
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from sopro_amd import hip  # noqa: E402

DEV = torch.device("cuda:0")


def build_probe():
    so = "/tmp/clock_probe.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(R, "tools", "micro", "clock_probe.hip")])
    lib = C.CDLL(so)
    lib.clock_probe_launch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    return lib


def power_files():
    out = []
    for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        out += glob.glob(pat)
    return sorted(out)


class Sampler:
    """Background thread: a clock-probe wave every `period` s (each spins `spin_us` of RTC time) + the hwmon power reading."""

    def __init__(self, lib, spin_us=200, period=0.002):
        self.lib, self.spin, self.period = lib, int(spin_us * 100), period
        self.host = C.c_void_p()
        assert lib.clock_probe_alloc(C.byref(self.host), 1) == 0
        self.st = C.c_void_p()
        assert lib.clock_probe_stream(C.byref(self.st)) == 0
        self.view = (C.c_uint64 * 4).from_address(self.host.value)
        self.pf = power_files()
        self.rows, self._stop, self.th = [], False, None

    def _run(self):
        while not self._stop:
            t = time.perf_counter()
            self.lib.clock_probe_launch(self.host, self.spin, self.st)
            self.lib.clock_probe_sync(self.st)
            dr, dc, xcc = self.view[0], self.view[1], self.view[2]
            p = None
            if self.pf:
                try:
                    p = int(open(self.pf[0]).read()) / 1e6
                except Exception:  # noqa: BLE001
                    p = None
            if dr:
                self.rows.append((t, 100.0 * dc / dr, int(xcc), p, (time.perf_counter() - t) * 1e3))
            time.sleep(self.period)

    def start(self):
        self.rows, self._stop = [], False
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def stop(self):
        self._stop = True
        self.th.join()
        r = self.rows
        if not r:
            return "no samples"
        mhz = np.array([x[1] for x in r])
        pw = np.array([x[3] for x in r if x[3] is not None])
        lat = np.array([x[4] for x in r])
        s = f"clock MHz p10/p50/p90 {np.percentile(mhz, 10):.0f}/{np.percentile(mhz, 50):.0f}/{np.percentile(mhz, 90):.0f} (n={len(r)}, probe round trip p50 {np.percentile(lat, 50):.2f} ms)"
        if len(pw):
            s += f"; power W p50/max {np.percentile(pw, 50):.0f}/{pw.max():.0f}"
        byx = {}
        for x in r:
            byx.setdefault(x[2], []).append(x[1])
        s += "; by XCC " + " ".join(f"{k}:{np.median(v):.0f}" for k, v in sorted(byx.items()))
        return s


def main():
    torch.cuda.set_device(0)
    hip.set_host_wait(True, 0)
    lib = build_probe()
    print("hwmon power files:", power_files())
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")[:1]:
        print(f, open(f).read().replace("\n", " | "))
    for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap")[:1]:
        print(f, open(f).read().strip())
    try:
        print(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower"], capture_output=True, text=True, timeout=30).stdout[-1500:])
    except Exception as e:  # noqa: BLE001
        print("rocm-smi:", e)
    sm = Sampler(lib)
    sm.start(); time.sleep(0.5); print("idle:", sm.stop())

    tts, cfg, mc, wn, mn = bench.build_engine("cuda:0", "f32")
    rng = np.random.default_rng(0)
    toks = torch.from_numpy(rng.integers(0, 2048, size=(64, 200, 32)).astype(np.int32)).to(DEV)
    ids, _ = bench.make_inputs(0)
    voices = [tts.prepare_reference(ref_tokens_tq=v) for v in bench.make_voices(0, 32)]
    ids128, refs128 = (ids * 4), (voices * 4)
    codec, model = tts.codec, tts.model
    saved = (codec.stream, model.stream, model.bulk_stream, model.prep_stream)

    def on(cus, lo=None):
        return hip.cu_range_stream(256 - cus if lo is None else lo, cus, DEV)

    def time_decode(n=12):
        for _ in range(3):
            codec.decode_batch(toks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            codec.decode_batch(toks)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    # (a) decode alone at several partition sizes
    print("\n(a) codec decode of 64 x 200 frames alone (generation partition idle), recorded sequence, ms per decode:")
    for cus in (256, 224, 208, 192, 176, 160, 128):
        st = on(cus)
        codec.stream = st
        codec._graphs.clear()
        time_decode(2)
        sm.start()
        ms = time_decode(16)
        print(f"  {cus:3d} CUs: {ms:7.3f} ms   {sm.stop()}", flush=True)
        codec._graphs.clear()
        torch.cuda.synchronize()
    # (b) refinement alone on 192 CUs
    print("\n(b) refinement of 128 x 200 alone on 192 / 256 CUs:")
    prep = model.phase_cond(ids128, refs128, max_frames=199, style_strength=float(cfg.style_strength))
    cond = prep["cond_ar"][:, :200].contiguous()
    cb0 = torch.from_numpy(rng.integers(0, 2048, size=(128, 200)).astype(np.int32)).to(DEV)
    for cus in (256, 192):
        st = on(cus)
        model.bulk_stream = st
        model._nar_graphs.clear()
        for _ in range(3):
            model._nar_pass(cond, cond.stride(0), cb0, cb0.stride(0), [200] * 128, 128, 200, sync=True, raw=True)
        sm.start()
        t0 = time.perf_counter()
        for _ in range(16):
            model._nar_pass(cond, cond.stride(0), cb0, cb0.stride(0), [200] * 128, 128, 200, sync=True, raw=True)
        ms = (time.perf_counter() - t0) / 16 * 1e3
        print(f"  {cus:3d} CUs: {ms:7.3f} ms per 128-row pass   {sm.stop()}")
        model._nar_graphs.clear()
    model.bulk_stream = saved[2]
    # (c) generation alone on 64 CUs (128 rows) and (e) decode on 192 CUs beside it
    print("\n(c) one 128-row generation phase alone on 64 CUs; (e) decode on 192 CUs beside generation phases:")
    model.stream = on(64, 0)
    model.ar_tiles_wide = "1x2"
    model._ar_cache.clear()
    arkw = dict(max_frames=199, top_p=0.9, temperature=1.05, anti_loop=True, style_strength=float(cfg.style_strength), min_gen_frames=None, prep=prep, seed=1)
    model.phase_ar(ids128, refs128, **arkw)
    model.phase_ar(ids128, refs128, **arkw)
    torch.cuda.synchronize()
    sm.start()
    t0 = time.perf_counter()
    for _ in range(3):
        model.phase_ar(ids128, refs128, **arkw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"  generation alone: {ms:7.2f} ms per 200-frame phase ({ms / 200 * 1e3:.1f} us per frame)   {sm.stop()}")
    codec.stream = on(192)
    codec._graphs.clear()
    time_decode(2)
    ms_alone = time_decode(12)
    stop = [False]

    def gen_loop():
        torch.cuda.set_device(0)
        while not stop[0]:
            model.phase_ar(ids128, refs128, **arkw)

    th = threading.Thread(target=gen_loop, daemon=True)
    th.start()
    time.sleep(0.2)
    sm.start()
    ms_beside = time_decode(16)
    s = sm.stop()
    stop[0] = True
    th.join()
    torch.cuda.synchronize()
    print(f"  decode on 192 CUs: alone {ms_alone:.3f} ms, beside ONE running generation phase {ms_beside:.3f} ms   {s}")
    codec._graphs.clear()
    model._ar_cache.clear()
    codec.stream, model.stream, model.bulk_stream, model.prep_stream = saved
    torch.cuda.synchronize()


if __name__ == "__main__":
    try:
        main()
    except Exception:  # noqa: BLE001
        import traceback

        traceback.print_exc(file=sys.stdout)
        sys.stdout.flush()
        os._exit(1)
