"""Diagnostic: build the kernel with -DCTR_WAVELOG on the GPU box, render the bench frame and
analyse per-wave start/end times (resident waves over time, tail, cost per tile).
usage (through gpurun): python scripts/gpu_wavelog.py [extra -D flags]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cutrace_amd import build

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import cutrace_amd as ca
s = ca.HostScene.load("scene/bunny.json")
ds = ca.DeviceScene(s)
ds.render(bounces=5)
r = ds.render(bounces=5)
n = r["normal"].reshape(1080, 1920, 3)[::8, ::8].copy().view(np.uint32)
st, en = n[..., 0].astype(np.int64), n[..., 1].astype(np.int64)
t0 = st.min()
st -= t0; en -= t0
dur = (en - st) / 100.0   # us
print("kernel_ms", r["kernel_ms"], "waves", st.size, "span_us", en.max() / 100.0)
print("wave lifetime us: mean %%.1f  p50 %%.1f  p90 %%.1f  max %%.1f" %% (dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()))
T = int(en.max()) + 1
res = np.zeros(T + 1)
np.add.at(res, st.ravel(), 1); np.add.at(res, en.ravel(), -1)
res = np.cumsum(res)[:T]
B = 20
print("resident waves (of 4096 slots) per 1/%%d of the span:" %% B)
print(" ".join("%%4d" %% res[i * T // B:(i + 1) * T // B].mean() for i in range(B)))
print("slot-time used / (4096 * span): %%.3f" %% (dur.sum() * 100 / (4096.0 * T)))
print("start time of wave by tile row (us), every 9th row:")
for y in range(0, 135, 9):
    print("  row %%3d start %%7.1f..%%7.1f  mean dur %%6.1f max %%6.1f" %% (y, st[y].min() / 100, st[y].max() / 100, dur[y].mean(), dur[y].max()))
late = en > 0.9 * T
print("waves still running in the last 10%%%% of the span:", int(late.sum()), "mean dur", dur[late].mean())
np.save("gpurun_out/wavelog.npy", np.stack([st, en, n[..., 2].astype(np.int64)]))
''' % ROOT


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    build.build_host()
    lib = os.path.join(ROOT, "gpurun_out", "libwavelog.so")
    cmd = [build.hipcc(), *build.HIP_FLAGS, "-DCTR_WAVELOG", *sys.argv[1:], "-shared", "-o", lib, *build.HIP_SRCS]
    subprocess.run(cmd, check=True)
    env = dict(os.environ, CUTRACE_AMD_LIB=lib)
    subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=ROOT, timeout=300, check=True)
    os.remove(lib)


if __name__ == "__main__":
    main()
