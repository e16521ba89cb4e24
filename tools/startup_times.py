"""Where a short run's start-up goes: dlopen of libsvo_hip.so, the first HIP call, svo_create for the runner's
configuration (batch 256, 1241x376), host buffers.  Usage: python tools/startup_times.py [lk|orb] [batch]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
t1 = time.perf_counter()
lib = pkg.load_library()
t2 = time.perf_counter()
n = C.c_int(0)
lib.svo_device_count(C.byref(n))
t3 = time.perf_counter()
mode = sys.argv[1] if len(sys.argv) > 1 else "lk"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
kw = dict(track_mode=pkg.MODE_ORB) if mode == "orb" else {}
ctx = pkg.Context(1241, 376, max_batch=B, **kw)
t4 = time.perf_counter()
ctx2 = pkg.Context(1241, 376, max_batch=B, **kw)
t5 = time.perf_counter()
h = ctx.host_frames(B + 1, 1280)
t6 = time.perf_counter()
ctx.sync()
print(f"import package {t1 - t0:.3f} s, dlopen {t2 - t1:.3f}, first HIP call (device count) {t3 - t2:.3f}, "
      f"svo_create #1 {t4 - t3:.3f}, svo_create #2 {t5 - t4:.3f}, svo_host_alloc({(B + 1) * 376 * 1280 / 1e6:.0f} MB) {t6 - t5:.3f}")
