"""The oracle is the parity checker: it must not rely on undefined behaviour or out-of-bounds reads
(a descriptor sample beside the image, a window over the pyramid frame ...).  Builds oracle/*.c with
-fsanitize=address,undefined and drives every entry point (CPU only; GPU sanitizers are not
available on the pool)."""
import os
import subprocess
import sys

import conftest


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    odir = os.path.join(conftest.ROOT, "oracle")
    so = str(tmp_path / "libsvo_oracle_asan.so")
    srcs = [os.path.join(odir, f) for f in sorted(os.listdir(odir)) if f.endswith(".c")]
    subprocess.check_call(["gcc", "-O1", "-g", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-fopenmp",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-std=gnu11", "-shared", "-o", so] + srcs + ["-lm"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "tests", "_asan_driver.py"), so], env=env,
                       capture_output=True, timeout=600)
    out = r.stdout.decode() + r.stderr.decode()
    assert r.returncode == 0 and "sanitizer run complete" in out and "ERROR: AddressSanitizer" not in out and \
        "runtime error" not in out, out[-4000:]
