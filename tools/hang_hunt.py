#!/usr/bin/env python3
"""GPU box: several ttmlrender elements in one gst-launch under rocgdb; if the process is
still running after SECONDS, every thread's backtrace.   python tools/hang_hunt.py [attempts] [seconds]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gstenv  # noqa: E402

NAMES = ["k16_unknown_glyphs", "k18_recolour", "k13_vertical_group", "k20_hexbox_vertical", "k06_animation"]


def main():
    attempts = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    seconds = sys.argv[2] if len(sys.argv) > 2 else "40"
    tmp = tempfile.mkdtemp()
    cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "--no-fault"]
    for n in NAMES:
        cmd += ["filesrc", "location=" + os.path.join(ROOT, "tests", "golden", "scenes", n + ".ttml"),
                "!", "ttmlrender", "!", "video/x-raw,width=320,height=180",
                "!", "queue", "max-size-buffers=64", "!", "filesink", "location=" + os.path.join(tmp, n + ".raw")]
    env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR)
    import time
    if os.environ.get("HUNT_HOLD_GPU"):      # (as pytest and bench.py do: another process on the device)
        import torch
        hold = torch.zeros(1 << 20, device="cuda")
        torch.cuda.synchronize()
    for a in range(attempts):
        t0 = time.time()
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp)
        try:
            out, err = p.communicate(timeout=float(seconds))
            print(f"attempt {a}: rc {p.returncode}, {time.time() - t0:.1f} s", flush=True)
            if p.returncode:
                print(err[-2000:])
            continue
        except subprocess.TimeoutExpired:
            pass
        print(f"attempt {a}: still running after {seconds} s: the threads' backtraces", flush=True)
        g = subprocess.run(["/opt/rocm/bin/rocgdb", "-q", "-batch", "-p", str(p.pid), "-ex", "set pagination off",
                            "-ex", "thread apply all bt 30"], capture_output=True, text=True, timeout=300)
        print(g.stdout[-60000:])
        print(g.stderr[-2000:])
        p.kill()
        p.communicate()
        return 0
    print("no attempt hung")
    return 0


if __name__ == "__main__":
    sys.exit(main())
