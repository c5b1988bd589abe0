#!/usr/bin/env python3
"""GPU box: several ttmlrender elements in one gst-launch, each of the two pipelines ONCE; if
a process is still running after SECONDS, every thread's backtrace (rocgdb).  A single-shot
capture tool: it is not a reproduction loop and takes no repeat count (a hang is diagnosed from
the backtrace of the run that showed it; a CPU-side race is then reproduced on the CPU build,
FLUTL_DUMP_ONLY / TSan, not by looping pipelines on the GPU pool).
    python tools/hang_hunt.py [seconds]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gstenv  # noqa: E402

NAMES = ["k16_unknown_glyphs", "k18_recolour", "k13_vertical_group", "k20_hexbox_vertical", "k06_animation"]


def main():
    attempts = 2                # (the two pipelines below, once each)
    seconds = sys.argv[1] if len(sys.argv) > 1 else "40"
    tmp = tempfile.mkdtemp()
    cmd = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q", "--no-fault"]
    for n in NAMES:
        cmd += ["filesrc", "location=" + os.path.join(ROOT, "tests", "golden", "scenes", n + ".ttml"),
                "!", "ttmlrender", "!", "video/x-raw,width=320,height=180",
                "!", "queue", "max-size-buffers=64", "!", "filesink", "location=" + os.path.join(tmp, n + ".raw")]
    # ... and bench.py's four chains of the 4K document
    cmd4k = [os.path.join(gstenv.CONDA, "bin", "gst-launch-1.0"), "-q"]
    for _ in range(4):
        cmd4k += ["filesrc", "location=" + os.path.join(ROOT, "bench_data", "cfg2.ttml"), "!", "ttmlrender", "!",
                  "video/x-raw,width=3840,height=2160", "!", "fakesink"]
    cmds = [cmd, cmd4k]
    env = gstenv.env_for(gstenv.HIP_PLUGIN_DIR)
    # (a debugger that is not the pipeline's parent may attach only if the pipeline says so)
    src = os.path.join(tmp, "ptracer_any.c")
    with open(src, "w") as f:
        f.write("#include <sys/prctl.h>\n__attribute__((constructor)) static void any(void)"
                "{ prctl(PR_SET_PTRACER, PR_SET_PTRACER_ANY, 0, 0, 0); }\n")
    so = os.path.join(tmp, "ptracer_any.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", src, "-o", so])
    env["LD_PRELOAD"] = (env.get("LD_PRELOAD", "") + ":" + so).strip(":")
    import time
    for a in range(attempts):
        t0 = time.time()
        cmd = cmds[a % 2]
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
        print(g.stdout[-120000:])
        print(g.stderr[-2000:])
        # (whatever the debugger could do: what every thread is blocked in)
        for t in sorted(os.listdir(f"/proc/{p.pid}/task"), key=int):
            def rd(name):
                try:
                    return open(f"/proc/{p.pid}/task/{t}/{name}").read().strip()
                except OSError as e:
                    return repr(e)
            print("task", t, rd("comm"), "| state", rd("stat").rsplit(")", 1)[-1].split()[0], "| wchan", rd("wchan"),
                  "| syscall", rd("syscall")[:80])
        p.kill()
        p.communicate()
        return 0
    print("no attempt hung")
    return 0


if __name__ == "__main__":
    sys.exit(main())
