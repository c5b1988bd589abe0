#!/usr/bin/env python3
"""GPU box: where the element's host threads spend their time on a stream with new
outlined text in every cue -- a long one (default 3000 cues, as bench_data/typical_4k.ttml
makes them) under tools/sprof.c.   python tools/sprof_stream.py [cues] [top]"""
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_timing as E  # noqa: E402
import gstenv  # noqa: E402

WORDS = ("the quick brown fox jumps over lazy dog subtitles scroll along bottom of picture "
         "while a tonight").split()


def document(cues, path):
    rng = random.Random(5)
    head = open(os.path.join(ROOT, "bench_data", "typical_4k.ttml")).read().split("<body>")[0]
    with open(path, "w") as f:
        f.write(head + "<body><div>\n")
        for i in range(cues):
            a = " ".join(rng.choice(WORDS) for _ in range(6))
            b = " ".join(rng.choice(WORDS) for _ in range(5))
            f.write(f'  <p region="r" begin="{i * 0.5:.2f}s" end="{i * 0.5 + 0.5:.2f}s">{a}<br/>{b}</p>\n')
        f.write(" </div></body>\n</tt>\n")


def main():
    cues = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    top = sys.argv[2] if len(sys.argv) > 2 else "30"
    tmp = tempfile.gettempdir()
    doc = os.path.join(tmp, "typical_long.ttml")
    if os.environ.get("SPROF_DOC") == "cfg2":      # (the cfg2 kind of stream, `cues` states long)
        import gen_config_ttml
        open(doc, "w").write(gen_config_ttml.gen("cfg2", 2, n_states=cues))
    else:
        document(cues, doc)
    so = os.path.join(tmp, "sprof.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "sprof.c"),
                           "-o", so, "-lpthread"])
    runs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    first = E.measure(doc, 3840, 2160)                       # (warm, and the rate without the profiler)
    print(first, flush=True)
    first = E.measure(doc, 3840, 2160)
    # (sampling starts where the stream does: the parse and the start-up before it are left out)
    delay = max(300, int((first["pipeline_wall_s"] - first["span_s"] - 0.12) * 1000))
    print("sampling from", delay, "ms", flush=True)
    pre = gstenv.env_for(gstenv.HIP_PLUGIN_DIR).get("LD_PRELOAD", "")
    outs = []
    for r in range(runs):
        outs.append(os.path.join(tmp, f"sprof{r}.txt"))
        print(E.measure(doc, 3840, 2160, extra_env={"LD_PRELOAD": (pre + ":" + so).strip(":"),
                                                     "SPROF_OUT": outs[-1], "SPROF_THREAD": "filesrc",
                                                     "SPROF_HZ": "5000", "SPROF_DELAY_MS": str(delay)}), flush=True)
    # (the base class sorts spans in by insertion, O(cues^2): the parse is left out)
    subprocess.call([sys.executable, os.path.join(ROOT, "tools", "sprof_report.py"), *outs, "top=" + top,
                     "under=gst_ttmlrender_gen_buffer"])
    subprocess.call([sys.executable, os.path.join(ROOT, "tools", "sprof_report.py"), *outs, "top=12"])


if __name__ == "__main__":
    main()
