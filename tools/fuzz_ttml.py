#!/usr/bin/env python3
"""Differential fuzzing of the whole path: random TTML documents are rendered by the
REFERENCE plugin and recorded by the fluttml-hip element (record-only, CPU) + composited
by the oracle; every frame must be bit-identical.  Builder container (needs oracle/_ref).

    [FUZZ_SCALE=k] python tools/fuzz_ttml.py [first_seed [count]]
"""
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import compare_ref  # noqa: E402

WORDS = ("subtitle overlay Glyph MAVERICK jumpy quiz, fix & go; Why? 12:45 öäü naïve "
         "Test_under line ÆØ x y z").split()
# FUZZ_UNKNOWN=1: words no font of this image covers (Pango's hex boxes)
if os.environ.get("FUZZ_UNKNOWN"):
    WORDS += ["字幕", "한글", "テ", "\U00020000x", "中",
              # (round 4: code points Pango's unknown-glyph renderer has branches of their own
              # for -- the line separator's arrow where the font has no glyph -- and the ones
              # HarfBuzz / Pango take care of before they get there)
              "li\u2028ne", "\u2028", "a\u00a0b", "z\u200bw", "s\u00adhy", "\ufeffbom", "i\u3000d",
              "l\u200er", "\u2060j"]
FAMS = ["DejaVu Sans", "DejaVu Serif", "DejaVu Sans Mono", "sansSerif", "serif", "monospace",
        "default"]


# FUZZ_SETS=1: colour animations inside paragraphs and spans (<set> on tts:color /
# tts:backgroundColor, several per element, alphas of every class): what the element's
# region cache plays back in other colours instead of recording anew
SETS = bool(os.environ.get("FUZZ_SETS"))


def color_sets(r):
    if not SETS or r.random() < 0.35:
        return ""
    out = ""
    for _ in range(r.randrange(1, 3)):
        attr = r.choice(["tts:color", "tts:color", "tts:backgroundColor"])
        t = r.choice([0.0, 0.25, 0.5])
        # mostly opaque -> opaque (the common case), sometimes across alpha classes
        alpha = 255 if r.random() < 0.6 else None
        for k in range(r.randrange(1, 5)):
            out += f'<set begin="{t:.2f}s" end="{t + 0.25:.2f}s" {attr}="{col(r, alpha)}"/>'
            t += r.choice([0.25, 0.25, 0.5])
    return out


# FUZZ_E2E=1 (GPU box): the element renders for real and its frames are compared with
# the reference's byte for byte -- kernel, read-back and device-made atlas entries
# included; only then are blurred outlines generated (they need the device)
E2E = bool(os.environ.get("FUZZ_E2E"))
BLUR = E2E


def col(r, alpha=None):
    a = alpha if alpha is not None else r.choice([255, 255, 255, 192, 128, 64, 0])
    return "#%02X%02X%02X%02X" % (r.randrange(256), r.randrange(256), r.randrange(256), a)


def length(r, lo, hi, units=("px", "px", "%", "c", "em")):
    u = r.choice(units)
    v = r.uniform(lo, hi)
    if u == "%":
        return "%.1f%%" % (v / 3.2)
    if u == "c":
        return "%.2fc" % (v / 10.0)
    if u == "em":
        return "%.2fem" % (v / 16.0)
    return ("%d" % v if r.random() < 0.7 else "%.2f" % v) + "px"


def span_style(r):
    s = []
    if r.random() < 0.5:
        s.append(f'tts:color="{col(r)}"')
    if r.random() < 0.25:
        s.append(f'tts:backgroundColor="{col(r)}"')
    if r.random() < 0.3:
        s.append(f'tts:fontSize="{r.randrange(10, 34)}px"')
    if r.random() < 0.1:
        s.append(f'tts:fontSize="{r.randrange(10, 30)}px {r.randrange(10, 30)}px"')
    if r.random() < 0.25:
        s.append('tts:fontStyle="%s"' % r.choice(["italic", "oblique", "reverseOblique", "normal"]))
    if r.random() < 0.25:
        s.append('tts:fontWeight="%s"' % r.choice(["bold", "normal"]))
    if r.random() < 0.3:
        s.append('tts:textDecoration="%s"' % r.choice(
            ["underline", "lineThrough", "overline", "underline overline", "noUnderline",
             "underline lineThrough overline"]))
    if r.random() < 0.2:
        s.append(f'tts:fontFamily="{r.choice(FAMS)}"')
    if r.random() < 0.08:
        s.append('tts:visibility="hidden"')
    if r.random() < 0.08:
        s.append('tts:unicodeBidi="bidiOverride" tts:direction="rtl"')
    seen, out = set(), []
    for a in " ".join(s).split(" tts:"):         # no attribute twice: XML would be invalid
        name = a.split("=")[0].replace("tts:", "")
        if a and name not in seen:
            seen.add(name)
            out.append(a if a.startswith("tts:") else "tts:" + a)
    return " ".join(out)


def text(r, n):
    return " ".join(r.choice(WORDS) for _ in range(n)).replace("&", "&amp;")


def gen(seed):
    r = random.Random(seed)
    regions, ps = [], []
    for i in range(r.randrange(1, 5)):
        x, y = r.randrange(0, 200), r.randrange(0, 110)
        w, h = r.randrange(60, 320 - x + 40), r.randrange(30, 180 - y + 30)
        a = [f'xml:id="r{i}"', f'tts:origin="{x}px {y}px"', f'tts:extent="{w}px {h}px"']
        if r.random() < 0.7:
            a.append(f'tts:backgroundColor="{col(r)}"')
        if r.random() < 0.4:
            a.append('tts:opacity="%.2f"' % r.choice([0.0, 0.25, 0.5, 0.81, 1.0, r.random()]))
        if r.random() < 0.5:
            a.append('tts:displayAlign="%s"' % r.choice(["before", "center", "after"]))
        if r.random() < 0.5:
            a.append('tts:textAlign="%s"' % r.choice(["left", "center", "right", "start", "end"]))
        if r.random() < 0.35:
            n = r.choice([1, 2, 4])
            a.append('tts:padding="%s"' % " ".join(length(r, 0, 12, ("px", "px", "%")) for _ in range(n)))
        if r.random() < 0.4:
            a.append('tts:textOutline="%s %s%s"' % (r.choice(["black", col(r, 255), col(r)]),
                                                     r.choice(["1px", "2px", "1.5px", "3px", "0.5px"]),
                                                     BLUR and r.random() < 0.5 and
                                                     r.choice([" 1px", " 2px", " 3.5px", " 0.5px"]) or ""))
        if r.random() < 0.25:
            a.append(f'tts:zIndex="{r.randrange(-3, 4)}"')
        if r.random() < 0.15:
            a.append('tts:writingMode="%s"' % r.choice(["tbrl", "tblr", "rltb", "lrtb", "tb", "rl"]))
        if r.random() < 0.2:
            a.append('tts:wrapOption="%s"' % r.choice(["wrap", "noWrap"]))
        if r.random() < 0.2:
            a.append('tts:overflow="%s"' % r.choice(["visible", "hidden"]))
        if r.random() < 0.2:
            a.append('tts:showBackground="%s"' % r.choice(["always", "whenActive"]))
        if r.random() < 0.25:
            a.append('tts:lineHeight="%s"' % r.choice(["normal", "20px", "30px", "125%", "1.5c"]))
        sets = ""
        if r.random() < 0.3:
            attr = r.choice(["tts:backgroundColor", "tts:opacity", "tts:origin", "tts:color"])
            for k in range(r.randrange(1, 3)):
                v = {"tts:backgroundColor": col(r), "tts:opacity": "%.2f" % r.random(),
                     "tts:origin": f"{r.randrange(0, 200)}px {r.randrange(0, 100)}px",
                     "tts:color": col(r)}[attr]
                sets += f'<set begin="{0.5 + k * 0.5:.1f}s" end="{1.0 + k * 0.5:.1f}s" {attr}="{v}"/>'
        regions.append("  <region " + " ".join(a) + (f">{sets}</region>" if sets else "/>"))
        for k in range(r.randrange(0, 3)):
            t0 = r.choice([0, 0, 0.5, 1.0])
            t1 = t0 + r.choice([0.5, 1.0, 2.0])
            body = ""
            for _ in range(r.randrange(1, 4)):
                if r.random() < 0.5:
                    timing = ""
                    if SETS and r.random() < 0.25:      # a span with an interval of its own
                        b0 = r.choice([0.0, 0.25, 0.5])
                        timing = f' begin="{b0:.2f}s" ' + r.choice(
                            [f'end="{b0 + r.choice([0.25, 0.5, 1.0]):.2f}s"',
                             f'dur="{r.choice([0.25, 0.5, 0.75]):.2f}s"', ""])
                    body += f'<span{timing} {span_style(r)}>{color_sets(r)}{text(r, r.randrange(1, 4))}</span> '
                else:
                    body += text(r, r.randrange(1, 4)) + " "
                if r.random() < 0.2:
                    body += "<br/>"
            st = span_style(r)
            if "tts:fontSize" not in st:
                st += f' tts:fontSize="{r.randrange(12, 30)}px"'
            ps.append(f'  <p region="r{i}" begin="{t0}s" end="{t1}s" {st}>{color_sets(r)}{body}</p>')
    # The reference reads an uninitialised pointer when a paragraph holds BOTH an
    # anamorphic font size and text that breaks its Pango markup (a raw '&':
    # pango_parse_markup fails and leaves pango_attr_list unset,
    # gstttmlrender.c:769-771); what follows is undefined, also for later frames.
    import re
    if re.search(r'tts:fontSize="[0-9.]+px [0-9.]+px"', "".join(ps)):
        ps = [p.replace("&amp;", "and") for p in ps]
    cell = ' ttp:cellResolution="32 15"' if r.random() < 0.5 else ""
    return f'''<?xml version="1.0" encoding="UTF-8"?>
<tt xmlns="http://www.w3.org/ns/ttml" xmlns:tts="http://www.w3.org/ns/ttml#styling" xmlns:ttp="http://www.w3.org/ns/ttml#parameter" xml:lang="en"{cell}>
 <head><layout>
{chr(10).join(regions)}
 </layout></head>
 <body><div>
{chr(10).join(ps)}
 </div></body>
</tt>
'''


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    bad = []
    # FUZZ_SCALE=k: the same documents with every px length multiplied by k on a frame
    # k times as large (fonts of 30-100 px at k = 3: other hinting, wider outlines)
    scale = int(os.environ.get("FUZZ_SCALE", "1"))
    for seed in range(first, first + count):
        doc = gen(seed)
        if scale != 1:
            import re
            doc = re.sub(r'(\d+(?:\.\d+)?)px',
                         lambda m: ("%g" % (float(m.group(1)) * scale)) + "px", doc)
        with tempfile.NamedTemporaryFile("w", suffix=".ttml", delete=False) as f:
            f.write(doc)
        try:
            if E2E and os.environ.get("FUZZ_VIDEO"):
                # FUZZ_VIDEO=1: through the element's request pad (fused over-video,
                # SURVEY 8f-3): every output frame = the REFERENCE's overlay frame of
                # that time OVER the video frame, pixman's rule (oracle_lib.over)
                import numpy as np
                import gstenv
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_lib as O
                W, H, fps, n = 320 * scale, 180 * scale, 8, 24
                refb = gstenv.render(gstenv.REF_PLUGIN_DIR, f.name, W, H)
                ref = np.frombuffer(refb, np.uint32).reshape(-1, H, W)
                times = gstenv.buffer_times(gstenv.HIP_PLUGIN_DIR, f.name, W, H)
                vraw, oraw = gstenv.render_over_video(f.name, W, H, n, fps, pattern="smpte")
                video = np.frombuffer(vraw, np.uint32).reshape(n, H, W)
                out = np.frombuffer(oraw, np.uint32).reshape(n, H, W)
                ok = len(times) == len(ref)
                for k in range(n if ok else 0):
                    t = k * 10**9 // fps
                    st = [i for i, (p, d) in enumerate(times) if p <= t and (d is None or t < p + d)]
                    exp = O.over(ref[st[0]].reshape(-1), video[k].reshape(-1)).reshape(H, W) \
                        if st else video[k]
                    ok = ok and bool(np.array_equal(out[k], exp))
            elif E2E:
                import gstenv
                W, H = 320 * scale, 180 * scale
                ref = gstenv.render(gstenv.REF_PLUGIN_DIR, f.name, W, H)
                got = gstenv.render(gstenv.HIP_PLUGIN_DIR, f.name, W, H)
                ok = ref == got
                ref = range(len(ref) // (W * H * 4))
            else:
                ok, ref, pk = compare_ref.compare(f.name, 320 * scale, 180 * scale, verbose=False)
            note = f"{len(ref)} frames"
        except Exception as e:      # noqa: BLE001
            ok, note = None, "ERROR " + repr(e)[:200]
        print(f"seed {seed}: {'ok' if ok else ('MISMATCH' if ok is False else 'error')} ({note})",
              flush=True)
        if not ok:
            bad.append(seed)
            keep = os.path.join(os.environ.get("FUZZ_KEEP_DIR", tempfile.gettempdir()),
                                f"fuzz_{seed}.ttml" if scale == 1 else f"fuzz_x{scale}_{seed}.ttml")
            import shutil
            shutil.move(f.name, keep)
            print("   kept", keep)
        else:
            os.unlink(f.name)
    print("failing seeds:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
