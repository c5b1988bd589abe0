#!/usr/bin/env python3
"""Deterministic TTML generator for the BASELINE.json configs.

    python tools/gen_config_ttml.py cfg2 [seed] > cfg2.ttml

cfg0: 1280x720,  2 regions (EBU-TT-D like), 3 intervals
cfg1: 1920x1080, 4 styled regions, one with opacity, 60 states via <set> on color
cfg2: 3840x2160, 8 regions with background boxes + 3px black outline, 96px font,
      2 of 8 with opacity < 1, 120 states
cfg4: 7680x4320, 16 overlapping regions, animated origin/background/opacity,
      160px font, half with outline, 240 states
Only DejaVu fonts are named (the only families of this image).
"""
import sys

CORPUS = [
    "The quick brown fox jumps over the lazy dog",
    "Pack my box with five dozen liquor jugs now",
    "Sphinx of black quartz, judge my vow today",
    "How vexingly quick daft zebras jump around",
    "Bright vixens jump; dozy fowl quack loudly",
    "Jackdaws love my big sphinx of quartz rock",
    "Five quacking zephyrs jolt my wax bed here",
    "Waltz, bad nymph, for quick jigs vex them",
    "Crazy Fredrick bought many very exquisite opal jewels",
    "We promptly judged antique ivory buckles for the prize",
]
SIZES = {"cfg0": (1280, 720), "cfg1": (1920, 1080), "cfg2": (3840, 2160),
         "cfg4": (7680, 4320)}


class Rng:  # splitmix64
    def __init__(self, seed):
        self.s = (seed * 0x9E3779B97F4A7C15 + 0x1234567) & (2**64 - 1)

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)

    def rint(self, lo, hi):
        return lo + self.next() % (hi - lo + 1)

    def color(self, alpha=None, lo=0, hi=255):
        a = alpha if alpha is not None else 255
        return "#%02X%02X%02X%02X" % (self.rint(lo, hi), self.rint(lo, hi),
                                      self.rint(lo, hi), a)


HEAD = ('<?xml version="1.0" encoding="UTF-8"?>\n'
        '<tt xmlns="http://www.w3.org/ns/ttml" '
        'xmlns:tts="http://www.w3.org/ns/ttml#styling" '
        'xmlns:ttp="http://www.w3.org/ns/ttml#parameter" xml:lang="en">\n')


def text(rng, lines):
    out = []
    for _ in range(lines):
        s = CORPUS[rng.rint(0, len(CORPUS) - 1)]
        out.append(s[:rng.rint(24, min(48, len(s)))])
    return "<br/>".join(out)


def gen(cfg, seed, n_states=None):
    """n_states (cfg1 / cfg2 only): more scene states than the config names -- a longer
    stream of the same kind (bench.py's steady-state figure; other texts and colours: the
    generator draws them in sequence)."""
    rng = Rng(seed)
    more_states = n_states
    W, H = SIZES[cfg]
    regs, body = [], []
    if cfg == "cfg0":
        regs.append('<region xml:id="top" tts:origin="10% 5%" tts:extent="80% 20%" '
                    'tts:backgroundColor="#000000C0" tts:displayAlign="before" tts:textAlign="center"/>')
        regs.append('<region xml:id="bottom" tts:origin="10% 75%" tts:extent="80% 20%" '
                    'tts:backgroundColor="#00004080" tts:displayAlign="after" tts:textAlign="center"/>')
        body.append('<p region="top" begin="0s" end="2s" tts:fontFamily="DejaVu Sans" '
                    'tts:color="white">%s</p>' % text(rng, 1))
        body.append('<p region="bottom" begin="1s" end="3s" tts:fontFamily="DejaVu Sans" '
                    'tts:color="#FFFF00FF">%s</p>' % text(rng, 2))
    elif cfg in ("cfg1", "cfg2"):
        n = 4 if cfg == "cfg1" else 8
        font = 54 if cfg == "cfg1" else 96
        n_states = more_states or (60 if cfg == "cfg1" else 120)
        slot = 100.0 / n
        for i in range(n):
            alpha = rng.rint(0x99, 0xE0) if cfg == "cfg2" else (0x80 if i % 2 else 0xC0)
            extra = ""
            if cfg == "cfg2":
                extra += ' tts:textOutline="black 3px"'
                if i in (2, 5):
                    extra += ' tts:opacity="0.%d"' % rng.rint(55, 85)
            elif i == 1:
                extra += ' tts:opacity="0.5"'
            regs.append('<region xml:id="r%d" tts:origin="10%% %.2f%%" tts:extent="80%% %.2f%%" '
                        'tts:backgroundColor="%s" tts:displayAlign="center" tts:textAlign="%s"%s/>'
                        % (i, i * slot + 0.4, slot - 0.8, rng.color(alpha, 0, 96),
                           ("center", "left", "right")[i % 3], extra))
            sets = []
            for k in range(n_states):
                if k % n == i:
                    sets.append('<set begin="%.2fs" end="%.2fs" tts:color="%s"/>'
                                % (k * 0.5, k * 0.5 + 0.5, rng.color(255, 64, 255)))
            span = ""
            if cfg == "cfg1":
                span = ' <span tts:color="%s" tts:backgroundColor="%s">%s</span>' % (
                    rng.color(0xC0, 128, 255), rng.color(0x80), "styled span")
            body.append('<p region="r%d" begin="0s" end="%.2fs" tts:fontFamily="%s" '
                        'tts:fontSize="%dpx" tts:color="%s">%s%s%s</p>'
                        % (i, n_states * 0.5, ("DejaVu Sans", "DejaVu Serif")[i % 2], font,
                           rng.color(255, 160, 255), "".join(sets), text(rng, 2), span))
    elif cfg == "cfg4":
        n, font, n_states = 16, 160, 240
        for i in range(n):
            ox, oy = rng.rint(0, 45), rng.rint(0, 80)
            extra = ' tts:zIndex="%d"' % (i + 1)
            # An explicit base opacity: the reference gives an attribute that only
            # <set> children define a zero-filled base value (gstttmlbase.c:697-705,
            # "FIXME: This can be certainly improved"), i.e. opacity 0 outside the
            # set's interval -- without this line every frame of the document is
            # blank in the reference (as round 1's cfg4 was).  Half of the regions
            # are translucent, so that opacity groups overlap.
            extra += ' tts:opacity="%s"' % ("1.0" if i % 4 in (0, 3)
                                            else "0.%d" % rng.rint(55, 90))
            if i % 2:
                extra += ' tts:textOutline="black 4px"'
            sets = []
            for k in range(n_states):
                if k % n != i:
                    continue
                kind = (k // n) % 3
                t0, t1 = k / 120.0 * 2, (k + 1) / 120.0 * 2
                if kind == 0:
                    sets.append('<set begin="%.4fs" end="%.4fs" tts:origin="%d%% %d%%"/>'
                                % (t0, t1, rng.rint(0, 45), rng.rint(0, 80)))
                elif kind == 1:
                    sets.append('<set begin="%.4fs" end="%.4fs" tts:backgroundColor="%s"/>'
                                % (t0, t1, rng.color(rng.rint(0x60, 0xE0), 0, 128)))
                else:
                    sets.append('<set begin="%.4fs" end="%.4fs" tts:opacity="0.%d"/>'
                                % (t0, t1, rng.rint(30, 90)))
            regs.append('<region xml:id="r%d" tts:origin="%d%% %d%%" tts:extent="52%% 14%%" '
                        'tts:backgroundColor="%s" tts:displayAlign="center" tts:textAlign="center"%s>%s</region>'
                        % (i, ox, oy, rng.color(rng.rint(0x60, 0xE0), 0, 128), extra, "".join(sets)))
            body.append('<p region="r%d" begin="0s" end="%.4fs" tts:fontFamily="%s" '
                        'tts:fontSize="%dpx" tts:color="%s">%s</p>'
                        % (i, n_states / 120.0 * 2, ("DejaVu Sans", "DejaVu Serif")[i % 2], font,
                           rng.color(255, 160, 255), text(rng, 2)))
    return (HEAD + " <head><layout>\n  " + "\n  ".join(regs) + "\n </layout></head>\n"
            " <body><div>\n  " + "\n  ".join(body) + "\n </div></body>\n</tt>\n")


if __name__ == "__main__":
    cfg = sys.argv[1]
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else {"cfg0": 0, "cfg1": 1, "cfg2": 2, "cfg4": 4}[cfg]
    sys.stdout.write(gen(cfg, seed))
