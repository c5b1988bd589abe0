#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- builds the *reference's own* ttml plugin, from the
# sources where they lie under /root/reference, into oracle/_ref/ (git-ignored).
# Nothing is copied into the repo and no stand-in header/library is written:
#  * the meson-generated config.h is only included under HAVE_CONFIG_H, which we
#    do not define; its option macros (plugins/ttml/meson.build:21-28) are passed
#    with -D instead;
#  * fluc.h's only job is to include fluc_config.h (generated) and then the
#    threads + bwmeter headers (libs/fluc/flu-codec-sdk/fluc/fluc.h:11-20); we
#    pre-define its include guard and force-include those two real headers.
# Source list = plugins/ttml/meson.build:4-28 + libs/flu/downloader/meson.build:6-9
# + the fluc bwmeter/threads sources the downloader links.
# Third-party stack: /opt/conda (GStreamer 1.14.0, cairo 1.16.0, pango 1.45.3,
# pixman 0.40.0) -- see SURVEY.md §8c.
set -euo pipefail
REF=${REF:-/root/reference}
C=${CONDA:-/opt/conda}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/plugins/ttml" ]; then
  echo "build_ref: $REF not present (GPU box?) -- keeping prebuilt $OUT" >&2
  exit 0
fi
if [ ! -f "$C/include/gstreamer-1.0/gst/gst.h" ]; then
  echo "build_ref: GStreamer dev stack not found under $C" >&2
  exit 0
fi
mkdir -p "$OUT"
T="$REF/plugins/ttml"
SRCS="$T/gstfluttml.c $T/gstttmlbase.c $T/gstttmltype.c $T/gstttmlattribute.c
 $T/gstttmlstate.c $T/gstttmlevent.c $T/gstttmlspan.c $T/gstttmlutils.c
 $T/gstttmlnamespace.c $T/gstttmlstyle.c $T/gstttmlparse.c
 $T/gstttmlsegmentedparse.c $T/gstttmlrender.c $T/gstttmlblur.c
 $REF/libs/flu/downloader/lib/fludownloader.c
 $REF/libs/flu/downloader/lib/fludownloaderhelper.c
 $REF/libs/fluc/flu-codec-sdk/fluc/bwmeter/fluc_bwmeter.c
 $REF/libs/fluc/flu-codec-sdk/fluc/bwmeter/fluc_bwmeter_base.c
 $REF/libs/fluc/flu-codec-sdk/fluc/bwmeter/fluc_bwmeter_sock.c
 $REF/libs/fluc/flu-codec-sdk/fluc/threads/fluc_barrier.c
 $REF/libs/fluc/flu-codec-sdk/fluc/threads/fluc_monitor.c
 $REF/libs/fluc/flu-codec-sdk/fluc/threads/fluc_mutex.c"
INC="-I$T -I$REF/libs/common -I$REF/libs/flu/downloader/lib
 -I$REF/libs/fluc/flu-codec-sdk -I$REF/libs/fluc/flu-codec-sdk/fluc
 -I$C/include/gstreamer-1.0 -I$C/include/glib-2.0 -I$C/lib/glib-2.0/include
 -I$C/lib/gstreamer-1.0/include -I$C/include/libxml2 -I$C/include/pango-1.0
 -I$C/include/cairo -I$C/include/pixman-1 -I$C/include/harfbuzz
 -I$C/include/freetype2 -I$C/include/fribidi -I$C/include"
DEFS="-DVERSION=\"oracle\" -DPACKAGE=\"flu-plugins-oss\" -DPACKAGE_NAME=\"flu-plugins-oss\"
 -DBUILD_TTMLPARSE=1 -DBUILD_TTMLRENDER=1 -DFLUC_USE_THREADS=1 -D_FLUC_H_
 -include fluc/threads/fluc_threads.h -include fluc/bwmeter/fluc_bwmeter.h"
# shellcheck disable=SC2086
gcc -shared -fPIC -O2 -w $DEFS $INC $SRCS \
  -L$C/lib -Wl,-rpath,$C/lib -lgstreamer-1.0 -lgstbase-1.0 -lgstvideo-1.0 \
  -lgobject-2.0 -lglib-2.0 -lxml2 -lpangocairo-1.0 -lpango-1.0 -lcairo \
  -lpixman-1 -lcurl -lm -o "$OUT/libgstfluttml.so"
echo "build_ref: built $OUT/libgstfluttml.so"
