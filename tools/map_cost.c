/* GPU-box tool (tools/map_cost.py builds and runs it): what a consumer of ttmlrender's buffers
 * pays for touching them.  filesrc ! ttmlrender ! caps ! probe ! fakesink, where `probe` is a
 * GstBaseTransform of this file working in place:
 *   none   passes every buffer on untouched (a sink that never looks)
 *   read   maps it for reading (a converter, an encoder that uploads)
 *   write  maps it READWRITE, as an element that works in place does: on READONLY memory
 *          GStreamer copies the frame first
 * and the frames/s from the first buffer to the last are printed.
 * (An appsink was the first harness: GStreamer 1.14's loses a wake-up between its pusher and
 * its puller now and then when max-buffers is set -- both found waiting on its one condition
 * variable, the element's threads idle; gpurun_out/hang_backtraces.txt of round 4.)
 *   map_cost <ttml> <width> <height> none|read|write */
#include <gst/gst.h>
#include <gst/base/gstbasetransform.h>
#include <stdio.h>
#include <string.h>

typedef struct { GstBaseTransform parent; } Probe;
typedef struct { GstBaseTransformClass parent_class; } ProbeClass;
static int probe_mode;
static guint probe_n, probe_copied;
static gint64 probe_t0, probe_t1;
static guint64 probe_sum;
GType probe_get_type (void);
G_DEFINE_TYPE (Probe, probe, GST_TYPE_BASE_TRANSFORM);

static GstFlowReturn
probe_transform_ip (GstBaseTransform *bt, GstBuffer *buf)
{
  if (probe_mode) {
    GstMapInfo info;
    GstMemory *before = gst_buffer_peek_memory (buf, 0);
    if (gst_buffer_map (buf, &info, probe_mode == 2 ? GST_MAP_READWRITE : GST_MAP_READ)) {
      /* (a pixel per 64 KB: the frame is there; an in-place element would go on to touch all of it) */
      for (gsize k = 0; k < info.size; k += 65536) {
        probe_sum += info.data[k];
        if (probe_mode == 2)
          info.data[k] ^= 1;
      }
      gst_buffer_unmap (buf, &info);
    }
    if (gst_buffer_peek_memory (buf, 0) != before)
      probe_copied++;
  }
  if (probe_n == 0)
    probe_t0 = g_get_monotonic_time ();
  probe_t1 = g_get_monotonic_time ();
  probe_n++;
  return GST_FLOW_OK;
}

static void
probe_class_init (ProbeClass *klass)
{
  static GstStaticPadTemplate sink_t = GST_STATIC_PAD_TEMPLATE ("sink", GST_PAD_SINK, GST_PAD_ALWAYS, GST_STATIC_CAPS_ANY);
  static GstStaticPadTemplate src_t = GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS, GST_STATIC_CAPS_ANY);
  gst_element_class_add_static_pad_template (GST_ELEMENT_CLASS (klass), &sink_t);
  gst_element_class_add_static_pad_template (GST_ELEMENT_CLASS (klass), &src_t);
  gst_element_class_set_static_metadata (GST_ELEMENT_CLASS (klass), "probe", "Filter", "maps buffers", "flutl");
  GST_BASE_TRANSFORM_CLASS (klass)->transform_ip = probe_transform_ip;
  GST_BASE_TRANSFORM_CLASS (klass)->transform_ip_on_passthrough = TRUE;
}

static void
probe_init (Probe *p)
{
  /* none / read: the buffer goes on as it came (no writability asked for); write: in place */
  gst_base_transform_set_passthrough (GST_BASE_TRANSFORM (p), probe_mode != 2);
  gst_base_transform_set_in_place (GST_BASE_TRANSFORM (p), TRUE);
}

int
main (int argc, char **argv)
{
  gst_init (&argc, &argv);
  if (argc < 5)
    return 2;
  probe_mode = !strcmp (argv[4], "read") ? 1 : !strcmp (argv[4], "write") ? 2 : 0;
  gst_element_register (NULL, "flutlprobe", GST_RANK_NONE, probe_get_type ());
  gchar *desc = g_strdup_printf ("filesrc location=%s ! ttmlrender ! video/x-raw,width=%s,height=%s ! "
      "flutlprobe ! fakesink sync=false enable-last-sample=false", argv[1], argv[2], argv[3]);
  GError *err = NULL;
  GstElement *pipe = gst_parse_launch (desc, &err);
  if (!pipe) {
    fprintf (stderr, "%s\n", err ? err->message : "parse error");
    return 1;
  }
  gst_element_set_state (pipe, GST_STATE_PLAYING);
  GstBus *bus = gst_element_get_bus (pipe);
  GstMessage *msg = gst_bus_timed_pop_filtered (bus, GST_CLOCK_TIME_NONE, GST_MESSAGE_EOS | GST_MESSAGE_ERROR);
  int rc = 0;
  if (msg && GST_MESSAGE_TYPE (msg) == GST_MESSAGE_ERROR) {
    GError *e = NULL;
    gst_message_parse_error (msg, &e, NULL);
    fprintf (stderr, "error: %s\n", e ? e->message : "?");
    rc = 1;
  }
  if (msg)
    gst_message_unref (msg);
  gst_element_set_state (pipe, GST_STATE_NULL);
  printf ("mode=%s buffers=%u frames_per_s=%.1f memories_replaced_by_copies=%u checksum=%" G_GUINT64_FORMAT "\n",
      argv[4], probe_n, probe_n > 1 ? (probe_n - 1) / ((probe_t1 - probe_t0) * 1e-6) : 0.0, probe_copied, probe_sum);
  gst_object_unref (bus);
  gst_object_unref (pipe);
  return rc ? rc : (probe_n ? 0 : 1);
}
