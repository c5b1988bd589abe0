/*
 * flutl_state_fix.c -- memory-safe gst_ttml_state_save_data for the product
 * plugin.
 *
 * The reference's version (plugins/ttml/gstttmlstate.c:428-444) takes the
 * caller's text buffer with g_realloc () and keeps it in the hash of saved
 * data, although its only caller (gstttmlbase.c:428-431) goes on appending
 * characters to that very buffer and frees it at dispose (:1596): the saved
 * image dangles from the first smpte:image on, and the process aborts with
 * "double free" at teardown.  build_plugin.sh compiles the reference file with
 * that one definition renamed away; this one stores a private copy in the
 * layout gst_ttml_state_restore_data (:446-467) reads back: a gint32 length
 * followed by the bytes.
 */
#include <string.h>
#include <gst/gst.h>

#include "gstttmlstate.h"

void
gst_ttml_state_save_data (GstTTMLState *state, guint8 *data, gint length,
    const gchar *id)
{
  guint8 *copy;
  gint32 len32 = length;

  if (!state->saved_data)
    state->saved_data = g_hash_table_new_full (g_str_hash, g_str_equal, g_free, g_free);
  copy = g_malloc ((gsize) length + 4);
  memcpy (copy, &len32, 4);
  memcpy (copy + 4, data, (gsize) length);
  g_hash_table_insert (state->saved_data, g_strdup (id), copy);
}
