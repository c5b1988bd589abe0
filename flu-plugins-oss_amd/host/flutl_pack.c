/* flutl_pack.c -- see flutl_pack.h */
#include "flutl_pack.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct FlutlPackWriter
{
  char *path;
  int flushes;
  int width, height;
  uint8_t *states;       /* serialized states */
  size_t states_len, states_cap;
  uint32_t n_states;
  uint8_t *atlas;
  size_t atlas_size;
};

static void
append (FlutlPackWriter *w, const void *src, size_t n)
{
  if (w->states_len + n > w->states_cap) {
    size_t cap = w->states_cap ? w->states_cap : 1 << 16;
    while (cap < w->states_len + n)
      cap *= 2;
    w->states = realloc (w->states, cap);
    if (!w->states)
      abort ();
    w->states_cap = cap;
  }
  memcpy (w->states + w->states_len, src, n);
  w->states_len += n;
}

FlutlPackWriter *
flutl_pack_writer_new (const char *path)
{
  FlutlPackWriter *w = calloc (1, sizeof *w);
  if (!w)
    abort ();
  w->path = strdup (path);
  return w;
}

void
flutl_pack_writer_add (FlutlPackWriter *w, int width, int height,
    const flutl_region *regions, uint32_t n_regions, const flutl_prim *prims,
    uint32_t n_prims, const uint8_t *atlas, size_t atlas_size)
{
  uint32_t hdr[2] = { n_regions, n_prims };
  if (w->n_states && (w->width != width || w->height != height))
    flutl_pack_writer_flush (w);
  w->width = width;
  w->height = height;
  append (w, hdr, sizeof hdr);
  append (w, regions, (size_t) n_regions * sizeof (flutl_region));
  append (w, prims, (size_t) n_prims * sizeof (flutl_prim));
  w->n_states++;
  /* the atlas only grows within a document: keep the latest copy */
  w->atlas = realloc (w->atlas, atlas_size ? atlas_size : 1);
  if (!w->atlas)
    abort ();
  memcpy (w->atlas, atlas, atlas_size);
  w->atlas_size = atlas_size;
}

void
flutl_pack_writer_flush (FlutlPackWriter *w)
{
  char name[4096];
  FILE *f;
  uint32_t hdr[4];
  if (!w->n_states)
    return;
  if (w->flushes == 0)
    snprintf (name, sizeof name, "%s", w->path);
  else
    snprintf (name, sizeof name, "%s.%d", w->path, w->flushes);
  f = fopen (name, "wb");
  if (f) {
    hdr[0] = (uint32_t) w->width;
    hdr[1] = (uint32_t) w->height;
    hdr[2] = w->n_states;
    hdr[3] = (uint32_t) w->atlas_size;
    fwrite ("FLPK1\0\0\0", 1, 8, f);
    fwrite (hdr, 4, 4, f);
    fwrite (w->states, 1, w->states_len, f);
    fwrite (w->atlas, 1, w->atlas_size, f);
    fclose (f);
  }
  w->flushes++;
  w->n_states = 0;
  w->states_len = 0;
}

void
flutl_pack_writer_free (FlutlPackWriter *w)
{
  if (!w)
    return;
  flutl_pack_writer_flush (w);
  free (w->states);
  free (w->atlas);
  free (w->path);
  free (w);
}
