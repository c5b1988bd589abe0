/*
 * flutl_pack.h -- scene-pack writer: the display lists + atlas of the scene
 * states of one TTML document, as recorded by the element (FLUTL_DUMP_PACK).
 * Read back by scene_pack.py for the parity tests and for bench.py, so that
 * the GPU box needs neither fonts nor Pango to replay real scenes.
 *
 * Layout (little endian):
 *   "FLPK1\0\0\0"  u32 width, height, n_states, atlas_bytes
 *   n_states x { u32 n_regions, n_prims; flutl_region[]; flutl_prim[] }
 *   atlas bytes
 */
#ifndef FLUTL_PACK_H
#define FLUTL_PACK_H

#include <stddef.h>
#include <stdint.h>
#include "flutl.h"

typedef struct FlutlPackWriter FlutlPackWriter;

FlutlPackWriter *flutl_pack_writer_new (const char *path);
void flutl_pack_writer_add (FlutlPackWriter *w, int width, int height,
    const flutl_region *regions, uint32_t n_regions, const flutl_prim *prims,
    uint32_t n_prims, const uint8_t *atlas, size_t atlas_size);
/* Write the states accumulated so far (with the latest atlas) and start over.
 * The first flush goes to <path>, later ones to <path>.1, <path>.2, ... */
void flutl_pack_writer_flush (FlutlPackWriter *w);
void flutl_pack_writer_free (FlutlPackWriter *w);

#endif
