/* tests/oggshim/oggshim.c - TEST INFRASTRUCTURE: see ogg/ogg.h. */
#include <string.h>
#include "ogg/ogg.h"

static ogg_uint32_t crc_table[256];
static int crc_ready;

static void crc_init(void) {
  int i;
  int j;
  for (i = 0; i < 256; i++) {
    ogg_uint32_t r;
    r = (ogg_uint32_t)i << 24;
    for (j = 0; j < 8; j++) r = (r << 1) ^ ((r & 0x80000000U) ? 0x04c11db7U : 0);
    crc_table[i] = r;
  }
  crc_ready = 1;
}

static ogg_uint32_t crc_update(ogg_uint32_t crc, const unsigned char *p, long n) {
  while (n-- > 0) crc = (crc << 8) ^ crc_table[((crc >> 24) & 0xff) ^ *p++];
  return crc;
}

int ogg_stream_init(ogg_stream_state *os, int serialno) {
  if (!crc_ready) crc_init();
  memset(os, 0, sizeof(*os));
  os->serialno = serialno;
  return 0;
}

int ogg_stream_clear(ogg_stream_state *os) {
  free(os->body_data);
  free(os->lacing_vals);
  free(os->granule_vals);
  free(os->page_body);
  memset(os, 0, sizeof(*os));
  return 0;
}

int ogg_stream_packetin(ogg_stream_state *os, ogg_packet *op) {
  long nseg;
  long i;
  if (os->e_o_s) return -1;
  nseg = op->bytes/255 + 1;
  if (os->body_fill + op->bytes > os->body_storage) {
    os->body_storage = 2*(os->body_fill + op->bytes) + 1024;
    os->body_data = (unsigned char *)realloc(os->body_data, os->body_storage);
  }
  if (os->lacing_fill + nseg > os->lacing_storage) {
    os->lacing_storage = 2*(os->lacing_fill + nseg) + 32;
    os->lacing_vals = (int *)realloc(os->lacing_vals, os->lacing_storage*sizeof(int));
    os->granule_vals = (ogg_int64_t *)realloc(os->granule_vals,
     os->lacing_storage*sizeof(ogg_int64_t));
  }
  if (!os->body_data || !os->lacing_vals || !os->granule_vals) return -1;
  memcpy(os->body_data + os->body_fill, op->packet, op->bytes);
  os->body_fill += op->bytes;
  for (i = 0; i < nseg - 1; i++) {
    os->lacing_vals[os->lacing_fill + i] = 255;
    os->granule_vals[os->lacing_fill + i] = -1;
  }
  os->lacing_vals[os->lacing_fill + nseg - 1] = (int)(op->bytes%255);
  os->granule_vals[os->lacing_fill + nseg - 1] = op->granulepos;
  os->lacing_vals[os->lacing_fill] |= 0x100;
  os->lacing_fill += nseg;
  os->packetno++;
  if (op->e_o_s) os->e_o_s = 1;
  return 0;
}

/* Builds one page out of the pending segments.  force: page out whatever is
   pending; otherwise only once more than 4096 body bytes end at a packet
   boundary or 255 segments are pending (the nominal page size libogg aims
   for).  The first page of a stream holds exactly the first packet. */
static int page_out(ogg_stream_state *os, ogg_page *og, int force) {
  long nseg;
  long bytes;
  long i;
  ogg_int64_t granule;
  int flags;
  ogg_uint32_t crc;
  long maxseg;
  if (os->lacing_fill == 0) return 0;
  maxseg = os->lacing_fill > 255 ? 255 : os->lacing_fill;
  nseg = 0;
  bytes = 0;
  if (!os->b_o_s) {
    /* initial page: the first packet only */
    for (nseg = 0; nseg < maxseg; nseg++) {
      bytes += os->lacing_vals[nseg] & 0xff;
      if ((os->lacing_vals[nseg] & 0xff) < 255) {
        nseg++;
        break;
      }
    }
    force = 1;
  }
  else {
    long done_bytes;
    long done_seg;
    done_bytes = 0;
    done_seg = 0;
    for (nseg = 0; nseg < maxseg; nseg++) {
      if (done_bytes > 4096 && (os->lacing_vals[nseg] & 0x100)) break;
      bytes += os->lacing_vals[nseg] & 0xff;
      if ((os->lacing_vals[nseg] & 0xff) < 255) {
        done_bytes = bytes;
        done_seg = nseg + 1;
      }
    }
    (void)done_seg;
    if (os->e_o_s) force = 1;
    if (!force && !(nseg == 255 || done_bytes > 4096)) return 0;
  }
  granule = -1;
  for (i = 0; i < nseg; i++) if ((os->lacing_vals[i] & 0xff) < 255) granule = os->granule_vals[i];
  flags = 0;
  if (os->continued) flags |= 0x01;
  if (!os->b_o_s) flags |= 0x02;
  if (os->e_o_s && nseg == os->lacing_fill) flags |= 0x04;
  memcpy(os->header, "OggS", 4);
  os->header[4] = 0;
  os->header[5] = (unsigned char)flags;
  for (i = 0; i < 8; i++) os->header[6 + i] = (unsigned char)((uint64_t)granule >> (8*i));
  for (i = 0; i < 4; i++) os->header[14 + i] = (unsigned char)((ogg_uint32_t)os->serialno >> (8*i));
  for (i = 0; i < 4; i++) os->header[18 + i] = (unsigned char)((ogg_uint32_t)os->pageno >> (8*i));
  memset(os->header + 22, 0, 4);
  os->header[26] = (unsigned char)nseg;
  for (i = 0; i < nseg; i++) os->header[27 + i] = (unsigned char)(os->lacing_vals[i] & 0xff);
  if (bytes > os->page_body_storage) {
    os->page_body_storage = bytes + 1024;
    os->page_body = (unsigned char *)realloc(os->page_body, os->page_body_storage);
    if (!os->page_body) return 0;
  }
  memcpy(os->page_body, os->body_data, bytes);
  og->header = os->header;
  og->header_len = 27 + nseg;
  og->body = os->page_body;
  og->body_len = bytes;
  crc = crc_update(0, og->header, og->header_len);
  crc = crc_update(crc, og->body, og->body_len);
  for (i = 0; i < 4; i++) os->header[22 + i] = (unsigned char)(crc >> (8*i));
  /* consume */
  os->continued = (os->lacing_vals[nseg - 1] & 0xff) == 255;
  memmove(os->body_data, os->body_data + bytes, os->body_fill - bytes);
  os->body_fill -= bytes;
  memmove(os->lacing_vals, os->lacing_vals + nseg, (os->lacing_fill - nseg)*sizeof(int));
  memmove(os->granule_vals, os->granule_vals + nseg,
   (os->lacing_fill - nseg)*sizeof(ogg_int64_t));
  os->lacing_fill -= nseg;
  os->pageno++;
  os->b_o_s = 1;
  if (flags & 0x04) os->eos_paged = 1;
  return 1;
}

int ogg_stream_pageout(ogg_stream_state *os, ogg_page *og) {
  return page_out(os, og, 0);
}

int ogg_stream_flush(ogg_stream_state *os, ogg_page *og) {
  return page_out(os, og, 1);
}

int ogg_stream_eos(ogg_stream_state *os) {
  return os->eos_paged;
}

ogg_int64_t ogg_page_granulepos(const ogg_page *og) {
  uint64_t g;
  int i;
  g = 0;
  for (i = 7; i >= 0; i--) g = (g << 8) | og->header[6 + i];
  return (ogg_int64_t)g;
}
