/* tests/oggshim/ogg/ogg.h - TEST INFRASTRUCTURE.

   Minimal stand-in for the part of libogg's <ogg/ogg.h> that the reference's
   examples/encoder_example.c uses (libogg is a container-framing dependency of
   the example, `ogg >= 1.3` in the reference's configure.ac:91, and is not
   installed here; it is not part of the codec arithmetic).  Written from the
   Ogg bitstream format (RFC 3533): pages of up to 255 lacing values with the
   CRC-32 (polynomial 0x04c11db7, no reflection) over header + body.  With it
   the UNMODIFIED encoder_example.c compiles and writes a well-formed .ogv
   whose packets tests/test_encoder_example.py extracts again. */
#ifndef ODHIP_TEST_OGG_H
#define ODHIP_TEST_OGG_H
#include <stdint.h>
#include <stdlib.h>

typedef int64_t ogg_int64_t;
typedef int32_t ogg_int32_t;
typedef uint32_t ogg_uint32_t;

#define _ogg_malloc malloc
#define _ogg_calloc calloc
#define _ogg_realloc realloc
#define _ogg_free free

typedef struct {
  unsigned char *packet;
  long bytes;
  long b_o_s;
  long e_o_s;
  ogg_int64_t granulepos;
  ogg_int64_t packetno;
} ogg_packet;

typedef struct {
  unsigned char *header;
  long header_len;
  unsigned char *body;
  long body_len;
} ogg_page;

typedef struct {
  /* pending (not yet paged) data */
  unsigned char *body_data;
  long body_storage;
  long body_fill;
  int *lacing_vals;          /* 0..255, | 0x100 on the first segment of a packet */
  ogg_int64_t *granule_vals; /* granulepos of the packet ending at this segment, else -1 */
  long lacing_storage;
  long lacing_fill;
  /* the page handed out last (valid until the next pageout/flush) */
  unsigned char header[282];
  unsigned char *page_body;
  long page_body_storage;
  int e_o_s;      /* an e_o_s packet has been submitted */
  int eos_paged;  /* ... and the page carrying it has been produced */
  int b_o_s;      /* the first page has been produced */
  long serialno;
  long pageno;
  int continued;  /* the next page starts inside a packet */
  ogg_int64_t packetno;
} ogg_stream_state;

int ogg_stream_init(ogg_stream_state *os, int serialno);
int ogg_stream_clear(ogg_stream_state *os);
int ogg_stream_packetin(ogg_stream_state *os, ogg_packet *op);
int ogg_stream_pageout(ogg_stream_state *os, ogg_page *og);
int ogg_stream_flush(ogg_stream_state *os, ogg_page *og);
int ogg_stream_eos(ogg_stream_state *os);
ogg_int64_t ogg_page_granulepos(const ogg_page *og);
#endif
