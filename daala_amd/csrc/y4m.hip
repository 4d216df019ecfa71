/* y4m.hip - YUV4MPEG2 input for the frame-batch path (host code; SURVEY.md 8(f) rank 4,
   input side).  The reference reads its input in examples/encoder_example.c:89-160
   (stream header tags), :190-400 (chroma types) and :449-508 (FRAME headers + planes);
   this reader accepts what the batched path can take today - progressive 8-bit 4:2:0
   (C420, C420jpeg, C420mpeg2, C420paldv, or no C tag) - and refuses everything else
   with ODHIP_EIMPL instead of guessing.  Planes come out tightly packed: luma w x h,
   chroma ((w + 1) >> 1) x ((h + 1) >> 1), the layout odhip_pipe_set_pictures and
   odhip_image_planes_copy_pad take. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/daala_hip.h"

struct odhip_y4m {
  FILE *f;
  int w;
  int h;
};

extern "C" odhip_y4m *odhip_y4m_open(const char *path, int *pic_w, int *pic_h, int *fps_n, int *fps_d,
 int *err) {
  int dummy = 0;
  if (!err) err = &dummy;
  *err = ODHIP_EINVAL;
  if (!path) return nullptr;
  FILE *f = fopen(path, "rb");
  if (!f) return nullptr;
  char buf[256];
  int n = 0;
  int c;
  while ((c = fgetc(f)) != EOF && c != '\n' && n < (int)sizeof(buf) - 1) buf[n++] = (char)c;
  buf[n] = '\0';
  if (c != '\n' || strncmp(buf, "YUV4MPEG2", 9) != 0) {
    fclose(f);
    return nullptr;
  }
  int w = 0;
  int h = 0;
  int fn = 0;
  int fd = 0;
  char interlace = 'p';
  char chroma[16] = "420";
  for (char *tag = strtok(buf + 9, " "); tag; tag = strtok(nullptr, " ")) {
    switch (tag[0]) {
      case 'W': w = atoi(tag + 1); break;
      case 'H': h = atoi(tag + 1); break;
      case 'F': if (sscanf(tag + 1, "%d:%d", &fn, &fd) != 2) fn = fd = 0; break;
      case 'I': interlace = tag[1]; break;
      case 'C': strncpy(chroma, tag + 1, sizeof(chroma) - 1); chroma[sizeof(chroma) - 1] = '\0'; break;
      default: break;     /* A (aspect) and X (comment) tags do not change the sample layout */
    }
  }
  if (w <= 0 || h <= 0) {
    fclose(f);
    return nullptr;
  }
  const bool c420 = !strcmp(chroma, "420") || !strcmp(chroma, "420jpeg") || !strcmp(chroma, "420mpeg2")
   || !strcmp(chroma, "420paldv");
  if (!c420 || (interlace != 'p' && interlace != '?')) {
    /* 4:4:4, 4:2:2, 4:1:1, mono, high bit depths, interlaced material: not implemented */
    *err = ODHIP_EIMPL;
    fclose(f);
    return nullptr;
  }
  odhip_y4m *y = (odhip_y4m *)calloc(1, sizeof(*y));
  if (!y) {
    fclose(f);
    return nullptr;
  }
  y->f = f;
  y->w = w;
  y->h = h;
  if (pic_w) *pic_w = w;
  if (pic_h) *pic_h = h;
  if (fps_n) *fps_n = fn;
  if (fps_d) *fps_d = fd;
  *err = ODHIP_SUCCESS;
  return y;
}

extern "C" int odhip_y4m_read(odhip_y4m *y, uint8_t *luma, uint8_t *cb, uint8_t *cr) {
  if (!y || !luma || !cb || !cr) return ODHIP_EINVAL;
  char frame[6];
  const size_t got = fread(frame, 1, 6, y->f);
  if (got == 0) return 0;                                /* end of stream */
  if (got != 6 || memcmp(frame, "FRAME", 5) != 0) return ODHIP_EFAULT;    /* loss of framing */
  if (frame[5] != '\n') {
    /* frame parameters up to the end of the line */
    int c;
    int k = 0;
    while ((c = fgetc(y->f)) != EOF && c != '\n' && k < 121) k++;
    if (c != '\n') return ODHIP_EFAULT;
  }
  const size_t ny = (size_t)y->w*y->h;
  const size_t nc = (size_t)((y->w + 1) >> 1)*((y->h + 1) >> 1);
  if (fread(luma, 1, ny, y->f) != ny || fread(cb, 1, nc, y->f) != nc || fread(cr, 1, nc, y->f) != nc) {
    return ODHIP_EFAULT;
  }
  return 1;
}

/* Steps over one FRAME without reading its samples (a rank of a frame-sharded encode reads
   only the frames it owns): 1, 0 at the end of the stream, negative on loss of framing. */
extern "C" int odhip_y4m_skip(odhip_y4m *y) {
  if (!y) return ODHIP_EINVAL;
  char frame[6];
  const size_t got = fread(frame, 1, 6, y->f);
  if (got == 0) return 0;
  if (got != 6 || memcmp(frame, "FRAME", 5) != 0) return ODHIP_EFAULT;
  if (frame[5] != '\n') {
    int c;
    int k = 0;
    while ((c = fgetc(y->f)) != EOF && c != '\n' && k < 121) k++;
    if (c != '\n') return ODHIP_EFAULT;
  }
  const size_t ny = (size_t)y->w*y->h;
  const size_t nc = (size_t)((y->w + 1) >> 1)*((y->h + 1) >> 1);
  const size_t nbytes = ny + 2*nc;
  /* Every byte but the last is stepped over; the last one is READ, so that a truncated frame is
     reported here, by the rank that skips it, exactly as the rank that owns it sees it from
     odhip_y4m_read (a seek past the end of a file succeeds).  A stream that cannot seek (a pipe:
     ESPIPE) is read and discarded. */
  if (nbytes > 1 && fseek(y->f, (long)(nbytes - 1), SEEK_CUR) != 0) {
    char buf[4096];
    size_t left = nbytes - 1;
    clearerr(y->f);
    while (left > 0) {
      const size_t n = left < sizeof(buf) ? left : sizeof(buf);
      if (fread(buf, 1, n, y->f) != n) return ODHIP_EFAULT;
      left -= n;
    }
  }
  if (fgetc(y->f) == EOF) return ODHIP_EFAULT;
  return 1;
}

extern "C" void odhip_y4m_close(odhip_y4m *y) {
  if (!y) return;
  fclose(y->f);
  free(y);
}
