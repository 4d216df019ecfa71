/* pipeline.hip - odhip_pipe: the frame-batch step as ONE C call.

   One step = one pass of the block-transform hot path over F resident 4:2:0
   pictures (what bench.py times and a frame-parallel all-intra encoder would run
   per batch; round 1 drove it from Python with ~100 ctypes calls per step):

     luma chain   (context / stream A)       chroma chain (context / stream B)
       od_img_plane_copy_pad                   od_img_plane_copy_pad
       forward pyramid, 5 levels               forward pyramid, 4 levels
       PVQ band stage, no reference            [wait: references of this step]
       choice                                  PVQ band stage WITH the chroma-from-luma
       chroma-from-luma references  ------>      reference (src/encode.c:1680-1687)
       dequantise + inverse, 5 levels          choice, dequantise + inverse, 4 levels

   The two chains are software-pipelined over steps: the luma chain of step i+1
   overlaps the chroma chain of step i (two reference buffers, events both ways).
   Each chain has its OWN odhip_ctx - scratch, edge strips, job tables and side
   streams are never shared between streams - and the pipe owns every device
   buffer and both streams.

   With cfg.chroma_cfl == 0 chroma goes through the no-reference stage together with
   luma on one stream (round 1's first workload).  Rate tables (the host's od_pvq_rate
   results, one double per candidate) are optional per plane set: without them the
   choice is made on distortion alone.

   Host code only; the kernels are the batched entry points of daala_hip.h. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "od_ctx.cuh"
#include "gen/od_scan_tables.h"

namespace {

constexpr int kStages = ODHIP_PIPE_NSTAGES;
constexpr int kMaxTimed = 4096;

struct PlaneSet {
  int dec;
  int pli;
  int nplanes;
  int w, h;          /* coded plane size */
  int pw, ph;        /* picture size in this plane */
  int nlev;
  uint8_t *pic;      /* the pictures the next step reads: pic_buf[front] */
  uint8_t *pic_buf[2];
  uint8_t *px;
  /* inter mode: the motion-compensated prediction of every picture, its padded plane and
     its pyramid (the reference of every block at every level, mdtmp in the encoder) */
  uint8_t *pred_pic;
  uint8_t *pred_px;
  od_coeff *pred_levels[ODHIP_NBSIZES];
  od_coeff *levels[ODHIP_NBSIZES];
  uint8_t *recon[ODHIP_NBSIZES];
  int16_t *qm[ODHIP_NBSIZES];
  int16_t *qm_inv[ODHIP_NBSIZES];
  int32_t q[ODHIP_NBSIZES][ODHIP_MAX_BANDS];
  /* the chroma set holds F Cb planes, then F Cr planes: per-band steps of the second
     half (pvq_qm_q4[2], src/encode.c:3052-3072) */
  int32_t q2[ODHIP_NBSIZES][ODHIP_MAX_BANDS];
  int plane_split;   /* 0: one table */
  int32_t beta[ODHIP_NBSIZES][ODHIP_MAX_BANDS];
  long nblocks[ODHIP_NBSIZES];
};

}  // namespace

struct odhip_pipe {
  odhip_pipe_config cfg;
  int pic_w, pic_h, W, H;
  odhip_ctx *ctx[2];              /* 0: luma chain, 1: chroma chain */
  hipStream_t stream[2];
  bool serial;
  PlaneSet set[2];
  /* [parity]: luma 0..4, chroma (no-reference mode) 5..8.  With chroma from luma the
     chroma chain of step i reads the luma CHOICES of step i (pulses and choice records,
     odhip_pvq_refjob.luma) while the luma chain of step i + 1 already writes the next ones:
     two sets that share everything but those two buffers; otherwise only [0] is used */
  odhip_pvq_job jobs[2][2*ODHIP_NBSIZES];
  int njobs;
  odhip_pvq_refjob refjobs[2][ODHIP_NBSIZES];
  odhip_pvq_refjob interjobs[2][ODHIP_NBSIZES];   /* inter mode: [plane set][level] */
  bool inter_pending[2];
  double *rate[2][ODHIP_NBSIZES];
  hipEvent_t ev_refs[2];
  hipEvent_t ev_used[2];
  long nstep;
  int pending;                    /* parity of the step whose theta list is unchecked, -1 */
  long reruns;                    /* bands re-run with the host's theta so far */
  long price_reruns;              /* priced choices re-decided with the host libm so far */
  double wait_ms;                 /* host time spent waiting for the margin count */
  long k_range;                   /* bands above ODHIP_PVQ_MAX_K seen at syncs */
  bool record;
  /* odhip_pipe_feed: the pictures of the NEXT step arrive in the back buffers on their
     own stream while the current step computes */
  hipStream_t copy_stream;
  hipEvent_t ev_fed;              /* the back buffers hold the fed pictures */
  hipEvent_t ev_pad[2];           /* the padding kernel of a chain has read its pictures */
  int front;
  bool fed;
  /* odhip_pipe_set_export: what a host entropy coder consumes - choice records and pulse vectors of
     every band - leaves for pinned host memory on a third stream, behind the stage that produced it */
  hipStream_t export_stream;
  uint8_t *export_host;
  /* the packed decisions (export_kernels.hip) of the step being exported, by step parity: the used part of
     the streams of step i leaves while step i + 1 is being packed */
  uint8_t *export_dev[2];
  odhip_export_layout export_lay;
  odhip_export_header *export_hdr[2];   /* pinned: the totals of that step, read by the host one step late */
  hipEvent_t ev_exp_hdr[2];
  hipEvent_t ev_exp_sent[2];      /* the streams of that parity's buffer have left */
  int export_pending;             /* parity of the step whose streams have not been sent yet, -1 */
  long export_stale;              /* steps re-decided by a late resolve after their export had left */
  bool in_flush;
  hipEvent_t ev_exp_luma[2];      /* the luma outputs of parity [i] have left */
  hipEvent_t ev_exp_chroma;       /* the chroma outputs (shared between the parities) have left */
  hipEvent_t ev_chroma_done;
  std::vector<hipEvent_t> timed[kStages];    /* pairs */
  std::vector<void *> owned;
};

namespace {

int alloc(odhip_pipe *p, void **out, size_t bytes, bool zero) {
  void *d = nullptr;
  ODHIP_TRY(hipMalloc(&d, bytes ? bytes : 16));
  p->owned.push_back(d);
  if (zero) ODHIP_TRY(hipMemset(d, 0, bytes ? bytes : 16));
  *out = d;
  return ODHIP_SUCCESS;
}
#define STEP_TRY(expr) \
  do { \
    const int rc_ = (expr); \
    if (rc_) return rc_; \
  } while (0)

#define PIPE_ALLOC(p, ptr, bytes, zero) \
  do { \
    const int rc_ = alloc((p), (void **)&(ptr), (bytes), (zero)); \
    if (rc_) return rc_; \
  } while (0)

struct Timed {
  odhip_pipe *p;
  int stage;
  hipStream_t s;
  bool on;
  Timed(odhip_pipe *p_, int stage_, hipStream_t s_) : p(p_), stage(stage_), s(s_) {
    on = p->record && p->timed[stage].size() < (size_t)2*kMaxTimed;
    if (on) mark();
  }
  ~Timed() {
    if (on) mark();
  }
  void mark() {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(e, s);
    p->timed[stage].push_back(e);
  }
};

int setup_set(odhip_pipe *p, PlaneSet &s, int dec, int pli, int nplanes) {
  const odhip_quant *qt = p->cfg.quant;
  s.dec = dec;
  s.pli = pli;
  s.nplanes = nplanes;
  s.w = p->W >> dec;
  s.h = p->H >> dec;
  s.pw = (p->pic_w + dec) >> dec;
  s.ph = (p->pic_h + dec) >> dec;
  s.nlev = ODHIP_NBSIZES - dec;
  s.plane_split = pli == 1 ? nplanes/2 : 0;
  /* full-precision references: the coded planes and the reconstructions hold int16
     samples; the resident source pictures keep their own depth */
  const size_t px_bytes = p->cfg.fpr_bits ? 2 : 1;
  const size_t pic_bytes = p->cfg.fpr_bits > 8 ? 2 : 1;
  PIPE_ALLOC(p, s.pic_buf[0], (size_t)nplanes*s.pw*s.ph*pic_bytes, true);
  PIPE_ALLOC(p, s.pic_buf[1], (size_t)nplanes*s.pw*s.ph*pic_bytes, true);
  s.pic = s.pic_buf[0];
  PIPE_ALLOC(p, s.px, (size_t)nplanes*s.w*s.h*px_bytes, true);
  s.pred_pic = s.pred_px = nullptr;
  if (p->cfg.inter) {
    PIPE_ALLOC(p, s.pred_pic, (size_t)nplanes*s.pw*s.ph*pic_bytes, true);
    PIPE_ALLOC(p, s.pred_px, (size_t)nplanes*s.w*s.h*px_bytes, true);
  }
  for (int bs = 0; bs < s.nlev; bs++) {
    const int n = 4 << bs;
    const int len = n*n < OD_SCAN_LEN ? n*n : OD_SCAN_LEN;
    PIPE_ALLOC(p, s.levels[bs], sizeof(od_coeff)*(size_t)nplanes*s.w*s.h, true);
    s.pred_levels[bs] = nullptr;
    if (p->cfg.inter) PIPE_ALLOC(p, s.pred_levels[bs], sizeof(od_coeff)*(size_t)nplanes*s.w*s.h, true);
    PIPE_ALLOC(p, s.recon[bs], (size_t)nplanes*s.w*s.h*px_bytes, true);
    PIPE_ALLOC(p, s.qm[bs], sizeof(int16_t)*len, false);
    PIPE_ALLOC(p, s.qm_inv[bs], sizeof(int16_t)*len, false);
    const int off = odhip_qm_offset(bs, dec);
    ODHIP_TRY(hipMemcpy(s.qm[bs], qt->qm + off, sizeof(int16_t)*len, hipMemcpyHostToDevice));
    ODHIP_TRY(hipMemcpy(s.qm_inv[bs], qt->qm_inv + off, sizeof(int16_t)*len, hipMemcpyHostToDevice));
    if (odhip_quant_bands(qt, pli, bs, s.q[bs], s.beta[bs]) < 0) return ODHIP_EINVAL;
    if (pli == 1 && odhip_quant_bands(qt, 2, bs, s.q2[bs], nullptr) < 0) return ODHIP_EINVAL;
    s.nblocks[bs] = (long)nplanes*(s.w/n)*(s.h/n);
  }
  return ODHIP_SUCCESS;
}

int setup_job(odhip_pipe *p, odhip_pvq_job &j, PlaneSet &s, int bs, const odhip_pvq_job *share = nullptr) {
  int nb = 0;
  int len = 0;
  odhip_pvq_band_layout(bs, &nb, nullptr, &len);
  memset(&j, 0, sizeof(j));
  j.d_coef = s.levels[bs];
  j.nplanes = s.nplanes;
  j.w = s.w;
  j.h = s.h;
  j.bs = bs;
  j.d_qm = s.qm[bs];
  j.d_qm_inv = s.qm_inv[bs];
  j.q_band = s.q[bs];
  j.beta_band = s.beta[bs];
  if (s.plane_split) {
    j.q_band2 = s.q2[bs];
    j.plane_split = s.plane_split;
  }
  const long B = s.nblocks[bs];
  /* The two parities of the luma job set share ONE band-record buffer while the chroma chain of
     step i runs beside the luma chain of step i + 1.  That is race-free only because (a) the
     records are written by close-call bands alone and luma_choose resolves them on the host
     BEFORE the next step is enqueued, (b) lref_piece and the inverse read the choice and pulse
     buffers (which ARE per parity), never the records, (c) the unpriced choice kernel runs on the
     luma stream itself.  A new consumer of luma band records on the side stream (e.g. a deferred
     resolve, or a host dump through odhip_pipe_buffer(BUF_BAND) while a step is in flight) must
     give parity 1 its own buffer here. */
  if (share) j.cands.band = share->cands.band;       /* the other parity's set: own pulses and choices */
  else PIPE_ALLOC(p, j.cands.band, sizeof(odhip_pvq_band)*(size_t)B*nb, true);
  PIPE_ALLOC(p, j.cands.y, sizeof(int16_t)*(size_t)2*B*len, true);
  PIPE_ALLOC(p, j.cands.choice, sizeof(int32_t)*(size_t)B*nb*4, true);
  return ODHIP_SUCCESS;
}

int setup_refjob(odhip_pipe *p, odhip_pvq_refjob &j, PlaneSet &s, int bs, const od_coeff *ref,
 const odhip_pvq_refjob *share, const odhip_pvq_job *luma = nullptr) {
  int nb = 0;
  int len = 0;
  odhip_pvq_band_layout(bs, &nb, nullptr, &len);
  memset(&j, 0, sizeof(j));
  j.d_coef = s.levels[bs];
  j.d_ref = ref;
  j.luma = luma;
  j.nplanes = s.nplanes;
  j.w = s.w;
  j.h = s.h;
  j.bs = bs;
  j.is_keyframe = 1;
  j.pli = s.pli;
  j.d_qm = s.qm[bs];
  j.d_qm_inv = s.qm_inv[bs];
  j.q_band = s.q[bs];
  j.beta_band = s.beta[bs];
  if (s.plane_split) {
    j.q_band2 = s.q2[bs];
    j.plane_split = s.plane_split;
  }
  if (share) {
    /* same planes, the other reference buffer: outputs and work vectors are shared
       (the chroma chains of consecutive steps run in order on one stream) */
    j.band = share->band;
    j.items = share->items;
    j.y = share->y;
    j.r16 = share->r16;
    j.x16 = share->x16;
    j.xr = share->xr;
    j.choice = share->choice;
    return ODHIP_SUCCESS;
  }
  const long B = s.nblocks[bs];
  PIPE_ALLOC(p, j.band, sizeof(odhip_pvq_refband)*(size_t)B*nb, true);
  PIPE_ALLOC(p, j.items, (size_t)3*nb*ODHIP_PVQ_REF_SLOTS*B*16, true);
  PIPE_ALLOC(p, j.y, sizeof(int16_t)*(size_t)ODHIP_PVQ_REF_SLOTS*B*len, true);
  PIPE_ALLOC(p, j.r16, sizeof(int16_t)*(size_t)B*len, true);
  PIPE_ALLOC(p, j.x16, sizeof(int16_t)*(size_t)B*len, true);
  PIPE_ALLOC(p, j.xr, sizeof(int16_t)*(size_t)B*len, true);
  PIPE_ALLOC(p, j.choice, sizeof(int32_t)*(size_t)B*nb*16, true);
  return ODHIP_SUCCESS;
}

int pipe_init(odhip_pipe *p) {
  const odhip_pipe_config &c = p->cfg;
  ODHIP_TRY(hipSetDevice(c.device));
  p->pic_w = c.pic_w;
  p->pic_h = c.pic_h;
  p->W = (c.pic_w + 63) & ~63;     /* coded frame size, src/state.c:376-379 */
  p->H = (c.pic_h + 63) & ~63;
  p->serial = c.serial || odhip_env_serial();
  for (int i = 0; i < 2; i++) {
    p->ctx[i] = odhip_create(c.device);
    if (!p->ctx[i]) return ODHIP_EFAULT;
    /* with two chains side by side (chroma from luma, inter) the band stages do not fork
       their searches onto side streams: more concurrency only interleaves the searches of
       one chain (measured: 5.72 -> 5.58 ms per step; ODHIP_PIPE_FORK=3 restores the forks).
       The single chain of the chroma-without-reference mode keeps them (3.56 vs 3.45 ms). */
    const bool two_chains = c.chroma_cfl || c.inter;
    /* ODHIP_PIPE_FORK: a bit mask - bit 0 = the luma chain forks, bit 1 = the chroma chain forks.
       Rounds 1-2 read the variable as a flag ("set = both chains fork"): a value that is not a
       number in 0..3 (e.g. "yes", "true") keeps that meaning; the parsed mask is logged once. */
    int forkmask = 0;
    if (const char *fe = ODHIP_EXP_ENV("ODHIP_PIPE_FORK")) {
      char *end = nullptr;
      const long v = strtol(fe, &end, 10);
      forkmask = (end == fe || *end != '\0' || v < 0 || v > 3) ? 3 : (int)v;
      static bool logged = false;
      if (!logged) {
        fprintf(stderr, "odhip_pipe: ODHIP_PIPE_FORK=%s -> fork mask %d (bit 0 luma chain, bit 1 chroma chain)\n",
         fe, forkmask);
        logged = true;
      }
    }
    odhip_ctx_set_serial(p->ctx[i], p->serial || (two_chains && !(forkmask >> i & 1)));
    odhip_ctx_set_fpr(p->ctx[i], c.fpr_bits != 0);
  }
  /* Experiment knob: ODHIP_PIPE_CUSPLIT=n (1..7) gives the luma chain n of every 8 compute
     units and the chroma chain the other 8 - n (hipExtStreamCreateWithCUMask) instead of
     letting the two chains share every CU. */
  const char *split_env = ODHIP_EXP_ENV("ODHIP_PIPE_CUSPLIT");
  const int split = split_env ? atoi(split_env) : 0;
  if (!p->serial && split >= 1 && split <= 7) {
    hipDeviceProp_t prop;
    ODHIP_TRY(hipGetDeviceProperties(&prop, c.device));
    const int words = (prop.multiProcessorCount + 31)/32;
    std::vector<uint32_t> ma(words), mb(words);
    const uint32_t byte_a = (1u << split) - 1u;
    for (int i = 0; i < words; i++) {
      ma[i] = byte_a*0x01010101u;
      mb[i] = ~ma[i];
    }
    ODHIP_TRY(hipExtStreamCreateWithCUMask(&p->stream[0], (uint32_t)words, ma.data()));
    ODHIP_TRY(hipExtStreamCreateWithCUMask(&p->stream[1], (uint32_t)words, mb.data()));
  }
  else if (const char *prio_env = p->serial ? nullptr : ODHIP_EXP_ENV("ODHIP_PIPE_PRIO")) {
    /* Experiment knob: ODHIP_PIPE_PRIO=1 gives the luma chain the highest queue priority the device
       offers and the chroma chain the lowest, 2 the other way round. */
    int lo = 0;
    int hi = 0;
    ODHIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const bool luma_high = atoi(prio_env) == 1;
    ODHIP_TRY(hipStreamCreateWithPriority(&p->stream[0], hipStreamNonBlocking, luma_high ? hi : lo));
    ODHIP_TRY(hipStreamCreateWithPriority(&p->stream[1], hipStreamNonBlocking, luma_high ? lo : hi));
  }
  else {
    ODHIP_TRY(hipStreamCreateWithFlags(&p->stream[0], hipStreamNonBlocking));
    if (p->serial) p->stream[1] = p->stream[0];
    else ODHIP_TRY(hipStreamCreateWithFlags(&p->stream[1], hipStreamNonBlocking));
  }
  int rc = setup_set(p, p->set[0], 0, 0, c.frames);
  if (rc) return rc;
  rc = setup_set(p, p->set[1], 1, 1, 2*c.frames);
  if (rc) return rc;
  p->inter_pending[0] = p->inter_pending[1] = false;
  if (c.inter) {
    /* every plane through the with-reference stage against its own prediction pyramid */
    for (int si = 0; si < 2; si++) {
      for (int bs = 0; bs < p->set[si].nlev; bs++) {
        rc = setup_refjob(p, p->interjobs[si][bs], p->set[si], bs, p->set[si].pred_levels[bs], nullptr);
        if (rc) return rc;
        p->interjobs[si][bs].is_keyframe = 0;
      }
    }
    p->njobs = 0;
    p->pending = -1;
    ODHIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
    ODHIP_TRY(hipEventCreateWithFlags(&p->ev_fed, hipEventDisableTiming));
    ODHIP_TRY(hipEventCreateWithFlags(&p->ev_pad[0], hipEventDisableTiming));
    ODHIP_TRY(hipEventCreateWithFlags(&p->ev_pad[1], hipEventDisableTiming));
    ODHIP_TRY(hipDeviceSynchronize());
    return ODHIP_SUCCESS;
  }
  for (int bs = 0; bs < 5; bs++) {
    rc = setup_job(p, p->jobs[0][bs], p->set[0], bs);
    if (rc) return rc;
  }
  p->njobs = 5;
  if (!c.chroma_cfl) {
    for (int bs = 0; bs < 4; bs++) {
      rc = setup_job(p, p->jobs[0][5 + bs], p->set[1], bs);
      if (rc) return rc;
    }
    p->njobs = 9;
  }
  else {
    PlaneSet &ch = p->set[1];
    for (int bs = 0; bs < 5; bs++) {
      rc = setup_job(p, p->jobs[1][bs], p->set[0], bs, &p->jobs[0][bs]);
      if (rc) return rc;
    }
    for (int par = 0; par < 2; par++) {
      for (int bs = 0; bs < 4; bs++) {
        /* the chroma-from-luma reference of chroma level bs: the choices of luma level
           bs + 1 of the same step, read in place (no reference planes) */
        rc = setup_refjob(p, p->refjobs[par][bs], ch, bs, nullptr, par ? &p->refjobs[0][bs] : nullptr,
         &p->jobs[par][bs + 1]);
        if (rc) return rc;
      }
      ODHIP_TRY(hipEventCreateWithFlags(&p->ev_refs[par], hipEventDisableTiming));
      ODHIP_TRY(hipEventCreateWithFlags(&p->ev_used[par], hipEventDisableTiming));
    }
  }
  p->pending = -1;
  ODHIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
  ODHIP_TRY(hipEventCreateWithFlags(&p->ev_fed, hipEventDisableTiming));
  ODHIP_TRY(hipEventCreateWithFlags(&p->ev_pad[0], hipEventDisableTiming));
  ODHIP_TRY(hipEventCreateWithFlags(&p->ev_pad[1], hipEventDisableTiming));
  ODHIP_TRY(hipDeviceSynchronize());
  return ODHIP_SUCCESS;
}

/* Makes ctx current for the scope, restores the caller's selection afterwards. */
struct Current {
  odhip_ctx *prev;
  explicit Current(odhip_ctx *c) : prev(odhip_get_current()) {
    (void)odhip_make_current(c);
  }
  ~Current() {
    (void)odhip_make_current(prev);
  }
};

int stage_pad_run(odhip_pipe *p, int si, hipStream_t s);
int export_luma(odhip_pipe *p, int par);
int export_chroma(odhip_pipe *p, int par);
int export_finish(odhip_pipe *p);

/* Padding is the only reader of the resident pictures: its completion frees them for the
   next feed. */
int stage_pad(odhip_pipe *p, int si, hipStream_t s) {
  const int rc = stage_pad_run(p, si, s);
  if (rc) return rc;
  ODHIP_TRY(hipEventRecord(p->ev_pad[si], s));
  return ODHIP_SUCCESS;
}

int stage_pad_run(odhip_pipe *p, int si, hipStream_t s) {
  PlaneSet &t = p->set[si];
  Timed tm(p, si ? ODHIP_PIPE_PAD_CHROMA : ODHIP_PIPE_PAD_LUMA, s);
  if (p->cfg.fpr_bits) {
    return odhip_image_planes_copy_pad16(reinterpret_cast<uint16_t *>(t.px), t.w, (long)t.w*t.h, t.w, t.h, t.pic,
     p->cfg.fpr_bits, t.pw, (long)t.pw*t.ph, t.pw, t.ph, t.nplanes, s);
  }
  return odhip_image_planes_copy_pad(t.px, t.w, (long)t.w*t.h, t.w, t.h, t.pic, t.pw, (long)t.pw*t.ph,
   t.pw, t.ph, t.nplanes, s);
}

int stage_pyramid(odhip_pipe *p, int si, hipStream_t s) {
  PlaneSet &t = p->set[si];
  Timed tm(p, si ? ODHIP_PIPE_PYRAMID_CHROMA : ODHIP_PIPE_PYRAMID_LUMA, s);
  return odhip_forward_pyramid(t.levels, t.px, t.w, (long)t.w*t.h, t.nplanes, t.w, t.h, t.dec,
   p->pic_w, p->pic_h, s);
}

int stage_inverse_noref(odhip_pipe *p, int si, hipStream_t s, int jpar) {
  PlaneSet &t = p->set[si];
  Timed tm(p, si ? ODHIP_PIPE_INVERSE_CHROMA : ODHIP_PIPE_INVERSE_LUMA, s);
  return odhip_inverse_levels_pvq(t.recon, t.w, (long)t.w*t.h, p->jobs[jpar] + (si ? 5 : 0), t.nlev, t.dec,
   p->pic_w, p->pic_h, s);
}

/* Choice (only when the host prices or nobody does: with cfg.price the band stage decided
   every band itself, odhip_pvq_ref_bands_decided_multi) and the inverse. */
int chroma_tail(odhip_pipe *p, int par, hipStream_t s) {
  PlaneSet &ch = p->set[1];
  const double lam = p->cfg.pvq_norm_lambda;
  if (!p->cfg.price) {
    Timed tm(p, ODHIP_PIPE_CHOOSE_CHROMA, s);
    STEP_TRY(odhip_pvq_ref_choose_multi(p->refjobs[par], 4, lam, s));
  }
  Timed tm(p, ODHIP_PIPE_INVERSE_CHROMA, s);
  return odhip_inverse_levels_pvq_ref(ch.recon, ch.w, (long)ch.w*ch.h, p->refjobs[par], 4, 1, p->pic_w,
   p->pic_h, s);
}

/* The count of bands inside the device-acos margin of the previous step's
   with-reference stage is checked one step late, so the host never waits inside a
   step; a listed band whose theta the host corrects (never seen outside the forced
   tests) repeats what consumed it. */
int finish_pending(odhip_pipe *p) {
  if (p->pending < 0) return ODHIP_SUCCESS;
  const int par = p->pending;
  p->pending = -1;
  Current cur(p->ctx[1]);
  const bool exporting = p->export_host != nullptr;
  /* (a resolve rewrites choices and pulses of that step on the side stream: behind the pack kernels that read
     them, which run on the same stream) */
  const auto t0 = std::chrono::steady_clock::now();
  /* with cfg.price a band re-run with the host's theta is also decided again by the resolve */
  const int n = odhip_pvq_ref_resolve_finish(p->refjobs[par], 4, p->cfg.pvq_norm_lambda, p->stream[1]);
  p->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (n < 0) return n;
  int m = 0;
  if (p->cfg.price) {
    /* the chroma choices of that step: listed bands are re-decided with the host libm */
    const auto t1 = std::chrono::steady_clock::now();
    m = odhip_pvq_ref_choose_priced_resolve(p->refjobs[par], 4, p->cfg.pvq_norm_lambda, p->stream[1]);
    p->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    if (m < 0) return m;
  }
  p->reruns += n;
  p->price_reruns += m;
  if (n > 0 || m > 0) {
    /* what consumed the choices runs again (the buffers are intact until the next chroma
       chain is enqueued, below) */
    STEP_TRY(chroma_tail(p, par, p->stream[1]));
  }
  /* the resolves and the re-run read that step's luma pulses and choices (the chroma-from-luma
     references, in place): the luma chain of step + 2 reuses those buffers and waits for this */
  if (n > 0 || m > 0) ODHIP_TRY(hipEventRecord(p->ev_used[par], p->stream[1]));
  if (exporting && (n > 0 || m > 0)) {
    /* what left for the host is superseded.  Inside odhip_pipe_flush nothing newer has been packed: the step is
       exported again (step, flush, sync, read is exact); inside the NEXT step the host has already been told
       the buffer was complete - counted (odhip_pipe_export_stale) */
    if (p->in_flush) {
      p->export_pending = -1;        /* (the streams packed before the resolve are not sent) */
      ODHIP_TRY(hipStreamSynchronize(p->export_stream));
      ODHIP_TRY(hipStreamSynchronize(p->stream[0]));
      ODHIP_TRY(hipStreamSynchronize(p->stream[1]));
      ODHIP_TRY(hipMemset(p->export_dev[par], 0, sizeof(odhip_export_header)));
      STEP_TRY(export_luma(p, par));
      /* (the chroma packs, on the side stream, wait for the cleared header) */
      ODHIP_TRY(hipStreamWaitEvent(p->stream[1], p->ev_exp_luma[par], 0));
      STEP_TRY(export_chroma(p, par));
    }
    else p->export_stale++;
  }
  return ODHIP_SUCCESS;
}

int luma_bands(odhip_pipe *p, hipStream_t s, int jpar) {
  const double lam = p->cfg.pvq_norm_lambda;
  /* with pricing the searches make the choice themselves (no separate choice kernel) */
  Timed tm(p, ODHIP_PIPE_BANDS_LUMA, s);
  return p->cfg.price ? odhip_pvq_noref_bands_priced_multi(p->jobs[jpar], p->njobs, lam, s)
   : odhip_pvq_noref_bands_multi(p->jobs[jpar], p->njobs, lam, s);
}

int luma_front(odhip_pipe *p, hipStream_t s, int jpar) {
  STEP_TRY(stage_pad(p, 0, s));
  STEP_TRY(stage_pyramid(p, 0, s));
  if (!p->cfg.chroma_cfl) {
    STEP_TRY(stage_pad(p, 1, s));
    STEP_TRY(stage_pyramid(p, 1, s));
  }
  else {
    /* the chroma chain of step i - 2 has read the choices this band stage overwrites */
    ODHIP_TRY(hipStreamWaitEvent(s, p->ev_used[jpar], 0));
  }
  return luma_bands(p, s, jpar);
}

int luma_choose(odhip_pipe *p, hipStream_t s, int jpar) {
  const double lam = p->cfg.pvq_norm_lambda;
  if (!p->cfg.price) {
    Timed tm(p, ODHIP_PIPE_CHOOSE_LUMA, s);
    return odhip_pvq_choose_multi(p->jobs[jpar], p->njobs, lam, s);
  }
  /* (the choice was made inside the band stage: odhip_pvq_noref_bands_priced_multi) */
  /* The luma choices feed this step's chroma references and inverse: a band whose priced
     decision is too close to take from the device is settled (host libm) before they are
     enqueued.  The host waits here for the luma front of this step while the chroma chain
     of the previous step keeps the GPU busy. */
  const auto t0 = std::chrono::steady_clock::now();
  const int n = odhip_pvq_choose_priced_resolve(p->jobs[jpar], p->njobs, lam, s);
  p->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (n < 0) return n;
  p->price_reruns += n;
  return ODHIP_SUCCESS;
}

int chroma_bands(odhip_pipe *p, int par, hipStream_t s) {
  Timed tm(p, ODHIP_PIPE_BANDS_CHROMA, s);
  /* (the decided stage sends its two counts itself) */
  if (p->cfg.price) return odhip_pvq_ref_bands_decided_multi(p->refjobs[par], 4, p->cfg.pvq_norm_lambda, s);
  STEP_TRY(odhip_pvq_ref_bands_multi(p->refjobs[par], 4, p->cfg.pvq_norm_lambda, s));
  return odhip_pvq_ref_resolve_begin(s);
}

/* ---- inter mode: both plane sets through the with-reference stage against the pyramid of
   their prediction pictures (pvq_theta with is_keyframe = 0, src/encode.c:1326-1360); the two
   chains are independent, each in its own context on its own stream. */
int inter_tail(odhip_pipe *p, int si, hipStream_t s) {
  PlaneSet &t = p->set[si];
  const double lam = p->cfg.pvq_norm_lambda;
  odhip_pvq_refjob *jobs = p->interjobs[si];
  if (!p->cfg.price) {
    Timed tm(p, si ? ODHIP_PIPE_CHOOSE_CHROMA : ODHIP_PIPE_CHOOSE_LUMA, s);
    STEP_TRY(odhip_pvq_ref_choose_multi(jobs, t.nlev, lam, s));
  }
  Timed tm(p, si ? ODHIP_PIPE_INVERSE_CHROMA : ODHIP_PIPE_INVERSE_LUMA, s);
  return odhip_inverse_levels_pvq_ref(t.recon, t.w, (long)t.w*t.h, jobs, t.nlev, t.dec, p->pic_w, p->pic_h, s);
}

/* the counts of the previous step's chain si (theta margin, price margin), one step late */
int inter_finish(odhip_pipe *p, int si) {
  if (!p->inter_pending[si]) return ODHIP_SUCCESS;
  p->inter_pending[si] = false;
  PlaneSet &t = p->set[si];
  hipStream_t s = p->stream[si];
  const double lam = p->cfg.pvq_norm_lambda;
  Current cur(p->ctx[si]);
  const auto t0 = std::chrono::steady_clock::now();
  const int n = odhip_pvq_ref_resolve_finish(p->interjobs[si], t.nlev, lam, s);
  if (n < 0) return n;
  int m = 0;
  if (p->cfg.price) {
    m = odhip_pvq_ref_choose_priced_resolve(p->interjobs[si], t.nlev, lam, s);
    if (m < 0) return m;
  }
  p->reruns += n;
  p->price_reruns += m;
  if (n > 0 || m > 0) STEP_TRY(inter_tail(p, si, s));
  p->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return ODHIP_SUCCESS;
}

int inter_chain(odhip_pipe *p, int si) {
  PlaneSet &t = p->set[si];
  hipStream_t s = p->stream[si];
  const double lam = p->cfg.pvq_norm_lambda;
  Current cur(p->ctx[si]);
  STEP_TRY(stage_pad(p, si, s));
  STEP_TRY(stage_pyramid(p, si, s));
  {
    /* the prediction pictures: same padding, same pyramid */
    Timed tm(p, si ? ODHIP_PIPE_PAD_CHROMA : ODHIP_PIPE_PAD_LUMA, s);
    if (p->cfg.fpr_bits) {
      STEP_TRY(odhip_image_planes_copy_pad16(reinterpret_cast<uint16_t *>(t.pred_px), t.w, (long)t.w*t.h, t.w,
       t.h, t.pred_pic, p->cfg.fpr_bits, t.pw, (long)t.pw*t.ph, t.pw, t.ph, t.nplanes, s));
    }
    else {
      STEP_TRY(odhip_image_planes_copy_pad(t.pred_px, t.w, (long)t.w*t.h, t.w, t.h, t.pred_pic, t.pw,
       (long)t.pw*t.ph, t.pw, t.ph, t.nplanes, s));
    }
  }
  {
    Timed tm(p, si ? ODHIP_PIPE_PYRAMID_CHROMA : ODHIP_PIPE_PYRAMID_LUMA, s);
    STEP_TRY(odhip_forward_pyramid(t.pred_levels, t.pred_px, t.w, (long)t.w*t.h, t.nplanes, t.w, t.h, t.dec,
     p->pic_w, p->pic_h, s));
  }
  {
    Timed tm(p, si ? ODHIP_PIPE_BANDS_CHROMA : ODHIP_PIPE_BANDS_LUMA, s);
    if (p->cfg.price) STEP_TRY(odhip_pvq_ref_bands_decided_multi(p->interjobs[si], t.nlev, lam, s));
    else {
      STEP_TRY(odhip_pvq_ref_bands_multi(p->interjobs[si], t.nlev, lam, s));
      STEP_TRY(odhip_pvq_ref_resolve_begin(s));
    }
  }
  STEP_TRY(inter_tail(p, si, s));
  p->inter_pending[si] = true;
  return ODHIP_SUCCESS;
}

int step_inter(odhip_pipe *p) {
  for (int si = 0; si < 2; si++) {
    STEP_TRY(inter_finish(p, si));
    STEP_TRY(inter_chain(p, si));
  }
  return ODHIP_SUCCESS;
}

int step_noref(odhip_pipe *p) {
  hipStream_t s = p->stream[0];
  Current cur(p->ctx[0]);
  STEP_TRY(luma_front(p, s, 0));
  STEP_TRY(luma_choose(p, s, 0));
  STEP_TRY(stage_inverse_noref(p, 0, s, 0));
  return stage_inverse_noref(p, 1, s, 0);
}

/* ---- the output side of the PCIe-inclusive rate (odhip_pipe_set_export) ------------------
   Sections of the export buffer (include/daala_hip.h, export_kernels.hip): luma levels 0..4, chroma
   levels 0..3.  The luma sections are packed as soon as the luma choices are final, the chroma ones
   behind the chroma band stage; then ONE ship kernel moves the header, the records and the used part
   of every stream to the host.  The band stages may overwrite choices and pulses as soon as the PACK
   kernels have read them (ev_exp_luma / ev_exp_chroma), not only after the transfer. */
int export_layout(const odhip_pipe *p, odhip_export_layout *lay) {
  long nblocks[9];
  int bs[9];
  int with_ref[9];
  for (int i = 0; i < 5; i++) {
    nblocks[i] = p->set[0].nblocks[i];
    bs[i] = i;
    with_ref[i] = 0;
  }
  for (int i = 0; i < 4; i++) {
    nblocks[5 + i] = p->set[1].nblocks[i];
    bs[5 + i] = i;
    with_ref[5 + i] = 1;
  }
  return odhip_export_layout_make(lay, 9, nblocks, bs, with_ref);
}

/* The pack kernels run INSIDE the chains, behind the stage whose outputs they read (luma: main stream, behind the
   choice; chroma: side stream, behind the band stage): on a stream of their own they waited for the searches of
   both chains to leave registers free (profiles/r6_overlap.txt) and held back, through their events, the band
   stages that reuse the buffers they read - 7.2 ms per step instead of 3.6. */
/* ODHIP_EXPORT_DBG (experiments build): bit 0 no pack kernels, bit 1 no copy of the fixed part, bit 2 no
   stream copies - which part of the export a step pays for. */
int export_dbg() {
  static int v = -1;
  if (v < 0) {
    const char *e = ODHIP_EXP_ENV("ODHIP_EXPORT_DBG");
    v = e ? atoi(e) : 0;
  }
  return v;
}

int export_luma(odhip_pipe *p, int par) {
  hipStream_t s = p->stream[0];
  if (!(export_dbg() & 1)) {
    const int32_t *choice[5];
    const int16_t *y[5];
    long nblocks[5];
    int bss[5];
    for (int bs = 0; bs < 5; bs++) {
      choice[bs] = p->jobs[par][bs].cands.choice;
      y[bs] = p->jobs[par][bs].cands.y;
      nblocks[bs] = p->set[0].nblocks[bs];
      bss[bs] = bs;
    }
    STEP_TRY(odhip_export_pack_multi(p->export_dev[par], &p->export_lay, 0, 5, choice, y, nblocks, bss, 0, s));
  }
  ODHIP_TRY(hipEventRecord(p->ev_exp_luma[par], s));
  return ODHIP_SUCCESS;
}

/* The chroma sections, then everything whose size the host knows - header, records, group bases - on its way on
   the export stream (copy engine: no compute unit involved).  The streams follow in export_finish. */
int export_chroma(odhip_pipe *p, int par) {
  hipStream_t s = p->stream[1];
  hipStream_t x = p->export_stream;
  if (!(export_dbg() & 1)) {
    const int32_t *choice[4];
    const int16_t *y[4];
    long nblocks[4];
    int bss[4];
    for (int bs = 0; bs < 4; bs++) {
      choice[bs] = p->refjobs[par][bs].choice;
      y[bs] = p->refjobs[par][bs].y;
      nblocks[bs] = p->set[1].nblocks[bs];
      bss[bs] = bs;
    }
    STEP_TRY(odhip_export_pack_multi(p->export_dev[par], &p->export_lay, 5, 4, choice, y, nblocks, bss, 1, s));
  }
  ODHIP_TRY(hipEventRecord(p->ev_exp_chroma, s));
  /* the totals travel IN the chain (like the band stages' counts): on the export stream even this 128-byte copy
     waited for the searches */
  ODHIP_TRY(hipStreamWaitEvent(s, p->ev_exp_luma[par], 0));
  if (!(export_dbg() & 8)) {
    ODHIP_TRY(hipMemcpyAsync(p->export_hdr[par], p->export_dev[par], sizeof(odhip_export_header), hipMemcpyDeviceToHost, s));
  }
  ODHIP_TRY(hipEventRecord(p->ev_exp_hdr[par], s));
  ODHIP_TRY(hipStreamWaitEvent(x, p->ev_exp_hdr[par], 0));
  if (!(export_dbg() & 2)) {
    ODHIP_TRY(hipMemcpyAsync(p->export_host, p->export_dev[par], (size_t)p->export_lay.fixed_bytes, hipMemcpyDeviceToHost, x));
  }
  p->export_pending = par;
  return ODHIP_SUCCESS;
}

/* The used part of every stream of the step packed last, once its totals have reached the host: called one
   step late (behind finish_pending, when the chroma band stage of that step is known to have ended) and from
   odhip_pipe_sync. */
int export_finish(odhip_pipe *p) {
  if (p->export_pending < 0) return ODHIP_SUCCESS;
  const int par = p->export_pending;
  p->export_pending = -1;
  if (!p->export_host) return ODHIP_SUCCESS;
  if (!(export_dbg() & 16)) ODHIP_TRY(hipEventSynchronize(p->ev_exp_hdr[par]));
  const odhip_export_header *h = p->export_hdr[par];
  for (int s = 0; s < p->export_lay.nsections; s++) {
    const odhip_export_section &sec = p->export_lay.section[s];
    uint32_t words = h->total_words[s];
    if (words > sec.cap_words) words = sec.cap_words;
    const size_t bytes = ((size_t)words*2 + 15) & ~(size_t)15;
    if (bytes && !(export_dbg() & 4)) {
      ODHIP_TRY(hipMemcpyAsync(p->export_host + sec.stream_off, p->export_dev[par] + sec.stream_off, bytes,
       hipMemcpyDeviceToHost, p->export_stream));
    }
  }
  /* export_dev[par] is packed again two steps later: totals and flags cleared for it */
  STEP_TRY(odhip_export_begin(p->export_dev[par], &p->export_lay, p->export_stream));
  ODHIP_TRY(hipEventRecord(p->ev_exp_sent[par], p->export_stream));
  return ODHIP_SUCCESS;
}

int step_cfl(odhip_pipe *p) {
  hipStream_t main = p->stream[0];
  hipStream_t side = p->stream[1];
  const int par = (int)(p->nstep & 1);
  const bool exporting = p->export_host != nullptr;
  {
    Current cur(p->ctx[0]);
    STEP_TRY(luma_front(p, main, par));
    STEP_TRY(luma_choose(p, main, par));
    /* this parity's export buffer: its last contents (step i - 2) have left and its header has been cleared behind
       them, on the export stream (export_finish) - not here: even a 128-byte memset in this chain waits for the
       other chain's searches to leave registers free (profiles/r6_overlap.txt) */
    if (exporting) ODHIP_TRY(hipStreamWaitEvent(main, p->ev_exp_sent[par], 0));
    /* the luma choices of this step are final: the chroma chain takes its references from
       them (odhip_pvq_refjob.luma) */
    ODHIP_TRY(hipEventRecord(p->ev_refs[par], main));
    if (exporting) STEP_TRY(export_luma(p, par));
    STEP_TRY(stage_inverse_noref(p, 0, main, par));
  }
  STEP_TRY(finish_pending(p));
  /* (the chroma band stage of the previous step has ended: its export is packed or about to be) */
  if (exporting) STEP_TRY(export_finish(p));
  {
    Current cur(p->ctx[1]);
    STEP_TRY(stage_pad(p, 1, side));
    STEP_TRY(stage_pyramid(p, 1, side));
    ODHIP_TRY(hipStreamWaitEvent(side, p->ev_refs[par], 0));
    STEP_TRY(chroma_bands(p, par, side));
    /* only the preparation kernels of the band stage read the luma choices */
    ODHIP_TRY(hipEventRecord(p->ev_used[par], side));
    if (exporting) STEP_TRY(export_chroma(p, par));
    STEP_TRY(chroma_tail(p, par, side));
  }
  p->pending = par;
  return ODHIP_SUCCESS;
}

}  // namespace

extern "C" odhip_pipe *odhip_pipe_create(const odhip_pipe_config *cfg) {
  if (!cfg || !cfg->quant || cfg->frames <= 0 || cfg->pic_w <= 0 || cfg->pic_h <= 0
   || (cfg->pic_w & 1) || (cfg->pic_h & 1)
   || (cfg->fpr_bits != 0 && cfg->fpr_bits != 8 && cfg->fpr_bits != 10 && cfg->fpr_bits != 12)) {
    return nullptr;
  }
  odhip_pipe *p = new odhip_pipe();
  p->cfg = *cfg;
  p->ctx[0] = p->ctx[1] = nullptr;
  p->stream[0] = p->stream[1] = nullptr;
  p->njobs = 0;
  p->nstep = 0;
  p->reruns = 0;
  p->price_reruns = 0;
  p->wait_ms = 0;
  p->k_range = 0;
  p->record = false;
  memset(p->rate, 0, sizeof(p->rate));
  memset(p->ev_refs, 0, sizeof(p->ev_refs));
  memset(p->ev_used, 0, sizeof(p->ev_used));
  p->copy_stream = nullptr;
  p->ev_fed = nullptr;
  p->ev_pad[0] = p->ev_pad[1] = nullptr;
  p->front = 0;
  p->fed = false;
  p->export_stream = nullptr;
  p->export_host = nullptr;
  p->export_dev[0] = p->export_dev[1] = nullptr;
  p->export_hdr[0] = p->export_hdr[1] = nullptr;
  p->ev_exp_hdr[0] = p->ev_exp_hdr[1] = nullptr;
  p->ev_exp_sent[0] = p->ev_exp_sent[1] = nullptr;
  p->export_pending = -1;
  p->export_stale = 0;
  p->in_flush = false;
  p->ev_exp_luma[0] = p->ev_exp_luma[1] = p->ev_exp_chroma = p->ev_chroma_done = nullptr;
  if (pipe_init(p) != ODHIP_SUCCESS) {
    odhip_pipe_destroy(p);
    return nullptr;
  }
  /* the quantiser tables were copied to the device; do not keep the caller's pointer */
  p->cfg.quant = nullptr;
  return p;
}

extern "C" void odhip_pipe_destroy(odhip_pipe *p) {
  if (!p) return;
  (void)hipSetDevice(p->cfg.device);
  (void)hipDeviceSynchronize();
  for (int i = 0; i < 2; i++) {
    if (p->ctx[i]) odhip_destroy(p->ctx[i]);
    if (p->ev_refs[i]) (void)hipEventDestroy(p->ev_refs[i]);
    if (p->ev_used[i]) (void)hipEventDestroy(p->ev_used[i]);
    if (p->ev_pad[i]) (void)hipEventDestroy(p->ev_pad[i]);
  }
  if (p->ev_fed) (void)hipEventDestroy(p->ev_fed);
  if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
  for (hipEvent_t e : {p->ev_exp_luma[0], p->ev_exp_luma[1], p->ev_exp_chroma, p->ev_chroma_done}) {
    if (e) (void)hipEventDestroy(e);
  }
  for (int i = 0; i < 2; i++) {
    if (p->ev_exp_hdr[i]) (void)hipEventDestroy(p->ev_exp_hdr[i]);
    if (p->ev_exp_sent[i]) (void)hipEventDestroy(p->ev_exp_sent[i]);
    if (p->export_hdr[i]) (void)hipHostFree(p->export_hdr[i]);
  }
  if (p->export_stream) (void)hipStreamDestroy(p->export_stream);
  if (p->stream[1] && p->stream[1] != p->stream[0]) (void)hipStreamDestroy(p->stream[1]);
  if (p->stream[0]) (void)hipStreamDestroy(p->stream[0]);
  for (int i = 0; i < kStages; i++) {
    for (hipEvent_t e : p->timed[i]) (void)hipEventDestroy(e);
  }
  for (void *d : p->owned) (void)hipFree(d);
  delete p;
}

extern "C" int odhip_pipe_set_pictures(odhip_pipe *p, const uint8_t *luma, const uint8_t *chroma,
 int on_device) {
  if (!p || !luma || !chroma) return ODHIP_EINVAL;
  /* a step still in flight may be reading the pictures */
  const int rc = odhip_pipe_sync(p);
  if (rc) return rc;
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const PlaneSet &l = p->set[0];
  const PlaneSet &c = p->set[1];
  const size_t pic_bytes = p->cfg.fpr_bits > 8 ? 2 : 1;
  ODHIP_TRY(hipMemcpyAsync(l.pic, luma, (size_t)l.nplanes*l.pw*l.ph*pic_bytes, kind, p->stream[0]));
  ODHIP_TRY(hipMemcpyAsync(c.pic, chroma, (size_t)c.nplanes*c.pw*c.ph*pic_bytes, kind, p->stream[0]));
  ODHIP_TRY(hipStreamSynchronize(p->stream[0]));
  /* a feed that no step has taken yet is dropped: these are the pictures of the next step */
  ODHIP_TRY(hipStreamSynchronize(p->copy_stream));
  p->fed = false;
  return ODHIP_SUCCESS;
}

/* The pictures of the NEXT step, from host memory, while the steps already enqueued
   compute: copied on the pipe's own copy stream into the back buffers (after the padding
   kernels that may still read them), taken by the next odhip_pipe_step.  Pinned host
   memory (hipHostMalloc / hipHostRegister) makes the copy asynchronous; the buffers must
   stay valid until that step has been enqueued AND the copy has completed
   (odhip_pipe_sync waits for it too). */
extern "C" int odhip_pipe_feed(odhip_pipe *p, const uint8_t *luma, const uint8_t *chroma) {
  if (!p || !luma || !chroma) return ODHIP_EINVAL;
  ODHIP_TRY(hipSetDevice(p->cfg.device));
  const int back = p->front ^ 1;
  const PlaneSet &l = p->set[0];
  const PlaneSet &c = p->set[1];
  const size_t pic_bytes = p->cfg.fpr_bits > 8 ? 2 : 1;
  ODHIP_TRY(hipStreamWaitEvent(p->copy_stream, p->ev_pad[0], 0));
  ODHIP_TRY(hipStreamWaitEvent(p->copy_stream, p->ev_pad[1], 0));
  ODHIP_TRY(hipMemcpyAsync(l.pic_buf[back], luma, (size_t)l.nplanes*l.pw*l.ph*pic_bytes, hipMemcpyHostToDevice,
   p->copy_stream));
  ODHIP_TRY(hipMemcpyAsync(c.pic_buf[back], chroma, (size_t)c.nplanes*c.pw*c.ph*pic_bytes, hipMemcpyHostToDevice,
   p->copy_stream));
  ODHIP_TRY(hipEventRecord(p->ev_fed, p->copy_stream));
  p->fed = true;
  return ODHIP_SUCCESS;
}

/* Size of the export buffer (odhip_pipe_set_export), 0 for the modes that do not export. */
extern "C" size_t odhip_pipe_export_bytes(const odhip_pipe *p) {
  if (!p || !p->cfg.chroma_cfl || p->cfg.inter || !p->cfg.price) return 0;
  odhip_export_layout lay;
  if (export_layout(p, &lay) != ODHIP_SUCCESS) return 0;
  return (size_t)lay.total_bytes;
}

extern "C" int odhip_pipe_export_layout(const odhip_pipe *p, odhip_export_layout *out) {
  if (!p || !out) return ODHIP_EINVAL;
  if (odhip_pipe_export_bytes(p) == 0) return ODHIP_EIMPL;
  return export_layout(p, out);
}

extern "C" long odhip_pipe_export_stale(const odhip_pipe *p) {
  return p ? p->export_stale : 0;
}

/* host != NULL: every following step leaves its decisions - record and pulses of every band, compacted on the
   device (export_kernels.hip) - in `host`, odhip_pipe_export_bytes(p) bytes of pinned host memory, on the
   pipe's export stream, overlapped with the rest of the step; the buffer holds step i once odhip_pipe_sync()
   returns after step i.  NULL: stop exporting.  Keyframe steps with chroma from luma and pricing on the
   device only (ODHIP_EIMPL otherwise). */
extern "C" int odhip_pipe_set_export(odhip_pipe *p, void *host) {
  if (!p) return ODHIP_EINVAL;
  if (host && odhip_pipe_export_bytes(p) == 0) return ODHIP_EIMPL;
  const int rc = odhip_pipe_sync(p);
  if (rc) return rc;
  ODHIP_TRY(hipSetDevice(p->cfg.device));
  if (host) {
    /* each object on its own: a call that failed half way is completed by the next one */
    if (!p->export_stream) ODHIP_TRY(hipStreamCreateWithFlags(&p->export_stream, hipStreamNonBlocking));
    for (hipEvent_t *e : {&p->ev_exp_luma[0], &p->ev_exp_luma[1], &p->ev_exp_chroma, &p->ev_chroma_done}) {
      if (!*e) ODHIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
  }
  if (host && !p->export_dev[0]) {
    STEP_TRY(export_layout(p, &p->export_lay));
    for (int i = 0; i < 2; i++) {
      PIPE_ALLOC(p, p->export_dev[i], (size_t)p->export_lay.total_bytes, false);
      ODHIP_TRY(hipMemset(p->export_dev[i], 0, (size_t)p->export_lay.fixed_bytes));
      ODHIP_TRY(hipHostMalloc((void **)&p->export_hdr[i], sizeof(odhip_export_header), hipHostMallocDefault));
      ODHIP_TRY(hipEventCreateWithFlags(&p->ev_exp_hdr[i], hipEventDisableTiming));
      ODHIP_TRY(hipEventCreateWithFlags(&p->ev_exp_sent[i], hipEventDisableTiming));
    }
  }
  p->export_pending = -1;
  if (host) {
    /* (the pipe is idle: odhip_pipe_sync above) */
    for (int i = 0; i < 2; i++) ODHIP_TRY(hipMemset(p->export_dev[i], 0, sizeof(odhip_export_header)));
  }
  p->export_host = static_cast<uint8_t *>(host);
  return ODHIP_SUCCESS;
}

extern "C" int odhip_pipe_step(odhip_pipe *p) {
  if (!p) return ODHIP_EINVAL;
  ODHIP_TRY(hipSetDevice(p->cfg.device));
  if (p->fed) {
    /* odhip_pipe_feed: this step codes the fed pictures */
    p->front ^= 1;
    p->set[0].pic = p->set[0].pic_buf[p->front];
    p->set[1].pic = p->set[1].pic_buf[p->front];
    ODHIP_TRY(hipStreamWaitEvent(p->stream[0], p->ev_fed, 0));
    if (p->stream[1] != p->stream[0]) ODHIP_TRY(hipStreamWaitEvent(p->stream[1], p->ev_fed, 0));
    p->fed = false;
  }
  const int rc = p->cfg.inter ? step_inter(p) : p->cfg.chroma_cfl ? step_cfl(p) : step_noref(p);
  p->nstep++;
  return rc;
}

extern "C" int odhip_pipe_flush(odhip_pipe *p) {
  if (!p) return ODHIP_EINVAL;
  if (p->cfg.inter) {
    STEP_TRY(inter_finish(p, 0));
    return inter_finish(p, 1);
  }
  p->in_flush = true;
  const int rc = finish_pending(p);
  p->in_flush = false;
  return rc;
}

/* Inter mode: the prediction pictures (what motion compensation produced for each picture
   of the batch), same layouts and depth as odhip_pipe_set_pictures. */
extern "C" int odhip_pipe_set_reference_pictures(odhip_pipe *p, const uint8_t *luma, const uint8_t *chroma,
 int on_device) {
  if (!p || !luma || !chroma || !p->cfg.inter) return ODHIP_EINVAL;
  const int rc = odhip_pipe_sync(p);
  if (rc) return rc;
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const PlaneSet &l = p->set[0];
  const PlaneSet &c = p->set[1];
  const size_t pic_bytes = p->cfg.fpr_bits > 8 ? 2 : 1;
  ODHIP_TRY(hipMemcpyAsync(l.pred_pic, luma, (size_t)l.nplanes*l.pw*l.ph*pic_bytes, kind, p->stream[0]));
  ODHIP_TRY(hipMemcpyAsync(c.pred_pic, chroma, (size_t)c.nplanes*c.pw*c.ph*pic_bytes, kind, p->stream[0]));
  ODHIP_TRY(hipStreamSynchronize(p->stream[0]));
  return ODHIP_SUCCESS;
}

extern "C" int odhip_pipe_sync(odhip_pipe *p) {
  if (!p) return ODHIP_EINVAL;
  ODHIP_TRY(hipStreamSynchronize(p->stream[0]));
  if (p->stream[1] != p->stream[0]) ODHIP_TRY(hipStreamSynchronize(p->stream[1]));
  ODHIP_TRY(hipStreamSynchronize(p->copy_stream));
  if (p->export_stream) {
    STEP_TRY(export_finish(p));
    ODHIP_TRY(hipStreamSynchronize(p->export_stream));
  }
  /* bands whose reference candidate has more pulses than the pulse vectors hold (include/daala_hip.h,
     ODHIP_PVQ_MAX_K): their results are not the reference's - say so */
  unsigned kr[2] = {0, 0};
  for (int i = 0; i < 2; i++) {
    /* the counters belong to the contexts of the two chains */
    if (i == 1 && p->ctx[1] == p->ctx[0]) break;
    Current cur(p->ctx[i]);
    unsigned a = 0;
    unsigned b = 0;
    const int rc = odhip_pvq_k_range_take(&a, &b);
    if (rc != ODHIP_SUCCESS && rc != ODHIP_ERANGE) return rc;
    kr[0] += a;
    kr[1] += b;
  }
  p->k_range += (long)kr[0] + kr[1];
  if (kr[0] || kr[1]) {
    fprintf(stderr, "libdaalahip: %u + %u band(s) need more than %d pulses (quantiser too fine for the int16 pulse "
     "vectors): the steps since the last odhip_pipe_sync are not the reference's\n", kr[0], kr[1], ODHIP_PVQ_MAX_K);
    return ODHIP_ERANGE;
  }
  return ODHIP_SUCCESS;
}

/* Bands counted by odhip_pvq_k_range_take at this pipe's syncs so far. */
extern "C" long odhip_pipe_k_range(const odhip_pipe *p) {
  return p ? p->k_range : 0;
}

/* The stages one at a time, in order on the luma stream (tests, the priced
   verification flow of bench.py: the host prices candidates between the band
   stage and the choice).  parity selects the reference buffer. */
extern "C" int odhip_pipe_stage(odhip_pipe *p, int stage, int parity) {
  if (!p || stage < 0 || stage >= kStages || (parity != 0 && parity != 1)) return ODHIP_EINVAL;
  if (p->cfg.inter) return ODHIP_EINVAL;       /* inter mode runs whole steps only */
  ODHIP_TRY(hipSetDevice(p->cfg.device));
  hipStream_t s = p->stream[0];
  const bool cfl = p->cfg.chroma_cfl != 0;
  const double lam = p->cfg.pvq_norm_lambda;
  const bool chroma_stage = stage == ODHIP_PIPE_PAD_CHROMA || stage == ODHIP_PIPE_PYRAMID_CHROMA
   || stage == ODHIP_PIPE_BANDS_CHROMA || stage == ODHIP_PIPE_CHOOSE_CHROMA
   || stage == ODHIP_PIPE_INVERSE_CHROMA;
  Current cur(p->ctx[chroma_stage && cfl ? 1 : 0]);
  const int jpar = cfl ? parity : 0;     /* the luma set of that parity */
  (void)lam;
  switch (stage) {
    case ODHIP_PIPE_PAD_LUMA: return stage_pad(p, 0, s);
    case ODHIP_PIPE_PYRAMID_LUMA: return stage_pyramid(p, 0, s);
    case ODHIP_PIPE_PAD_CHROMA: return stage_pad(p, 1, s);
    case ODHIP_PIPE_PYRAMID_CHROMA: return stage_pyramid(p, 1, s);
    case ODHIP_PIPE_BANDS_LUMA: return luma_bands(p, s, jpar);
    case ODHIP_PIPE_CHOOSE_LUMA: return luma_choose(p, s, jpar);
    /* (the references are read in place from the luma choices: nothing to run) */
    case ODHIP_PIPE_CFL_REFS: return cfl ? ODHIP_SUCCESS : ODHIP_EINVAL;
    case ODHIP_PIPE_INVERSE_LUMA: return stage_inverse_noref(p, 0, s, jpar);
    case ODHIP_PIPE_BANDS_CHROMA: {
      if (!cfl) return ODHIP_SUCCESS;      /* part of ODHIP_PIPE_BANDS_LUMA */
      STEP_TRY(chroma_bands(p, parity, s));
      const int n = odhip_pvq_ref_resolve_finish(p->refjobs[parity], 4, lam, s);
      if (n < 0) return n;
      p->reruns += n;
      return ODHIP_SUCCESS;
    }
    case ODHIP_PIPE_CHOOSE_CHROMA: {
      if (!cfl) return ODHIP_SUCCESS;
      if (p->cfg.price) {
        /* the band stage decided; what is left is the host-libm resolve of listed bands */
        const int m = odhip_pvq_ref_choose_priced_resolve(p->refjobs[parity], 4, lam, s);
        if (m < 0) return m;
        p->price_reruns += m;
        return ODHIP_SUCCESS;
      }
      Timed tm(p, stage, s);
      return odhip_pvq_ref_choose_multi(p->refjobs[parity], 4, lam, s);
    }
    case ODHIP_PIPE_INVERSE_CHROMA: {
      if (!cfl) return stage_inverse_noref(p, 1, s, 0);
      PlaneSet &ch = p->set[1];
      Timed tm(p, stage, s);
      return odhip_inverse_levels_pvq_ref(ch.recon, ch.w, (long)ch.w*ch.h, p->refjobs[parity], 4, 1,
       p->pic_w, p->pic_h, s);
    }
    default: return ODHIP_EINVAL;
  }
}

extern "C" int odhip_pipe_buffer(odhip_pipe *p, int what, int set, int level, int parity, void **d_ptr,
 size_t *bytes) {
  if (!p || !d_ptr || !bytes || (set != 0 && set != 1) || parity < -1 || parity > 1) {
    return ODHIP_EINVAL;
  }
  /* -1: the buffers of the LAST odhip_pipe_step (the luma choices and the chroma references
     alternate between two sets from step to step) */
  if (parity < 0) parity = p->nstep > 0 ? (int)((p->nstep - 1) & 1) : 0;
  PlaneSet &t = p->set[set];
  if (what != ODHIP_PIPE_BUF_PIC && what != ODHIP_PIPE_BUF_PX && (level < 0 || level >= t.nlev)) {
    return ODHIP_EINVAL;
  }
  int nb = 0;
  int len = 0;
  if (level >= 0 && level < ODHIP_NBSIZES) odhip_pvq_band_layout(level, &nb, nullptr, &len);
  const long B = level >= 0 && level < ODHIP_NBSIZES ? t.nblocks[level] : 0;
  const bool inter = p->cfg.inter != 0;
  const bool ref = inter || (set == 1 && p->cfg.chroma_cfl);
  const int jpar = p->cfg.chroma_cfl && !inter ? parity : 0;
  const odhip_pvq_job *j = ref ? nullptr : set == 0 ? &p->jobs[jpar][level] : &p->jobs[0][5 + level];
  const odhip_pvq_refjob *r = inter ? &p->interjobs[set][level] : ref ? &p->refjobs[parity][level] : nullptr;
  void *ptr = nullptr;
  size_t n = 0;
  switch (what) {
    case ODHIP_PIPE_BUF_PIC: ptr = t.pic; n = (size_t)t.nplanes*t.pw*t.ph*(p->cfg.fpr_bits > 8 ? 2 : 1); break;
    case ODHIP_PIPE_BUF_PX: ptr = t.px; n = (size_t)t.nplanes*t.w*t.h*(p->cfg.fpr_bits ? 2 : 1); break;
    case ODHIP_PIPE_BUF_LEVEL: ptr = t.levels[level]; n = sizeof(od_coeff)*(size_t)t.nplanes*t.w*t.h; break;
    case ODHIP_PIPE_BUF_RECON: ptr = t.recon[level]; n = (size_t)t.nplanes*t.w*t.h*(p->cfg.fpr_bits ? 2 : 1); break;
    case ODHIP_PIPE_BUF_BAND:
      ptr = j ? (void *)j->cands.band : (void *)r->band;
      n = (size_t)64*B*nb;
      break;
    case ODHIP_PIPE_BUF_Y:
      ptr = j ? j->cands.y : r->y;
      n = sizeof(int16_t)*(size_t)(j ? 2 : ODHIP_PVQ_REF_SLOTS)*B*len;
      break;
    case ODHIP_PIPE_BUF_CHOICE:
      ptr = j ? j->cands.choice : r->choice;
      n = sizeof(int32_t)*(size_t)B*nb*(j ? 4 : 16);
      break;
    case ODHIP_PIPE_BUF_ITEMS:
      if (!r) return ODHIP_EINVAL;
      ptr = r->items;
      n = (size_t)3*nb*ODHIP_PVQ_REF_SLOTS*B*16;
      break;
    case ODHIP_PIPE_BUF_REF:
      if (!r) return ODHIP_EINVAL;
      /* inter mode: the reference of every block is the pyramid of its prediction picture;
         keyframe chroma takes its reference from the luma choices in place: no plane exists */
      if (!inter) return ODHIP_EINVAL;
      ptr = t.pred_levels[level];
      n = sizeof(od_coeff)*(size_t)t.nplanes*t.w*t.h;
      break;
    case ODHIP_PIPE_BUF_RATE: {
      /* allocated (zero: every candidate free = choice on distortion alone) and
         attached on first request: [B][nb][2] without reference,
         [B][nb][ODHIP_PVQ_REF_SLOTS + 1] with */
      n = sizeof(double)*(size_t)B*nb*(j ? 2 : ODHIP_PVQ_REF_SLOTS + 1);
      if (!p->rate[set][level]) {
        ODHIP_TRY(hipSetDevice(p->cfg.device));
        PIPE_ALLOC(p, p->rate[set][level], n, true);
        if (j) {
          p->jobs[0][set ? 5 + level : level].d_rate = p->rate[set][level];
          p->jobs[1][set ? 5 + level : level].d_rate = p->rate[set][level];
        }
        else if (inter) p->interjobs[set][level].d_rate = p->rate[set][level];
        else {
          p->refjobs[0][level].d_rate = p->rate[set][level];
          p->refjobs[1][level].d_rate = p->rate[set][level];
        }
      }
      ptr = p->rate[set][level];
      break;
    }
    default: return ODHIP_EINVAL;
  }
  *d_ptr = ptr;
  *bytes = n;
  return ODHIP_SUCCESS;
}

extern "C" int odhip_pipe_read(odhip_pipe *p, void *host, const void *d_ptr, size_t bytes) {
  if (!p || !host || !d_ptr) return ODHIP_EINVAL;
  const int rc = odhip_pipe_sync(p);
  if (rc) return rc;
  ODHIP_TRY(hipMemcpy(host, d_ptr, bytes, hipMemcpyDeviceToHost));
  return ODHIP_SUCCESS;
}

extern "C" int odhip_pipe_write(odhip_pipe *p, void *d_ptr, const void *host, size_t bytes) {
  if (!p || !host || !d_ptr) return ODHIP_EINVAL;
  const int rc = odhip_pipe_sync(p);
  if (rc) return rc;
  ODHIP_TRY(hipMemcpy(d_ptr, host, bytes, hipMemcpyHostToDevice));
  return ODHIP_SUCCESS;
}

extern "C" int odhip_pipe_record(odhip_pipe *p, int enable) {
  if (!p) return ODHIP_EINVAL;
  p->record = enable != 0;
  for (int i = 0; i < kStages; i++) {
    for (hipEvent_t e : p->timed[i]) (void)hipEventDestroy(e);
    p->timed[i].clear();
  }
  /* the dominant kernels of the two band stages, on the streams they run on */
  {
    Current cur(p->ctx[0]);
    const int rc = odhip_pvq_profile(enable);
    if (rc) return rc;
  }
  if (p->cfg.chroma_cfl) {
    Current cur(p->ctx[1]);
    const int rc = odhip_pvq_ref_profile(enable);
    if (rc) return rc;
  }
  return ODHIP_SUCCESS;
}

extern "C" int odhip_pipe_timings(odhip_pipe *p, double avg_ms[ODHIP_PIPE_NSTAGES],
 int count[ODHIP_PIPE_NSTAGES]) {
  if (!p || !avg_ms || !count) return ODHIP_EINVAL;
  const int rc = odhip_pipe_sync(p);
  if (rc) return rc;
  for (int i = 0; i < kStages; i++) {
    double sum = 0;
    int n = 0;
    for (size_t k = 0; k + 1 < p->timed[i].size(); k += 2) {
      float ms = 0;
      ODHIP_TRY(hipEventElapsedTime(&ms, p->timed[i][k], p->timed[i][k + 1]));
      sum += ms;
      n++;
    }
    avg_ms[i] = n ? sum/n : 0;
    count[i] = n;
  }
  return ODHIP_SUCCESS;
}

extern "C" int odhip_pipe_search_timings(odhip_pipe *p, int chroma, float *ms, int max_n) {
  if (!p || !ms) return ODHIP_EINVAL;
  if (chroma && !p->cfg.chroma_cfl) return 0;
  Current cur(p->ctx[chroma ? 1 : 0]);
  return chroma ? odhip_pvq_ref_profile_read(ms, max_n) : odhip_pvq_profile_read(ms, max_n);
}

/* The luma forward pyramid launched n times on an otherwise idle GPU: average
   milliseconds per launch (the filter + DCT stage of the north star, timed alone). */
extern "C" int odhip_pipe_time_pyramid(odhip_pipe *p, int n, double *avg_ms) {
  if (!p || n <= 0 || !avg_ms) return ODHIP_EINVAL;
  int rc = odhip_pipe_sync(p);
  if (rc) return rc;
  hipStream_t s = p->stream[0];
  PlaneSet &t = p->set[0];
  hipEvent_t a = nullptr;
  hipEvent_t b = nullptr;
  ODHIP_TRY(hipEventCreate(&a));
  ODHIP_TRY(hipEventCreate(&b));
  rc = odhip_forward_pyramid(t.levels, t.px, t.w, (long)t.w*t.h, t.nplanes, t.w, t.h, 0, p->pic_w,
   p->pic_h, s);
  (void)hipEventRecord(a, s);
  for (int i = 0; i < n && !rc; i++) {
    rc = odhip_forward_pyramid(t.levels, t.px, t.w, (long)t.w*t.h, t.nplanes, t.w, t.h, 0, p->pic_w,
     p->pic_h, s);
  }
  (void)hipEventRecord(b, s);
  float ms = 0;
  if (!rc && (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess)) {
    rc = ODHIP_EFAULT;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *avg_ms = ms/n;
  return rc;
}

/* One filter + DCT stage (padding, forward pyramid or dequantise + inverse of one plane set)
   launched n times on an otherwise idle GPU, on the buffers the last step left: average
   milliseconds per launch group.  parity as for odhip_pipe_stage (-1: the last step's). */
extern "C" int odhip_pipe_time_stage(odhip_pipe *p, int stage, int parity, int n, double *avg_ms) {
  if (!p || n <= 0 || !avg_ms || parity < -1 || parity > 1) return ODHIP_EINVAL;
  if (stage != ODHIP_PIPE_PAD_LUMA && stage != ODHIP_PIPE_PAD_CHROMA && stage != ODHIP_PIPE_PYRAMID_LUMA
   && stage != ODHIP_PIPE_PYRAMID_CHROMA && stage != ODHIP_PIPE_INVERSE_LUMA && stage != ODHIP_PIPE_INVERSE_CHROMA) {
    return ODHIP_EINVAL;     /* the other stages are not idempotent on their own buffers */
  }
  if (parity < 0) parity = p->nstep > 0 ? (int)((p->nstep - 1) & 1) : 0;
  int rc = odhip_pipe_flush(p);
  if (rc) return rc;
  rc = odhip_pipe_sync(p);
  if (rc) return rc;
  hipStream_t s = p->stream[0];
  hipEvent_t a = nullptr;
  hipEvent_t b = nullptr;
  /* the events first: a failure here must neither leak one nor leave recording switched off */
  ODHIP_TRY(hipEventCreate(&a));
  if (hipEventCreate(&b) != hipSuccess) {
    (void)hipEventDestroy(a);
    return ODHIP_EFAULT;
  }
  const bool rec = p->record;
  p->record = false;
  rc = odhip_pipe_stage(p, stage, parity);
  (void)hipEventRecord(a, s);
  for (int i = 0; i < n && !rc; i++) rc = odhip_pipe_stage(p, stage, parity);
  (void)hipEventRecord(b, s);
  float ms = 0;
  if (!rc && (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess)) {
    rc = ODHIP_EFAULT;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  p->record = rec;
  *avg_ms = ms/n;
  return rc;
}

/* Host milliseconds spent so far waiting for the margin counts (the only host waits
   of a step; everything else in odhip_pipe_step is launch work). */
extern "C" double odhip_pipe_host_wait_ms(const odhip_pipe *p) {
  return p ? p->wait_ms : 0;
}

extern "C" int odhip_pipe_set_test_hooks(odhip_pipe *p, double theta_margin, int theta_perturb,
 double price_tol_scale) {
  if (!p) return ODHIP_EINVAL;
  for (int i = 0; i < 2; i++) {
    const int rc = odhip_ctx_set_test_hooks(p->ctx[i], theta_margin, theta_perturb, price_tol_scale);
    if (rc) return rc;
  }
  return ODHIP_SUCCESS;
}

extern "C" long odhip_pipe_price_reruns(const odhip_pipe *p) {
  return p ? p->price_reruns : 0;
}

extern "C" long odhip_pipe_theta_reruns(const odhip_pipe *p) {
  return p ? p->reruns : 0;
}

/* Bands whose theta lay inside the margin of the device acos and were recomputed with the host's libm so far
   (odhip_pipe_theta_reruns counts the ones whose theta changed). */
extern "C" long odhip_pipe_theta_listed(odhip_pipe *p) {
  if (!p) return 0;
  long n = 0;
  for (int i = 0; i < 2; i++) {
    if (i == 0 && !p->cfg.inter) continue;      /* keyframes: only the chroma chain runs the with-reference stage */
    Current cur(p->ctx[i]);
    const long v = odhip_pvq_ref_theta_listed();
    if (v > 0) n += v;
  }
  return n;
}
