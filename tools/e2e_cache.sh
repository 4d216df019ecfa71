# End-to-end: unmodified reference encoder on 1080p all-intra frames, plain C
# vs. fdct_2d served from the GPU frame cache (filters, idct and PVQ search left
# in C).  1 and 3 frames, so that (t3 - t1)/2 is the steady per-frame time.
for nf in 1 3; do
  export NFRAMES=$nf
  python tests/interpose/run_interposed.py 0 1920 1080 > /tmp/e2e_plain.json
  echo "plain C nframes=$nf rc=$?"; cut -c1-300 /tmp/e2e_plain.json
  ODHIP_CACHE_FDCT_ONLY=1 ODHIP_INTERPOSE_PASSTHROUGH=1 python tests/interpose/run_interposed.py 2 1920 1080 > /tmp/e2e_cache.json
  echo "frame cache nframes=$nf rc=$?"; cut -c1-300 /tmp/e2e_cache.json
done
