/* od_krange.cuh - bands the device could not code the way the reference does because a candidate the reference
   SEARCHES has more pulses than ODHIP_PVQ_MAX_K (32767: the pulse vectors are int16).  The reference's own K is an
   int (od_pvq_compute_k, src/pvq.c:436-470) and reaches such counts only for the finest quantisers on saturated
   content in 64x64 blocks (first seen by tests/soak/parity_soak.py at a coded quantiser of 8, below encoder_example's
   range).  Such a candidate is never searched or chosen here; every occurrence is COUNTED on the device so that the
   caller is told (odhip_pvq_k_range_take, odhip_pipe_sync -> ODHIP_ERANGE) instead of receiving another band than
   the reference's.  Host side, one counter per translation unit. */
#pragma once
int od_k_range_take_noref(unsigned *count);     /* pvq_bands.hip: reads and clears */
int od_k_range_take_ref(unsigned *count);       /* pvq_refbands.hip */
