/* od_occupancy.cuh - how many wavefronts of a deciding search kernel a SIMD holds (round 6 experiment,
   profiles/r6_overlap.txt).

   The step runs two chains on two streams.  Workgroups of two kernels share a compute unit only when the
   second one's FIT next to the first one's (tools/ubench/coresidency.hip, profiles/r6_coresidency_ubench.txt):
   the dispatcher hands out a workgroup as soon as some CU has the wave slots, VGPRs and LDS for it, and a
   kernel whose workgroups fit nowhere waits until the other kernel's drain.  The searches are compiled for
   three or four wavefronts per SIMD at 168 / 128 VGPRs - 504 to 512 of the 512 registers of every SIMD lane
   (rocprofv3's VGPR column counts register PAIRS on gfx950: its 84 is 168) - so nothing of the other stream,
   not even a 16-register memset, starts before a search ends (profiles/r6_timeline_before.csv).

   ODHIP_SEARCH_OCC == 2 compiles every deciding search for TWO wavefronts per SIMD at 176 - 200 VGPRs
   (amdgpu_num_vgpr counts pairs too; OD_SEARCH_VGPR_FLOOR pins the lower end: 170 registers or fewer would
   admit a third wavefront), which leaves 112 registers per lane, six wave slots per SIMD and >= 64 KB of LDS:
   one workgroup of k_forward_pyramid64x2 (107 VGPRs, 53 KB) or several of the smaller filter + DCT kernels
   then DO run beside a search (the luma pyramid's wall time inside the step falls from 1.0 to 0.2 - 0.5 ms)
   - and the step gets SLOWER, 3.57 -> 4.00 ms: the searches lose 14 - 17 % at two wavefronts, and the
   "HBM-bound" kernels they now share SIMDs with are no free riders (they issue 0.48 G of the step's 1.7 G
   VALU wave-instructions).  The default therefore stays what rounds 3-5 shipped (0); 2 and 3 are kept for
   the A/B (python -m daala_amd.build --variant occ2 -DODHIP_SEARCH_OCC=2; tools/gpu_r6_variants.sh). */
#pragma once
#ifndef ODHIP_SEARCH_OCC
# define ODHIP_SEARCH_OCC 0
#endif
#if ODHIP_SEARCH_OCC == 2
# define OD_SEARCH_OCC_ATTR __attribute__((amdgpu_waves_per_eu(2), amdgpu_num_vgpr(100)))    /* (the attribute counts register PAIRS on gfx90a and later: 200) */
# define OD_SEARCH_VGPR_FLOOR() asm volatile("" ::: "v175")
#elif ODHIP_SEARCH_OCC == 3
/* three wavefronts of 129 - 136 registers: 104 left - every filter + DCT kernel but the luma pyramid (107) */
# define OD_SEARCH_OCC_ATTR __attribute__((amdgpu_waves_per_eu(3), amdgpu_num_vgpr(68)))
# define OD_SEARCH_VGPR_FLOOR() asm volatile("" ::: "v128")
#else      /* rounds 3-5: the with-reference searches at three wavefronts, the no-reference ones as they come (four) */
# define OD_SEARCH_OCC_ATTR __attribute__((amdgpu_waves_per_eu(3)))
# define OD_DECIDE_OCC_ATTR
# define OD_SEARCH_VGPR_FLOOR() ((void)0)
#endif
#ifndef OD_DECIDE_OCC_ATTR
# define OD_DECIDE_OCC_ATTR OD_SEARCH_OCC_ATTR
#endif
/* Independent wavefronts per search workgroup sharing one 1/sqrt table (with-reference searches). */
#ifndef ODHIP_SEARCH_WAVES
# define ODHIP_SEARCH_WAVES 1
#endif
