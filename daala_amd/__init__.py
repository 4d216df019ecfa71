"""daala_amd - MI355X-native block-transform hot path of the Daala encoder.

The product is the C-ABI shared library daala_amd/lib/libdaalahip.so
(include/daala_hip.h).  This package is only the thin host-side mirror used by
tests, bench.py and the frame-sharded driver: it loads the library with ctypes
and passes torch device pointers / streams to it.  PyTorch is plumbing here
(device memory, streams, torch.distributed), not the compute path.

There is NO CPU fallback: importing `daala_amd.api` raises if the HIP library
has not been built, and every call raises if the library reports an error.
"""
from .api import (DaalaHipError, PulseRangeError, pvq_k_range_take, lib, lib_path, EXPERIMENTS_LIB, init, fdct2d_batch, idct2d_batch,  # noqa: F401
                  fdct2d_plane, idct2d_plane, filter_batch, dering_planes, forward_pyramid, inverse_level,
                  pvq_search_batch, pvq_search_row_batch, copy_ceiling, decode_export_sections, export_layout_make, pvq_band_layout, alloc_pvq_cands, unpack_cands, BAND_RECORD,
                  pvq_noref_bands,
                  pvq_select_synth_noref, PvqJob, pvq_noref_bands_multi,
                  pvq_select_synth_noref_multi, pvq_choose_multi, inverse_level_pvq, inverse_levels_pvq, pvq_profile, pvq_profile_read, pvq_ref_prepare,
                  pvq_ref_candidates, pvq_synthesis, REFPREP_RECORD, REFCAND_RECORD, host,
                  PvqRefJob, pvq_ref_bands_multi, pvq_ref_select_synth_multi,
                  pvq_ref_set_theta_margin, pvq_ref_theta_probe, REF_SLOTS, REFBAND_RECORD,
                  REFITEM_RECORD, REFBAND_R_NULL, REFBAND_THETA, REFBAND_NOREF, REFBAND_FLIP,
                  REFBAND_UNCERTAIN, REFITEM_SEARCHED, REFITEM_WITH_REF, REFITEM_K_RANGE, REFITEM_MOMENT_SHIFT, pvq_ref_profile,
                  pvq_ref_profile_read, image_planes_copy_pad, inverse_levels, pvq_ref_resolve_finish, cfl_refs_from_luma, pvq_ref_set_context,
                  pvq_ref_choose_multi, inverse_levels_pvq_ref, Context, Pipe, PIPE_STAGES,
                  BUF_PIC, BUF_PX, BUF_LEVEL, BUF_RECON, BUF_BAND, BUF_Y, BUF_CHOICE, BUF_ITEMS,
                  BUF_REF, BUF_RATE, compute_dist, set_price_tol_scale, px_dtype,
                  image_planes_copy_pad16, pvq_choose_priced_multi,
                  pvq_ref_choose_priced_multi, pvq_ref_bands_decided_multi, BAND_RECORD,
                  pvq_decode_bands)
from .quant import QuantTables, OD_PVQ_LAMBDA  # noqa: F401
